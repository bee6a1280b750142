#!/usr/bin/env python3
"""Benchmark of the Morton-key radix sort hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one sort of the resident batch: 2^28 (30-bit Morton key, u32 index)
pairs per GPU through an MRS_ALGO_HYBRID plan over key bits [0, 30): an exact
histogram of the leading 15 bits, two chained-scan scatters by those bits and a
sort of the 32768 sub-buckets inside LDS (three moves of every pair; sub-buckets
that do not fit a workgroup get two extra passes of their own, keys that crowd
into few sub-buckets are scattered by tables the device cuts to their density,
and keys that cannot be cut either take four 8-bit passes: the line reports what
the device chose as `schedule`, and `hybrid_other_inputs` times such inputs).
`--algo onesweep` times the four least-significant-digit-first passes over all
32 key bits instead.  Inputs are synthetic: points from the library's counter-based generator (seed 7) encoded to
Morton keys on the GPU before the timed region.  Every step sorts a FRESH
unsorted buffer from a pool built beforehand (sorting mutates its input, and
re-sorting sorted data would be a different workload), so the timed region
holds nothing but the K sorts.

With N > 1 a step is the sharded sort of 2^31 pairs in all, 2^31 / N per GPU (strong scaling: the
metric's curve is quoted on 2^31 keys; `--log2n L` fixes 2^L pairs per GPU instead, weak scaling),
one mrs_dist_sort_pairs call per rank: top-digit histogram, RCCL all-reduce + all-gather,
splitter, local partition pass, all-to-all-v exchange over xGMI, local sort.  The ranks are
one process per GPU: either torch.distributed.run started them (RANK / WORLD_SIZE set), or this
script does so itself when called as `python bench.py --gpus N` -- the parent then only spawns
`python -m torch.distributed.run ... bench.py` and passes on its exit status, without ever
touching the GPU.  `n_gpus` in the line is the world the communicator reports.

Rank 0 prints one JSON line.  `value` is whole-job Gkeys/s.  `roofline` is for
the dominant kernel (the chained-scan digit pass: 16 algorithmic bytes per pair per
launch; two launches per hybrid sort, four per onesweep sort) timed with HIP events
on the launch stream; `cpu_baseline` is
__gnu_parallel::sort of the whole step input on the host cores (N = 1 only);
`c4` is the same sort at 2^30 pairs (N = 1 only, when memory allows).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch and the library are imported by the ranks only (load_modules): the process that merely
# starts the ranks must not touch the GPU, nor map libmrs.so
torch = None
mrs = None

HBM_PEAK_BYTES_PER_S = 8.0e12          # MI355X HBM3E peak
SORT_BYTES_PER_PAIR = 68.0             # 4 (key read for histograms) + 4 passes * (8 read + 8 write)
PASS_BYTES_PER_PAIR = 16.0             # one digit pass: read key+index, write key+index
SEED = 7


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=None,
                    help="pairs per GPU = 2^log2n (default: 28 on one GPU; with N > 1 GPUs 2^31 / N, the fixed total the "
                         "metric's curve is quoted on)")
    ap.add_argument("--algo", default="hybrid", choices=["hybrid", "onesweep", "three_kernel"])
    ap.add_argument("--end-bit", type=int, default=0,
                    help="one past the last key bit sorted on (0: 30 for --algo hybrid, whose keys are 30-bit codes, else 32)")
    ap.add_argument("--digit-bits", type=int, default=8)
    ap.add_argument("--tile-items", type=int, default=0)
    ap.add_argument("--pool", type=int, default=0, help="distinct unsorted inputs (0 = steps + warmup, capped by memory)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log2", type=int, default=0, help="pairs of the parallel CPU sort = 2^this (0 = the whole step input)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-c4", action="store_true", help="skip the 2^30-pair leg")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed steps and the roofline's three event-timed sorts: no other sizes or inputs, no "
                         "encode pipelines, no CPU baseline, no context number, no 2^30 / 2^31 legs (for profiler runs, so "
                         "that every launch in a kernel summary is a launch of the headline size)")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="every rank prints its RANK / WORLD_SIZE / pairs per GPU as one JSON line and exits "
                         "without touching the GPU or the library (checks the rank launcher)")
    return ap.parse_args(argv)


TOTAL_PAIRS_CURVE = 1 << 31   # the metric's 1/2/4/8-GPU curve is quoted on 2^31 keys in all


def pairs_per_gpu(log2n, world):
    """(pairs each rank holds, "weak" | "strong").  --log2n fixes the per-GPU size (weak scaling);
    without it one GPU sorts the headline 2^28 pairs and N > 1 GPUs share 2^31 (strong scaling)."""
    if log2n is not None:
        return 1 << log2n, "weak"
    if world <= 1:
        return 1 << 28, "weak"
    return TOTAL_PAIRS_CURVE // world, "strong"


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start N ranks of this script as
    child processes, one per GPU, and return their exit status.  Nothing here touches the GPU,
    imports torch or maps libmrs.so: a process that has initialised the GPU must not be the one
    that execs or forks the ranks.  Rank 0's JSON line reaches stdout through the inherited pipe."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    rc = subprocess.run(cmd, env=env).returncode
    if args.launch_selftest:
        print(json.dumps({"selftest_parent": True, "children_status": rc, "torch_imported": "torch" in sys.modules,
                          "library_imported": "wgpu_morton_code_radix_sort_test_amd" in sys.modules}), flush=True)
    return rc


def size_label(n):
    return f"2^{n.bit_length() - 1}" if n & (n - 1) == 0 else str(n)


def load_modules():
    global torch, mrs
    import torch as _torch

    import wgpu_morton_code_radix_sort_test_amd as _mrs

    torch, mrs = _torch, _mrs


SCHEDULE_NAMES = {0: "passes", 1: "msd", 2: "msd refused by the device", 3: "msd adaptive"}


def make_input(ctx, n, seed, shape="uniform"):
    """points -> (keys, idx) on the device; returns the two tensors.  shape: "uniform" (the metric's
    input: points uniform in the unit cube); "cluster": the same with 5 % of the points inside a cube
    of 2 % of the side; "surface": half of the points on a sphere shell, the other half on three
    axis-aligned planes (what the centroids of a triangle mesh look like); "equal": one point."""
    xyz = ctx.generate_points(seed, n)
    if shape == "cluster":
        m = n // 20
        xyz[:m] = xyz[:m] * 0.02 + 0.4
    elif shape == "surface":
        half = n // 2
        d = xyz[:half] - 0.5
        d /= d.norm(dim=1, keepdim=True).clamp_min(1e-6)
        xyz[:half] = 0.5 + 0.45 * d
        del d
        third = (n - half) // 3
        xyz[half:half + third, 0] = 0.3
        xyz[half + third:half + 2 * third, 1] = 0.55
        xyz[half + 2 * third:, 2] = 0.8
    elif shape == "equal":
        xyz[:] = 0.3
    if shape == "uniform":
        mn = xyz.amin(dim=0).cpu().numpy()
        mx = xyz.amax(dim=0).cpu().numpy()
    else:
        mn, mx = [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]
    keys, idx = ctx.morton_encode30(xyz, mn, mx)
    del xyz
    return keys, idx


def verify_sorted(keys, idx, ref_keys_unsorted):
    """Size-independent checks on the device: non-decreasing keys, stable ties, a permutation
    whose gathered input keys equal the output keys."""
    k = keys.to(torch.int64) & 0xFFFFFFFF
    i = idx.to(torch.int64) & 0xFFFFFFFF
    dk = k[1:] - k[:-1]
    ok_sorted = bool((dk >= 0).all())
    ok_stable = bool(((dk > 0) | (i[1:] > i[:-1])).all())
    gathered = ref_keys_unsorted[i]
    ok_perm = bool((gathered == keys).all())
    return ok_sorted and ok_stable and ok_perm


def cpu_baseline(sample_keys, sample_idx, whole):
    """std::sort on the host (oracle/cpu_sort.cpp): single thread and all usable cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    cs = oracle_lib.CpuSort()
    hw = cs.hardware_threads()
    threads = max(1, min(hw, int(os.environ.get("OMP_NUM_THREADS", hw))))
    n = sample_keys.size
    k1, i1 = sample_keys.copy(), sample_idx.copy()
    secs_par, used = cs.sort_pairs(k1, i1, threads)
    n_single = max(1, min(n // 4, 1 << 23))
    k2, i2 = sample_keys[:n_single].copy(), sample_idx[:n_single].copy()
    secs_one, _ = cs.sort_pairs(k2, i2, 1)
    return {
        "value": n / secs_par / 1e9,
        "unit": "Gkeys/s",
        "cores": used,
        "kind": "port",
        "sample": f"std::sort (__gnu_parallel, {used} threads) of {'all' if whole else 'the first'} 2^{int(np.log2(n))} (key,index) pairs of the "
                  f"step input, packed as u64; single-thread std::sort on 2^{int(np.log2(n_single))} pairs: "
                  f"{n_single / secs_one / 1e9:.4f} Gkeys/s; host reports {hw} hardware threads",
        "single_thread_gkeys_per_s": n_single / secs_one / 1e9,
    }, k1, i1


def rocprim_context(keys, vals, end_bit, want_k, want_v, sorts=3):
    """Same-box context number (not a target, not part of the product): rocprim::radix_sort_pairs
    on `keys`/`vals` (left untouched) over key bits [0, end_bit), timed with HIP events on torch's
    current stream, its output compared with the library's (`want_k`, `want_v`).  None when
    tools/librocprim_context.so has not been built."""
    import ctypes

    path = os.path.join(ROOT, "tools", "librocprim_context.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.rocprim_context_temp_bytes.restype = ctypes.c_size_t
    lib.rocprim_context_temp_bytes.argtypes = [ctypes.c_size_t, ctypes.c_uint, ctypes.c_uint]
    lib.rocprim_context_sort_pairs.restype = ctypes.c_int
    lib.rocprim_context_sort_pairs.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p]
    n = keys.numel()
    temp_bytes = lib.rocprim_context_temp_bytes(n, 0, end_bit)
    if temp_bytes == 0:
        return {"error": "rocprim::radix_sort_pairs refused the size query"}
    temp = torch.empty(temp_bytes, dtype=torch.uint8, device=keys.device)
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    times = []
    for j in range(sorts + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.rocprim_context_sort_pairs(temp.data_ptr(), temp_bytes, keys.data_ptr(), ko.data_ptr(), vals.data_ptr(),
                                            vo.data_ptr(), n, 0, end_bit, stream)
        e1.record()
        torch.cuda.synchronize()
        if rc != 0:
            return {"error": f"rocprim::radix_sort_pairs returned hipError {rc}"}
        if j:  # the first sort warms up
            times.append(e0.elapsed_time(e1))
    ms = float(np.mean(times))
    return {
        "rocprim_gkeys_per_s": n / (ms * 1e-3) / 1e9,
        "rocprim_ms_per_sort": round(ms, 4),
        "pairs": n,
        "key_bits": end_bit,
        "temp_bytes": int(temp_bytes),
        "verified": bool(torch.equal(ko, want_k) and torch.equal(vo, want_v)),
        "what": f"rocprim::radix_sort_pairs (ROCm's device-wide radix sort, default config) on the same (key, index) pairs, "
                f"input to output buffers, mean of {sorts} sorts after 1 warm-up, HIP events; `verified` = its keys and "
                f"payload equal the library's output bit for bit",
    }


def c4_leg(ctx, log2n=30, sorts=3, context=False):
    """The 2^30-pair configuration (and, with log2n=31, the whole 2^31-pair job of the 8-GPU
    configuration on one GPU); HIP events of the library around every launch.  2^30 pairs take an
    MRS_ALGO_HYBRID plan over key bits [0, 30) (three moves; the four passes over all 32 bits, with
    64-bit tile status words and offsets, are timed beside it as `four_passes`), 2^31 the four
    passes.  Returns None when the device is short of memory."""
    n = 1 << log2n
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 44 * n:  # points 12 B + pairs 8 B x 3 (input, pristine copy, ping-pong) + slack
        return None
    xyz = ctx.generate_points(SEED + 30, n)
    keys, idx = ctx.morton_encode30(xyz, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    del xyz
    pk, pi = keys.clone(), idx.clone()
    kb, vb = torch.empty_like(keys), torch.empty_like(idx)
    def timed(plan, labels):
        plan.set_timing(True)
        totals, passes, launches = [], [], {}
        for j in range(sorts + 1):
            keys.copy_(pk)
            idx.copy_(pi)
            plan.sort_pairs(keys, idx, kb, vb)
            tm = plan.timings()
            if j:  # the first sort warms up
                totals.append(sum(ms for _, ms in tm))
                passes = [ms for lab, ms in tm if lab in labels]
                launches = {}
                for lab, ms in tm:
                    launches[lab] = round(launches.get(lab, 0.0) + ms, 4)
        plan.check()
        return float(np.mean(totals)), passes, launches

    three_moves = log2n == 30
    four = None
    if three_moves:
        plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, algo=mrs.ALGO_ONESWEEP)
        ms4, passes4, _ = timed(plan, ("onesweep_pass",))
        four = {"ms_per_sort": round(ms4, 4), "gkeys_per_s": n / (ms4 * 1e-3) / 1e9,
                "per_pass_ms": [round(float(x), 4) for x in passes4], "verified": bool(verify_sorted(keys, idx, pk)),
                "algo": "onesweep: four passes over all 32 key bits, 64-bit tile status words"}
        plan.close()
        plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, end_bit=30, algo=mrs.ALGO_HYBRID)
        ms, passes, launches = timed(plan, ("msd_pass_a", "msd_pass_b"))
        schedule = plan.schedule()[0]
    else:
        plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, algo=mrs.ALGO_ONESWEEP)
        ms, passes, launches = timed(plan, ("onesweep_pass",))
        schedule = 0
    ok = verify_sorted(keys, idx, pk)
    plan.close()
    del kb, vb
    rp = None
    if context:
        try:
            rp = rocprim_context(pk, pi, 30 if three_moves else 32, keys, idx, sorts=2)
        except Exception as exc:  # a context number: the leg stands without it
            rp = {"error": str(exc)}
    return {
        "rocprim_context": rp,
        "pairs": n,
        "ms_per_sort": round(ms, 4),
        "gkeys_per_s": n / (ms * 1e-3) / 1e9,
        "frac_68B": SORT_BYTES_PER_PAIR * n / (ms * 1e-3) / HBM_PEAK_BYTES_PER_S,
        "per_pass_ms": [round(float(x), 4) for x in passes],
        "pass_frac_16B": PASS_BYTES_PER_PAIR * n / (float(np.mean(passes)) * 1e-3) / HBM_PEAK_BYTES_PER_S,
        "schedule": SCHEDULE_NAMES[schedule],
        "launch_ms": launches,
        "four_passes": four,
        "algo": ("hybrid over key bits [0, 30): exact 16-bit histogram, two chained-scan scatters, 65536 sub-buckets of about 16384 pairs "
                 "sorted in LDS by workgroups of 1024 threads (20480 pairs each); 32-bit tile status words")
                if three_moves else
                ("onesweep: four passes over all 32 key bits, 64-bit tile status words (the three-move schedule needs sub-buckets of at "
                 "most 20480 pairs among at most 65536, n / 65536 here is above that)"),
        "timing": f"HIP events recorded by the library around every launch of {sorts} sorts after 1 warm-up",
        "verified": bool(ok),
    }


def load_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC run, if present."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(path):
        try:
            with open(path) as f:
                return json.load(f)
        except Exception:
            return None
    return None


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: this process only starts the ranks
        raise SystemExit(launch_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n, scaling = pairs_per_gpu(args.log2n, world)
    if args.launch_selftest:
        print(json.dumps({"selftest": True, "rank": rank, "local_rank": local_rank, "world": world, "pairs_per_gpu": n,
                          "scaling": scaling, "torch_imported": "torch" in sys.modules,
                          "library_imported": "wgpu_morton_code_radix_sort_test_amd" in sys.modules}), flush=True)
        return
    load_modules()
    # MRS_BENCH_FORCE_DIST=1 runs the sharded path with a single rank (rehearsal on a one-GPU box)
    distributed = world > 1 or os.environ.get("MRS_BENCH_FORCE_DIST") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the sort has no CPU path")
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist

    if world > 1:
        # torch.distributed carries the RCCL unique id to the ranks, the barriers and the max over
        # ranks of the elapsed time; the sort's own collectives are RCCL calls inside libmrs.so
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ctx = mrs.Context(local_rank)
    algo = {"hybrid": mrs.ALGO_HYBRID, "onesweep": mrs.ALGO_ONESWEEP}.get(args.algo, mrs.ALGO_THREE_KERNEL)
    end_bit = args.end_bit or (30 if algo == mrs.ALGO_HYBRID else 32)

    total_steps = args.steps + args.warmup
    free_b, _ = torch.cuda.mem_get_info()
    per_input = 8 * n
    budget = int(free_b * 0.6) - 4 * per_input
    pool = args.pool or total_steps
    pool = max(1, min(pool, max(1, budget // per_input)))

    if distributed:
        from wgpu_morton_code_radix_sort_test_amd.dist import (Communicator, DistributedSorter, pair_checksum,
                                                               verify_distributed)

        comm = Communicator.rccl(rank, world)
        # the keys are 30-bit Morton codes: all 4096 buckets of the top digit are live
        # MRS_BENCH_DIST_CHUNKS: pieces a rank's share arrives in (0 = the library's choice; rehearsal aid)
        chunks = int(os.environ.get("MRS_BENCH_DIST_CHUNKS", "0"))
        sorter = DistributedSorter(ctx, comm, n, key_words=1, payload_words=1, end_bit=30, exchange_chunks=chunks)
        # every rank shares the scene bounds, so that the key ranges of the slices overlap fully
        inputs = []
        for j in range(pool):
            xyz = ctx.generate_points(SEED + 1000 * rank + j, n)
            k, i = ctx.morton_encode30(xyz, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
            del xyz
            i += rank * n  # payload = global index
            inputs.append((k, i))
        restore = False

        def step(j):
            k, i = inputs[j % pool]
            return sorter.sort(k, i)
    else:
        plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=args.digit_bits, end_bit=end_bit, algo=algo,
                            tile_items=args.tile_items)
        inputs = [make_input(ctx, n, SEED + j) for j in range(pool)]
        kb = torch.empty(n, dtype=torch.int32, device=ctx.device)
        vb = torch.empty(n, dtype=torch.int32, device=ctx.device)
        restore = pool < total_steps
        if restore:
            pristine = [(k.clone(), i.clone()) for k, i in inputs[:1]]

        def step(j):
            k, i = inputs[j % pool]
            if restore and j >= pool:
                # the pool is smaller than the run: put unsorted data back (inside the timed region)
                k.copy_(pristine[0][0])
                i.copy_(pristine[0][1])
            in_b = plan.sort_pairs(k, i, kb, vb)
            return (kb, vb) if in_b else (k, i)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for j in range(args.warmup):
        step(j)
    barrier()
    t0 = time.perf_counter()
    for j in range(args.warmup, total_steps):
        out = step(j)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    total_pairs = n * world
    value = total_pairs / (elapsed / args.steps) / 1e9

    # ---- correctness of what was just timed (size-independent properties) -------------------
    verified = None
    if not args.no_verify and not distributed:
        ref_keys, _ = make_input(ctx, n, SEED + ((total_steps - 1) % pool)) if not (restore and total_steps > pool) \
            else (pristine[0][0], None)
        plan.check()
        verified = verify_sorted(out[0], out[1], ref_keys)
        del ref_keys
        if not verified:
            raise SystemExit("bench.py: the sorted output failed verification")
    elif distributed and not args.no_verify:
        sorter.check()
        k_in, i_in = inputs[(total_steps - 1) % pool]
        verified = verify_distributed(out[0], out[1], pair_checksum(k_in, i_in), n)
        if not verified:
            raise SystemExit("bench.py: the distributed output failed verification")

    # ---- roofline of the dominant kernel: HIP events on the launch stream ----------------------
    roofline = None
    extra = {}
    if not distributed and algo in (mrs.ALGO_ONESWEEP, mrs.ALGO_HYBRID):
        hybrid = algo == mrs.ALGO_HYBRID
        pass_labels = ("msd_pass_a", "msd_pass_b") if hybrid else ("onesweep_pass",)
        hist_label = "msd_histogram" if hybrid else "global_histogram"
        plan.set_timing(True)
        pass_ms, hist_ms, sort_ms, launches = [], [], [], {}
        fresh = [make_input(ctx, n, SEED + 500 + j) for j in range(3)]
        for k, i in fresh:
            plan.sort_pairs(k, i, kb, vb)
            tm = plan.timings()
            pass_ms += [ms for lab, ms in tm if lab in pass_labels]
            hist_ms += [ms for lab, ms in tm if lab == hist_label]
            sort_ms.append(sum(ms for _, ms in tm))
            for lab, ms in tm:
                launches.setdefault(lab, []).append(ms)
        plan.set_timing(False)
        schedule, largest = plan.schedule()
        del fresh
        avg_pass = float(np.mean(pass_ms)) * 1e-3
        achieved = PASS_BYTES_PER_PAIR * n / avg_pass / 1e9
        traffic = load_traffic()
        if traffic and hybrid:
            traffic = traffic.get("hybrid_pass") or None
        roofline = {
            "bound": "hbm",
            "kernel": "hybrid_pass_kernel (the chained-scan digit pass: pass A over the array, pass B bucket by bucket)"
                      if hybrid else "digit_pass_kernel<onesweep>",
            "achieved": achieved,
            "peak": HBM_PEAK_BYTES_PER_S / 1e9,
            "unit": "GB/s",
            "frac": achieved * 1e9 / HBM_PEAK_BYTES_PER_S,
            "traffic": traffic.get("bytes_per_launch") if traffic else None,
            "traffic_source": (f"profiles/traffic.json: builder-side rocprofv3 --pmc runs (FETCH_SIZE doubled as the gfx950 "
                               f"guide prescribes, WRITE_SIZE), {traffic.get('measured', 'date unknown')}; not measured in this run")
            if traffic else None,
            "algorithmic_bytes_per_launch": PASS_BYTES_PER_PAIR * n,
            "avg_launch_ms": avg_pass * 1e3,
        }
        extra = {
            "per_pass_ms": [round(float(x), 4) for x in pass_ms[-(2 if hybrid else plan.info.num_passes):]],
            "global_histogram_ms": round(float(np.mean(hist_ms)), 4),
            "global_histogram_GBps": 4.0 * n / (float(np.mean(hist_ms)) * 1e-3) / 1e9,
            "sort_event_ms": round(float(np.mean(sort_ms)), 4),
            "sort_roofline_frac_68B": SORT_BYTES_PER_PAIR * n / (float(np.mean(sort_ms)) * 1e-3) / HBM_PEAK_BYTES_PER_S,
        }
        if hybrid:
            # every launch of one sort, mean of three: both schedules are enqueued, the one the
            # device did not choose returns at once (its launches are the sub-0.01 ms entries)
            extra["launch_ms"] = {lab: round(float(np.mean(v)) * (len(v) / 3.0), 4) for lab, v in launches.items()}
            extra["schedule"] = SCHEDULE_NAMES[schedule]
            extra["largest_sub_bucket"] = largest
            local = float(np.mean(launches.get("msd_local_sort", [0.0]))) * 1e-3
            if local > 0:
                extra["local_sort_frac_16B"] = PASS_BYTES_PER_PAIR * n / local / HBM_PEAK_BYTES_PER_S
            extra["moved_bytes_per_pair"] = 4 + 3 * 16  # what this schedule reads and writes; the 68 above is the metric's figure
            # the same plan type at the other sizes it serves and on other inputs: clustered points (a few
            # oversized sub-buckets, which two extra segmented passes sort where they lie), points on
            # surfaces (most fixed sub-buckets oversized: the device cuts buckets and groups to the keys'
            # density, "msd adaptive"), equal keys (nothing to cut: the four ordinary passes), and 30-bit
            # codes declared as 32-bit keys (three quarters of the leading buckets empty: adaptive);
            # library events, mean of 3 sorts after 1 warm-up; not part of `value`
            sweep = {}
            for label, log2m, eb, shape in () if args.no_extras else (("2^26", 26, 30, "uniform"), ("2^27", 27, 30, "uniform"), ("2^29", 29, 30, "uniform"),
                                            ("2^28 uniform + 5 % cluster", 28, 30, "cluster"),
                                            ("2^28 surface (sphere shell + three planes)", 28, 30, "surface"),
                                            ("2^28 all equal", 28, 30, "equal"), ("2^28 30-bit codes declared as 32-bit keys", 28, 32, "uniform")):
                m = 1 << log2m
                if m > n and log2m != 29:  # the ping-pong buffers are the headline's
                    continue
                p2 = mrs.SortPlan(ctx, m, key_words=1, payload_bytes=4, digit_bits=8, end_bit=eb, algo=mrs.ALGO_HYBRID)
                p2.set_timing(True)
                k2, i2 = make_input(ctx, m, SEED + 900 + log2m, shape)
                kb2, vb2 = (kb[:m], vb[:m]) if m <= n else (torch.empty_like(k2), torch.empty_like(i2))
                pk, pi = k2.clone(), i2.clone()
                ms2 = []
                for j in range(4):
                    k2.copy_(pk)
                    i2.copy_(pi)
                    p2.sort_pairs(k2, i2, kb2, vb2)
                    if j:
                        ms2.append(sum(t for _, t in p2.timings()))
                p2.check()
                which, biggest = p2.schedule()
                over_subs, over_pairs, cap = p2.oversized()
                sweep[label] = {"ms_per_sort": round(float(np.mean(ms2)), 4), "gkeys_per_s": m / (float(np.mean(ms2)) * 1e-3) / 1e9,
                                "schedule": SCHEDULE_NAMES[which],
                                "largest_sub_bucket": biggest, "oversized_sub_buckets": over_subs, "oversized_pairs": over_pairs,
                                "workgroup_capacity": cap, "verified": bool(verify_sorted(k2, i2, pk))}
                if which == 3:
                    sweep[label]["adaptive_buckets"], sweep[label]["adaptive_groups"] = p2.adaptive()
                    sweep[label]["launch_ms"] = {lab: round(float(t), 4) for lab, t in p2.timings()}
                p2.close()
                del k2, i2, pk, pi, kb2, vb2
            extra["hybrid_other_inputs"] = sweep
        if not args.no_extras:
            # Morton encode, reported separately: 12 B read + 8 B written per point
            xyz = ctx.generate_points(SEED, n)
            mn, mx = xyz.amin(dim=0).cpu().numpy(), xyz.amax(dim=0).cpu().numpy()
            ek, ei = ctx.morton_encode30(xyz, mn, mx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ctx.morton_encode30(xyz, mn, mx, ek, ei)
            e1.record()
            torch.cuda.synchronize()
            enc_ms = e0.elapsed_time(e1) / 5
            extra["morton_encode30_ms"] = round(enc_ms, 4)
            extra["morton_encode30_GBps"] = 20.0 * n / (enc_ms * 1e-3) / 1e9
            # encode + sort as one pipeline, with the encoder that leaves the digit histograms in the
            # plan (the sort then skips its own read of the keys) and without; not part of `value`
            # ... and with no index array at all: the encoder writes keys only (12 + 4 B per point) and the
            # sort makes the indices up in its first pass (mrs_sort_pairs_indexed, 64 B per pair)
            for label, primed in (("encode_then_sort_ms", False), ("encode_for_sort_then_sort_ms", True),
                                  ("encode_keys_then_sort_indexed_ms", None)):
                times = []
                for _ in range(4):
                    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    p0.record()
                    if primed is None:
                        ctx.morton_encode30(xyz, mn, mx, ek, keys_only=True)
                        plan.sort_pairs_indexed(ek, ei, kb, vb)
                    elif primed:
                        plan.morton_encode30_for_sort(xyz, mn, mx, ek, ei)
                        plan.sort_pairs_primed(ek, ei, kb, vb)
                    else:
                        ctx.morton_encode30(xyz, mn, mx, ek, ei)
                        plan.sort_pairs(ek, ei, kb, vb)
                    p1.record()
                    torch.cuda.synchronize()
                    times.append(p0.elapsed_time(p1))
                extra[label] = round(float(np.median(times[1:])), 4)
            del xyz, ek, ei

    context = None
    if not distributed and rank == 0 and algo in (mrs.ALGO_ONESWEEP, mrs.ALGO_HYBRID) and not args.no_extras:
        # same-box context (never a target): the vendor's radix sort on one of the step inputs
        try:
            k_in, i_in = make_input(ctx, n, SEED + 777)
            pk, pi = k_in.clone(), i_in.clone()
            in_b = plan.sort_pairs(k_in, i_in, kb, vb)
            plan.check()
            want = (kb, vb) if in_b else (k_in, i_in)
            context = rocprim_context(pk, pi, end_bit, want[0], want[1])
            del k_in, i_in, pk, pi, want
        except Exception as exc:
            context = {"error": str(exc)}

    breakdown = {}
    if distributed:
        # the same roofline object at N > 1: digit passes of rank 0's local sort, HIP events on
        # the launch stream (one extra, untimed sort of the last input; collective, so every rank runs it)
        sorter.set_timing(True)
        k, i = inputs[(total_steps - 1) % pool]
        sorter.sort(k, i)
        torch.cuda.synchronize()
        breakdown = sorter.last_breakdown()
        sorter.set_timing(False)
        if breakdown.get("dist_local_sort_passes") and rank == 0:
            # the timed launches are the digit passes of the LAST piece of rank 0's share only
            avg_pass = breakdown["dist_local_sort_pass_ms"] * 1e-3
            piece_pairs = breakdown["dist_local_sort_piece_pairs"]
            achieved = PASS_BYTES_PER_PAIR * piece_pairs / avg_pass / 1e9
            roofline = {
                "bound": "hbm",
                "kernel": "the chained-scan digit pass (local sort of rank 0: " + ", ".join(breakdown.get("dist_local_sort_schedule", [])) + ")",
                "achieved": achieved,
                "peak": HBM_PEAK_BYTES_PER_S / 1e9,
                "unit": "GB/s",
                "frac": achieved * 1e9 / HBM_PEAK_BYTES_PER_S,
                "traffic": None,
                "algorithmic_bytes_per_launch": PASS_BYTES_PER_PAIR * piece_pairs,
                "avg_launch_ms": avg_pass * 1e3,
            }
        if world > 1:
            dist.barrier()

    cpu = None
    if rank == 0 and not distributed and not args.no_cpu_baseline and not args.no_extras:
        m = n if args.cpu_sample_log2 == 0 else min(n, 1 << args.cpu_sample_log2)
        sk, si = make_input(ctx, n, SEED + ((total_steps - 1) % pool)) if not (restore and total_steps > pool) \
            else pristine[0]
        sample_k = sk[:m].cpu().numpy().view(np.uint32)
        sample_i = si[:m].cpu().numpy().view(np.uint32)
        del sk, si
        cpu, _, _ = cpu_baseline(sample_k, sample_i, m == n)

    c4 = None
    c5_one_gpu = None
    if rank == 0 and not distributed and not args.no_c4 and not args.no_extras and algo in (mrs.ALGO_ONESWEEP, mrs.ALGO_HYBRID) and n == (1 << 28):
        # free the pool first: the 2^30 leg wants about 45 GiB
        del inputs, kb, vb, out
        if restore:
            del pristine
        torch.cuda.empty_cache()
        try:
            c4 = c4_leg(ctx, context=True)
        except Exception as exc:  # an extra: the headline line stands without it
            c4 = {"error": str(exc)}
        # the whole 8-GPU job (2^31 pairs) on this one GPU: the one-GPU point of a curve over 2^31 keys
        torch.cuda.empty_cache()
        try:
            c5_one_gpu = c4_leg(ctx, log2n=31, sorts=2)
        except Exception as exc:
            c5_one_gpu = {"error": str(exc)}

    if rank == 0:
        line = {
            "metric": "Gkeys/s sorted (u32 Morton key + u32 payload)",
            "value": value,
            "unit": "Gkeys/s",
            "n_gpus": comm.world if distributed else 1,  # what the communicator reports, not the argument
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"{size_label(n)} (30-bit Morton key, u32 index) pairs per GPU"
                            + (f" ({size_label(total_pairs)} in all)" if world > 1 else "") + " from uniform random 3D points, "
                            + (f"key bits [0, 30), local sorts by the three-move schedule over each piece's window of the key space (four 8-bit passes where the device refuses it)" if distributed else
                               f"{args.digit_bits}-bit digits over key bits [0, {end_bit}), {args.algo}"),
                "pairs_per_gpu": n,
                "total_pairs": total_pairs,
                "digit_bits": args.digit_bits,
                "algo": args.algo,
                "distinct_inputs": pool,
                "parallelism": f"{world} x MI355X, sharded by top-digit bucket (mrs_dist_sort_pairs over RCCL)" if distributed else "1 x MI355X",
            },
            "verified": verified,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "context": context,
        }
        line.update(extra)
        if c4 is not None:
            line["c4"] = c4
        if c5_one_gpu is not None:
            line["c5_one_gpu"] = c5_one_gpu
        line.update(breakdown)
        print(json.dumps(line))
    if distributed:
        # every rank is past its last collective before any communicator goes away
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        sorter.close()
        comm.close()
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
