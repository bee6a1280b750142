"""GPU tests at the benchmark's full sizes, through size-independent properties
(sortedness, stability, permutation, checksums), plus the distributed driver on one rank."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mrs():
    import wgpu_morton_code_radix_sort_test_amd as m

    return m


def check_sorted_pairs(keys_out, idx_out, keys_in):
    """keys_out non-decreasing as u32; ties keep ascending original index; idx_out is a permutation
    that maps the input keys onto the output keys; key checksums (sum, xor) are conserved."""
    k = keys_out.to(torch.int64) & 0xFFFFFFFF
    i = idx_out.to(torch.int64) & 0xFFFFFFFF
    dk = k[1:] - k[:-1]
    assert bool((dk >= 0).all()), "keys are not sorted"
    assert bool(((dk > 0) | (i[1:] > i[:-1])).all()), "equal keys lost their input order"
    del dk
    assert bool((keys_in[i] == keys_out).all()), "payload does not point at the key's origin"
    seen = torch.zeros(k.numel(), dtype=torch.bool, device=k.device)
    seen[i] = True
    assert bool(seen.all()), "payload is not a permutation"
    kin = keys_in.to(torch.int64) & 0xFFFFFFFF
    assert int(kin.sum()) == int(k.sum())


@pytest.mark.parametrize("log2n,algo_name,bits", [(28, "onesweep", 8), (28, "three_kernel", 8), (30, "onesweep", 8),
                                                  (31, "onesweep", 8)])
def test_full_size_sort_properties(ctx, mrs, log2n, algo_name, bits):
    """The metric size, the largest single-GPU configuration and the whole 8-GPU job on one GPU
    (the one-GPU point of the scaling curve): 2^28, 2^30 and 2^31 (Morton key, index) pairs."""
    n = 1 << log2n
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 60 * n:
        pytest.skip("not enough device memory for this size")
    algo = mrs.ALGO_ONESWEEP if algo_name == "onesweep" else mrs.ALGO_THREE_KERNEL
    xyz = ctx.generate_points(7, n)
    keys, idx = ctx.morton_encode30(xyz, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    del xyz
    assert int(keys.max()) < (1 << 30)
    keys_in = keys.clone()
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=bits, algo=algo)
    kb, vb = torch.empty_like(keys), torch.empty_like(idx)
    in_b = plan.sort_pairs(keys, idx, kb, vb)
    plan.check()
    assert not in_b
    check_sorted_pairs(keys, idx, keys_in)
    # idempotence: sorting the sorted array changes nothing
    k2, v2 = keys.clone(), idx.clone()
    plan.sort_pairs(k2, v2, kb, vb)
    plan.check()
    assert bool((k2 == keys).all()) and bool((v2 == idx).all())
    # the sort that makes the indices up gives the same pairs
    k2.copy_(keys_in)
    v2.fill_(-1)
    in_b = plan.sort_pairs_indexed(k2, v2, kb, vb)
    plan.check()
    assert not in_b
    assert bool((k2 == keys).all()) and bool((v2 == idx).all())
    plan.close()


@pytest.mark.parametrize("log2n,end_bit,want", [(28, 30, "msd"), (28, 32, "adaptive"), (27, 30, "msd"), (26, 30, "msd"),
                                                (26, 32, "passes"), (29, 30, "msd"), (30, 30, "msd"), (30, 32, "refused")])
def test_full_size_hybrid_plan_equals_four_passes(ctx, mrs, log2n, end_bit, want):
    """MRS_ALGO_HYBRID plans at the sizes they exist for, against the four-pass plan on the same
    input, pair for pair (a stable sort has one answer), plus the size-independent properties.
    30-bit codes declared as 32-bit keys leave three quarters of the fixed sub-buckets empty and
    the rest too full: the device cuts buckets and groups to the keys instead, and the result is
    still that of the four passes; 2^26 pairs of 32-bit keys are
    below the size the three-move schedule pays for: the host does not enqueue it; 2^29 pairs leave
    sub-buckets of 8192 pairs, which fit a workgroup that stages a pair in 6 bytes; 2^30 pairs leave
    sub-buckets of 16384, which fit the local sort's workgroups of 1024 threads (the ordinary passes a
    refusal falls back to use 64-bit tile status words there: 30-bit codes declared as 32-bit keys)."""
    n = 1 << log2n
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 48 * n:
        pytest.skip("not enough device memory for this size")
    xyz = ctx.generate_points(23, n)
    keys, idx = ctx.morton_encode30(xyz, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    del xyz
    keys_in = keys.clone()
    kb, vb = torch.empty_like(keys), torch.empty_like(idx)
    ref = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, end_bit=end_bit, algo=mrs.ALGO_ONESWEEP)
    rk, rv = keys.clone(), idx.clone()
    assert not ref.sort_pairs(rk, rv, kb, vb)
    ref.check()
    ref.close()
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, end_bit=end_bit, algo=mrs.ALGO_HYBRID)
    assert not plan.sort_pairs(keys, idx, kb, vb)
    plan.check()
    schedule, largest = plan.schedule()
    assert schedule == {"msd": plan.SCHEDULE_MSD, "refused": plan.SCHEDULE_MSD_REFUSED, "passes": plan.SCHEDULE_PASSES,
                        "adaptive": plan.SCHEDULE_MSD_ADAPTIVE}[want], largest
    assert bool((keys == rk).all()) and bool((idx == rv).all())
    check_sorted_pairs(keys, idx, keys_in)
    # the sort that makes the indices up, and a second call on the plan
    keys.copy_(keys_in)
    idx.fill_(-1)
    assert not plan.sort_pairs_indexed(keys, idx, kb, vb)
    plan.check()
    assert bool((keys == rk).all()) and bool((idx == rv).all())
    # all-equal keys at this size: one sub-bucket holds everything, the device takes the four passes
    keys.fill_(0x155A5A5A)
    idx.copy_(torch.arange(n, dtype=torch.int32, device=ctx.device))
    plan.sort_pairs(keys, idx, kb, vb)
    plan.check()
    if want != "passes":
        assert plan.schedule()[0] == plan.SCHEDULE_MSD_REFUSED
    assert bool((idx == torch.arange(n, dtype=torch.int32, device=ctx.device)).all())
    plan.close()


def _shaped_keys(ctx, n, shape, seed):
    """30-bit codes at sizes the CPU checker is too slow for.  "hot20": a fifth of the keys share their
    16 leading bits (one sub-bucket far beyond any 16-bit counter of the histogram); "cluster": 5 % of
    the points in a small cube; "surface": points on a sphere shell and on three planes; "equal": one point."""
    xyz = ctx.generate_points(seed, n)
    if shape == "cluster":
        m = n // 20
        xyz[:m] = xyz[:m] * 0.02 + 0.4
    elif shape == "surface":
        half = n // 2
        d = xyz[:half] - 0.5
        d /= d.norm(dim=1, keepdim=True).clamp_min(1e-6)
        xyz[:half] = 0.5 + 0.45 * d
        third = (n - half) // 3
        xyz[half:half + third, 0] = 0.3
        xyz[half + third:half + 2 * third, 1] = 0.55
        xyz[half + 2 * third:, 2] = 0.8
    elif shape == "equal":
        xyz[:] = 0.3
    keys, idx = ctx.morton_encode30(xyz, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    del xyz
    if shape == "hot20":
        m = n // 5
        keys[:m] = (keys[:m] & 0x3FFF) | (0x2BCD << 14)
        perm = torch.randperm(n, device=keys.device)
        keys = keys[perm].contiguous()
    return keys, idx


@pytest.mark.parametrize("log2n,shape,want", [(26, "hot20", "msd"), (28, "cluster", "msd"), (27, "surface", "adaptive"),
                                              (28, "surface", "adaptive"), (29, "cluster", "msd"), (28, "equal", "refused")])
def test_full_size_uneven_keys(ctx, mrs, log2n, shape, want):
    """Inputs whose sub-buckets are far from even, at sizes only the GPU can check: the result must be
    the four-pass plan's, pair for pair, whatever the device decides.  A sub-bucket with a fifth of
    2^26 keys overflows the 16-bit counters of every histogram workgroup many times over (the
    counts must still be exact) and is sorted by the two extra passes; clustered points leave a
    handful of oversized sub-buckets; points on surfaces leave thousands, and the device cuts buckets
    and groups to the keys' density instead; keys that are all equal cannot be cut at all, and the
    device takes the ordinary passes with the histograms it already has."""
    n = 1 << log2n
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 60 * n:
        pytest.skip("not enough device memory for this size")
    keys, idx = _shaped_keys(ctx, n, shape, 31 + log2n)
    keys_in = keys.clone()
    kb, vb = torch.empty_like(keys), torch.empty_like(idx)
    ref = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, end_bit=30, algo=mrs.ALGO_ONESWEEP)
    rk, rv = keys.clone(), idx.clone()
    assert not ref.sort_pairs(rk, rv, kb, vb)
    ref.check()
    ref.close()
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, end_bit=30, algo=mrs.ALGO_HYBRID)
    for again in range(2):  # the second call runs on the tables the first one left
        keys.copy_(keys_in)
        idx.copy_(torch.arange(n, dtype=torch.int32, device=ctx.device))
        assert not plan.sort_pairs(keys, idx, kb, vb)
        plan.check()
        schedule, largest = plan.schedule()
        subs, pairs, cap = plan.oversized()
        assert schedule == {"msd": plan.SCHEDULE_MSD, "refused": plan.SCHEDULE_MSD_REFUSED,
                            "adaptive": plan.SCHEDULE_MSD_ADAPTIVE}[want], (largest, subs, pairs)
        if want == "adaptive":
            buckets, groups = plan.adaptive()
            assert 128 <= buckets <= 256 and n // cap <= groups <= 65536, (buckets, groups)
        if want == "msd":
            assert 1 <= subs <= 256 and cap < pairs <= n // 4 and largest > cap, (largest, subs, pairs, cap)
            if shape == "hot20":
                assert subs == 1 and largest >= n // 5
        assert bool((keys == rk).all()), "keys differ from the four-pass plan's"
        assert bool((idx == rv).all()), "payload differs from the four-pass plan's"
    check_sorted_pairs(keys, idx, keys_in)
    plan.close()


@pytest.mark.parametrize("algo_name,mask", [("onesweep", 0x3FFFFFF), ("three_kernel", -1)])
def test_largest_supported_sort(ctx, mrs, algo_name, mask):
    """2^32 - 2 (key, source index) pairs, the most a plan takes: element numbers and byte offsets
    beyond 2^31 and 2^34 (64-bit tile status words and offsets).  Checked in slices of 2^28 elements:
    ascending keys, ties in index order (26-bit keys: 64 of each), every index pointing at its key,
    and the index sum of a permutation."""
    n = (1 << 32) - 2
    torch.cuda.empty_cache()  # the plan allocates beside torch's cache
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 24 * n:
        pytest.skip("not enough free device memory for 2^32 pairs")
    algo = {"onesweep": mrs.ALGO_ONESWEEP, "three_kernel": mrs.ALGO_THREE_KERNEL}[algo_name]
    keys = ctx.generate_u32(9, 1, n)
    if mask >= 0:
        keys &= mask
    ka, va = keys.clone(), torch.empty(n, dtype=torch.int32, device=ctx.device)
    kb, vb = torch.empty_like(ka), torch.empty_like(va)
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, algo=algo)
    in_b = plan.sort_pairs_indexed(ka, va, kb, vb)
    plan.check()
    ok, ov = (kb, vb) if in_b else (ka, va)
    step, prev, index_sum = 1 << 28, None, 0
    for at in range(0, n, step):
        k = ok[at:at + step].to(torch.int64) & 0xFFFFFFFF
        v = ov[at:at + step].to(torch.int64) & 0xFFFFFFFF
        dk = k[1:] - k[:-1]
        assert bool(((dk > 0) | ((dk == 0) & (v[1:] > v[:-1]))).all()), f"order breaks inside slice {at >> 28}"
        first = (int(k[0]), int(v[0]))
        assert prev is None or prev < first, f"order breaks before slice {at >> 28}"
        prev = (int(k[-1]), int(v[-1]))
        assert bool(((keys[v].to(torch.int64) & 0xFFFFFFFF) == k).all()), "an index does not point at its key"
        index_sum += int(v.sum())
        del k, v, dk
    assert index_sum == n * (n - 1) // 2
    plan.close()


@pytest.mark.parametrize("log2n,record_words,algo_name", [(28, 4, "hybrid"), (26, 16, "hybrid"), (27, 9, "onesweep")])
def test_full_size_sort_gather(ctx, mrs, log2n, record_words, algo_name):
    """mrs_sort_gather at the metric's size: (30-bit code, source index) pairs sorted by the plan, then
    every record (16, 64 or 36 bytes) moved once.  Word 0 of a record is its source index and word 1
    its key, the rest a function of the index, so the output tells whether every record arrived
    whole, behind its key, in index order among equal keys."""
    n = 1 << log2n
    torch.cuda.empty_cache()
    xyz = ctx.generate_points(21, n)
    keys, idx = ctx.morton_encode30(xyz, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    del xyz
    recs = torch.empty((n, record_words), dtype=torch.int32, device=ctx.device)
    recs[:, 0] = idx
    recs[:, 1] = keys
    for w in range(2, record_words):
        recs[:, w] = idx * (2 * w + 1) + w
    algo = {"hybrid": mrs.ALGO_HYBRID, "onesweep": mrs.ALGO_ONESWEEP}[algo_name]
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, end_bit=30, algo=algo)
    ka, kb = keys.clone(), torch.empty_like(keys)
    rout = torch.empty_like(recs)
    in_b, index = plan.sort_gather(ka, kb, recs, rout, 4 * record_words)
    plan.check()
    ok = kb if in_b else ka
    k = ok.to(torch.int64)
    v = rout[:, 0].to(torch.int64)
    dk = k[1:] - k[:-1]
    assert bool(((dk > 0) | ((dk == 0) & (v[1:] > v[:-1]))).all()), "order or stability"
    assert torch.equal(rout[:, 1], ok), "a record is not behind its key"
    assert torch.equal(index[:n], rout[:, 0]), "the sorted source indices differ from the records'"
    assert int(v.sum()) == n * (n - 1) // 2
    for w in range(2, record_words):
        assert torch.equal(rout[:, w], rout[:, 0] * (2 * w + 1) + w), f"word {w} of a record was torn"
    assert torch.equal(keys[v], ok)
    plan.close()


@pytest.mark.parametrize("payload_words", [1, 4])
def test_large_u64_keys_properties(ctx, mrs, payload_words):
    """64-bit keys (the reference's code type) with an index or a 16-byte TriangleIndices-sized
    payload at 2^26 elements: sorted as u64, stable, payload follows its key."""
    n = 1 << 26
    lo = ctx.generate_u32(11, 1, n)
    hi = ctx.generate_u32(11, 2, n) & 0x3FFFF  # some equal high words, as short codes have
    keys = torch.stack([lo, hi], dim=1).reshape(-1).contiguous()
    src = torch.arange(n, dtype=torch.int32, device=ctx.device)
    if payload_words == 1:
        vals = src.clone()
    else:
        vals = torch.stack([src, src ^ 0x5A5A5A5A, lo, hi], dim=1).reshape(-1).contiguous()
    k64_in = (hi.to(torch.int64) << 32) | (lo.to(torch.int64) & 0xFFFFFFFF)
    plan = mrs.SortPlan(ctx, n, key_words=2, payload_bytes=4 * payload_words)
    kb, vb = torch.empty_like(keys), torch.empty_like(vals)
    in_b = plan.sort_pairs(keys, vals, kb, vb)
    plan.check()
    assert not in_b  # 8 passes
    ko = keys.view(n, 2)
    k64 = (ko[:, 1].to(torch.int64) << 32) | (ko[:, 0].to(torch.int64) & 0xFFFFFFFF)
    vo = vals.view(n, payload_words)
    origin = vo[:, 0].to(torch.int64)
    dk = k64[1:] - k64[:-1]
    assert bool((dk >= 0).all()), "keys are not sorted"
    assert bool(((dk > 0) | (origin[1:] > origin[:-1])).all()), "equal keys lost their input order"
    assert bool((k64_in[origin] == k64).all()), "payload does not point at the key's origin"
    if payload_words == 4:
        assert bool((vo[:, 1] == (vo[:, 0] ^ 0x5A5A5A5A)).all()) and bool((vo[:, 2] == ko[:, 0]).all()) \
            and bool((vo[:, 3] == ko[:, 1]).all()), "the 16-byte payload was torn"
    plan.close()


def test_full_size_head_matches_oracle(ctx, mrs, oracle):
    """The first 2^20 outputs of a 2^26 sort equal the oracle's, bit for bit (the oracle sorts
    the whole 2^26 in seconds)."""
    from gpu_util import to_u32

    n = 1 << 26
    xyz = ctx.generate_points(11, n)
    keys, idx = ctx.morton_encode30(xyz, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    hk, hi = to_u32(keys).copy(), to_u32(idx).copy()
    del xyz
    plan = mrs.SortPlan(ctx, n)
    kb, vb = torch.empty_like(keys), torch.empty_like(idx)
    plan.sort_pairs(keys, idx, kb, vb)
    plan.check()
    order = np.argsort(hk, kind="stable")
    assert np.array_equal(to_u32(keys), hk[order])
    assert np.array_equal(to_u32(idx), hi[order])
    rk, rv, _ = oracle.lsd_sort(hk[: 1 << 22], hi[: 1 << 22], 1 << 22, 1, 1, 8, 0, 32, 256)
    assert np.array_equal(rk, hk[: 1 << 22][np.argsort(hk[: 1 << 22], kind="stable")])


def _props(keys_in, idx_in, ok, ov):
    """Size-independent checks on the device: sorted, ties in payload order, a permutation whose
    gathered input keys equal the output keys."""
    k = ok.to(torch.int64) & 0xFFFFFFFF
    v = ov.to(torch.int64) & 0xFFFFFFFF
    dk = k[1:] - k[:-1]
    assert bool((dk >= 0).all()), "not sorted"
    assert bool(((dk > 0) | (v[1:] > v[:-1])).all()), "ties out of order"
    assert bool((keys_in[v] == ok).all()), "payload does not point at its key"
    assert int(v.sum()) == ok.numel() * (ok.numel() - 1) // 2, "payload is not a permutation"


@pytest.mark.parametrize("dist_name,log2n", [("morton30", 28), ("all_equal", 28), ("one_hot_bucket", 28), ("morton30", 29)])
def test_sharded_sort_one_rccl_rank_full_size(ctx, mrs, dist_name, log2n):
    """mrs_dist_sort_pairs over RCCL with one rank at the per-GPU size of the 8-GPU configuration
    (2^28 pairs): ncclAllReduce / ncclAllGather of the histogram, grouped self send/receive, local sort.
    At 2^29 pairs each array of the transfer holds 2 GiB: a single ncclSend of that size lost data, so
    the transport cuts transfers into operations of at most 1 GiB."""
    from wgpu_morton_code_radix_sort_test_amd.dist import Communicator, DistributedSorter

    n = 1 << log2n
    if dist_name == "morton30":
        xyz = ctx.generate_points(5, n)
        keys, idx = ctx.morton_encode30(xyz, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
        del xyz
    else:
        keys = ctx.generate_u32(3, 1, n)
        if dist_name == "all_equal":
            keys.fill_(0x1234567)
        else:  # every key inside ONE of the 4096 top-digit buckets (bits 29..18 fixed)
            keys &= 0x3FFFF
            keys |= 0x3A500000
        idx = torch.arange(n, dtype=torch.int32, device=ctx.device)
    comm = Communicator.rccl(0, 1)
    sorter = DistributedSorter(ctx, comm, n, end_bit=30)
    ok, ov = sorter.sort(keys, idx)
    sorter.check()
    assert sorter.n_out == n
    _props(keys, idx, ok, ov)
    sorter.close()
    comm.close()


@pytest.mark.parametrize("world,log2n,dist_name", [(2, 26, "morton30"), (2, 26, "all_equal"), (2, 26, "one_hot_bucket"),
                                                   (8, 24, "morton30"), (8, 24, "one_hot_bucket"),
                                                   (2, 27, "morton30"), (2, 27, "all_equal"), (2, 27, "one_hot_bucket")])
def test_sharded_sort_ranks_on_one_gpu_large(ctx, mrs, world, log2n, dist_name):
    """In-process ranks share the GPU, two of 2^26 pairs each and eight of 2^24 (the rank count of
    the 8-GPU configuration: 16 destination pieces in the partition pass): the partition pass, the
    splitter's deeper levels (skewed inputs) and the exchange layout at sizes where every kernel
    runs many tiles.  Checked through the properties of a global stable sort."""
    import threading

    from wgpu_morton_code_radix_sort_test_amd.dist import Communicator, DistributedSorter

    n = 1 << log2n
    comms = Communicator.local(world)
    outs, ins, errors = [None] * world, [None] * world, []

    def body(r):
        try:
            torch.cuda.set_device(0)
            c = mrs.Context(0)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                if dist_name == "morton30":
                    xyz = c.generate_points(5 + r, n, stream=stream)
                    keys, _ = c.morton_encode30(xyz, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], stream=stream)
                    del xyz
                else:
                    keys = c.generate_u32(3 + r, 1, n, stream=stream)
                    if dist_name == "all_equal":
                        keys.fill_(0x1234567)
                    else:
                        keys &= 0x3FFFF
                        keys |= 0x3A500000
                idx = torch.arange(r * n, (r + 1) * n, dtype=torch.int32, device=c.device)
                sorter = DistributedSorter(c, comms[r], n, end_bit=30)
                ok, ov = sorter.sort(keys, idx, stream=stream)
                sorter.check(stream=stream)
                stream.synchronize()
                outs[r] = (ok.clone(), ov.clone(), sorter.info())
                ins[r] = keys
                sorter.close()
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    [t.start() for t in threads]
    [t.join(600) for t in threads]
    assert not any(t.is_alive() for t in threads), "a rank thread is still inside the library after 600 s"
    [c.close() for c in comms]
    if errors:
        raise errors[0]
    all_in = torch.cat(ins)
    ok = torch.cat([o[0] for o in outs])
    ov = torch.cat([o[1] for o in outs])
    assert ok.numel() == world * n
    _props(all_in, None, ok, ov)
    tol = n // 256
    for o in outs:
        assert abs(o[0].numel() - n) <= 2 * tol + 1
    if dist_name == "all_equal":
        # each rank receives its share in two pieces: 2 * world - 1 cuts, all inside the run of equal keys
        assert all(o[2].last_exact_cuts == 2 * world - 1 for o in outs) and outs[0][0].numel() == n
    if dist_name == "one_hot_bucket":
        assert all(o[2].last_levels >= 2 for o in outs)
    if log2n >= 27:
        # pieces of 2^26 pairs: the local sorts are offered the three-move schedule over the piece's
        # window of the key space; spread keys take it, equal keys make the device refuse
        # (a piece between two equal cut keys is not even offered it: 0; one next to the end of the key
        # space has a wide window, is offered it, and the device refuses: 2)
        want = {"morton30": (1,), "all_equal": (0, 2)}.get(dist_name)
        if want is not None:
            assert all(int(o[2].last_sort_schedule[c]) in want for o in outs for c in range(2)), \
                [list(o[2].last_sort_schedule) for o in outs]


def test_first_sorts_of_in_process_ranks_never_starve(mrs):
    """Eight in-process ranks leave their host-side collectives together, so their partition passes
    and local sorts begin within microseconds of each other on one device.  Only one sort per
    device may deal its tiles per die at a time (two can each hold the compute units the other
    waits for); a second call that begins while the first is still being enqueued must see it.
    When it did not, about one first sort in 40 of fresh handles ended in the look-back time-out,
    which the handle's check reports.  60 fresh sets of handles, one sort each, every key inside
    one bucket of the first split."""
    import threading

    from wgpu_morton_code_radix_sort_test_amd.dist import Communicator, DistributedSorter

    world, n = 8, 1 << 24
    for rep in range(60):
        comms = Communicator.local(world)
        errors, got = [], [0] * world

        def body(r):
            try:
                torch.cuda.set_device(0)
                c = mrs.Context(0)
                stream = torch.cuda.Stream()
                with torch.cuda.stream(stream):
                    keys = c.generate_u32(3 + r + 10 * rep, 1, n, stream=stream)
                    keys &= 0x3FFFF
                    keys |= 0x3A500000
                    idx = torch.arange(r * n, (r + 1) * n, dtype=torch.int32, device=c.device)
                    sorter = DistributedSorter(c, comms[r], n, end_bit=30)
                    ok, _ = sorter.sort(keys, idx, stream=stream)
                    sorter.check(stream=stream)
                    k64 = ok.to(torch.int64)
                    assert bool((k64[1:] >= k64[:-1]).all())
                    got[r] = int(ok.numel())
                    sorter.close()
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)

        threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
        [t.start() for t in threads]
        [t.join(600) for t in threads]
        assert not any(t.is_alive() for t in threads), "a rank thread is still inside the library after 600 s"
        [c.close() for c in comms]
        if errors:
            raise AssertionError(f"fresh set {rep}: {errors[0]}")
        assert sum(got) == world * n
