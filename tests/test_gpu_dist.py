"""The sharded sort through the C ABI (mrs_comm_*, mrs_dist_*) on one GPU.

Several ranks are played by host threads of this process that share the GPU (the in-process
transport of the library: same histograms, splitter, partition pass, exchange layout and local
sort as over RCCL, with device copies in place of ncclSend/ncclRecv), and compared bit for bit
with one stable sort of the concatenated input by the CPU checker.  RCCL itself runs with one
rank (a one-GPU box cannot host two RCCL ranks)."""
import threading

import numpy as np
import pytest
import torch

from gpu_util import to_dev, to_u32
from inputs import make_keys

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mrs():
    import wgpu_morton_code_radix_sort_test_amd as m

    return m


def rank_keys(oracle, n, key_words, dist, seed, end_bit):
    if n == 0:
        return np.zeros(0, dtype=np.uint32)
    if dist == "all_equal":
        k = np.full(n * key_words, 0x2A5A5A5 & ((1 << min(end_bit, 32)) - 1), dtype=np.uint32)
    elif dist == "hot_bucket":  # 90 % of the keys inside one top-digit bucket
        k = make_keys(oracle, n, key_words, "random", seed=seed)
        hot = oracle.gen_u32(seed + 1, 5, n) % 10 != 0
        lo = k.reshape(n, key_words)
        top = lo[:, key_words - 1]
        width = end_bit - 32 * (key_words - 1)
        bucket_shift = max(width - 12, 0)
        top[hot] = (np.uint32(0x5A5) << np.uint32(bucket_shift)) | (top[hot] & np.uint32((1 << bucket_shift) - 1))
        k = lo.reshape(-1)
    elif dist == "few_values":
        vals = oracle.gen_u32(seed + 2, 6, 4 * key_words).reshape(4, key_words)
        pick = oracle.gen_u32(seed + 3, 6, n) % 4
        k = vals[pick].reshape(-1)
    else:
        k = make_keys(oracle, n, key_words, dist, seed=seed)
    # keys live below end_bit
    k = k.reshape(n, key_words).copy()
    width = end_bit - 32 * (key_words - 1)
    if width < 32:
        k[:, key_words - 1] &= np.uint32((1 << width) - 1)
    return k.reshape(-1)


def run_ranks(mrs, world, sizes, host_keys, key_words, end_bit, tolerance, rounds=1, chunks=0, hybrid_min_n=0,
              big_local=False):
    """One thread per rank on GPU 0; returns per-rank (keys, payload) numpy arrays and infos."""
    from wgpu_morton_code_radix_sort_test_amd.dist import Communicator, DistributedSorter

    comms = Communicator.local(world)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    results, infos, errors = [None] * world, [None] * world, []

    def body(r):
        try:
            torch.cuda.set_device(0)
            ctx = mrs.Context(0)
            ctx.set_attribute(ctx.ATTR_TEST_HYBRID_MIN_N, hybrid_min_n)  # read by the plans the handle creates
            if big_local:
                ctx.set_attribute(ctx.ATTR_TEST_HYBRID_BIG_LOCAL, 1)
            stream = torch.cuda.Stream()
            n = sizes[r]
            with torch.cuda.stream(stream):
                keys = to_dev(host_keys[r], ctx.device) if n else torch.zeros(key_words, dtype=torch.int32, device=ctx.device)
                vals = (torch.arange(n, dtype=torch.int64, device=ctx.device) + int(offsets[r])).to(torch.int32)
                sorter = DistributedSorter(ctx, comms[r], max(max(sizes), 1), key_words=key_words, payload_words=1,
                                           end_bit=end_bit, tolerance=tolerance, exchange_chunks=chunks)
                for _ in range(rounds):
                    ok, ov = sorter.sort(keys, vals, n=n, stream=stream)
                sorter.check(stream=stream)
                stream.synchronize()
                results[r] = (to_u32(ok).copy(), to_u32(ov).copy())
                infos[r] = sorter.info()
                # the input slice is left untouched
                if n:
                    assert np.array_equal(to_u32(keys), host_keys[r])
                sorter.close()
        except Exception as exc:  # noqa: BLE001 - reported by the main thread
            errors.append((r, exc))

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    join_all(threads)
    for c in comms:
        c.close()
    if errors:
        raise errors[0][1]
    return results, infos


def join_all(threads, seconds=300):
    """A rank that never comes back is a failure to report, not a run to wait out."""
    for t in threads:
        t.join(seconds)
    stuck = [t.name for t in threads if t.is_alive()]
    assert not stuck, f"rank threads still inside the library after {seconds} s: {stuck}"


def check_against_oracle(oracle, results, host_keys, sizes, key_words, end_bit, tolerance):
    total = int(sum(sizes))
    all_k = np.concatenate([k for k in host_keys if k.size]) if total else np.zeros(0, dtype=np.uint32)
    all_v = np.arange(total, dtype=np.uint32)
    rk, rv = oracle.stable_sort_pairs(all_k, all_v, total, key_words, 1, 0, end_bit)
    got_k = np.concatenate([r[0] for r in results])
    got_v = np.concatenate([r[1] for r in results])
    assert got_v.size == total
    assert np.array_equal(got_k, rk), "keys differ from one stable sort of the concatenated input"
    assert np.array_equal(got_v, rv), "payload differs: the sharded sort is not the global stable sort"
    world = len(sizes)
    for r, res in enumerate(results):
        share = total * (r + 1) // world - total * r // world
        assert res[1].size <= share + 2 * tolerance, (r, res[1].size, share)


CASES = [
    # world, key_words, end_bit, distribution, tolerance (None = default), pieces per rank (0 = library's choice)
    (2, 1, 30, "morton30", None, 0),
    (2, 1, 30, "morton30", None, 1),
    (3, 1, 32, "random", None, 0),
    (8, 1, 30, "morton30", None, 0),
    (8, 1, 32, "all_equal", None, 0),
    (8, 1, 32, "all_equal", None, 1),
    (4, 1, 30, "all_equal", 1, 2),
    (8, 1, 32, "hot_bucket", None, 0),
    (8, 1, 32, "hot_bucket", None, 1),
    (5, 1, 32, "few_values", None, 0),
    (4, 1, 32, "sorted", None, 2),
    (3, 1, 32, "random", 1, 0),
    (4, 2, 64, "random", None, 0),
    (4, 2, 64, "all_equal", None, 0),
    (3, 2, 50, "hot_bucket", None, 1),
    (16, 1, 32, "few_values", None, 0),
    (1, 1, 32, "random", None, 2),
]


@pytest.mark.parametrize("world,key_words,end_bit,dist,tolerance,chunks", CASES)
def test_local_ranks_match_one_stable_sort(oracle, mrs, world, key_words, end_bit, dist, tolerance, chunks):
    rng = np.random.default_rng(world * 7 + end_bit)
    sizes = [int(rng.integers(20000, 90000)) for _ in range(world)]
    if world >= 3:
        sizes[1] = 0       # a rank with nothing to contribute
        sizes[2] = 4097    # and one just above the one-workgroup sort
    host_keys = [rank_keys(oracle, n, key_words, dist, seed=31 * r + 1, end_bit=end_bit) for r, n in enumerate(sizes)]
    # tolerance 0 in the C ABI selects the default (n_local_max / 256); 1 forces every cut down to the key
    tol = max(sizes) // 256 if tolerance is None else tolerance
    results, infos = run_ranks(mrs, world, sizes, host_keys, key_words, end_bit, 0 if tolerance is None else tolerance,
                               chunks=chunks)
    check_against_oracle(oracle, results, host_keys, sizes, key_words, end_bit, tol)
    pieces = chunks or (2 if 2 <= world <= 8 else 1)
    if dist == "all_equal" and world > 1:
        # every cut falls inside the one run of equal keys: all levels, all cuts by element order
        assert all(i.last_exact_cuts == world * pieces - 1 for i in infos)
        total = sum(sizes)
        assert [r[1].size for r in results] == [total * (r + 1) // world - total * r // world for r in range(world)]
    if dist == "morton30" and tolerance is None and world <= 3:
        assert all(i.last_levels == 1 and i.last_exact_cuts == 0 for i in infos)


WINDOW_CASES = [
    # world, end_bit, distribution, pieces per rank, schedule every local sort must report (None: any)
    (2, 30, "morton30", 0, "msd"),
    (2, 30, "morton30", 1, "msd"),
    (3, 32, "random", 0, None),      # one rank is empty, one tiny: their pieces differ
    (8, 30, "morton30", 0, "msd"),
    (4, 32, "sorted", 2, None),
    (8, 32, "all_equal", 0, None),       # a piece of equal keys fits one sub-bucket at this size
    (8, 32, "hot_bucket", 0, None),
    (5, 32, "few_values", 0, None),
    (16, 32, "random", 0, "msd"),
    (1, 32, "random", 0, "msd"),
]


@pytest.mark.parametrize("world,end_bit,dist,chunks,want", WINDOW_CASES)
def test_local_sorts_take_the_three_move_schedule(oracle, mrs, world, end_bit, dist, chunks, want):
    """The local sorts of (u32 key, u32 payload) slices are MRS_ALGO_HYBRID plans whose sub-buckets
    divide the slice's window of the key space (the cut keys on either side).  The library keeps
    pieces below 2^26 pairs on the ordinary passes, so the test lowers that threshold; the result
    must be the one stable sort of the concatenated input whatever the device decides per piece."""
    rng = np.random.default_rng(world * 11 + end_bit)
    sizes = [int(rng.integers(30000, 120000)) for _ in range(world)]
    if world == 3:
        sizes[1] = 0
        sizes[2] = 4097
    host_keys = [rank_keys(oracle, n, 1, dist, seed=17 * r + 3, end_bit=end_bit) for r, n in enumerate(sizes)]
    results, infos = run_ranks(mrs, world, sizes, host_keys, 1, end_bit, 0, chunks=chunks, hybrid_min_n=1)
    check_against_oracle(oracle, results, host_keys, sizes, 1, end_bit, max(sizes) // 256)
    pieces = chunks or (2 if 2 <= world <= 8 else 1)
    seen = {int(i.last_sort_schedule[c]) for i in infos for c in range(pieces)}
    if want == "msd":
        assert seen == {1}, seen


@pytest.mark.parametrize("world,end_bit,dist,chunks", [(2, 30, "morton30", 0), (2, 30, "morton30", 1), (4, 32, "random", 2),
                                                       (8, 32, "hot_bucket", 0), (3, 28, "few_values", 0)])
def test_local_sorts_with_workgroups_of_1024_threads(oracle, mrs, world, end_bit, dist, chunks):
    """Pieces of about 2^29 pairs (a quarter of 2^31) whose window is a little wider than a quarter of
    the key space get sub-buckets of about 16384 pairs, which only the local sort's workgroups of 1024
    threads hold; the test asks for those workgroups at a size the checker can follow."""
    rng = np.random.default_rng(world * 13 + end_bit)
    sizes = [int(rng.integers(60000, 200000)) for _ in range(world)]
    if world == 3:
        sizes[1] = 0
    host_keys = [rank_keys(oracle, n, 1, dist, seed=19 * r + 5, end_bit=end_bit) for r, n in enumerate(sizes)]
    results, infos = run_ranks(mrs, world, sizes, host_keys, 1, end_bit, 0, chunks=chunks, hybrid_min_n=1, big_local=True)
    check_against_oracle(oracle, results, host_keys, sizes, 1, end_bit, max(sizes) // 256)
    if dist == "morton30":
        pieces = chunks or 2
        assert {int(i.last_sort_schedule[c]) for i in infos for c in range(pieces)} == {1}


def test_local_ranks_fuzz(oracle, mrs):
    """Seeded random shapes of the sharded sort: rank count, slice sizes (empty and tiny slices
    included), key width and live bits, distribution, tolerance and pieces per rank drawn together.
    MRS_DIST_FUZZ_CASES / MRS_DIST_FUZZ_SEED widen a run; MRS_DIST_FUZZ_HYBRID_MIN_N=1 lets the local sorts of
    these small pieces take the three-move schedule, MRS_DIST_FUZZ_BIG_LOCAL=1 with the local sort's workgroups of
    1024 threads."""
    import os

    rng = np.random.default_rng(int(os.environ.get("MRS_DIST_FUZZ_SEED", "5")))
    dists = ["random", "morton30", "all_equal", "hot_bucket", "few_values", "sorted", "few_distinct", "one_hot_digit"]
    for case in range(int(os.environ.get("MRS_DIST_FUZZ_CASES", "10"))):
        world = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 11, 16]))
        key_words = int(rng.choice([1, 1, 2]))
        dist = dists[int(rng.integers(len(dists)))]
        end_bit = 30 if dist == "morton30" else int(rng.integers(32 * (key_words - 1) + 13, 32 * key_words + 1))
        if dist == "morton30":
            key_words = 1
        sizes = [int(rng.choice([0, 1, int(rng.integers(2, 300)), int(rng.integers(300, 6000)), int(rng.integers(6000, 60000))],
                                p=[0.08, 0.04, 0.13, 0.25, 0.5])) for _ in range(world)]
        chunks = int(rng.choice([0, 1, 2])) if 2 * world <= 16 else int(rng.choice([0, 1]))
        tolerance = [None, None, 1, int(rng.integers(1, 500))][int(rng.integers(4))]
        host_keys = [rank_keys(oracle, n, key_words, dist, seed=1000 * case + r, end_bit=end_bit) for r, n in enumerate(sizes)]
        tol = max(max(sizes), 1) // 256 if tolerance is None else tolerance
        what = (case, world, key_words, end_bit, dist, sizes, chunks, tolerance)
        try:
            results, _ = run_ranks(mrs, world, sizes, host_keys, key_words, end_bit, 0 if tolerance is None else tolerance,
                                   chunks=chunks, hybrid_min_n=int(os.environ.get("MRS_DIST_FUZZ_HYBRID_MIN_N", "0")),
                                   big_local=os.environ.get("MRS_DIST_FUZZ_BIG_LOCAL") == "1")
            check_against_oracle(oracle, results, host_keys, sizes, key_words, end_bit, max(tol, 1))
        except AssertionError as exc:
            raise AssertionError(f"{what}: {exc}") from exc
        except Exception as exc:  # noqa: BLE001
            raise RuntimeError(f"{what}: {exc}") from exc


def test_repeated_calls_and_shrinking_input(oracle, mrs):
    """The same handles sort twice in a row, the second time a smaller input (stale status words,
    tables and counts of the first call must not leak into the second)."""
    from wgpu_morton_code_radix_sort_test_amd.dist import Communicator, DistributedSorter

    world = 4
    comms = Communicator.local(world)
    rounds = [[70000, 50000, 81234, 60001], [3000, 0, 41000, 17]]
    host = [[rank_keys(oracle, n, 1, "hot_bucket" if j else "morton30", seed=100 * j + r, end_bit=30) for r, n in enumerate(sizes)]
            for j, sizes in enumerate(rounds)]
    outs = [[None] * world for _ in rounds]
    errors = []

    def body(r):
        try:
            torch.cuda.set_device(0)
            ctx = mrs.Context(0)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                sorter = DistributedSorter(ctx, comms[r], 90000, end_bit=30)
                for j, sizes in enumerate(rounds):
                    n = sizes[r]
                    off = int(sum(sizes[:r]))
                    keys = to_dev(host[j][r], ctx.device) if n else torch.zeros(1, dtype=torch.int32, device=ctx.device)
                    vals = (torch.arange(n, dtype=torch.int64, device=ctx.device) + off).to(torch.int32)
                    ok, ov = sorter.sort(keys, vals, n=n, stream=stream)
                    sorter.check(stream=stream)
                    outs[j][r] = (to_u32(ok).copy(), to_u32(ov).copy())
                sorter.close()
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    [t.start() for t in threads]
    join_all(threads)
    [c.close() for c in comms]
    if errors:
        raise errors[0]
    for j, sizes in enumerate(rounds):
        check_against_oracle(oracle, outs[j], host[j], sizes, 1, 30, 90000 // 256)


def test_rccl_single_rank(ctx, oracle, mrs):
    """The RCCL transport with a communicator of one rank: ncclAllReduce, ncclAllGather and the
    grouped self send/receive run for real; the result is the plain stable sort."""
    from wgpu_morton_code_radix_sort_test_amd.dist import Communicator, DistributedSorter

    comm = Communicator.rccl(0, 1)
    n = 300007
    for dist, end_bit in (("morton30", 30), ("all_equal", 32), ("random", 32)):
        hk = rank_keys(oracle, n, 1, dist, seed=3, end_bit=end_bit)
        hv = np.arange(n, dtype=np.uint32)
        sorter = DistributedSorter(ctx, comm, n, end_bit=end_bit)
        ok, ov = sorter.sort(to_dev(hk, ctx.device), to_dev(hv, ctx.device))
        sorter.check()
        rk, rv = oracle.stable_sort_pairs(hk, hv, n, 1, 1, 0, end_bit)
        assert sorter.n_out == n
        assert np.array_equal(to_u32(ok), rk) and np.array_equal(to_u32(ov), rv)
        sorter.close()
    comm.close()


def test_dist_argument_errors(ctx, mrs):
    from wgpu_morton_code_radix_sort_test_amd.dist import Communicator, DistributedSorter

    comm = Communicator.local(1)[0]
    with pytest.raises(mrs.MrsError):
        DistributedSorter(ctx, comm, 1000, key_words=3)
    with pytest.raises(mrs.MrsError):
        DistributedSorter(ctx, comm, 1000, end_bit=40)
    sorter = DistributedSorter(ctx, comm, 1000)
    keys = torch.zeros(2000, dtype=torch.int32, device=ctx.device)
    with pytest.raises(mrs.MrsError) as e:
        sorter.sort(keys, keys.clone(), n=2000)
    assert "n_local_max" in str(e.value)
    sorter.close()
    comm.close()


@pytest.mark.parametrize("world,bad_rank", [(2, 0), (3, 1), (8, 7)])
def test_a_failing_rank_releases_its_peers(mrs, world, bad_rank):
    """One rank of an in-process group leaves mrs_dist_sort_pairs with an error (here: more pairs
    than its handle was created for, found before the first collective).  The other ranks, which
    are waiting for it at the host barrier of the first collective, must come back with
    MRS_ERR_PEER_FAILED instead of waiting for ever, and so must every later call on the group."""
    from wgpu_morton_code_radix_sort_test_amd import _lib
    from wgpu_morton_code_radix_sort_test_amd.dist import Communicator, DistributedSorter

    comms = Communicator.local(world)
    n_max = 20000
    codes, second = [None] * world, [None] * world

    def body(r):
        torch.cuda.set_device(0)
        ctx = mrs.Context(0)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            n = n_max + 1 if r == bad_rank else n_max
            keys = ctx.generate_u32(r, 1, n, stream=stream)
            vals = torch.arange(n, dtype=torch.int32, device=ctx.device)
            sorter = DistributedSorter(ctx, comms[r], n_max)
            for slot, count in ((codes, n), (second, n_max)):
                try:
                    sorter.sort(keys, vals, n=count, stream=stream)
                    slot[r] = 0
                except mrs.MrsError as exc:
                    slot[r] = exc.code
            stream.synchronize()
            sorter.close()

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    join_all(threads, 120)
    for c in comms:
        c.close()
    for r in range(world):
        assert codes[r] == (_lib.MRS_ERR_INVALID_ARGUMENT if r == bad_rank else _lib.MRS_ERR_PEER_FAILED), (r, codes)
        # the group stays dead: a correct call afterwards fails at once on every rank
        assert second[r] == _lib.MRS_ERR_PEER_FAILED, (r, second)
