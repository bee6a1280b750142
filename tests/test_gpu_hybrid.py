"""GPU parity tests of MRS_ALGO_HYBRID plans: the most-significant-digit-first schedule and the
ordinary passes the device falls back to, through the C ABI, bit for bit against the CPU checker's
stable sort (a stable sort has one answer, whatever the order of its passes)."""
import os

import numpy as np
import pytest
import torch

from gpu_util import gpu_sort, to_dev, to_u32
from inputs import DISTRIBUTIONS, make_keys, make_vals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mrs():
    import wgpu_morton_code_radix_sort_test_amd as m

    return m


@pytest.fixture()
def small_inputs_take_the_schedule(ctx, aids):
    """The library keeps inputs below 2^26 pairs on the ordinary passes; the checker is too slow
    for more, so the tests lower the threshold (a plan reads it from its context at creation)."""
    aids(ctx.ATTR_TEST_HYBRID_MIN_N, 1)
    # this file is about the fixed sub-buckets and their refusal (tests/test_gpu_hybrid_adaptive.py:
    # the cut that follows the keys)
    aids(ctx.ATTR_TEST_HYBRID_NO_ADAPTIVE, 1)


def _sort_with_plan(ctx, mrs, keys, vals, n, begin, end, capacity=None, tile_items=0, indexed=False, oversized=None, twice=False):
    """One sort through an MRS_ALGO_HYBRID plan.  `oversized`: a list that receives what
    mrs_plan_get_oversized reports; `twice`: the plan sorts the same input a second time (what the
    first call left in the plan's tables must not leak into the second)."""
    plan = mrs.SortPlan(ctx, capacity or n, key_words=1, payload_bytes=4, digit_bits=8, begin_bit=begin, end_bit=end,
                        algo=mrs.ALGO_HYBRID, tile_items=tile_items)
    ka, va = to_dev(keys, ctx.device), to_dev(vals, ctx.device)
    if indexed:
        va.fill_(0x3C3C3C3C)
    kb, vb = torch.full_like(ka, -1), torch.full_like(va, -1)
    in_b = plan.sort_pairs_indexed(ka, va, kb, vb, n=n) if indexed else plan.sort_pairs(ka, va, kb, vb, n=n)
    plan.check()
    if twice:
        ka.copy_(to_dev(keys, ctx.device))
        if not indexed:
            va.copy_(to_dev(vals, ctx.device))
        in_b = plan.sort_pairs_indexed(ka, va, kb, vb, n=n) if indexed else plan.sort_pairs(ka, va, kb, vb, n=n)
        plan.check()
    schedule, largest = plan.schedule()
    if oversized is not None:
        oversized.extend(plan.oversized())
    ko, vo = to_u32(kb if in_b else ka)[:n], to_u32(vb if in_b else va)[:n]
    plan.close()
    return ko, vo, in_b, schedule, largest


# sizes around the tile (4096 and 32768 keys) and workgroup (512) boundaries of the three kernels
SIZES = [4097, 4608, 5000, 8191, 8192, 8193, 32767, 32768, 32769, 65536, 100003, 262144 + 511, (1 << 20) + 123]


@pytest.mark.parametrize("n", SIZES)
def test_schedule_matches_oracle(ctx, oracle, mrs, small_inputs_take_the_schedule, n):
    """30-bit Morton keys of uniform points and uniform 32-bit keys: the device takes the new schedule."""
    for dist, end in (("morton30", 30), ("random", 32)):
        keys = make_keys(oracle, n, 1, dist, seed=n)
        vals = make_vals(oracle, n, 1)
        ko, vo, in_b, schedule, largest = _sort_with_plan(ctx, mrs, keys, vals, n, 0, end)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, end)
        assert schedule == mrs.SortPlan.SCHEDULE_MSD, (dist, largest)
        assert not in_b  # four passes' side
        assert np.array_equal(ko, rk), (n, dist, "keys")
        assert np.array_equal(vo, rv), (n, dist, "payload")


@pytest.mark.parametrize("dist", DISTRIBUTIONS)
def test_every_distribution_either_schedule(ctx, oracle, mrs, small_inputs_take_the_schedule, dist):
    """Whatever the device decides (equal keys, one live digit and few distinct keys overflow a
    sub-bucket and make it refuse), the result is the stable sort."""
    n = 300007
    keys = make_keys(oracle, n, 1, dist)
    vals = make_vals(oracle, n, 1)
    ko, vo, in_b, schedule, largest = _sort_with_plan(ctx, mrs, keys, vals, n, 0, 32)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
    assert schedule in (mrs.SortPlan.SCHEDULE_MSD, mrs.SortPlan.SCHEDULE_MSD_REFUSED)
    if dist == "all_equal":  # one sub-bucket holds everything: more than the quarter the extra passes may take
        assert schedule == mrs.SortPlan.SCHEDULE_MSD_REFUSED and largest == n, (dist, largest)
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv), dist
    # the reference-structured pass loop of the checker agrees (src/main.rs:594-665 restated)
    lk, lv, _ = oracle.lsd_sort(keys, vals, n, 1, 1, 8, 0, 32, 256)
    assert np.array_equal(ko, lk) and np.array_equal(vo, lv)


def _keys_with_sub_buckets(oracle, n, shift, sizes, low_kinds, begin=0, end=32, seed=3):
    """n keys, evenly spread except that sub-bucket ids 0x2BCD, 0x2BCE, ... (the value of the bits
    [shift, end)) hold exactly sizes[j] keys each, whose lower bits are random ("random"), all the
    same ("equal") or drawn from five values ("few").  Shuffled."""
    rnd = oracle.gen_u32(seed, 9, n)
    keys = rnd.copy()
    width = end - shift
    ids = [(0x2BCD + j) & ((1 << width) - 1) for j in range(len(sizes))]
    field = np.uint32(((1 << width) - 1) << shift)
    low_mask = np.uint32((1 << shift) - 1)
    # keep the evenly spread rest out of the chosen sub-buckets
    taken = np.isin((keys & field) >> np.uint32(shift), np.array(ids, dtype=np.uint32))
    keys[taken] ^= np.uint32(1 << (end - 1))
    at = 0
    for j, (size, kind) in enumerate(zip(sizes, low_kinds)):
        low = rnd[at:at + size] & low_mask
        if kind == "equal":
            low[:] = np.uint32(0x1A5A5) & low_mask
        elif kind == "few":
            low = (np.array([1, 77, 300, 4097, 65535], dtype=np.uint32) & low_mask)[rnd[at:at + size] % 5]
        keys[at:at + size] = (keys[at:at + size] & ~(field | low_mask)) | np.uint32(ids[j] << shift) | low
        at += size
    assert at <= n
    perm = np.random.RandomState(5).permutation(n)
    return np.ascontiguousarray(keys[perm])


# (begin, end, leading bits, where the sub-bucket's own bits start, pairs a workgroup holds, keys above `end` differ)
CAPACITY_CASES = [
    (0, 32, 16, 16, 10240, False),  # 16 bits below the sub-bucket's: a pair is staged in 6 bytes
    (0, 30, 16, 14, 10240, False),
    (2, 32, 14, 18, 7680, False),   # 18 bits below: 8 bytes
    (4, 31, 14, 17, 7680, True),    # 17 bits below: 8 bytes; the bit above the sorted range differs between keys and is carried along
]


def test_keys_that_differ_above_the_sorted_range(ctx, oracle, mrs, small_inputs_take_the_schedule, aids):
    """Where a pair would be staged in 6 bytes (the sub-bucket's own bits start at bit 16 or below) the
    key bits above the sorted range must be the same for every key; the device finds out while it
    counts, and takes the ordinary passes if they are not: bits outside [begin, end) are carried
    along untouched either way."""
    aids(ctx.ATTR_TEST_HYBRID_LEADING_BITS, 14)
    n = 200003
    keys = make_keys(oracle, n, 1, "random", seed=77)
    vals = make_vals(oracle, n, 1)
    for high_vary in (True, False):
        k = keys if high_vary else ((keys & np.uint32((1 << 25) - 1)) | np.uint32(0x5A000000))
        ko, vo, in_b, schedule, _ = _sort_with_plan(ctx, mrs, k, vals, n, 0, 25)
        rk, rv = oracle.stable_sort_pairs(k, vals, n, 1, 1, 0, 25)
        assert schedule == (mrs.SortPlan.SCHEDULE_MSD_REFUSED if high_vary else mrs.SortPlan.SCHEDULE_MSD) and not in_b
        assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


@pytest.mark.parametrize("begin,end,leading,shift,cap,high_vary", CAPACITY_CASES)
def test_sub_buckets_at_the_capacity(ctx, oracle, mrs, small_inputs_take_the_schedule, aids, begin, end, leading, shift, cap,
                                     high_vary):
    """A sub-bucket of exactly as many pairs as a workgroup holds is sorted inside LDS; one pair more
    and it is sorted where it lies by the two extra passes (the schedule still runs)."""
    aids(ctx.ATTR_TEST_HYBRID_LEADING_BITS, leading)
    for size in (cap, cap + 1):
        n = size + 60000
        keys = _keys_with_sub_buckets(oracle, n, shift, [size], ["random"], begin, end, seed=size)
        if not high_vary and end < 32:
            keys &= np.uint32((1 << end) - 1)
        vals = make_vals(oracle, n, 1)
        info = []
        ko, vo, in_b, schedule, largest = _sort_with_plan(ctx, mrs, keys, vals, n, begin, end, oversized=info)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, begin, end)
        assert schedule == mrs.SortPlan.SCHEDULE_MSD and not in_b
        assert largest == size
        assert info == ([0, 0, cap] if size == cap else [1, size, cap]), info
        assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


@pytest.mark.parametrize("begin,end,leading,shift,cap,high_vary", CAPACITY_CASES)
def test_mix_of_empty_normal_and_oversized_sub_buckets(ctx, oracle, mrs, small_inputs_take_the_schedule, aids, begin, end,
                                                       leading, shift, cap, high_vary):
    """Most sub-buckets empty or small, one exactly full, and four oversized ones whose lower bits are
    random, all equal, few distinct values, random again: the oversized ones (under a quarter of the
    pairs) are sorted by the two extra chained-scan passes, everything else inside LDS; twice on one
    plan, as pairs and with indices made up by the first pass."""
    aids(ctx.ATTR_TEST_HYBRID_LEADING_BITS, leading)
    sizes = [cap, cap + 1, 40000, 23456, 3 * cap + 17]
    kinds = ["random", "random", "equal", "few", "random"]
    big = sum(sizes[1:])
    n = 4 * big + cap + 12345
    keys = _keys_with_sub_buckets(oracle, n, shift, sizes, kinds, begin, end, seed=leading)
    if not high_vary and end < 32:
        keys &= np.uint32((1 << end) - 1)
    for indexed in (False, True):
        vals = np.arange(n, dtype=np.uint32) if indexed else make_vals(oracle, n, 1)
        info = []
        ko, vo, in_b, schedule, largest = _sort_with_plan(ctx, mrs, keys, vals, n, begin, end, oversized=info, twice=True,
                                                          indexed=indexed)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, begin, end)
        assert schedule == mrs.SortPlan.SCHEDULE_MSD and not in_b
        assert largest == max(sizes) and info == [4, big, cap], (largest, info)
        assert np.array_equal(ko, rk), "keys"
        assert np.array_equal(vo, rv), "payload"


@pytest.mark.parametrize("end,leading,shift", [(32, 16, 16), (30, 16, 14), (30, 15, 15)])
def test_workgroups_of_1024_threads(ctx, oracle, mrs, small_inputs_take_the_schedule, aids, end, leading, shift):
    """The local sort that inputs of 2^29.1 to 2^30 pairs take (one workgroup of 1024 threads per compute
    unit, 20480 pairs each), switched on at a size the checker can follow: sub-buckets that are
    empty, small, beyond what the 512-thread workgroups hold, exactly full, and one pair more than
    full (that one is sorted by the two extra passes); twice on one plan, also with made-up indices."""
    aids(ctx.ATTR_TEST_HYBRID_LEADING_BITS, leading)
    aids(ctx.ATTR_TEST_HYBRID_BIG_LOCAL, 1)
    cap = 20480
    sizes = [cap, cap + 1, 10241, 15000, 20479, 1, 4097]
    kinds = ["random", "random", "few", "random", "equal", "random", "random"]
    n = 6 * (cap + 1) + sum(sizes) + 54321
    keys = _keys_with_sub_buckets(oracle, n, shift, sizes, kinds, 0, end, seed=end + leading)
    if end < 32:
        keys &= np.uint32((1 << end) - 1)
    for indexed in (False, True):
        vals = np.arange(n, dtype=np.uint32) if indexed else make_vals(oracle, n, 1)
        info = []
        ko, vo, in_b, schedule, largest = _sort_with_plan(ctx, mrs, keys, vals, n, 0, end, oversized=info, twice=True,
                                                          indexed=indexed)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, end)
        assert schedule == mrs.SortPlan.SCHEDULE_MSD and not in_b
        assert largest == cap + 1 and info == [1, cap + 1, cap], (largest, info)
        assert np.array_equal(ko, rk), "keys"
        assert np.array_equal(vo, rv), "payload"


def test_one_sub_bucket_with_most_of_the_pairs_is_refused(ctx, oracle, mrs, small_inputs_take_the_schedule):
    """90 % of the pairs in one sub-bucket: the extra passes would move most pairs four times, so the
    device takes the plan's ordinary passes; the result is the same stable sort."""
    n = 500000
    keys = _keys_with_sub_buckets(oracle, n, 16, [n * 9 // 10], ["random"])
    vals = make_vals(oracle, n, 1)
    info = []
    ko, vo, in_b, schedule, largest = _sort_with_plan(ctx, mrs, keys, vals, n, 0, 32, oversized=info)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
    assert schedule == mrs.SortPlan.SCHEDULE_MSD_REFUSED and not in_b
    assert largest == n * 9 // 10 and info[:2] == [0, 0]
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


@pytest.mark.parametrize("count,want", [(256, "msd"), (257, "refused")])
def test_at_most_256_oversized_sub_buckets(ctx, oracle, mrs, small_inputs_take_the_schedule, aids, count, want):
    """The table of oversized sub-buckets has 256 entries: one more and the device takes the ordinary
    passes (8-byte staging, so that the sub-buckets are as small as an oversized one can be)."""
    aids(ctx.ATTR_TEST_HYBRID_LEADING_BITS, 14)
    size = 7681
    n = 4 * count * size + 100000
    keys = _keys_with_sub_buckets(oracle, n, 18, [size] * count, ["random"] * count, 2, 32, seed=count)
    vals = make_vals(oracle, n, 1)
    info = []
    ko, vo, in_b, schedule, largest = _sort_with_plan(ctx, mrs, keys, vals, n, 2, 32, oversized=info)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 2, 32)
    assert not in_b
    if want == "msd":
        assert schedule == mrs.SortPlan.SCHEDULE_MSD and info == [count, count * size, 7680], info
    else:
        assert schedule == mrs.SortPlan.SCHEDULE_MSD_REFUSED and info[:2] == [0, 0], info
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


@pytest.mark.parametrize("msd_bits", [14, 15, 16])
@pytest.mark.parametrize("begin,end", [(0, 25), (0, 27), (3, 29), (2, 32), (7, 32), (0, 31), (0, 30)])
def test_bit_ranges(ctx, oracle, mrs, small_inputs_take_the_schedule, aids, begin, end, msd_bits):
    """Bits outside [begin, end) are carried along untouched and never compared; every split of the
    leading bits between the two scatters (7 + 7, 8 + 7, 8 + 8; the library picks it from n)."""
    aids(ctx.ATTR_TEST_HYBRID_LEADING_BITS, msd_bits)
    n = 150001
    keys = make_keys(oracle, n, 1, "random", seed=end * 32 + begin)
    vals = make_vals(oracle, n, 1)
    ko, vo, in_b, schedule, _ = _sort_with_plan(ctx, mrs, keys, vals, n, begin, end)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, begin, end)
    # the keys are random in all 32 bits: where the local sort would stage a pair in 6 bytes (the
    # sub-bucket's own bits start at bit 16 or below) the device finds bits that differ above `end`
    # and takes the ordinary passes
    leading = max(msd_bits, end - begin - 16)
    refused = end < 32 and end - leading <= 16
    assert schedule == (mrs.SortPlan.SCHEDULE_MSD_REFUSED if refused else mrs.SortPlan.SCHEDULE_MSD) and not in_b
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)
    if refused:  # the same keys without the bits above `end`: the schedule runs
        keys = keys & np.uint32((1 << end) - 1)
        ko, vo, in_b, schedule, _ = _sort_with_plan(ctx, mrs, keys, vals, n, begin, end)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, begin, end)
        assert schedule == mrs.SortPlan.SCHEDULE_MSD and not in_b
        assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


def test_device_refusal_can_be_forced(ctx, oracle, mrs, small_inputs_take_the_schedule, aids):
    """The gated ordinary passes (persistent workgroups drawing tiles) on keys that would have taken
    the new schedule."""
    aids(ctx.ATTR_TEST_HYBRID_FORCE_FALLBACK, 1)
    for n in (4097, 70001, 300000, (1 << 20) + 5):
        keys = make_keys(oracle, n, 1, "morton30", seed=n)
        vals = make_vals(oracle, n, 1)
        ko, vo, in_b, schedule, _ = _sort_with_plan(ctx, mrs, keys, vals, n, 0, 30)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 30)
        assert schedule == mrs.SortPlan.SCHEDULE_MSD_REFUSED and not in_b
        assert np.array_equal(ko, rk) and np.array_equal(vo, rv), n


def test_indexed_sort_and_plan_reuse(ctx, oracle, mrs, small_inputs_take_the_schedule):
    """mrs_sort_pairs_indexed on a hybrid plan (pass A makes the indices up), then the same plan on
    smaller and differently distributed inputs: nothing of one call survives into the next."""
    cap = 400000
    plan = mrs.SortPlan(ctx, cap, key_words=1, payload_bytes=4, digit_bits=8, end_bit=32, algo=mrs.ALGO_HYBRID)
    for n, dist, indexed in ((400000, "random", True), (123457, "all_equal", False), (399999, "morton30", True),
                             (5000, "random", False), (3000, "random", True), (250000, "few_distinct", True),
                             (400000, "random", False)):
        keys = make_keys(oracle, n, 1, dist, seed=n)
        vals = np.arange(n, dtype=np.uint32) if indexed else make_vals(oracle, n, 1)
        ka, va = to_dev(keys, ctx.device), to_dev(vals, ctx.device)
        if indexed:
            va.fill_(0x77777777)
        kb, vb = torch.full_like(ka, -1), torch.full_like(va, -1)
        in_b = plan.sort_pairs_indexed(ka, va, kb, vb, n=n) if indexed else plan.sort_pairs(ka, va, kb, vb, n=n)
        plan.check()
        schedule, _ = plan.schedule()
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
        assert not in_b
        if n <= 4096:
            assert schedule == plan.SCHEDULE_PASSES  # the one-workgroup sort
        assert np.array_equal(to_u32(ka)[:n], rk) and np.array_equal(to_u32(va)[:n], rv), (n, dist, indexed)
    plan.close()


def test_small_inputs_stay_on_the_ordinary_passes(ctx, oracle, mrs):
    """Without the test switch an input below 2^22 pairs never sees the new schedule."""
    n = 200000
    keys = make_keys(oracle, n, 1, "random")
    vals = make_vals(oracle, n, 1)
    ko, vo, in_b, schedule, largest = _sort_with_plan(ctx, mrs, keys, vals, n, 0, 32)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
    assert schedule == mrs.SortPlan.SCHEDULE_PASSES and largest == 0 and not in_b
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


def test_misaligned_buffers(ctx, oracle, mrs, small_inputs_take_the_schedule):
    """Key and payload arrays that start 4 bytes past a 16-byte boundary."""
    n = 200003
    keys = make_keys(oracle, n, 1, "random", seed=11)
    vals = make_vals(oracle, n, 1)
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, algo=mrs.ALGO_HYBRID)
    bufs = [torch.zeros(n + 4, dtype=torch.int32, device=ctx.device) for _ in range(4)]
    ka, va, kb, vb = (b[1:n + 1] for b in bufs)
    ka.copy_(to_dev(keys, ctx.device))
    va.copy_(to_dev(vals, ctx.device))
    in_b = plan.sort_pairs(ka, va, kb, vb)
    plan.check()
    assert plan.schedule()[0] == plan.SCHEDULE_MSD and not in_b
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
    assert np.array_equal(to_u32(ka), rk) and np.array_equal(to_u32(va), rv)
    for b in bufs:  # nothing written outside the slices
        assert int(b[0]) == 0 and int(b[n + 1]) == 0
    plan.close()


def test_replays_from_a_graph(ctx, oracle, mrs, small_inputs_take_the_schedule):
    """Both schedules are enqueued by every call, so one captured call serves inputs of either kind."""
    n = 150000
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, algo=mrs.ALGO_HYBRID)
    ka = torch.zeros(n, dtype=torch.int32, device=ctx.device)
    va, kb, vb = torch.zeros_like(ka), torch.zeros_like(ka), torch.zeros_like(ka)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ka.copy_(to_dev(make_keys(oracle, n, 1, "random", seed=1), ctx.device))
        plan.sort_pairs(ka, va, kb, vb, stream=side)  # warm-up outside the capture
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        in_b = plan.sort_pairs(ka, va, kb, vb, stream=torch.cuda.current_stream())
    assert not in_b
    for dist in ("random", "all_equal", "morton30", "few_distinct", "random"):
        keys = make_keys(oracle, n, 1, dist, seed=3)
        vals = make_vals(oracle, n, 1)
        ka.copy_(to_dev(keys, ctx.device))
        va.copy_(to_dev(vals, ctx.device))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
        assert np.array_equal(to_u32(ka), rk) and np.array_equal(to_u32(va), rv), dist
    plan.check()
    del graph
    plan.close()


def test_unsupported_shapes_are_refused(ctx, mrs):
    from wgpu_morton_code_radix_sort_test_amd._lib import MrsError

    for kwargs in (dict(key_words=2), dict(payload_bytes=0), dict(payload_bytes=16), dict(digit_bits=4),
                   dict(end_bit=24), dict(begin_bit=10, end_bit=30), dict(tile_items=8192)):
        args = dict(key_words=1, payload_bytes=4, digit_bits=8, algo=mrs.ALGO_HYBRID)
        args.update(kwargs)
        with pytest.raises(MrsError):
            mrs.SortPlan(ctx, 100000, **args)
    with pytest.raises(MrsError):
        mrs.SortPlan(ctx, (1 << 30) + 1, key_words=1, payload_bytes=4, digit_bits=8, algo=mrs.ALGO_HYBRID)


def test_fuzz(ctx, oracle, mrs, small_inputs_take_the_schedule):
    """Seeded random sizes, bit ranges, distributions and payload modes on hybrid plans."""
    seed = int(os.environ.get("MRS_FUZZ_SEED", "20241112"))
    rng = np.random.RandomState(seed)
    for case in range(int(os.environ.get("MRS_FUZZ_CASES", "40"))):
        n = int(rng.choice([rng.randint(4097, 9000), rng.randint(9000, 70000), rng.randint(70000, 500000)]))
        width = int(rng.randint(25, 33))
        begin = int(rng.randint(0, 32 - width + 1))
        end = begin + width
        dist = DISTRIBUTIONS[rng.randint(len(DISTRIBUTIONS))]
        indexed = bool(rng.randint(2))
        tile = int(rng.choice([0, 4096, 32768]))
        keys = make_keys(oracle, n, 1, dist, seed=case)
        if rng.randint(3) == 0:
            # some sub-buckets near the capacity of a workgroup (7680 pairs), some beyond it
            hot = int(rng.choice([7000, 7679, 7680, 7681, 9000]))
            if hot < n:
                at = rng.choice(n, size=hot, replace=False)
                low = end - (16 if width > 30 else 14) - begin
                keys[at] = (keys[at[0]] & ~np.uint32(((1 << low) - 1) << begin)) | ((keys[at] >> np.uint32(begin) & np.uint32((1 << low) - 1)) << np.uint32(begin))
        vals = np.arange(n, dtype=np.uint32) if indexed else make_vals(oracle, n, 1)
        ko, vo, in_b, schedule, _ = _sort_with_plan(ctx, mrs, keys, vals, n, begin, end, tile_items=tile, indexed=indexed)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, begin, end)
        what = (case, n, begin, end, dist, indexed, tile, schedule)
        assert not in_b, what
        assert np.array_equal(ko, rk), what
        assert np.array_equal(vo, rv), what


def test_two_streams_at_once(oracle, mrs):
    """Two hybrid plans of two contexts sorting on two streams at the same time, every attribute at
    its default: the second call finds the first one's die-aware passes in flight and deals its
    tiles from a single list (pass B included), so neither can starve the other."""
    n = 1 << 21
    ctxs = [mrs.Context(0), mrs.Context(0)]
    for c in ctxs:  # inputs a checker can follow take the schedule; nothing else differs from the defaults
        c.set_attribute(c.ATTR_TEST_HYBRID_MIN_N, 1)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    plans, bufs, host = [], [], []
    for j in range(2):
        keys = make_keys(oracle, n, 1, "morton30" if j == 0 else "random", seed=40 + j)
        vals = make_vals(oracle, n, 1)
        host.append((keys, vals))
        plans.append(mrs.SortPlan(ctxs[j], n, key_words=1, payload_bytes=4, digit_bits=8, end_bit=30 if j == 0 else 32,
                                  algo=mrs.ALGO_HYBRID))
        ka, va = to_dev(keys, ctxs[j].device), to_dev(vals, ctxs[j].device)
        bufs.append((ka, va, torch.empty_like(ka), torch.empty_like(va)))
    torch.cuda.synchronize()
    for rounds in range(3):
        for j in range(2):
            ka, va, kb, vb = bufs[j]
            ka.copy_(to_dev(host[j][0], ctxs[j].device))
            va.copy_(to_dev(host[j][1], ctxs[j].device))
        torch.cuda.synchronize()
        for j in range(2):
            ka, va, kb, vb = bufs[j]
            assert not plans[j].sort_pairs(ka, va, kb, vb, stream=streams[j])
        for j in range(2):
            streams[j].synchronize()
            plans[j].check(stream=streams[j])
            assert plans[j].schedule(stream=streams[j])[0] == plans[j].SCHEDULE_MSD
            end = 30 if j == 0 else 32
            rk, rv = oracle.stable_sort_pairs(host[j][0], host[j][1], n, 1, 1, 0, end)
            assert np.array_equal(to_u32(bufs[j][0]), rk) and np.array_equal(to_u32(bufs[j][1]), rv), (rounds, j)
    for p in plans:
        p.close()


def test_primed_call_on_a_hybrid_plan(ctx, oracle, mrs, small_inputs_take_the_schedule):
    """mrs_morton_encode30_for_sort + mrs_sort_pairs_primed on a hybrid plan: the histograms the
    encoder left serve the four passes only, the three-move schedule takes its own and runs."""
    n = 300001
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, end_bit=30, algo=mrs.ALGO_HYBRID)
    xyz = ctx.generate_points(9, n)
    keys, idx = plan.morton_encode30_for_sort(xyz, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    hk, hv = to_u32(keys).copy(), to_u32(idx).copy()
    kb, vb = torch.empty_like(keys), torch.empty_like(idx)
    plan.set_timing(True)
    assert not plan.sort_pairs_primed(keys, idx, kb, vb)
    plan.check()
    assert plan.schedule()[0] == plan.SCHEDULE_MSD
    # one read of the keys before the passes, the schedule's own; nothing else counts digits
    labels = [lab for lab, _ in plan.timings()]
    assert labels.count("msd_histogram") == 1 and not any("global_histogram" in lab or lab == "fallback_histogram" for lab in labels), labels
    rk, rv = oracle.stable_sort_pairs(hk, hv, n, 1, 1, 0, 30)
    assert np.array_equal(to_u32(keys), rk) and np.array_equal(to_u32(idx), rv)
    plan.close()


@pytest.mark.parametrize("tile,log2n", [(4096, 20), (32768, 23)])
def test_buckets_of_exactly_one_full_tile(ctx, oracle, mrs, small_inputs_take_the_schedule, aids, tile, log2n):
    """Every first-level bucket holds exactly one full tile of pass B (no partial tile anywhere, every
    tile the first and only one of its chain), in shuffled and in sorted input order."""
    aids(ctx.ATTR_TEST_HYBRID_LEADING_BITS, 16)
    n = 1 << log2n
    assert n // 256 == tile
    base = (np.arange(n, dtype=np.uint64) << np.uint64(32 - log2n)).astype(np.uint32)  # n distinct keys, evenly spaced
    for order in ("sorted", "shuffled"):
        keys = base if order == "sorted" else np.ascontiguousarray(base[np.random.RandomState(1).permutation(n)])
        vals = make_vals(oracle, n, 1)
        ko, vo, in_b, schedule, largest = _sort_with_plan(ctx, mrs, keys, vals, n, 0, 32, tile_items=tile)
        assert schedule == mrs.SortPlan.SCHEDULE_MSD and largest == n >> 16 and not in_b
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
        assert np.array_equal(ko, rk) and np.array_equal(vo, rv), order


def test_empty_buckets_between_full_ones(ctx, oracle, mrs, small_inputs_take_the_schedule):
    """Only 3 of the 256 first-level buckets hold keys (one of them the last), and inside them only
    every other sub-bucket: pass B's tile table has runs of empty buckets, the local sort empty visits."""
    n = 250001
    r = oracle.gen_u32(21, 4, n)
    top = np.array([3, 200, 255], dtype=np.uint32)[oracle.gen_u32(21, 5, n) % 3]
    keys = (top << np.uint32(24)) | (r & np.uint32(0x00FEFFFF))  # bit 16 clear: odd sub-buckets stay empty
    vals = make_vals(oracle, n, 1)
    ko, vo, in_b, schedule, largest = _sort_with_plan(ctx, mrs, keys, vals, n, 0, 32)
    assert schedule == mrs.SortPlan.SCHEDULE_MSD and 0 < largest <= 7680 and not in_b
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)
