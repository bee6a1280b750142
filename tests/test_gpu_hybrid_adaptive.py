"""GPU parity tests of the adaptive form of the most-significant-digit-first schedule
(MRS_SCHEDULE_MSD_ADAPTIVE): keys that crowd into few of the fixed sub-buckets are scattered by
tables the device cuts to their density.  Through the C ABI, bit for bit against the CPU checker's
stable sort.  The reference sorts codes of triangle centroids (src/main.rs:114-173), which lie on
surfaces: the inputs here are made like those."""
import numpy as np
import pytest
import torch

from gpu_util import to_dev, to_u32
from inputs import make_keys, make_vals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mrs():
    import wgpu_morton_code_radix_sort_test_amd as m

    return m


@pytest.fixture()
def small(ctx, aids):
    aids(ctx.ATTR_TEST_HYBRID_MIN_N, 1)


def _sort(ctx, mrs, keys, vals, n, end, twice=False, indexed=False):
    """`indexed`: the payload is made up by the first pass (mrs_sort_pairs_indexed; `vals` must be 0 .. n-1)."""
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, begin_bit=0, end_bit=end, algo=mrs.ALGO_HYBRID)
    ka, va = to_dev(keys, ctx.device), to_dev(vals, ctx.device)
    if indexed:
        va.fill_(0x5A5A5A5A)
    kb, vb = torch.full_like(ka, -1), torch.full_like(va, -1)
    sort = plan.sort_pairs_indexed if indexed else plan.sort_pairs
    in_b = sort(ka, va, kb, vb, n=n)
    plan.check()
    if twice:
        ka.copy_(to_dev(keys, ctx.device))
        if not indexed:
            va.copy_(to_dev(vals, ctx.device))
        in_b = sort(ka, va, kb, vb, n=n)
        plan.check()
    schedule, largest = plan.schedule()
    over = plan.oversized()
    ko, vo = to_u32(kb if in_b else ka)[:n], to_u32(vb if in_b else va)[:n]
    plan.close()
    return ko, vo, in_b, schedule, largest, over


def _part1by2(v):
    v = v.astype(np.uint32) & np.uint32(0x3FF)
    v = (v | (v << np.uint32(16))) & np.uint32(0x030000FF)
    v = (v | (v << np.uint32(8))) & np.uint32(0x0300F00F)
    v = (v | (v << np.uint32(4))) & np.uint32(0x030C30C3)
    v = (v | (v << np.uint32(2))) & np.uint32(0x09249249)
    return v


def _surface_codes(n, seed, shell=0.5, thickness=0.0):
    """30-bit Morton codes of points on a sphere shell (`shell` of them) and on three axis-aligned planes."""
    rng = np.random.default_rng(seed)
    xyz = rng.random((n, 3), dtype=np.float32)
    half = int(n * shell)
    d = xyz[:half] - 0.5
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6)
    xyz[:half] = 0.5 + (0.45 + thickness * rng.standard_normal((half, 1)).astype(np.float32)) * d
    third = (n - half) // 3
    xyz[half:half + third, 0] = 0.3
    xyz[half + third:half + 2 * third, 1] = 0.55
    xyz[half + 2 * third:, 2] = 0.8
    q = np.clip((xyz * 1024).astype(np.int64), 0, 1023)
    return np.ascontiguousarray((_part1by2(q[:, 0]) << np.uint32(2)) | (_part1by2(q[:, 1]) << np.uint32(1)) | _part1by2(q[:, 2]))


@pytest.mark.parametrize("n", [5000, 70001, 300007, (1 << 20) + 77, 3000003])
@pytest.mark.parametrize("end,dist", [(30, "morton30"), (28, "morton30"), (32, "one_hot_digit")])
def test_forced_on_spread_keys(ctx, oracle, mrs, small, aids, n, end, dist):
    """The device passes over the fixed sub-buckets; buckets and groups are cut to the keys."""
    aids(ctx.ATTR_TEST_HYBRID_FORCE_ADAPTIVE, 1)
    keys = make_keys(oracle, n, 1, dist, seed=n)
    if end == 28:
        keys = keys >> np.uint32(2)
    vals = make_vals(oracle, n, 1)
    ko, vo, in_b, schedule, largest, over = _sort(ctx, mrs, keys, vals, n, end)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, end)
    assert not in_b
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv), (n, end, dist, schedule)
    if dist == "morton30":
        assert schedule == mrs.SortPlan.SCHEDULE_MSD_ADAPTIVE, (n, end, largest, over)


@pytest.mark.parametrize("n,capacity", [(400003, 0), (400003, 2048), (400003, 24), (2500000, 0), (2500000, 300), (2500000, 150)])
def test_points_on_surfaces(ctx, oracle, mrs, small, aids, n, capacity):
    """Codes of points on a sphere shell and three planes: with a workgroup taken to hold few pairs
    the fixed sub-buckets are refused at these sizes as they are at 2^28 pairs with the real
    capacity, and the adaptive cut takes over by the device's own decision."""
    aids(ctx.ATTR_TEST_HYBRID_CAPACITY, capacity)
    keys = _surface_codes(n, seed=n + capacity)
    vals = make_vals(oracle, n, 1)
    ko, vo, in_b, schedule, largest, over = _sort(ctx, mrs, keys, vals, n, 30, twice=True)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 30)
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv), (n, capacity, schedule)
    if 0 < capacity <= 300:
        assert schedule == mrs.SortPlan.SCHEDULE_MSD_ADAPTIVE, (n, capacity, largest, over)
        assert over[2] == capacity


def test_heavy_duplicates_become_oversized_fine_cells(ctx, oracle, mrs, small, aids):
    """A few key values with more copies than a workgroup holds: fine cells that cannot be cut are
    sorted where they lie by the two extra passes, inside the adaptive schedule."""
    aids(ctx.ATTR_TEST_HYBRID_CAPACITY, 500)
    aids(ctx.ATTR_TEST_HYBRID_FORCE_ADAPTIVE, 1)
    n = 600011
    keys = _surface_codes(n, seed=11)
    rng = np.random.default_rng(3)
    for j, copies in enumerate([501, 900, 5000, 20000]):
        at = rng.integers(0, n - copies)
        keys[at:at + copies] = keys[at]
    keys = np.ascontiguousarray(keys[rng.permutation(n)])
    vals = make_vals(oracle, n, 1)
    ko, vo, in_b, schedule, largest, over = _sort(ctx, mrs, keys, vals, n, 30)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 30)
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv), schedule
    assert schedule == mrs.SortPlan.SCHEDULE_MSD_ADAPTIVE, (largest, over)
    assert over[0] >= 4 and over[1] >= 501 + 900 + 5000 + 20000


def test_a_bucket_that_cannot_be_cut_hands_over_to_the_ordinary_passes(ctx, oracle, mrs, small, aids):
    """32 sorted bits, keys in every one of the 65536 cells of the leading 16: a bucket of 1/224 of
    the keys spans about 290 of them and a group never spans two, so the bucket needs more than 256
    groups.  The device finds out after pass A and the ordinary passes sort the input."""
    aids(ctx.ATTR_TEST_HYBRID_FORCE_ADAPTIVE, 1)
    n = 1 << 20
    keys = make_keys(oracle, n, 1, "random", seed=5)
    vals = make_vals(oracle, n, 1)
    ko, vo, in_b, schedule, largest, over = _sort(ctx, mrs, keys, vals, n, 32)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
    assert schedule == mrs.SortPlan.SCHEDULE_MSD_REFUSED
    assert not in_b
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


def test_all_equal_and_few_values(ctx, oracle, mrs, small, aids):
    aids(ctx.ATTR_TEST_HYBRID_FORCE_ADAPTIVE, 1)
    n = 200003
    for dist in ("all_equal", "few_distinct", "sorted", "reversed", "one_hot_digit"):
        keys = make_keys(oracle, n, 1, dist) & np.uint32(0x3FFFFFFF)
        vals = make_vals(oracle, n, 1)
        ko, vo, in_b, schedule, largest, over = _sort(ctx, mrs, keys, vals, n, 30)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 30)
        assert np.array_equal(ko, rk) and np.array_equal(vo, rv), (dist, schedule)


def test_constant_bits_above_the_sorted_range(ctx, oracle, mrs, small, aids):
    """Bits above end_bit that are the same in every key are carried along (cells are numbered by
    the sorted bits only)."""
    aids(ctx.ATTR_TEST_HYBRID_FORCE_ADAPTIVE, 1)
    n = 150001
    keys = _surface_codes(n, seed=9) | np.uint32(0x80000000)
    vals = make_vals(oracle, n, 1)
    ko, vo, in_b, schedule, largest, over = _sort(ctx, mrs, keys, vals, n, 30)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 30)
    assert schedule == mrs.SortPlan.SCHEDULE_MSD_ADAPTIVE
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


def test_not_offered_when_switched_off(ctx, oracle, mrs, small, aids):
    aids(ctx.ATTR_TEST_HYBRID_CAPACITY, 150)
    aids(ctx.ATTR_TEST_HYBRID_NO_ADAPTIVE, 1)
    n = 2500000
    keys = _surface_codes(n, seed=1)
    vals = make_vals(oracle, n, 1)
    ko, vo, in_b, schedule, largest, over = _sort(ctx, mrs, keys, vals, n, 30)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 30)
    assert schedule == mrs.SortPlan.SCHEDULE_MSD_REFUSED
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


def test_fuzz(ctx, oracle, mrs, small):
    """Seeded random sizes, widths, capacities and key shapes with the adaptive form offered (and, in
    half of the cases, forced): whatever the device decides, the result is the stable sort."""
    import os

    seed = int(os.environ.get("MRS_FUZZ_SEED", "20261112"))
    rng = np.random.RandomState(seed)
    seen = set()
    for case in range(int(os.environ.get("MRS_FUZZ_CASES", "30"))):
        n = int(rng.choice([rng.randint(4097, 20000), rng.randint(20000, 200000), rng.randint(200000, 900000)]))
        end = int(rng.randint(28, 33))
        shape = ["surface", "thick", "random", "morton30", "few_distinct", "dup", "two_cells"][rng.randint(7)]
        if shape in ("surface", "thick"):
            keys = _surface_codes(n, seed=case, shell=float(rng.rand()), thickness=0.0 if shape == "surface" else 0.01)
        elif shape == "dup":
            keys = make_keys(oracle, n, 1, "morton30", seed=case) & np.uint32(~0x3FF & 0xFFFFFFFF)
        elif shape == "two_cells":  # everything in two cells of the leading 12 bits, one of them a single key value
            keys = (make_keys(oracle, n, 1, "random", seed=case) & np.uint32(0x3FFFF)) | np.uint32(0x155 << 18)
            keys[: n // 3] = np.uint32(0x2AA << 18) | np.uint32(12345)
        else:
            keys = make_keys(oracle, n, 1, shape, seed=case)
        if end < 32 and shape not in ("surface", "thick", "two_cells", "dup", "morton30"):
            keys = keys & np.uint32((1 << end) - 1)  # nothing above the sorted range: 6-byte staging stays possible
        if end > 30 and shape in ("surface", "thick", "dup", "morton30", "two_cells") and rng.randint(2):
            keys = keys << np.uint32(end - 30)  # use the declared width
        capacity = int(rng.choice([0, 0, int(rng.randint(40, 400)), int(rng.randint(400, 4000))]))
        forced = bool(rng.randint(2))
        ctx.set_attribute(ctx.ATTR_TEST_HYBRID_CAPACITY, capacity)
        ctx.set_attribute(ctx.ATTR_TEST_HYBRID_FORCE_ADAPTIVE, 1 if forced else 0)
        indexed = bool(rng.randint(3) == 0)
        try:
            vals = np.arange(n, dtype=np.uint32) if indexed else make_vals(oracle, n, 1)
            ko, vo, in_b, schedule, largest, over = _sort(ctx, mrs, keys, vals, n, end, twice=bool(rng.randint(2)), indexed=indexed)
        finally:
            ctx.set_attribute(ctx.ATTR_TEST_HYBRID_CAPACITY, 0)
            ctx.set_attribute(ctx.ATTR_TEST_HYBRID_FORCE_ADAPTIVE, 0)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, end)
        what = (case, n, end, shape, capacity, forced, indexed, schedule, largest, over)
        seen.add(schedule)
        assert not in_b, what
        assert np.array_equal(ko, rk), what
        assert np.array_equal(vo, rv), what
    assert mrs.SortPlan.SCHEDULE_MSD_ADAPTIVE in seen


def test_replays_from_a_graph(ctx, oracle, mrs, small, aids):
    """All three schedules are enqueued by every call and nothing on the host waits for the device's
    choice, so one captured call serves spread keys (fixed sub-buckets), surface keys (buckets and
    groups cut to them) and equal keys (ordinary passes) alike."""
    aids(ctx.ATTR_TEST_HYBRID_CAPACITY, 60)
    n = 300000
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, end_bit=30, algo=mrs.ALGO_HYBRID)
    ka = torch.zeros(n, dtype=torch.int32, device=ctx.device)
    va, kb, vb = torch.zeros_like(ka), torch.zeros_like(ka), torch.zeros_like(ka)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ka.copy_(to_dev(make_keys(oracle, n, 1, "morton30", seed=1), ctx.device))
        plan.sort_pairs(ka, va, kb, vb, stream=side)  # warm-up outside the capture
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        in_b = plan.sort_pairs(ka, va, kb, vb, stream=torch.cuda.current_stream())
    assert not in_b
    seen = []
    for shape in ("morton30", "surface", "all_equal", "surface", "morton30"):
        if shape == "surface":
            keys = _surface_codes(n, seed=len(seen))
        else:
            keys = make_keys(oracle, n, 1, shape, seed=3) & np.uint32(0x3FFFFFFF)
        vals = make_vals(oracle, n, 1)
        ka.copy_(to_dev(keys, ctx.device))
        va.copy_(to_dev(vals, ctx.device))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        seen.append(plan.schedule()[0])
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 30)
        assert np.array_equal(to_u32(ka), rk) and np.array_equal(to_u32(va), rv), (shape, seen)
    assert seen[1] == mrs.SortPlan.SCHEDULE_MSD_ADAPTIVE and seen[2] == mrs.SortPlan.SCHEDULE_MSD_REFUSED, seen
    plan.check()
    del graph
    plan.close()


@pytest.mark.parametrize("offset", [1, 2, 3])
def test_misaligned_buffers(ctx, oracle, mrs, small, aids, offset):
    """Key and payload arrays that start 4, 8 or 12 bytes past a 16-byte boundary (the kernel that
    counts a bucket reads 16 bytes at a time where it can); nothing is written outside them."""
    aids(ctx.ATTR_TEST_HYBRID_CAPACITY, 100)
    n = 333337
    keys = _surface_codes(n, seed=offset)
    vals = make_vals(oracle, n, 1)
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, end_bit=30, algo=mrs.ALGO_HYBRID)
    bufs = [torch.zeros(n + 8, dtype=torch.int32, device=ctx.device) for _ in range(4)]
    ka, va, kb, vb = (b[offset:n + offset] for b in bufs)
    ka.copy_(to_dev(keys, ctx.device))
    va.copy_(to_dev(vals, ctx.device))
    in_b = plan.sort_pairs(ka, va, kb, vb)
    plan.check()
    assert plan.schedule()[0] == plan.SCHEDULE_MSD_ADAPTIVE and not in_b
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 30)
    assert np.array_equal(to_u32(ka), rk) and np.array_equal(to_u32(va), rv)
    for b in bufs:
        assert int(b[:offset].abs().sum()) == 0 and int(b[n + offset:].abs().sum()) == 0
    plan.close()
