"""CPU tests: the oracle against the reference's own unit tests (src/tests.rs), against an
independent stable sort, and against the committed golden vectors."""
import glob
import os

import numpy as np
import pytest

from inputs import DISTRIBUTIONS, make_keys, make_vals

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- restated unit tests of the reference (src/tests.rs:30-143) ---------------------
def test_workgroup_local_index(oracle):
    assert oracle.L.orc_selftest_workgroup_local_index(1000) == 0


def test_workgroup_calculation(oracle):
    assert oracle.L.orc_selftest_workgroup_calculation() == 0


def test_digit_test_6bit(oracle):
    assert oracle.L.orc_selftest_digit6() == 0


def test_digit_test_8(oracle):
    assert oracle.L.orc_selftest_digit8() == 0
    for i in range(8):  # src/tests.rs:132-143 from the Python side as well
        y = 255 << (i * 8)
        assert oracle.L.orc_select_digit8(i, y & 0xFFFFFFFF, y >> 32) == 255


def test_select_bits(oracle):
    v = 0x1122334455667788
    assert oracle.L.orc_select_bits(v, 8) == v
    assert oracle.L.orc_select_bits(v, 3) == 0x667788


# ---- host helpers ---------------------------------------------------------------------
def test_prefix_buffer_sizes(oracle):
    # src/main.rs:201-233 evaluated by hand for the shipped size and the benchmark sizes
    assert oracle.prefix_buffer_sizes(3000) == [256]
    assert oracle.prefix_buffer_sizes(1024) == []
    assert oracle.prefix_buffer_sizes(1025) == [256]
    assert oracle.prefix_buffer_sizes(1 << 16) == [256]
    assert oracle.prefix_buffer_sizes(1 << 20) == [4096, 256]
    assert oracle.prefix_buffer_sizes(1 << 28) == [1 << 20, 4096, 256]


def test_ballot_rank_with_tail(oracle):
    rng = np.random.default_rng(3)
    for nvalid in (1, 31, 32, 33, 200, 255, 256):
        digits = rng.integers(0, 256, size=256, dtype=np.uint32)
        got = oracle.ballot_rank_256(digits, nvalid)
        seen = {}
        for t in range(nvalid):
            assert got[t] == seen.get(int(digits[t]), 0)
            seen[int(digits[t])] = seen.get(int(digits[t]), 0) + 1


@pytest.mark.parametrize("length", [256, 1024, 1280, 65536, 4096 * 256 + 512])
def test_hierarchical_scan_equals_flat_scan(oracle, length):
    vals = oracle.gen_u32(11, 5, length) >> np.uint32(20)
    assert np.array_equal(oracle.hierarchical_scan(vals), oracle.exclusive_scan(vals))


def test_scan_wraps_like_u32(oracle):
    vals = np.full(1024, 0x80000000, dtype=np.uint32)
    out = oracle.exclusive_scan(vals)
    assert out[1] == 0x80000000 and out[2] == 0 and out[3] == 0x80000000


# ---- the sort: reference-structured LSD vs independent stable sort ---------------------
SHAPES = [(1, 0), (1, 1), (2, 0), (2, 1), (2, 4)]
SIZES = [0, 1, 2, 255, 256, 257, 1500, 3000, 10000]


@pytest.mark.parametrize("key_words,payload_words", SHAPES)
@pytest.mark.parametrize("n", SIZES)
def test_lsd_matches_stable_sort(oracle, key_words, payload_words, n):
    for dist in ("random", "few_distinct", "all_equal"):
        keys = make_keys(oracle, n, key_words, dist)
        vals = make_vals(oracle, n, payload_words)
        end = 32 * key_words
        k1, v1, in_b = oracle.lsd_sort(keys, vals, n, key_words, payload_words, 8, 0, end, 256)
        k2, v2 = oracle.stable_sort_pairs(keys, vals, n, key_words, payload_words, 0, end)
        assert in_b == ((end // 8) % 2 == 1)
        assert np.array_equal(k1, k2)
        if vals is not None:
            assert np.array_equal(v1, v2)


@pytest.mark.parametrize("dist", DISTRIBUTIONS)
def test_lsd_distributions_and_digit_widths(oracle, dist):
    n = 5000
    keys = make_keys(oracle, n, 1, dist)
    vals = make_vals(oracle, n, 1)
    ref_k, ref_v = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
    for bits, tile in ((8, 256), (4, 256), (8, 8192), (4, 1024)):
        k, v, _ = oracle.lsd_sort(keys, vals, n, 1, 1, bits, 0, 32, tile)
        assert np.array_equal(k, ref_k) and np.array_equal(v, ref_v)


def test_partial_bit_range(oracle):
    n = 4000
    keys = make_keys(oracle, n, 1, "random")
    vals = make_vals(oracle, n, 1)
    for begin, end, bits in ((0, 30, 8), (4, 20, 8), (0, 30, 4), (3, 11, 8)):
        k, v, in_b = oracle.lsd_sort(keys, vals, n, 1, 1, bits, begin, end, 256)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, begin, end)
        assert np.array_equal(k, rk) and np.array_equal(v, rv)
        assert in_b == ((-(-(end - begin) // bits)) % 2 == 1)


def test_config0_64k_points_morton30(oracle):
    """The reference-sized CPU case: 64K random 3D points -> 30-bit Morton keys, CPU sort (no GPU)."""
    n = 1 << 16
    xyz = oracle.gen_points(7, n)
    mn, mx = oracle.minmax3(xyz)
    keys, idx = oracle.morton_encode30(xyz, mn, mx)
    assert keys.max() < (1 << 30)
    assert np.array_equal(idx, np.arange(n, dtype=np.uint32))
    k, v, in_b = oracle.lsd_sort(keys, idx, n, 1, 1, 8, 0, 32, 256)
    assert not in_b
    rk, rv = oracle.stable_sort_pairs(keys, idx, n, 1, 1, 0, 32)
    assert np.array_equal(k, rk) and np.array_equal(v, rv)
    # numpy's stable argsort as a third opinion
    order = np.argsort(keys, kind="stable").astype(np.uint32)
    assert np.array_equal(v, order) and np.array_equal(k, keys[order])
    assert oracle.count_unsorted(k, n, 1, 4) == 0


def test_morton30_known_answers(oracle):
    mn = np.zeros(3, dtype=np.float32)
    mx = np.ones(3, dtype=np.float32)
    pts = np.array([[0, 0, 0], [1, 1, 1], [0.5, 0, 0], [0, 0.5, 0], [0, 0, 0.5], [2, -1, 0.25]], dtype=np.float32)
    keys, _ = oracle.morton_encode30(pts, mn, mx)
    assert keys[0] == 0
    assert keys[1] == (1 << 30) - 1
    assert keys[2] == 1 << 29 and keys[3] == 1 << 28 and keys[4] == 1 << 27
    # clamped: x -> 1023, y -> 0, z -> 256
    assert keys[5] == (0x49249249 << 2) & 0x3FFFFFFF | (1 << 24)


def test_histogram_layout_is_digit_major(oracle):
    n = 600
    keys = (np.arange(n, dtype=np.uint32) % np.uint32(3)).astype(np.uint32)
    hist = oracle.block_histogram(keys, n, 1, 0, 8, 256)
    W = 3
    assert hist.size == 256 * W
    for d in range(3):
        for b in range(W):
            lo, hi = b * 256, min(n, (b + 1) * 256)
            assert hist[W * d + b] == np.count_nonzero(keys[lo:hi] == d)


# ---- golden vectors ---------------------------------------------------------------------
def test_golden_vectors(oracle):
    files = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))
    assert files, "golden fixtures are missing (tests/golden/make_golden.py writes them)"
    for f in files:
        g = np.load(f)
        kind = str(g["kind"])
        if kind == "sort":
            kw, pw = int(g["key_words"]), int(g["payload_words"])
            n = int(g["n"])
            vals = g["vals_in"] if pw else None
            k, v, in_b = oracle.lsd_sort(g["keys_in"], vals, n, kw, pw, int(g["bits"]), int(g["begin_bit"]),
                                         int(g["end_bit"]), int(g["tile"]))
            assert np.array_equal(k, g["keys_out"]), f
            if pw:
                assert np.array_equal(v, g["vals_out"]), f
            assert in_b == bool(g["result_in_b"]), f
        elif kind == "encode30":
            k, i = oracle.morton_encode30(g["xyz"], g["mn"], g["mx"])
            assert np.array_equal(k, g["keys"]) and np.array_equal(i, g["idx"]), f
        elif kind == "encode_lut":
            codes = oracle.morton_encode_lut(g["blob"], g["vertices"], g["triangles"])
            assert np.array_equal(codes, g["codes"]), f
        else:
            raise AssertionError(f"unknown golden kind {kind}")
