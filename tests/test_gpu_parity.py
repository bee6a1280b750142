"""GPU parity tests: every call goes through the C ABI of libmrs.so and is compared
bit-for-bit with the CPU oracle on the same seeded input."""
import glob
import os

import numpy as np
import pytest
import torch

from gpu_util import gpu_sort, to_dev, to_u32
from inputs import DISTRIBUTIONS, make_keys, make_vals

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def mrs():
    import wgpu_morton_code_radix_sort_test_amd as m

    return m


# ---- generators and encoders ---------------------------------------------------------
def test_generators_match_oracle(ctx, oracle):
    for n in (1, 1000, 65537):
        xyz = ctx.generate_points(7, n).cpu().numpy()
        assert np.array_equal(xyz.view(np.uint32), oracle.gen_points(7, n).view(np.uint32))
        u = to_u32(ctx.generate_u32(9, 3, n))
        assert np.array_equal(u, oracle.gen_u32(9, 3, n))


@pytest.mark.parametrize("n", [1, 3, 4, 5, 1023, 4095, 4096, 4097, 65536, 100003, 4096 * 256, 4096 * 600 + 7])
def test_morton_encode30_matches_oracle(ctx, oracle, n):
    """Sizes on both sides of the encoder's 4096-point tile, one tile per workgroup, and more tiles
    than workgroups (every workgroup loops, loading one tile ahead) with a ragged end."""
    xyz = oracle.gen_points(7, n)
    mn, mx = oracle.minmax3(xyz)
    keys, idx = ctx.morton_encode30(to_dev(xyz, ctx.device), mn, mx)
    rk, ri = oracle.morton_encode30(xyz, mn, mx)
    assert np.array_equal(to_u32(keys), rk)
    assert np.array_equal(to_u32(idx), ri)
    # bounds tighter than the data: exercises the clamp on both sides
    mn2 = np.array([0.2, 0.1, 0.3], dtype=np.float32)
    mx2 = np.array([0.7, 0.9, 0.8], dtype=np.float32)
    keys, _ = ctx.morton_encode30(to_dev(xyz, ctx.device), mn2, mx2)
    assert np.array_equal(to_u32(keys), oracle.morton_encode30(xyz, mn2, mx2)[0])


@pytest.mark.parametrize("nv,nt", [(3, 1), (1000, 3000), (5000, 70001)])
def test_morton_encode_lut_matches_oracle(ctx, oracle, mrs, nv, nt):
    positions, indices, gen = mrs.scene.create_scene(nv, nt, 7)
    verts = mrs.scene.pack_vertices(positions)
    tris = mrs.scene.pack_triangles(indices)
    for blob in (mrs.scene.create_morton_uniforms(gen),
                 mrs.scene.random_uniforms_blob(5, gen.offset, gen.multiplier, gen.size_multiplier),
                 # offset inside the scene and a huge multiplier: negative and overflowing quantised values
                 mrs.scene.random_uniforms_blob(6, [0.5, 0.5, 0.5], [3e9, 1e10, 512.0], 1e6)):
        codes = ctx.morton_encode_lut(to_dev(blob, ctx.device), to_dev(verts, ctx.device), to_dev(tris, ctx.device))
        assert np.array_equal(to_u32(codes), oracle.morton_encode_lut(blob, verts, tris))


def test_morton_encode_lut_out_of_range_vertex_index(ctx, oracle, mrs):
    """A triangle that names a vertex past the end of the vertex buffer (undefined in the reference,
    whose wgpu robustness layer decides): library and checker both read the last vertex."""
    positions, indices, gen = mrs.scene.create_scene(100, 500, 3)
    indices = indices.copy()
    indices[7, 1] = 100          # first index past the buffer
    indices[11, 2] = 0xFFFFFFFF  # and the largest one
    verts = mrs.scene.pack_vertices(positions)
    tris = mrs.scene.pack_triangles(indices)
    blob = mrs.scene.create_morton_uniforms(gen)
    codes = ctx.morton_encode_lut(to_dev(blob, ctx.device), to_dev(verts, ctx.device), to_dev(tris, ctx.device))
    want = oracle.morton_encode_lut(blob, verts, tris)
    assert np.array_equal(to_u32(codes), want)
    clamped = indices.copy()
    clamped[clamped >= 100] = 99
    assert np.array_equal(want, oracle.morton_encode_lut(blob, verts, mrs.scene.pack_triangles(clamped)))


# ---- single stages of the three-kernel pass ------------------------------------------
@pytest.mark.parametrize("tile", [256, 1024, 8192])
@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("n", [1, 255, 256, 257, 3000, 70000])
def test_stages_match_oracle(ctx, oracle, mrs, tile, bits, n):
    keys = make_keys(oracle, n, 1, "random")
    vals = make_vals(oracle, n, 1)
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=bits, algo=mrs.ALGO_THREE_KERNEL,
                        tile_items=tile)
    assert plan.info.tile_items == tile
    dk, dv = to_dev(keys, ctx.device), to_dev(vals, ctx.device)
    for p in (0, plan.info.num_passes - 1):
        shift = p * bits
        hist = plan.histogram(dk, p)
        ref_hist = oracle.block_histogram(keys, n, 1, shift, bits, tile)
        assert np.array_equal(to_u32(hist), ref_hist)
        plan.scan(hist)
        ref_scan = oracle.exclusive_scan(ref_hist)
        assert np.array_equal(to_u32(hist), ref_scan)
        ko, vo = torch.empty_like(dk), torch.empty_like(dv)
        plan.scatter(dk, dv, ko, vo, hist, p)
        rk, rv = oracle.scatter(keys, vals, n, 1, 1, shift, bits, tile, ref_scan)
        assert np.array_equal(to_u32(ko), rk)
        assert np.array_equal(to_u32(vo), rv)
    plan.close()


def test_reference_histogram_layout(ctx, oracle, mrs):
    """tile_items = 256 reproduces the reference's [256][ceil(n/256)] histogram and its
    hierarchical prefix sum for the shipped 3000-element case (src/main.rs:377-381)."""
    n = 3000
    keys = make_keys(oracle, n, 2, "random")
    plan = mrs.SortPlan(ctx, n, key_words=2, payload_bytes=16, digit_bits=8, algo=mrs.ALGO_THREE_KERNEL,
                        tile_items=256)
    assert plan.info.max_tiles == 12 and plan.info.histogram_len == 3072 and plan.info.num_passes == 8
    dk = to_dev(keys, ctx.device)
    for p in range(8):
        hist = plan.histogram(dk, p)
        ref = oracle.block_histogram(keys, n, 2, 8 * p, 8, 256)
        assert np.array_equal(to_u32(hist), ref)
        plan.scan(hist)
        assert np.array_equal(to_u32(hist), oracle.hierarchical_scan(ref))


def test_large_scan(ctx, oracle, mrs):
    # 2^20 keys at 256 per tile: a 2^20-entry histogram (three scan kernels)
    n = 1 << 20
    keys = make_keys(oracle, n, 1, "random")
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=0, digit_bits=8, algo=mrs.ALGO_THREE_KERNEL,
                        tile_items=256)
    hist = plan.histogram(to_dev(keys, ctx.device), 1)
    ref = oracle.block_histogram(keys, n, 1, 8, 8, 256)
    assert np.array_equal(to_u32(hist), ref)
    plan.scan(hist)
    assert np.array_equal(to_u32(hist), oracle.exclusive_scan(ref))


# ---- the whole sort ---------------------------------------------------------------------
SHAPES = [(1, 0), (1, 1), (2, 0), (2, 1), (2, 4)]
SIZES = [0, 1, 255, 256, 257, 3000, 8191, 8192, 8193, 65536, 200001]


@pytest.mark.parametrize("algo_name", ["three_kernel", "onesweep"])
@pytest.mark.parametrize("key_words,payload_words", SHAPES)
@pytest.mark.parametrize("bits", [4, 8])
def test_sort_matches_oracle(ctx, oracle, mrs, algo_name, key_words, payload_words, bits):
    algo = mrs.ALGO_ONESWEEP if algo_name == "onesweep" else mrs.ALGO_THREE_KERNEL
    end = 32 * key_words
    for n in SIZES:
        keys = make_keys(oracle, n, key_words, "random", seed=n + 1)
        vals = make_vals(oracle, n, payload_words)
        ko, vo, in_b = gpu_sort(ctx, mrs, keys, vals, n, key_words, payload_words, bits, 0, end, algo)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, key_words, payload_words, 0, end)
        assert in_b == ((end // bits) % 2 == 1)
        assert np.array_equal(ko, rk), (n, "keys")
        if payload_words:
            assert np.array_equal(vo, rv), (n, "payload")


@pytest.mark.parametrize("algo_name", ["three_kernel", "onesweep"])
@pytest.mark.parametrize("dist", DISTRIBUTIONS)
def test_sort_distributions(ctx, oracle, mrs, algo_name, dist):
    algo = mrs.ALGO_ONESWEEP if algo_name == "onesweep" else mrs.ALGO_THREE_KERNEL
    for key_words, payload_words, n in ((1, 1, 100000), (2, 4, 20000)):
        keys = make_keys(oracle, n, key_words, dist)
        vals = make_vals(oracle, n, payload_words)
        end = 32 * key_words
        ko, vo, _ = gpu_sort(ctx, mrs, keys, vals, n, key_words, payload_words, 8, 0, end, algo)
        # oracle with the reference's pass structure, then the independent stable sort
        rk, rv, _ = oracle.lsd_sort(keys, vals, n, key_words, payload_words, 8, 0, end, 256)
        assert np.array_equal(ko, rk) and np.array_equal(vo, rv)
        sk, sv = oracle.stable_sort_pairs(keys, vals, n, key_words, payload_words, 0, end)
        assert np.array_equal(ko, sk) and np.array_equal(vo, sv)


@pytest.mark.parametrize("tile", [256, 1024, 4096, 8192, 16384, 32768])
@pytest.mark.parametrize("algo_name", ["three_kernel", "onesweep"])
def test_sort_tile_shapes(ctx, oracle, mrs, tile, algo_name):
    algo = mrs.ALGO_ONESWEEP if algo_name == "onesweep" else mrs.ALGO_THREE_KERNEL
    n = 50001 if tile <= 8192 else 12 * tile + 77
    keys = make_keys(oracle, n, 1, "morton30")
    vals = make_vals(oracle, n, 1)
    ko, vo, _ = gpu_sort(ctx, mrs, keys, vals, n, 1, 1, 8, 0, 32, algo, tile_items=tile)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


@pytest.mark.parametrize("tile", [256, 1024, 4096, 8192])
@pytest.mark.parametrize("algo_name", ["three_kernel", "onesweep"])
def test_sort_tile_shapes_triangle_payload(ctx, oracle, mrs, tile, algo_name):
    """The reference's element type: 64-bit code, 16-byte TriangleIndices payload."""
    algo = mrs.ALGO_ONESWEEP if algo_name == "onesweep" else mrs.ALGO_THREE_KERNEL
    n = 9 * tile + 5
    keys = make_keys(oracle, n, 2, "random")
    vals = make_vals(oracle, n, 4)
    ko, vo, _ = gpu_sort(ctx, mrs, keys, vals, n, 2, 4, 8, 0, 64, algo, tile_items=tile)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 2, 4, 0, 64)
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


@pytest.mark.parametrize("key_words,payload_words,tile", [(1, 1, 0), (1, 1, 32768), (2, 4, 8192), (2, 1, 16384), (1, 0, 256)])
def test_onesweep_wide_status_words(ctx, oracle, mrs, aids, key_words, payload_words, tile):
    """The 64-bit status words and 64-bit addressing that plans of 2^30 elements (or arrays above
    4 GiB) use, forced on an input the checker can sort."""
    aids(ctx.ATTR_TEST_WIDE_STATUS, 1)
    n = 150001 if tile <= 8192 else 10 * tile + 9
    keys = make_keys(oracle, n, key_words, "random")
    vals = make_vals(oracle, n, payload_words)
    end = 32 * key_words
    ko, vo, _ = gpu_sort(ctx, mrs, keys, vals, n, key_words, payload_words, 8, 0, end, mrs.ALGO_ONESWEEP, tile_items=tile)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, key_words, payload_words, 0, end)
    assert np.array_equal(ko, rk)
    if payload_words:
        assert np.array_equal(vo, rv)


@pytest.mark.parametrize("tile", [0, 256, 32768])
def test_match_rank_fallback(oracle, mrs, tile):
    """A context whose rank mode is set to 0 ranks by ballot matching (what a device that fails
    the LDS-atomic order check would use); the default context ranks by LDS atomics."""
    ctx2 = mrs.Context(0)
    assert ctx2.rank_mode == "atomic"
    ctx2.set_attribute(ctx2.ATTR_RANK_MODE, 0)
    assert ctx2.rank_mode == "match"
    n = 100003 if tile <= 8192 else 6 * tile + 1
    for key_words, payload_words in ((1, 1), (2, 4)):
        if tile > 8192 and key_words == 2:
            continue
        keys = make_keys(oracle, n, key_words, "few_distinct")
        vals = make_vals(oracle, n, payload_words)
        end = 32 * key_words
        ko, vo, _ = gpu_sort(ctx2, mrs, keys, vals, n, key_words, payload_words, 8, 0, end, mrs.ALGO_ONESWEEP, tile_items=tile)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, key_words, payload_words, 0, end)
        assert np.array_equal(ko, rk) and np.array_equal(vo, rv)
    ctx2.close()


@pytest.mark.parametrize("key_words,payload_words", SHAPES)
@pytest.mark.parametrize("bits", [4, 8])
def test_small_inputs_both_paths(ctx, oracle, mrs, aids, key_words, payload_words, bits):
    """Inputs of at most 4096 elements are sorted by one workgroup in one launch; with
    MRS_ATTR_TEST_NO_SMALL_SORT they take the general multi-launch path.  Both must agree with the checker,
    including the side the result lands on."""
    end = 32 * key_words
    for general in (False, True):
        if general:
            aids(ctx.ATTR_TEST_NO_SMALL_SORT, 1)
        for n in (1, 2, 63, 64, 65, 255, 256, 257, 1000, 3000, 4095, 4096):
            for dist in ("random", "all_equal"):
                keys = make_keys(oracle, n, key_words, dist, seed=n)
                vals = make_vals(oracle, n, payload_words)
                ko, vo, in_b = gpu_sort(ctx, mrs, keys, vals, n, key_words, payload_words, bits, 0, end, mrs.ALGO_ONESWEEP)
                rk, rv = oracle.stable_sort_pairs(keys, vals, n, key_words, payload_words, 0, end)
                assert in_b == ((end // bits) % 2 == 1)
                assert np.array_equal(ko, rk), (general, n, dist)
                if payload_words:
                    assert np.array_equal(vo, rv), (general, n, dist)


def test_small_sort_bit_ranges(ctx, oracle, mrs):
    """Odd and even pass counts (result in the second or the first buffer pair), partial last digit."""
    n = 3000
    for key_words, payload_words in ((1, 1), (2, 4)):
        keys = make_keys(oracle, n, key_words, "random")
        vals = make_vals(oracle, n, payload_words)
        for begin, end, bits in ((0, 30, 8), (0, 24, 8), (3, 11, 8), (0, 8, 8), (5, 31, 4), (0, 12, 4)):
            ko, vo, in_b = gpu_sort(ctx, mrs, keys, vals, n, key_words, payload_words, bits, begin, end, mrs.ALGO_ONESWEEP)
            rk, rv = oracle.stable_sort_pairs(keys, vals, n, key_words, payload_words, begin, end)
            assert in_b == ((-(-(end - begin) // bits)) % 2 == 1)
            assert np.array_equal(ko, rk) and np.array_equal(vo, rv), (key_words, begin, end, bits)


def test_sort_fuzz(ctx, oracle, mrs):
    """Seeded random plans: size, element type, digit width, bit range, tile size, algorithm and
    key distribution drawn together; every result must equal the checker's stable sort."""
    seed = int(os.environ.get("MRS_FUZZ_SEED", "20240611"))
    rng = np.random.RandomState(seed)
    mask_rng = np.random.RandomState(seed % 1000 + 77)
    tiles_small = {(1, 0): [0, 256, 1024, 4096, 8192, 16384, 32768], (1, 1): [0, 256, 1024, 4096, 8192, 16384, 32768],
                   (2, 0): [0, 256, 1024, 4096, 16384], (2, 1): [0, 256, 1024, 4096, 16384],
                   (2, 4): [0, 256, 1024, 4096, 8192], (1, 4): [0, 256, 1024, 4096, 8192]}
    for case in range(int(os.environ.get("MRS_FUZZ_CASES", "60"))):
        key_words, payload_words = [(1, 0), (1, 1), (2, 0), (2, 1), (2, 4), (1, 4)][rng.randint(6)]
        bits = int(rng.choice([4, 8]))
        key_bits = 32 * key_words
        begin = int(rng.randint(0, key_bits - 1))
        end = int(rng.randint(begin + 1, key_bits + 1))
        if (end - begin + bits - 1) // bits > 16:
            end = begin + 16 * bits
        tile = int(rng.choice(tiles_small[(key_words, payload_words)]))
        n = int(rng.choice([rng.randint(1, 70), rng.randint(70, 5000), rng.randint(5000, 120000), rng.randint(120000, 400000)]))
        dist = DISTRIBUTIONS[rng.randint(len(DISTRIBUTIONS))]
        algo = mrs.ALGO_ONESWEEP if rng.randint(3) else mrs.ALGO_THREE_KERNEL
        keys = make_keys(oracle, n, key_words, dist, seed=case)
        live = None
        if key_words == 2 and mask_rng.randint(2):
            # some nibbles the same for every key: passes the device may leave out
            live = [int(sum(0xF << (4 * j) for j in range(8) if mask_rng.rand() < 0.6)) for _ in range(2)]
            k2 = keys.reshape(n, 2).copy()
            for w in range(2):
                k2[:, w] = (k2[:, w] & np.uint32(live[w])) | np.uint32(0x5A5A5A5A & ~live[w] & 0xFFFFFFFF)
            keys = k2.reshape(-1)
        vals = make_vals(oracle, n, payload_words)
        # every third case with a 4-byte payload sorts (key, source index) with the indices made up on the device
        indexed = payload_words == 1 and case % 3 == 0
        if indexed:
            vals = np.arange(n, dtype=np.uint32)
        ko, vo, in_b = gpu_sort(ctx, mrs, keys, vals, n, key_words, payload_words, bits, begin, end, algo, tile_items=tile,
                                indexed=indexed)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, key_words, payload_words, begin, end)
        what = (case, key_words, payload_words, bits, begin, end, tile, n, dist, algo, live, indexed)
        assert in_b == (((end - begin + bits - 1) // bits) % 2 == 1), what
        assert np.array_equal(ko, rk), what
        if payload_words:
            assert np.array_equal(vo, rv), what


@pytest.mark.parametrize("key_words", [1, 2])
@pytest.mark.parametrize("record_words", [1, 2, 3, 4, 9])
def test_sort_gather_matches_oracle(ctx, oracle, mrs, key_words, record_words):
    """mrs_sort_gather (sort key + source index, move each record once) gives what carrying the
    record through every pass gives: the checker's stable sort of (key, record) pairs."""
    end = 32 * key_words
    for n in (1, 3000, 4097, 100001):
        keys = make_keys(oracle, n, key_words, "few_distinct", seed=n)
        recs = make_vals(oracle, n, record_words)
        plan = mrs.SortPlan(ctx, n, key_words=key_words, payload_bytes=4)
        ka = to_dev(keys, ctx.device)
        kb = torch.full_like(ka, -1)
        rin = to_dev(recs, ctx.device)
        rout = torch.full_like(rin, -1)
        in_b, index = plan.sort_gather(ka, kb, rin, rout, 4 * record_words, n=n)
        plan.check()
        assert in_b == ((end // 8) % 2 == 1)
        rk, rv = oracle.stable_sort_pairs(keys, recs, n, key_words, record_words, 0, end)
        assert np.array_equal(to_u32(kb if in_b else ka)[: n * key_words], rk)
        assert np.array_equal(to_u32(rout)[: n * record_words], rv)
        # the sorted source indices are left behind as well
        src = to_u32(index)[:n].astype(np.int64)
        assert np.array_equal(recs.reshape(n, record_words)[src].reshape(-1), rv)
        plan.close()


@pytest.mark.parametrize("key_words,digit_bits,algo_name,dist,n,tile", [
    (1, 8, "onesweep", "random", 1, 0), (1, 8, "onesweep", "few_distinct", 3000, 0),   # one-workgroup path
    (2, 8, "onesweep", "random", 4096, 0), (1, 4, "onesweep", "random", 4000, 0),
    (1, 8, "onesweep", "random", 4097, 0), (1, 8, "onesweep", "morton30", 300007, 0),
    (1, 8, "onesweep", "random", 70001, 256), (1, 8, "onesweep", "few_distinct", 200003, 32768),
    (1, 4, "onesweep", "random", 150001, 0), (1, 8, "three_kernel", "random", 150001, 0),
    (1, 8, "three_kernel", "all_equal", 40000, 1024),
    (2, 8, "onesweep", "random", 150001, 0), (2, 8, "three_kernel", "random", 50001, 0),
    # 64-bit keys whose passes are partly or all identities: passes 0 and 1 must run all the same
    (2, 8, "onesweep", "one_hot_digit", 150001, 0), (2, 8, "onesweep", "all_equal", 150001, 0),
    (2, 8, "onesweep", "few_distinct", 150001, 0), (2, 8, "onesweep", "sorted", 100000, 0),
])
def test_sort_pairs_indexed_matches_oracle(ctx, oracle, mrs, key_words, digit_bits, algo_name, dist, n, tile):
    """mrs_sort_pairs_indexed (the first pass makes the source indices up) gives what mrs_sort_pairs
    gives for a payload of 0..n-1, whatever the payload buffers held before the call."""
    end = 32 * key_words
    algo = mrs.ALGO_ONESWEEP if algo_name == "onesweep" else mrs.ALGO_THREE_KERNEL
    keys = make_keys(oracle, n, key_words, dist, seed=n)
    iota = np.arange(n, dtype=np.uint32)
    plan = mrs.SortPlan(ctx, n, key_words=key_words, payload_bytes=4, digit_bits=digit_bits, algo=algo, tile_items=tile)
    ka = to_dev(keys, ctx.device)
    kb = torch.full_like(ka, -1)
    va = torch.full((n,), 0x5A5A5A5A, dtype=torch.int32, device=ctx.device)  # never read
    vb = torch.full((n,), -1, dtype=torch.int32, device=ctx.device)
    in_b = plan.sort_pairs_indexed(ka, va, kb, vb, n=n)
    plan.check()
    assert in_b == (plan.info.num_passes % 2 == 1)
    rk, rv = oracle.stable_sort_pairs(keys, iota, n, key_words, 1, 0, end)
    assert np.array_equal(to_u32(kb if in_b else ka)[: n * key_words], rk)
    assert np.array_equal(to_u32(vb if in_b else va)[:n], rv)
    if key_words == 2 and algo_name == "onesweep" and n > 4096:
        left_out = plan.passes_left_out()
        assert left_out & 3 == 0, "the passes that make the payload up and keep the count even must run"
        assert bin(left_out).count("1") % 2 == 0
        if dist == "all_equal":
            assert left_out == 0b11111100
    # a plain sort on the same plan afterwards is unaffected
    ka2, va2 = to_dev(keys, ctx.device), to_dev(iota[::-1].copy(), ctx.device)
    in_b = plan.sort_pairs(ka2, va2, kb, vb, n=n)
    plan.check()
    rk, rv = oracle.stable_sort_pairs(keys, iota[::-1].copy(), n, key_words, 1, 0, end)
    assert np.array_equal(to_u32(kb if in_b else ka2)[: n * key_words], rk)
    assert np.array_equal(to_u32(vb if in_b else va2)[:n], rv)
    plan.close()


def test_sort_pairs_indexed_rejects_other_payloads(ctx, mrs):
    n = 1000
    for pb in (0, 16):
        plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=pb)
        k = torch.zeros(n, dtype=torch.int32, device=ctx.device)
        v = torch.zeros(4 * n, dtype=torch.int32, device=ctx.device)
        with pytest.raises(mrs.MrsError):
            plan.sort_pairs_indexed(k, v, torch.empty_like(k), torch.empty_like(v), n=n)
        plan.close()


def test_encode_keys_only_then_indexed_sort(ctx, oracle, mrs):
    """The pipeline without a stored index array: the encoder writes keys only, the sort makes the
    indices up.  Same keys and same sorted pairs as the two-array pipeline."""
    n = 4096 * 300 + 77
    xyz = oracle.gen_points(3, n)
    mn, mx = oracle.minmax3(xyz)
    rk, ri = oracle.morton_encode30(xyz, mn, mx)
    keys, idx = ctx.morton_encode30(to_dev(xyz, ctx.device), mn, mx, keys_only=True)
    assert idx is None
    assert np.array_equal(to_u32(keys), rk)
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, end_bit=30)
    va, kb, vb = (torch.full((n,), -1, dtype=torch.int32, device=ctx.device) for _ in range(3))
    in_b = plan.sort_pairs_indexed(keys, va, kb, vb)
    plan.check()
    sk, sv = oracle.stable_sort_pairs(rk, ri, n, 1, 1, 0, 30)
    assert np.array_equal(to_u32(kb if in_b else keys), sk) and np.array_equal(to_u32(vb if in_b else va), sv)
    plan.close()


def _karras_tree(codes, key_bits):
    """Binary radix tree over sorted codes (Karras 2012: determineRange + findSplit), with the
    leaf number breaking ties between equal codes.  Returns (left, right, first, last) per internal
    node; children >= n - 1 are leaves (leaf i is n - 1 + i)."""
    n = len(codes)

    def delta(i, j):
        if j < 0 or j >= n:
            return -1
        x = int(codes[i]) ^ int(codes[j])
        if x:
            return key_bits - x.bit_length()
        return key_bits + (32 - (i ^ j).bit_length())

    left, right = [0] * (n - 1), [0] * (n - 1)
    first, last = [0] * (n - 1), [0] * (n - 1)
    for i in range(n - 1):
        d = 1 if delta(i, i + 1) > delta(i, i - 1) else -1
        dmin = delta(i, i - d)
        lmax = 2
        while delta(i, i + lmax * d) > dmin:
            lmax *= 2
        length, t = 0, lmax // 2
        while t >= 1:
            if delta(i, i + (length + t) * d) > dmin:
                length += t
            t //= 2
        j = i + length * d
        dnode = delta(i, j)
        s, t = 0, length
        while t > 1:
            t = (t + 1) // 2
            if delta(i, i + (s + t) * d) > dnode:
                s += t
        gamma = i + s * d + min(d, 0)
        lo, hi = min(i, j), max(i, j)
        left[i] = (n - 1 + gamma) if lo == gamma else gamma
        right[i] = (n - 1 + gamma + 1) if hi == gamma + 1 else gamma + 1
        first[i], last[i] = lo, hi
    return left, right, first, last


@pytest.mark.parametrize("code_mask", [0xFFFFFFFFFFFFFFFF, 0x0000000000FF0000])
def test_sort_gather_output_feeds_a_bvh_builder(ctx, oracle, mrs, code_mask):
    """The BVH-facing output (sorted codes, leaf -> primitive map, records in leaf order): a
    Karras-style radix tree built from it on the host is a proper binary tree whose node ranges
    nest, also when most codes are equal (the second mask leaves 256 distinct codes)."""
    n = 3000
    positions, indices, gen = mrs.scene.create_scene(1000, n, 7)
    verts, tris = mrs.scene.pack_vertices(positions), mrs.scene.pack_triangles(indices)
    codes = ctx.morton_encode_lut(to_dev(mrs.scene.create_morton_uniforms(gen), ctx.device), to_dev(verts, ctx.device),
                                  to_dev(tris, ctx.device))
    host_codes = (to_u32(codes).view(np.uint64) & np.uint64(code_mask))
    ka = to_dev(host_codes.view(np.uint32), ctx.device)
    kb = torch.empty_like(ka)
    recs = to_dev(tris.view(np.uint32), ctx.device)
    rout = torch.empty_like(recs)
    plan = mrs.SortPlan(ctx, n, key_words=2, payload_bytes=4, digit_bits=8)
    in_b, index = plan.sort_gather(ka, kb, recs, rout, 16)
    plan.check()
    sorted_codes = to_u32(kb if in_b else ka).view(np.uint64)
    leaf_to_prim = to_u32(index)[:n].astype(np.int64)
    # the three arrays agree with each other and with a stable sort
    order = np.argsort(host_codes, kind="stable")
    assert np.array_equal(sorted_codes, host_codes[order]) and np.array_equal(leaf_to_prim, order)
    assert np.array_equal(to_u32(rout).reshape(n, 4), tris.view(np.uint32).reshape(n, 4)[leaf_to_prim])
    left, right, first, last = _karras_tree(sorted_codes, 64)
    # node 0 is the root and covers every leaf; children split their parent's range exactly; every
    # leaf and every internal node other than the root has exactly one parent
    assert (first[0], last[0]) == (0, n - 1)
    seen = np.zeros(2 * n - 1, dtype=np.int64)
    for i in range(n - 1):
        for child, lo, hi in ((left[i], first[i], None), (right[i], None, last[i])):
            seen[child] += 1
            if child < n - 1:
                if lo is not None:
                    assert first[child] == lo and last[child] < last[i]
                else:
                    assert last[child] == hi and first[child] > first[i]
        lmax = last[left[i]] if left[i] < n - 1 else left[i] - (n - 1)
        rmin = first[right[i]] if right[i] < n - 1 else right[i] - (n - 1)
        assert lmax + 1 == rmin
    assert seen[0] == 0 and np.all(seen[1:] == 1)
    plan.close()


@pytest.mark.parametrize("bits,begin,end,n", [(8, 0, 32, 200003), (8, 0, 30, 200003), (4, 0, 30, 200003),
                                              (8, 3, 27, 200003), (8, 0, 30, 4096 * 530 + 11)])
def test_encode_for_sort_primes_the_plan(ctx, oracle, mrs, bits, begin, end, n):
    """The encoder that leaves the digit histograms in the plan: same keys as the plain encoder,
    and the sort that follows (which then skips its histogram launch) gives the checker's result;
    a later sort of other keys with the same plan is not affected.  The largest size has more
    4096-point tiles than the GPU has CUs, so every workgroup of the encoder counts several."""
    xyz = oracle.gen_points(5, n)
    mn, mx = oracle.minmax3(xyz)
    rk, ri = oracle.morton_encode30(xyz, mn, mx)
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=bits, begin_bit=begin, end_bit=end)
    plan.set_timing(True)
    keys, idx = plan.morton_encode30_for_sort(to_dev(xyz, ctx.device), mn, mx)
    assert np.array_equal(to_u32(keys), rk) and np.array_equal(to_u32(idx), ri)
    kb, vb = torch.empty_like(keys), torch.empty_like(idx)
    in_b = plan.sort_pairs_primed(keys, idx, kb, vb)
    plan.check()
    assert "global_histogram" not in [lab for lab, _ in plan.timings()]
    sk, sv = oracle.stable_sort_pairs(rk, ri, n, 1, 1, begin, end)
    assert np.array_equal(to_u32(kb if in_b else keys), sk) and np.array_equal(to_u32(vb if in_b else idx), sv)
    # the histograms are used once: a second primed call is refused, a plain sort computes its own
    other = make_keys(oracle, n, 1, "random")
    ok, ov = to_dev(other, ctx.device), to_dev(ri, ctx.device)
    with pytest.raises(mrs.MrsError):
        plan.sort_pairs_primed(ok, ov, kb, vb)
    in_b = plan.sort_pairs(ok, ov, kb, vb)
    plan.check()
    assert "global_histogram" in [lab for lab, _ in plan.timings()]
    sk, sv = oracle.stable_sort_pairs(other, ri, n, 1, 1, begin, end)
    assert np.array_equal(to_u32(kb if in_b else ok), sk) and np.array_equal(to_u32(vb if in_b else ov), sv)
    plan.close()


def test_left_over_histograms_are_never_used_by_accident(ctx, oracle, mrs):
    """Sequences that once made a sort run on stale digit histograms: the encoder, then (a) other
    keys written to the same address and a plain sort, (b) a partition on the plan and a sort of
    the encoder's keys.  Both sorts must compute their own histograms; a primed sort after the
    partition is refused."""
    n = 150001
    xyz = oracle.gen_points(11, n)
    mn, mx = oracle.minmax3(xyz)
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8)
    kb, vb = (torch.empty(n, dtype=torch.int32, device=ctx.device) for _ in range(2))
    # (a) same pointer, same n, different contents
    keys, idx = plan.morton_encode30_for_sort(to_dev(xyz, ctx.device), mn, mx)
    other = make_keys(oracle, n, 1, "random", seed=99)
    keys.copy_(to_dev(other, ctx.device))
    hv = to_u32(idx)
    in_b = plan.sort_pairs(keys, idx, kb, vb)
    plan.check()
    sk, sv = oracle.stable_sort_pairs(other, hv, n, 1, 1, 0, 32)
    assert np.array_equal(to_u32(kb if in_b else keys), sk) and np.array_equal(to_u32(vb if in_b else idx), sv)
    # (b) encoder, partition on the same plan, then sorts of the encoder's keys
    keys, idx = plan.morton_encode30_for_sort(to_dev(xyz, ctx.device), mn, mx, keys, idx)
    hk, hv = to_u32(keys), to_u32(idx)
    table = torch.zeros(256, dtype=torch.uint8, device=ctx.device)
    table[128:] = 1
    plan.partition(keys, idx, kb, vb, 22, 8, table, 2)
    with pytest.raises(mrs.MrsError):
        plan.sort_pairs_primed(keys, idx, kb, vb)
    in_b = plan.sort_pairs(keys, idx, kb, vb)
    plan.check()
    sk, sv = oracle.stable_sort_pairs(hk, hv, n, 1, 1, 0, 32)
    assert np.array_equal(to_u32(kb if in_b else keys), sk) and np.array_equal(to_u32(vb if in_b else idx), sv)
    plan.close()


def test_sort_from_misaligned_slices(ctx, oracle, mrs):
    """Key, payload and scratch buffers that start 4 bytes past a 16-byte boundary: the histogram
    and index kernels use 16-byte vector accesses, which gfx950 serves at element alignment."""
    n = 70001
    keys = make_keys(oracle, n, 1, "random", seed=5)
    vals = make_vals(oracle, n, 1)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
    big = [torch.zeros(n + 8, dtype=torch.int32, device=ctx.device) for _ in range(4)]
    ka, va, kb, vb = (t[1:n + 1] for t in big)
    assert ka.data_ptr() % 16 == 4
    for algo in (mrs.ALGO_ONESWEEP, mrs.ALGO_THREE_KERNEL):
        ka.copy_(to_dev(keys, ctx.device))
        va.copy_(to_dev(vals, ctx.device))
        plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8, algo=algo)
        in_b = plan.sort_pairs(ka, va, kb, vb)
        plan.check()
        assert np.array_equal(to_u32(kb if in_b else ka), rk) and np.array_equal(to_u32(vb if in_b else va), rv)
        plan.close()
    # mrs_sort_gather: the index buffers are filled with 16-byte stores
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=8)
    ka.copy_(to_dev(keys, ctx.device))
    recs = to_dev(vals, ctx.device)
    rout = torch.empty_like(recs)
    in_b, index = plan.sort_gather(ka, kb, recs, rout, 4, index_a=va, index_b=vb)
    plan.check()
    assert np.array_equal(to_u32(kb if in_b else ka), rk) and np.array_equal(to_u32(rout), rv)
    plan.close()
    # the bounds of the slices were respected
    for t in big:
        assert int(t[0]) == 0 and int(t[n + 1:].abs().sum()) == 0


def test_context_attributes(ctx):
    assert ctx.attribute(ctx.ATTR_COMPUTE_UNITS) == 256
    # gfx950 serves the lanes of a returning LDS atomic in lane order; the creation-time check sees it
    assert ctx.rank_mode == "atomic"
    # all 8 dies of an MI355X in its default partition mode receive workgroups
    assert ctx.attribute(ctx.ATTR_DIE_AWARE_TICKETS) == 1


def test_single_ticket_list(oracle, mrs):
    """Die-aware tickets switched off through the C ABI: one ticket list, same results."""
    ctx2 = mrs.Context(0)
    ctx2.set_attribute(ctx2.ATTR_DIE_AWARE_TICKETS, 0)
    assert ctx2.attribute(ctx2.ATTR_DIE_AWARE_TICKETS) == 0
    with pytest.raises(mrs.MrsError):
        ctx2.set_attribute(ctx2.ATTR_COMPUTE_UNITS, 0)  # found on the device, not a setting
    n = 300007
    keys = make_keys(oracle, n, 1, "morton30")
    vals = make_vals(oracle, n, 1)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
    for tile in (256, 8192):
        ko, vo, _ = gpu_sort(ctx2, mrs, keys, vals, n, 1, 1, 8, 0, 32, mrs.ALGO_ONESWEEP, tile_items=tile)
        assert np.array_equal(ko, rk) and np.array_equal(vo, rv)
    ctx2.set_attribute(ctx2.ATTR_DIE_AWARE_TICKETS, 1)
    assert ctx2.attribute(ctx2.ATTR_DIE_AWARE_TICKETS) == 1
    ctx2.close()


def test_concurrent_sorts_on_two_streams(oracle, mrs):
    """Two plans sorting at the same time on two streams of one device, every attribute at its
    default: a call that finds a die-aware sort of another stream still in flight runs with a
    single ticket list (the lowest unfinished tile of such a kernel is always held by a running
    workgroup), so the kernels can share the compute units in any interleaving."""
    ctx2 = mrs.Context(0)
    assert ctx2.attribute(ctx2.ATTR_DIE_AWARE_TICKETS) == 1
    n = (1 << 21) + 4321
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    plans = [mrs.SortPlan(ctx2, n), mrs.SortPlan(ctx2, n)]
    host = [(make_keys(oracle, n, 1, "morton30", seed=s), make_vals(oracle, n, 1)) for s in (1, 2)]
    want = [oracle.stable_sort_pairs(k, v, n, 1, 1, 0, 32) for k, v in host]
    rounds = 6
    work = [[(to_dev(k, ctx2.device), to_dev(v, ctx2.device)) for _ in range(rounds)] for k, v in host]
    scratch = [(torch.empty(n, dtype=torch.int32, device=ctx2.device), torch.empty(n, dtype=torch.int32, device=ctx2.device))
               for _ in range(2)]
    torch.cuda.synchronize()
    for r in range(rounds):
        for j in range(2):
            with torch.cuda.stream(streams[j]):
                ka, va = work[j][r]
                assert not plans[j].sort_pairs(ka, va, scratch[j][0], scratch[j][1], stream=streams[j])
    for j in range(2):
        plans[j].check(stream=streams[j])
    torch.cuda.synchronize()
    for j in range(2):
        for ka, va in work[j]:
            assert np.array_equal(to_u32(ka), want[j][0]) and np.array_equal(to_u32(va), want[j][1])
        plans[j].close()
    ctx2.close()


def test_sort_large_tile_u64_keys(ctx, oracle, mrs):
    """64-bit keys with a 4-byte payload on the 16384-element tile."""
    n = 7 * 16384 + 3
    keys = make_keys(oracle, n, 2, "random")
    vals = make_vals(oracle, n, 1)
    for algo in (mrs.ALGO_THREE_KERNEL, mrs.ALGO_ONESWEEP):
        ko, vo, _ = gpu_sort(ctx, mrs, keys, vals, n, 2, 1, 8, 0, 64, algo, tile_items=16384)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 2, 1, 0, 64)
        assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


# (mask of the low word, mask of the high word): which bytes of a 64-bit key vary
LIVE_BYTES = [
    (0xFFFFFFFF, 0x00000000),  # passes 4-7 move nothing (even count: all four are left out)
    (0xFFFFFFFF, 0x000000FF),  # passes 5-7 (odd count: the last of them runs after all)
    (0xFF00FFFF, 0xFFFFFFFF),  # pass 2 alone (runs after all)
    (0xFF0000FF, 0xFFFFFFFF),  # passes 1 and 2
    (0xFFFF00FF, 0x00FFFFFF),  # passes 1 and 7: the passes between read from the other side
    (0x00FFFF00, 0xFF0000FF),  # passes 0, 3, 5, 6
    (0x00000000, 0x00000000),  # every pass
    (0x000F0000, 0x00000000),  # one live nibble
]


@pytest.mark.parametrize("payload_words", [1, 4])
@pytest.mark.parametrize("lo_mask,hi_mask", LIVE_BYTES)
def test_u64_passes_left_out(ctx, oracle, mrs, aids, lo_mask, hi_mask, payload_words):
    """64-bit keys some of whose bytes are the same for every key: the device leaves those
    passes out (in pairs, so that the result still lands on the promised side) and the passes
    after them read from the other buffer pair.  Results must not change, with 8- and 4-bit
    digits, with an odd number of passes, and on the general path for a small input."""
    n = 50021
    keys = make_keys(oracle, n, 2, "random").reshape(n, 2).copy()
    keys[:, 0] = (keys[:, 0] & np.uint32(lo_mask)) | np.uint32(0x12345678 & ~lo_mask & 0xFFFFFFFF)
    keys[:, 1] = (keys[:, 1] & np.uint32(hi_mask)) | np.uint32(0x9ABCDEF0 & ~hi_mask & 0xFFFFFFFF)
    keys = keys.reshape(-1)
    vals = make_vals(oracle, n, payload_words)
    for bits, begin, end in ((8, 0, 64), (4, 0, 64), (8, 0, 40), (8, 8, 64)):
        ko, vo, in_b = gpu_sort(ctx, mrs, keys, vals, n, 2, payload_words, bits, begin, end, mrs.ALGO_ONESWEEP)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 2, payload_words, begin, end)
        assert np.array_equal(ko, rk) and np.array_equal(vo, rv), (bits, begin, end)
        assert in_b == ((-(-(end - begin) // bits)) % 2 == 1)
    # which passes the device reports as left out: the constant bytes, minus the last of them
    # when their number is odd
    constant = [p for p in range(8) if (((hi_mask << 32) | lo_mask) >> (8 * p)) & 0xFF == 0]
    if len(constant) % 2:
        constant = constant[:-1]
    plan = mrs.SortPlan(ctx, n, key_words=2, payload_bytes=4 * payload_words)
    ka, va = to_dev(keys, ctx.device), to_dev(vals, ctx.device)
    plan.sort_pairs(ka, va, torch.empty_like(ka), torch.empty_like(va))
    assert plan.passes_left_out() == sum(1 << p for p in constant)
    plan.sort_pairs(ka[: 2 * 100], va[: 100 * payload_words], torch.empty_like(ka), torch.empty_like(va), n=100)
    assert plan.passes_left_out() == 0  # the one-workgroup path keeps no record
    plan.close()
    aids(ctx.ATTR_TEST_NO_SMALL_SORT, 1)
    m = 777
    ko, vo, _ = gpu_sort(ctx, mrs, keys[: 2 * m], vals[: m * payload_words], m, 2, payload_words, 8, 0, 64, mrs.ALGO_ONESWEEP)
    rk, rv = oracle.stable_sort_pairs(keys[: 2 * m], vals[: m * payload_words], m, 2, payload_words, 0, 64)
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


@pytest.mark.parametrize("tile", [256, 1024])
@pytest.mark.parametrize("dist", ["random", "all_equal"])
def test_onesweep_long_chains(ctx, oracle, mrs, tile, dist):
    """Thousands of tiles: the look-back walks over many windows and, with all keys equal,
    every tile's count of one digit goes through the chain."""
    n = (1 << 20) + 123
    keys = make_keys(oracle, n, 1, dist)
    vals = make_vals(oracle, n, 1)
    ko, vo, _ = gpu_sort(ctx, mrs, keys, vals, n, 1, 1, 8, 0, 32, mrs.ALGO_ONESWEEP, tile_items=tile)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


@pytest.mark.parametrize("begin,end,bits", [(0, 30, 8), (4, 20, 8), (0, 30, 4), (3, 11, 8), (0, 8, 8)])
def test_sort_bit_ranges(ctx, oracle, mrs, begin, end, bits):
    n = 30000
    keys = make_keys(oracle, n, 1, "random")
    vals = make_vals(oracle, n, 1)
    for algo in (mrs.ALGO_THREE_KERNEL, mrs.ALGO_ONESWEEP):
        ko, vo, in_b = gpu_sort(ctx, mrs, keys, vals, n, 1, 1, bits, begin, end, algo)
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, begin, end)
        assert np.array_equal(ko, rk) and np.array_equal(vo, rv)
        assert in_b == ((-(-(end - begin) // bits)) % 2 == 1)


def test_plan_capacity_larger_than_n(ctx, oracle, mrs):
    """A plan sized for more elements sorts fewer, repeatedly (state is reset per call)."""
    cap = 100000
    plan = mrs.SortPlan(ctx, cap, key_words=1, payload_bytes=4, digit_bits=8, algo=mrs.ALGO_ONESWEEP)
    for n in (cap, 12345, 1, 77777):
        keys = make_keys(oracle, n, 1, "random", seed=n)
        vals = make_vals(oracle, n, 1)
        ka, va = to_dev(keys, ctx.device), to_dev(vals, ctx.device)
        kb, vb = torch.empty_like(ka), torch.empty_like(va)
        in_b = plan.sort_pairs(ka, va, kb, vb, n=n)
        plan.check()
        rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
        assert np.array_equal(to_u32(kb if in_b else ka), rk)
        assert np.array_equal(to_u32(vb if in_b else va), rv)


def test_config1_2pow20_4bit_three_kernel(ctx, oracle, mrs):
    """2^20 u32 Morton keys, 4-bit-digit LSD radix (8 passes)."""
    n = 1 << 20
    keys = make_keys(oracle, n, 1, "morton30")
    vals = make_vals(oracle, n, 1)
    ko, vo, in_b = gpu_sort(ctx, mrs, keys, vals, n, 1, 1, 4, 0, 32, mrs.ALGO_THREE_KERNEL)
    rk, rv = oracle.stable_sort_pairs(keys, vals, n, 1, 1, 0, 32)
    assert not in_b
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


# ---- golden vectors --------------------------------------------------------------------------
def test_golden_vectors_on_gpu(ctx, mrs):
    files = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))
    assert files
    for f in files:
        g = np.load(f)
        kind = str(g["kind"])
        if kind == "sort":
            kw, pw, n = int(g["key_words"]), int(g["payload_words"]), int(g["n"])
            vals = g["vals_in"] if pw else None
            for algo in (mrs.ALGO_THREE_KERNEL, mrs.ALGO_ONESWEEP):
                for tile in (int(g["tile"]), 0):
                    ko, vo, in_b = gpu_sort(ctx, mrs, g["keys_in"], vals, n, kw, pw, int(g["bits"]),
                                            int(g["begin_bit"]), int(g["end_bit"]), algo, tile_items=tile)
                    assert np.array_equal(ko, g["keys_out"]), f
                    if pw:
                        assert np.array_equal(vo, g["vals_out"]), f
                    assert in_b == bool(g["result_in_b"]), f
        elif kind == "encode30":
            k, i = ctx.morton_encode30(to_dev(g["xyz"], ctx.device), g["mn"], g["mx"])
            assert np.array_equal(to_u32(k), g["keys"]) and np.array_equal(to_u32(i), g["idx"]), f
        elif kind == "encode_lut":
            codes = ctx.morton_encode_lut(to_dev(g["blob"], ctx.device), to_dev(g["vertices"], ctx.device),
                                          to_dev(g["triangles"], ctx.device))
            assert np.array_equal(to_u32(codes), g["codes"]), f


# ---- the reference's entry points ---------------------------------------------------------------
@pytest.mark.parametrize("algo_name", ["three_kernel", "onesweep"])
def test_run_headless(ctx, oracle, mrs, algo_name):
    """run_headless on the shipped scene size: the reference's adjacent-pair check
    (src/main.rs:831-864) plus full key AND payload parity with the oracle."""
    algo = mrs.ALGO_ONESWEEP if algo_name == "onesweep" else mrs.ALGO_THREE_KERNEL
    res = mrs.run_headless(0, algo=algo, tile_items=256 if algo_name == "three_kernel" else 0)
    assert res["sorted"] and not res["all_zero"] and res["times_unsorted"] == 0
    host = res["buffers"].host
    verts = mrs.scene.pack_vertices(host["positions"])
    tris = mrs.scene.pack_triangles(host["indices"])
    codes = oracle.morton_encode_lut(host["uniforms"], verts, tris)
    n = host["indices"].shape[0]
    rk, rv, in_b = oracle.lsd_sort(codes, tris.view(np.uint32), n, 2, 4, 8, 0, 64, 256)
    assert in_b == res["buffers"].result_in_2 == False  # noqa: E712  (8 passes: result in the first buffers)
    assert np.array_equal(res["keys"].view(np.uint32), rk)
    assert np.array_equal(res["triangles"], rv.reshape(-1, 4)[:, :3])


def test_run_headless_chacha8_scene(ctx, oracle, mrs):
    """The shipped scene size from the restated ChaCha8 stream: sorted, and equal to the checker."""
    res = mrs.run_headless(0, scene_rng="chacha8")
    assert res["sorted"] and not res["all_zero"]
    host = res["buffers"].host
    verts = mrs.scene.pack_vertices(host["positions"])
    tris = mrs.scene.pack_triangles(host["indices"])
    codes = oracle.morton_encode_lut(host["uniforms"], verts, tris)
    n = host["indices"].shape[0]
    rk, rv, _ = oracle.lsd_sort(codes, tris.view(np.uint32), n, 2, 4, 8, 0, 64, 256)
    assert np.array_equal(res["keys"].view(np.uint32), rk)
    assert np.array_equal(res["triangles"], rv.reshape(-1, 4)[:, :3])


@pytest.mark.parametrize("num_triangles", [3000, 70001])
def test_run_headless_gathered_payload(ctx, oracle, mrs, num_triangles):
    """The same entry point with the TriangleIndices moved once at the end instead of through
    every pass: identical codes and triangles."""
    carried = mrs.run_headless(0, num_triangles=num_triangles)
    gathered = mrs.run_headless(0, num_triangles=num_triangles, gather_payload=True)
    assert gathered["sorted"] and gathered["times_unsorted"] == 0
    assert np.array_equal(carried["keys"], gathered["keys"])
    assert np.array_equal(carried["triangles"], gathered["triangles"])
    assert gathered["buffers"].triangles_in_2


# ---- multi-GPU building blocks ----------------------------------------------------------------------
def test_bucket_histogram(ctx, oracle):
    n = 300001
    keys = make_keys(oracle, n, 1, "morton30")
    for shift, bits in ((18, 12), (22, 8), (0, 4)):
        counts = ctx.bucket_histogram(to_dev(keys, ctx.device), n, 1, shift, bits).cpu().numpy()
        ref = np.bincount((keys >> np.uint32(shift)) & np.uint32((1 << bits) - 1), minlength=1 << bits)
        assert np.array_equal(counts, ref)
    # a key array that is not 16-byte aligned (no vector loads), tiny inputs, 64-bit keys
    dev = to_dev(keys, ctx.device)
    for first, m in ((1, n - 1), (3, 5), (2, 4099), (0, 1), (0, 3)):
        counts = ctx.bucket_histogram(dev[first:], m, 1, 20, 12).cpu().numpy()
        assert np.array_equal(counts, np.bincount(keys[first: first + m] >> np.uint32(20), minlength=4096))
    k64 = make_keys(oracle, 70001, 2, "random")
    hi = k64.reshape(-1, 2)[:, 1]
    for first, m in ((0, 70001), (1, 70000)):
        counts = ctx.bucket_histogram(to_dev(k64, ctx.device)[2 * first:], m, 2, 52, 12).cpu().numpy()
        assert np.array_equal(counts, np.bincount(hi[first: first + m] >> np.uint32(20), minlength=4096))


@pytest.mark.parametrize("bits_plan", [4, 8])
def test_partition(ctx, oracle, mrs, bits_plan):
    n = 250007
    keys = make_keys(oracle, n, 1, "morton30")
    vals = make_vals(oracle, n, 1)
    shift, bits, parts = 18, 12, 8
    bucket = (keys >> np.uint32(shift)) & np.uint32((1 << bits) - 1)
    bmap = (np.arange(1 << bits) * parts // (1 << bits)).astype(np.uint8)
    bmap[5] = 7  # a non-monotone entry: the map is a table, not a shift
    plan = mrs.SortPlan(ctx, n, key_words=1, payload_bytes=4, digit_bits=bits_plan, algo=mrs.ALGO_ONESWEEP)
    ka, va = to_dev(keys, ctx.device), to_dev(vals, ctx.device)
    kb, vb = torch.empty_like(ka), torch.empty_like(va)
    counts = plan.partition(ka, va, kb, vb, shift, bits, torch.from_numpy(bmap).to(ctx.device), parts)
    plan.check()
    dest = bmap[bucket]
    order = np.argsort(dest, kind="stable")
    assert np.array_equal(to_u32(kb), keys[order])
    assert np.array_equal(to_u32(vb), vals[order])
    assert np.array_equal(to_u32(counts)[:parts], np.bincount(dest, minlength=parts))


# ---- error behaviour ------------------------------------------------------------------------------------
def test_errors_are_reported(ctx, mrs):
    with pytest.raises(mrs.MrsError):
        mrs.SortPlan(ctx, 1000, key_words=3)
    with pytest.raises(mrs.MrsError):
        mrs.SortPlan(ctx, 1000, digit_bits=5)
    with pytest.raises(mrs.MrsError):
        mrs.SortPlan(ctx, 0)
    with pytest.raises(mrs.MrsError):
        mrs.SortPlan(ctx, 1000, key_words=1, payload_bytes=16, tile_items=16384)
    with pytest.raises(mrs.MrsError):
        mrs.SortPlan(ctx, 1000, key_words=2, payload_bytes=4, tile_items=8192)
    plan = mrs.SortPlan(ctx, 1000)
    k = torch.zeros(2000, dtype=torch.int32, device=ctx.device)
    with pytest.raises(mrs.MrsError):
        plan.sort_pairs(k, k, k, k, n=1000)  # A and B must differ
    with pytest.raises(mrs.MrsError):
        plan.sort_pairs(k, k, k.clone(), k.clone(), n=2000)  # beyond capacity
    table = torch.zeros(1 << 13, dtype=torch.uint8, device=ctx.device)
    with pytest.raises(mrs.MrsError):
        plan.partition(k, k.clone(), k.clone(), k.clone(), 0, 13, table, 2, n=1000)  # tables hold at most 2^12 entries
    with pytest.raises(mrs.MrsError):
        plan.partition(k, k.clone(), k.clone(), k.clone(), 0, 12, table, 17, n=1000)  # at most 16 parts


@pytest.mark.parametrize("extra", [[], ["--gather"], ["--algo", "three_kernel"], ["--triangles", "70001", "--vertices", "5000", "--gather"],
                                   ["--triangles", "70001", "--vertices", "5000", "--passes", "5"],
                                   ["--frames", "6"], ["--frames", "3", "--gather", "--triangles", "70001", "--vertices", "5000"]])
def test_cpp_host_driver(extra):
    """mrs_run, the compiled counterpart of run_headless (src/main.rs:690-1033): scene, encode,
    sort, read back, the reference's adjacent-pair check, and equality with std::stable_sort of the
    encoded input for codes and triangles, carrying the triangles through the passes or gathering them."""
    import json
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       "wgpu_morton_code_radix_sort_test_amd", "csrc", "mrs_run")
    assert os.path.exists(exe), "mrs_run is built by `make` in csrc (and by __graft_entry__.build())"
    r = subprocess.run([exe] + extra, capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["sorted"] and out["matches_stable_sort"] and not out.get("all_zero", False)
    if "--frames" in extra:
        # the per-frame caller (State::render, src/state.rs:109-148) on buffers and a plan made once
        assert out["frames"] == int(extra[extra.index("--frames") + 1]) and out["device_ms_per_frame"] > 0
