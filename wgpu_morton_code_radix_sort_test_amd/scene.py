"""Host-side input construction: the role of `create_scene` and `create_morton_uniforms`.

The reference draws its scene from ChaCha8Rng(seed 7) (src/main.rs:124-146) and gets
its LUTs from the `extended_morton_coder` crate (src/main.rs:165); neither is
available here.  This module uses the repository's own counter-based generator
(the same hash the HIP generator kernels and the CPU checker implement) and a
documented stand-in LUT generator, and packs everything into the reference's
byte layouts.
"""
import numpy as np

# src/constants.rs:1-21
NUM_VERTICES = 1000
NUM_TRIANGLES = 3000
BITS_PER_PASS = 8
HISTOGRAM_SIZE = 1 << BITS_PER_PASS
WORKGROUP_SIZE = 256
ITEMS_PER_HISTOGRAM_PASS = 256
ITEMS_PER_LARGE_PREFIX = 256
MAX_ITEMS_IN_SMALL_PREFIX = 1024
NUM_PASSES = 8
RNG_SEED = 7

MORTON_UNIFORMS_BYTES = 51248
LUT_WORDS = 4608
SIZE_LUT_WORDS = 8192
VERTEX_STRIDE = 48
TRIANGLE_STRIDE = 16

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


# ---- src/utilities.rs:10-55 -------------------------------------------------
def div_ceil(l, r):
    return (l + r - 1) // r


def calculate_number_of_workgroups(element_count, workgroup_size):
    return div_ceil(element_count, workgroup_size)


def round_up(l, r):
    rem = l % r
    return l if rem == 0 else l + r - rem


def calculate_num_items_prefix_buffers(num_elements):
    """Level sizes of the reference's hierarchical prefix sum (src/main.rs:201-233)."""
    items = calculate_number_of_workgroups(num_elements, ITEMS_PER_HISTOGRAM_PASS) * HISTOGRAM_SIZE
    items = round_up(items, ITEMS_PER_LARGE_PREFIX)
    if items <= MAX_ITEMS_IN_SMALL_PREFIX:
        return []
    items //= 256
    items = round_up(items, ITEMS_PER_LARGE_PREFIX)
    out = []
    while items > MAX_ITEMS_IN_SMALL_PREFIX:
        out.append(items)
        items //= ITEMS_PER_LARGE_PREFIX
        items = round_up(items, ITEMS_PER_LARGE_PREFIX)
    out.append(items)
    return out


# ---- counter-based generator ---------------------------------------------------
def _mix64(z):
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= _M1
    z ^= z >> np.uint64(27)
    z *= _M2
    z ^= z >> np.uint64(31)
    return z


def rand_u32(seed, stream, n, start=0):
    """hash(seed, stream, i) for i in [start, start+n): uniform u32."""
    with np.errstate(over="ignore"):
        key = _mix64(np.array([np.uint64(seed) + _GOLDEN * np.uint64(stream + 1)], dtype=np.uint64))[0]
        idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        return (_mix64(key + _GOLDEN * idx) >> np.uint64(32)).astype(np.uint32)


def gen_points(seed, n):
    """[n, 3] f32 points uniform in [0, 1): top 24 bits of the hash times 2^-24."""
    cols = [(rand_u32(seed, c, n) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0) for c in range(3)]
    return np.stack(cols, axis=1)


# ---- scene ------------------------------------------------------------------------
class StandInMortonGenerator:
    """Stand-in for `extended_morton_coder::MortonCodeGenerator::new(min, max)`.

    The crate (0.1.0) is not vendored in the reference, so its table contents are
    unknown; this produces tables of the same SHAPE (lut [[u64; 256]; 9],
    size_lut [u64; 4096]) with a simple, documented meaning: table 3i+c
    spreads bits 0..2 of its index (= bits 3i..3i+2 of coordinate c, which is how
    the shader addresses it, src/shaders/morton_code.wgsl:113-115) to every third
    bit of a 27-bit Morton code held above 12 size bits; size_lut[j] = j.
    Codes made with it are NOT the reference's codes.
    """

    SIZE_BITS = 12

    def __init__(self, scene_min, scene_max):
        scene_min = np.asarray(scene_min, dtype=np.float32)
        scene_max = np.asarray(scene_max, dtype=np.float32)
        extent = np.maximum(scene_max - scene_min, np.float32(1e-30))
        self.offset = scene_min
        self.multiplier = (np.float32(512.0) / extent).astype(np.float32)
        diag = np.float32(np.sqrt(np.sum(extent.astype(np.float64) ** 2)))
        self.size_multiplier = np.float32(4095.0) / diag
        self.morton_index_scale = np.float32(1.0)
        lut = np.zeros((9, 256), dtype=np.uint64)
        for i in range(3):
            for c in range(3):
                for b in range(256):
                    code = 0
                    for k in range(3):
                        if (b >> k) & 1:
                            code |= 1 << (3 * (3 * i + k) + (2 - c))
                    lut[3 * i + c, b] = code << self.SIZE_BITS
        self.lut = lut
        self.size_lut = np.arange(4096, dtype=np.uint64)


def create_morton_uniforms(gen):
    """Packs the generator into the MortonCodeConstants blob (src/main.rs:177-198).

    u64 tables are flattened to (lo, hi) u32 pairs; std430 offsets as in
    src/shaders/morton_code.wgsl:19-26.
    """
    blob = np.zeros(MORTON_UNIFORMS_BYTES, dtype=np.uint8)
    lut = np.empty(LUT_WORDS, dtype=np.uint32)
    flat = gen.lut.reshape(-1)
    lut[0::2] = (flat & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    lut[1::2] = (flat >> np.uint64(32)).astype(np.uint32)
    assert lut.size == 4608, "lut is the wrong length"
    size_lut = np.empty(SIZE_LUT_WORDS, dtype=np.uint32)
    size_lut[0::2] = (gen.size_lut & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    size_lut[1::2] = (gen.size_lut >> np.uint64(32)).astype(np.uint32)
    assert size_lut.size == 8192
    blob[0:18432] = lut.view(np.uint8)
    blob[18432:51200] = size_lut.view(np.uint8)
    blob[51200:51204] = np.array([gen.morton_index_scale], dtype=np.float32).view(np.uint8)
    blob[51216:51228] = np.asarray(gen.offset, dtype=np.float32).view(np.uint8)
    blob[51228:51232] = np.array([gen.size_multiplier], dtype=np.float32).view(np.uint8)
    blob[51232:51244] = np.asarray(gen.multiplier, dtype=np.float32).view(np.uint8)
    return blob


def random_uniforms_blob(seed, offset, multiplier, size_multiplier):
    """A blob with RANDOM table words: for LUT-agnostic encoder tests."""
    blob = np.zeros(MORTON_UNIFORMS_BYTES, dtype=np.uint8)
    blob[0:51200] = rand_u32(seed, 17, (LUT_WORDS + SIZE_LUT_WORDS)).view(np.uint8)
    blob[51200:51204] = np.array([1.0], dtype=np.float32).view(np.uint8)
    blob[51216:51228] = np.asarray(offset, dtype=np.float32).view(np.uint8)
    blob[51228:51232] = np.array([size_multiplier], dtype=np.float32).view(np.uint8)
    blob[51232:51244] = np.asarray(multiplier, dtype=np.float32).view(np.uint8)
    return blob


def pack_vertices(positions):
    """[n,3] f32 -> stride-48 Vertex records: position @0, normal (0,1,0) @16, uv (0,0.5) @32
    (src/types.rs:12-17, src/main.rs:132-136)."""
    n = positions.shape[0]
    rec = np.zeros((n, 12), dtype=np.float32)
    rec[:, 0:3] = positions
    rec[:, 4:7] = np.array([0.0, 1.0, 0.0], dtype=np.float32)
    rec[:, 8:10] = np.array([0.0, 0.5], dtype=np.float32)
    return rec.reshape(-1).view(np.uint8)


def pack_triangles(indices):
    """[n,3] u32 -> stride-16 TriangleIndices records (src/types.rs:30-36)."""
    n = indices.shape[0]
    rec = np.zeros((n, 4), dtype=np.uint32)
    rec[:, 0:3] = indices
    return rec.reshape(-1).view(np.uint8)


class ChaCha8Rng:
    """`rand_chacha::ChaCha8Rng::seed_from_u64` as the reference seeds it (src/main.rs:124),
    restated from the published algorithms; NOT checked against the Rust crates (no Rust
    toolchain or crate sources here), so scenes made with it are labelled unverified.

    * seed expansion (rand_core 0.6 `SeedableRng::seed_from_u64`): eight PCG32 outputs
      (multiplier 6364136223846793005, increment 11634580027462260723, XSH-RR), little
      endian, form the 32-byte key;
    * ChaCha with 8 rounds, 64-bit block counter in words 12-13, stream id 0 in words 14-15;
      the output words of consecutive blocks are consumed in order as `next_u32`;
    * `gen::<f32>()` (rand 0.8 `Standard`): `(next_u32() >> 8) * 2^-24`;
    * `Uniform::new(0, n).sample()` for u32 (rand 0.8.5 `UniformInt`): widening multiply of a
      u32 draw by n, accepted when the low half is at most `u32::MAX - ((2^32 - n) % n)`.
    """

    _SIGMA = (0x61707865, 0x3320646E, 0x79622D32, 0x6B206574)

    def __init__(self, seed_u64=None, key_bytes=None):
        if key_bytes is None:
            key_bytes = self.seed_bytes_from_u64(seed_u64)
        self.key = np.frombuffer(bytes(key_bytes), dtype="<u4").astype(np.uint32)
        self.counter = 0
        self._words = np.zeros(0, dtype=np.uint32)
        self._at = 0

    @staticmethod
    def seed_bytes_from_u64(state):
        mask = (1 << 64) - 1
        out = bytearray()
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & mask
            xorshifted = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
            rot = state >> 59
            x = ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & 0xFFFFFFFF
            out += int(x).to_bytes(4, "little")
        return bytes(out)

    def blocks(self, first_counter, count):
        """`count` consecutive 16-word output blocks starting at block `first_counter`."""
        ctr = np.arange(first_counter, first_counter + count, dtype=np.uint64)
        x = np.empty((16, count), dtype=np.uint32)
        for i in range(4):
            x[i] = self._SIGMA[i]
        for i in range(8):
            x[4 + i] = self.key[i]
        x[12] = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        x[13] = (ctr >> np.uint64(32)).astype(np.uint32)
        x[14] = 0
        x[15] = 0
        init = x.copy()

        def rotl(v, r):
            return (v << np.uint32(r)) | (v >> np.uint32(32 - r))

        def quarter(a, b, c, d):
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16)
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12)
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8)
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7)

        with np.errstate(over="ignore"):
            for _ in range(4):  # 8 rounds = 4 double rounds
                quarter(0, 4, 8, 12); quarter(1, 5, 9, 13); quarter(2, 6, 10, 14); quarter(3, 7, 11, 15)
                quarter(0, 5, 10, 15); quarter(1, 6, 11, 12); quarter(2, 7, 8, 13); quarter(3, 4, 9, 14)
            x += init
        return x.T.reshape(-1)  # block-major, word order inside a block

    def next_u32_array(self, count):
        """The next `count` outputs of `next_u32`."""
        out = np.empty(count, dtype=np.uint32)
        filled = 0
        while filled < count:
            if self._at == self._words.size:
                nblocks = max(4, min(1 << 16, (count - filled + 15) // 16))
                self._words = self.blocks(self.counter, nblocks)
                self.counter += nblocks
                self._at = 0
            take = min(count - filled, self._words.size - self._at)
            out[filled:filled + take] = self._words[self._at:self._at + take]
            self._at += take
            filled += take
        return out

    def gen_f32(self, count):
        return (self.next_u32_array(count) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / (1 << 24))

    def sample_uniform_u32(self, high, count):
        """`Uniform::new(0, high)` sampled `count` times: the accepted draws of the stream, in order."""
        zone = 0xFFFFFFFF - ((1 << 32) - high) % high
        out = np.empty(count, dtype=np.uint32)
        filled = 0
        while filled < count:
            # never more draws than results still missing: the stream position after the call is
            # then the one the scalar loop leaves
            v = self.next_u32_array(count - filled).astype(np.uint64) * np.uint64(high)
            kept = (v[(v & np.uint64(0xFFFFFFFF)) <= np.uint64(zone)] >> np.uint64(32)).astype(np.uint32)
            out[filled:filled + kept.size] = kept
            filled += kept.size
        return out


def create_scene_chacha8(num_vertices=NUM_VERTICES, num_triangles=NUM_TRIANGLES, seed=RNG_SEED):
    """`create_scene` with the reference's random stream (src/main.rs:124-161): positions are
    three `gen::<f32>()` per vertex, then three `Uniform(0, NUM_VERTICES)` draws per triangle from
    the same generator.  Unverified against the Rust crates: see ChaCha8Rng."""
    rng = ChaCha8Rng(seed)
    positions = rng.gen_f32(3 * num_vertices).reshape(num_vertices, 3)
    indices = rng.sample_uniform_u32(num_vertices, 3 * num_triangles).reshape(num_triangles, 3)
    used = positions[indices.reshape(-1)]
    return positions, indices, StandInMortonGenerator(used.min(axis=0), used.max(axis=0))


def create_scene(num_vertices=NUM_VERTICES, num_triangles=NUM_TRIANGLES, seed=RNG_SEED):
    """Random vertices in [0,1)^3 and triangles of random vertex indices; scene bounds
    over the vertices that triangles use (src/main.rs:114-173).

    Returns (positions [nv,3] f32, indices [nt,3] u32, generator).
    """
    positions = gen_points(seed, num_vertices)
    # Uniform(0, NUM_VERTICES): widening multiply of a u32 draw (no rejection step;
    # the bias is < 2^-22 for the sizes used here)
    draws = np.stack([rand_u32(seed, 3 + c, num_triangles) for c in range(3)], axis=1).astype(np.uint64)
    indices = ((draws * np.uint64(num_vertices)) >> np.uint64(32)).astype(np.uint32)
    used = positions[indices.reshape(-1)]
    scene_min = used.min(axis=0)
    scene_max = used.max(axis=0)
    return positions, indices, StandInMortonGenerator(scene_min, scene_max)
