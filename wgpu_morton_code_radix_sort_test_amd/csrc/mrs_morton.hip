// Morton encoders and the synthetic input generators.
//
// Built with -ffp-contract=off: the f32 expressions below are evaluated one
// IEEE operation at a time, as the CPU checker does.
#include "mrs_internal.h"

namespace mrs {

// ---- extended Morton code from the LUT blob --------------------------------------
// One thread per triangle.  The reference's morton_code_unrolled gives each
// thread eight consecutive triangles of one of eight slices
// (src/shaders/morton_code.wgsl:96-99); the per-item arithmetic
// (src/shaders/morton_code.wgsl:103-127) is what is kept here.
//
// The 18 KiB coordinate table is copied into LDS once per workgroup (the
// reference's README notes the pass was limited by the table lookups); the
// 32 KiB size table is read once per triangle and stays in global memory/L2.
constexpr int kLutWords = 4608;
constexpr int kEncodeThreads = 256;

// f32 -> u32 toward zero; NaN and negatives give 0, values >= 2^32 give 2^32-1.
__device__ __forceinline__ uint32_t f2u_sat(float f)
{
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return static_cast<uint32_t>(f);
}

struct MortonUniformsTail {  // byte offsets inside the blob, src/shaders/morton_code.wgsl:19-26
    static constexpr int kSizeLut = 18432;
    static constexpr int kOffset = 51216;
    static constexpr int kSizeMultiplier = 51228;
    static constexpr int kMultiplier = 51232;
};

__global__ __launch_bounds__(kEncodeThreads) void morton_encode_lut_kernel(
    const unsigned char* __restrict__ blob, const unsigned char* __restrict__ vertices,
    unsigned long long num_vertices, const uint4* __restrict__ triangles, unsigned long long n,
    uint2* __restrict__ codes)
{
    __shared__ uint32_t s_lut[kLutWords];
    const uint4* lut4 = reinterpret_cast<const uint4*>(blob);
    for (int j = threadIdx.x; j < kLutWords / 4; j += kEncodeThreads)
        reinterpret_cast<uint4*>(s_lut)[j] = lut4[j];
    __syncthreads();

    const uint32_t* __restrict__ size_lut = reinterpret_cast<const uint32_t*>(blob + MortonUniformsTail::kSizeLut);
    const float* offset = reinterpret_cast<const float*>(blob + MortonUniformsTail::kOffset);
    const float* multiplier = reinterpret_cast<const float*>(blob + MortonUniformsTail::kMultiplier);
    const float size_multiplier = *reinterpret_cast<const float*>(blob + MortonUniformsTail::kSizeMultiplier);

    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * kEncodeThreads;
    for (unsigned long long t = static_cast<unsigned long long>(blockIdx.x) * kEncodeThreads + threadIdx.x; t < n;
         t += stride) {
        const uint4 tri = triangles[t];
        float v[3][3];
        const uint32_t vi[3] = {tri.x, tri.y, tri.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // wgpu clamps out-of-range indices (robust buffer access); a vertex
            // index past the buffer reads the last vertex here
            unsigned long long idx = vi[k] < num_vertices ? vi[k] : num_vertices - 1;
            const float* p = reinterpret_cast<const float*>(vertices + idx * MRS_VERTEX_STRIDE);
            v[k][0] = p[0]; v[k][1] = p[1]; v[k][2] = p[2];
        }
        uint32_t p[3];
        float d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float mn = fminf(fminf(v[0][c], v[1][c]), v[2][c]);
            const float mx = fmaxf(fmaxf(v[0][c], v[1][c]), v[2][c]);
            const float centre = (mn + mx) * 0.5f;
            p[c] = f2u_sat((centre - offset[c]) * multiplier[c] - 0.001f);
            d[c] = mx - mn;
        }
        // sqrtf is correctly rounded under -fhip-fp32-correctly-rounded-divide-sqrt; __fsqrt_rn is the
        // 1-ulp hardware instruction
        const float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        uint32_t j = f2u_sat(len * size_multiplier);
        j = j > 4095u ? 4095u : j;  // the size table has 4096 entries
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // shift by i*3, as written at src/shaders/morton_code.wgsl:113-115
                const uint32_t byte = (p[c] >> (3 * i)) & 0xFFu;
                const uint2 e = *reinterpret_cast<const uint2*>(&s_lut[(c + 3 * i) * 512 + byte * 2]);
                lo |= e.x;
                hi |= e.y;
            }
        }
        const uint2 s = *reinterpret_cast<const uint2*>(&size_lut[j * 2]);
        codes[t] = make_uint2(lo | s.x, hi | s.y);
    }
}

// ---- 30-bit Morton key of a point -----------------------------------------------------
__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

struct Bounds3 {
    float mn[3];
    float mx[3];
};

// Digit histograms of the keys being written, for the sort that follows (see
// mrs_morton_encode30_for_sort): same content as global_histogram_kernel produces.
struct EncodeHistogram {
    uint32_t* ghist;  // [num_passes][1 << bits], zeroed
    uint32_t begin_bit, end_bit, bits, num_passes;
};

// Each thread encodes 4 consecutive points = 48 bytes.  A workgroup of 1024 threads takes a tile of
// 4096 points: its 48 KiB are read as three rows of 16 KiB, one fully coalesced 16-byte load per
// thread and row (a wave reads 1 KiB of consecutive addresses instead of 64 pieces 48 bytes apart),
// and handed to the threads through LDS.  One workgroup per CU, nontemporal loads and stores (the
// stream is touched once): 0.90 ms against 1.07 ms for direct strided loads at 2^28 points.
// The points after the last full tile take the direct path in one workgroup.
constexpr int kEncode30Threads = 1024;
constexpr int kEncode30TilePoints = kEncode30Threads * 4;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool HIST>
__global__ __launch_bounds__(kEncode30Threads) void morton_encode30_kernel(
    const float* __restrict__ xyz, unsigned long long n, const Bounds3 b, uint32_t* __restrict__ keys,
    uint32_t* __restrict__ idx, const EncodeHistogram h)
{
    __shared__ f32x4 s_tile[2][3 * kEncode30Threads];
    __shared__ uint32_t s_hist[HIST ? 1024 : 1];  // 4 x 256 or 8 x 16 counters (32-bit keys)
    if constexpr (HIST) {
        s_hist[threadIdx.x] = 0;
        __syncthreads();
    }
    auto count = [&](uint32_t k) {
        for (uint32_t p = 0; p < h.num_passes; ++p) {
            const uint32_t shift = h.begin_bit + p * h.bits;
            const uint32_t width = h.end_bit - shift < h.bits ? h.end_bit - shift : h.bits;
            atomicAdd(&s_hist[(p << h.bits) + ((k >> shift) & ((1u << width) - 1u))], 1u);
        }
    };
    const float ext[3] = {b.mx[0] - b.mn[0], b.mx[1] - b.mn[1], b.mx[2] - b.mn[2]};
    auto encode4 = [&](const float (&f)[12], uint32_t (&k)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t u[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float t = (f[3 * q + c] - b.mn[c]) / ext[c];
                u[c] = f2u_sat(fminf(fmaxf(t * 1024.0f, 0.0f), 1023.0f));
            }
            k[q] = (spread10(u[0]) << 2) | (spread10(u[1]) << 1) | spread10(u[2]);
        }
    };

    const unsigned long long full_tiles = n / kEncode30TilePoints;
    auto load_rows = [&](unsigned long long tile, f32x4 (&row)[3]) {
        const f32x4* p = reinterpret_cast<const f32x4*>(xyz) + tile * (3 * kEncode30Threads);
#pragma unroll
        for (int r = 0; r < 3; ++r) row[r] = __builtin_nontemporal_load(p + r * kEncode30Threads + threadIdx.x);
    };
    f32x4 row[3];
    if (blockIdx.x < full_tiles) load_rows(blockIdx.x, row);
    int buf = 0;
    for (unsigned long long tile = blockIdx.x; tile < full_tiles; tile += gridDim.x, buf ^= 1) {
        // two LDS tiles: a thread that is still reading this one cannot be overtaken by a writer
        // of the tile after the next, which has to pass the next barrier first
        f32x4* t = s_tile[buf];
#pragma unroll
        for (int r = 0; r < 3; ++r) t[r * kEncode30Threads + threadIdx.x] = row[r];
        __syncthreads();
        // the next tile's loads are in flight while this one is encoded
        if (tile + gridDim.x < full_tiles) load_rows(tile + gridDim.x, row);
        const f32x4 a0 = t[threadIdx.x * 3], a1 = t[threadIdx.x * 3 + 1], a2 = t[threadIdx.x * 3 + 2];
        const float f[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
        uint32_t k[4];
        encode4(f, k);
        const unsigned long long first = tile * kEncode30TilePoints + threadIdx.x * 4u;
        const uint32_t i0 = static_cast<uint32_t>(first);
        const u32x4 kv = {k[0], k[1], k[2], k[3]};
        const u32x4 iv = {i0, i0 + 1, i0 + 2, i0 + 3};
        __builtin_nontemporal_store(kv, reinterpret_cast<u32x4*>(keys + first));
        if (idx) __builtin_nontemporal_store(iv, reinterpret_cast<u32x4*>(idx + first));
        if constexpr (HIST) {
#pragma unroll
            for (int q = 0; q < 4; ++q) count(k[q]);
        }
    }
    // the ragged end: fewer than 4096 points, in the workgroup whose turn it would have been
    if (blockIdx.x == full_tiles % gridDim.x) {
        const unsigned long long first = full_tiles * kEncode30TilePoints + threadIdx.x * 4u;
        if (first < n) {
            float f[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) f[j] = first * 3 + j < n * 3 ? xyz[first * 3 + j] : 0.0f;
            uint32_t k[4];
            encode4(f, k);
            for (int q = 0; q < 4; ++q)
                if (first + q < n) {
                    keys[first + q] = k[q];
                    if (idx) idx[first + q] = static_cast<uint32_t>(first + q);
                    if constexpr (HIST) count(k[q]);
                }
        }
    }
    if constexpr (HIST) {
        __syncthreads();
        if (threadIdx.x < (h.num_passes << h.bits)) {
            const uint32_t c = s_hist[threadIdx.x];
            if (c) atomicAdd(&h.ghist[threadIdx.x], c);
        }
    }
}

// One workgroup per CU at most; fewer when there are fewer tiles (the ragged end counts as one).
static unsigned encode30_grid(unsigned long long n, int num_cus)
{
    const unsigned long long tiles = (n + kEncode30TilePoints - 1) / kEncode30TilePoints;
    return static_cast<unsigned>(tiles < static_cast<unsigned long long>(num_cus) ? (tiles ? tiles : 1) : num_cus);
}

// ---- generators: splitmix64-style counter hash -------------------------------------------
__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

__host__ __device__ __forceinline__ uint32_t rand_u32(unsigned long long stream_key, unsigned long long index)
{
    return static_cast<uint32_t>(mix64(stream_key + 0x9E3779B97F4A7C15ull * (index + 1ull)) >> 32);
}

__global__ __launch_bounds__(256) void generate_points_kernel(unsigned long long k0, unsigned long long k1,
                                                               unsigned long long k2, unsigned long long n,
                                                               float* __restrict__ xyz)
{
    const unsigned long long total = n * 3;
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * 256;
    for (unsigned long long j = static_cast<unsigned long long>(blockIdx.x) * 256 + threadIdx.x; j < total;
         j += stride) {
        const unsigned long long i = j / 3;
        const int c = static_cast<int>(j - i * 3);
        const unsigned long long key = c == 0 ? k0 : (c == 1 ? k1 : k2);
        xyz[j] = static_cast<float>(rand_u32(key, i) >> 8) * (1.0f / 16777216.0f);
    }
}

__global__ __launch_bounds__(256) void generate_u32_kernel(unsigned long long key, unsigned long long n,
                                                            uint32_t* __restrict__ out)
{
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * 256;
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride)
        out[i] = rand_u32(key, i);
}

static unsigned grid_for(unsigned long long work_items, int num_cus)
{
    unsigned long long blocks = (work_items + 255) / 256;
    const unsigned long long cap = static_cast<unsigned long long>(num_cus) * 8;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return static_cast<unsigned>(blocks);
}

}  // namespace mrs

using namespace mrs;

extern "C" int mrs_morton_encode_lut(mrs_context* ctx, const void* uniforms_dev, const void* vertices_dev,
                                     uint64_t num_vertices, const void* triangles_dev, uint64_t n,
                                     uint32_t* codes_dev, void* stream)
{
    clear_error();
    if (!ctx) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_morton_encode_lut: null context");
    if (n == 0) return MRS_OK;
    if (!uniforms_dev || !vertices_dev || !triangles_dev || !codes_dev)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_morton_encode_lut: null buffer");
    if (num_vertices == 0) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_morton_encode_lut: no vertices");
    if ((reinterpret_cast<uintptr_t>(uniforms_dev) | reinterpret_cast<uintptr_t>(triangles_dev)) & 15u)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_morton_encode_lut: uniforms and triangles must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(codes_dev) & 7u) || (reinterpret_cast<uintptr_t>(vertices_dev) & 3u))
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_morton_encode_lut: misaligned codes or vertices");
    const ProfilerRange range("Morton Coding");  // src/main.rs:583
    hipLaunchKernelGGL(morton_encode_lut_kernel, dim3(grid_for(n, ctx->num_cus)), dim3(kEncodeThreads), 0,
                       static_cast<hipStream_t>(stream), static_cast<const unsigned char*>(uniforms_dev),
                       static_cast<const unsigned char*>(vertices_dev), static_cast<unsigned long long>(num_vertices),
                       static_cast<const uint4*>(triangles_dev), static_cast<unsigned long long>(n),
                       reinterpret_cast<uint2*>(codes_dev));
    MRS_CHECK_LAUNCH("morton_encode_lut_kernel");
    return MRS_OK;
}

extern "C" int mrs_morton_encode30(mrs_context* ctx, const float* xyz_dev, uint64_t n, const float* mn,
                                   const float* mx, uint32_t* keys_dev, uint32_t* idx_dev, void* stream)
{
    clear_error();
    if (!ctx) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_morton_encode30: null context");
    if (n == 0) return MRS_OK;
    if (n > 0xFFFFFFFFull) return fail(MRS_ERR_UNSUPPORTED, "mrs_morton_encode30: n must fit in 32 bits");
    if (!xyz_dev || !mn || !mx || !keys_dev) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_morton_encode30: null pointer");
    if ((reinterpret_cast<uintptr_t>(xyz_dev) | reinterpret_cast<uintptr_t>(keys_dev) |
         reinterpret_cast<uintptr_t>(idx_dev)) & 15u)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_morton_encode30: buffers must be 16-byte aligned");
    Bounds3 b;
    for (int c = 0; c < 3; ++c) { b.mn[c] = mn[c]; b.mx[c] = mx[c]; }
    const ProfilerRange range("Morton Coding");  // src/main.rs:583
    hipLaunchKernelGGL(morton_encode30_kernel<false>, dim3(encode30_grid(n, ctx->num_cus)), dim3(kEncode30Threads), 0,
                       static_cast<hipStream_t>(stream), xyz_dev, static_cast<unsigned long long>(n), b, keys_dev,
                       idx_dev, EncodeHistogram{});
    MRS_CHECK_LAUNCH("morton_encode30_kernel");
    return MRS_OK;
}

namespace mrs {
// The encoder with the digit histograms of a 32-bit-key plan accumulated on the way (called by
// mrs_morton_encode30_for_sort in mrs_api.hip, which owns the plan's memory).
int launch_encode30_with_histogram(int num_cus, const float* xyz_dev, uint64_t n, const float* mn, const float* mx,
                                   uint32_t* keys_dev, uint32_t* idx_dev, uint32_t* ghist, uint32_t begin_bit,
                                   uint32_t end_bit, uint32_t bits, uint32_t num_passes, hipStream_t s)
{
    if ((reinterpret_cast<uintptr_t>(xyz_dev) | reinterpret_cast<uintptr_t>(keys_dev) |
         reinterpret_cast<uintptr_t>(idx_dev)) & 15u)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_morton_encode30_for_sort: buffers must be 16-byte aligned");
    if ((static_cast<uint64_t>(num_passes) << bits) > 1024)
        return fail(MRS_ERR_UNSUPPORTED, "mrs_morton_encode30_for_sort: more than 1024 digit counters");
    Bounds3 b;
    for (int c = 0; c < 3; ++c) { b.mn[c] = mn[c]; b.mx[c] = mx[c]; }
    const EncodeHistogram h{ghist, begin_bit, end_bit, bits, num_passes};
    // every workgroup ends with up to 1024 global atomics: one per CU keeps that to 256 K per call
    hipLaunchKernelGGL(morton_encode30_kernel<true>, dim3(encode30_grid(n, num_cus)), dim3(kEncode30Threads), 0, s,
                       xyz_dev, static_cast<unsigned long long>(n), b, keys_dev, idx_dev, h);
    MRS_CHECK_LAUNCH("morton_encode30_kernel<histogram>");
    return MRS_OK;
}
}  // namespace mrs

extern "C" int mrs_generate_points(mrs_context* ctx, uint64_t seed, uint64_t n, float* xyz_dev, void* stream)
{
    clear_error();
    if (!ctx) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_generate_points: null context");
    if (n == 0) return MRS_OK;
    if (!xyz_dev) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_generate_points: null buffer");
    unsigned long long k[3];
    for (unsigned long long c = 0; c < 3; ++c) k[c] = mix64(seed + 0x9E3779B97F4A7C15ull * (c + 1ull));
    hipLaunchKernelGGL(generate_points_kernel, dim3(grid_for(n * 3, ctx->num_cus)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), k[0], k[1], k[2], static_cast<unsigned long long>(n), xyz_dev);
    MRS_CHECK_LAUNCH("generate_points_kernel");
    return MRS_OK;
}

extern "C" int mrs_generate_u32(mrs_context* ctx, uint64_t seed, uint64_t stream_id, uint64_t n, uint32_t* out_dev,
                                void* stream)
{
    clear_error();
    if (!ctx) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_generate_u32: null context");
    if (n == 0) return MRS_OK;
    if (!out_dev) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_generate_u32: null buffer");
    const unsigned long long key = mix64(seed + 0x9E3779B97F4A7C15ull * (stream_id + 1ull));
    hipLaunchKernelGGL(generate_u32_kernel, dim3(grid_for(n, ctx->num_cus)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), key, static_cast<unsigned long long>(n), out_dev);
    MRS_CHECK_LAUNCH("generate_u32_kernel");
    return MRS_OK;
}
