// Exclusive prefix sum over the flat digit-major block histogram.
//
// Replaces the reference's three prefix-sum shaders and their level planner
// (radix_sort_block_sum_large / _after: src/shaders/radix_sort_block_sum_large.wgsl:146-172,
// radix_sort_block_sum_small: src/shaders/radix_sort_block_sum_small.wgsl:50-77,
// calculate_num_items_prefix_buffers + dispatch: src/main.rs:201-233,604-655).
// The result is the same array the reference's hierarchy produces -- entry i
// becomes the wrapping u32 sum of entries [0, i) -- computed as reduce /
// scan-of-chunk-sums / scan-with-offset over 4096-entry chunks, so the level
// count does not grow with the input (2L+1 dispatches in the reference, 3 here).
#include "mrs_internal.h"

namespace mrs {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanChunk = kScanThreads * kScanItems;  // 4096

// Exclusive scan of one value per thread across a 256-thread workgroup.
// Returns the exclusive prefix; *total receives the workgroup sum.
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t x, uint32_t* s_wave /*[4]*/, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t off = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t v = s_wave[w];
        if (w < wave) off += v;
        sum += v;
    }
    __syncthreads();  // s_wave may be reused by the caller's next round
    *total = sum;
    return off + incl - x;
}

__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(const uint32_t* __restrict__ vals,
                                                                   unsigned long long len,
                                                                   uint32_t* __restrict__ partials)
{
    __shared__ uint32_t s_wave[4];
    const unsigned long long base = static_cast<unsigned long long>(blockIdx.x) * kScanChunk;
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const unsigned long long idx = base + i * kScanThreads + threadIdx.x;
        if (idx < len) sum += vals[idx];
    }
    uint32_t total;
    block_exclusive_scan_256(sum, s_wave, &total);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// One workgroup scans an array of any length in place, 4096 entries per round.
__global__ __launch_bounds__(kScanThreads) void scan_single_block_kernel(uint32_t* __restrict__ vals,
                                                                         unsigned long long len)
{
    __shared__ uint32_t s_wave[4];
    uint32_t carry = 0;
    for (unsigned long long base = 0; base < len; base += kScanChunk) {
        uint32_t v[kScanItems];
        uint32_t sum = 0;
        const unsigned long long first = base + static_cast<unsigned long long>(threadIdx.x) * kScanItems;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) {
            v[i] = first + i < len ? vals[first + i] : 0u;
            sum += v[i];
        }
        uint32_t total;
        uint32_t run = carry + block_exclusive_scan_256(sum, s_wave, &total);
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) {
            if (first + i < len) vals[first + i] = run;
            run += v[i];
        }
        carry += total;
    }
}

__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(uint32_t* __restrict__ vals,
                                                                  unsigned long long len,
                                                                  const uint32_t* __restrict__ partials)
{
    __shared__ uint32_t s_wave[4];
    const unsigned long long base = static_cast<unsigned long long>(blockIdx.x) * kScanChunk;
    const unsigned long long first = base + static_cast<unsigned long long>(threadIdx.x) * kScanItems;
    uint32_t v[kScanItems];
    uint32_t sum = 0;
    if (first + kScanItems <= len) {
        const uint4* p = reinterpret_cast<const uint4*>(vals + first);
#pragma unroll
        for (int q = 0; q < kScanItems / 4; ++q) {
            const uint4 x = p[q];
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) v[i] = first + i < len ? vals[first + i] : 0u;
    }
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) sum += v[i];
    uint32_t total;
    uint32_t run = partials[blockIdx.x] + block_exclusive_scan_256(sum, s_wave, &total);
    if (first + kScanItems <= len) {
        uint4* p = reinterpret_cast<uint4*>(vals + first);
#pragma unroll
        for (int q = 0; q < kScanItems / 4; ++q) {
            uint4 x;
            x.x = run; run += v[4 * q];
            x.y = run; run += v[4 * q + 1];
            x.z = run; run += v[4 * q + 2];
            x.w = run; run += v[4 * q + 3];
            p[q] = x;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) {
            if (first + i < len) vals[first + i] = run;
            run += v[i];
        }
    }
}

uint64_t scan_partials_len(uint64_t len) { return (len + kScanChunk - 1) / kScanChunk; }

int launch_exclusive_scan(uint32_t* vals, uint64_t len, uint32_t* partials, hipStream_t s)
{
    if (len == 0) return MRS_OK;
    const uint64_t chunks = scan_partials_len(len);
    if (chunks <= 4) {
        const ProfilerRange range("Small Prefix Sum");  // src/main.rs:625
        hipLaunchKernelGGL(scan_single_block_kernel, dim3(1), dim3(kScanThreads), 0, s, vals,
                           static_cast<unsigned long long>(len));
        MRS_CHECK_LAUNCH("scan_single_block_kernel");
        return MRS_OK;
    }
    if (chunks > 0x7FFFFFFFull) return fail(MRS_ERR_UNSUPPORTED, "histogram too long to scan");
    {
        const ProfilerRange range("Large Prefix Sum part 1");  // src/main.rs:606
        hipLaunchKernelGGL(scan_reduce_kernel, dim3(static_cast<unsigned>(chunks)), dim3(kScanThreads), 0, s, vals,
                           static_cast<unsigned long long>(len), partials);
        MRS_CHECK_LAUNCH("scan_reduce_kernel");
    }
    {
        const ProfilerRange range("Small Prefix Sum");  // src/main.rs:625
        hipLaunchKernelGGL(scan_single_block_kernel, dim3(1), dim3(kScanThreads), 0, s, partials,
                           static_cast<unsigned long long>(chunks));
        MRS_CHECK_LAUNCH("scan_single_block_kernel");
    }
    {
        const ProfilerRange range("Large Prefix Sum part 2");  // src/main.rs:635
        hipLaunchKernelGGL(scan_apply_kernel, dim3(static_cast<unsigned>(chunks)), dim3(kScanThreads), 0, s, vals,
                           static_cast<unsigned long long>(len), partials);
        MRS_CHECK_LAUNCH("scan_apply_kernel");
    }
    return MRS_OK;
}

}  // namespace mrs
