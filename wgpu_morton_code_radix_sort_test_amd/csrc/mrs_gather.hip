// Final gather: instead of dragging a wide record through every digit pass (the reference moves
// the 16-byte TriangleIndices 8 times, src/shaders/radix_sort_index.wgsl:281-307), sort
// (key, source index) pairs and move each record once at the end.  The source indices are made
// up by the first digit pass (PassArgs::index_payload); they are never written as an input.
#include "mrs_internal.h"

namespace mrs {

namespace {

// 64-bit loop indices throughout: n may lie within one grid stride of 2^32, where a 32-bit index
// would wrap to a value below n and never leave the loop.
// out[j] = in[index[j]], records of sizeof(T) bytes
template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const T* __restrict__ in, const uint32_t* __restrict__ index,
                                                     T* __restrict__ out, uint32_t n)
{
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * 256u;
    for (unsigned long long j = static_cast<unsigned long long>(blockIdx.x) * 256u + threadIdx.x; j < n; j += stride)
        out[j] = in[index[j]];
}

// records of `words` 32-bit words, any count: one thread per word
__global__ __launch_bounds__(256) void gather_words_kernel(const uint32_t* __restrict__ in,
                                                           const uint32_t* __restrict__ index,
                                                           uint32_t* __restrict__ out, uint32_t n, uint32_t words)
{
    const unsigned long long total = static_cast<unsigned long long>(n) * words;
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * 256u;
    for (unsigned long long w = static_cast<unsigned long long>(blockIdx.x) * 256u + threadIdx.x; w < total; w += stride) {
        const unsigned long long j = w / words;
        const uint32_t k = static_cast<uint32_t>(w - j * words);
        out[w] = in[static_cast<unsigned long long>(index[j]) * words + k];
    }
}

int grid_for(uint64_t items, int num_cus)
{
    uint64_t blocks = (items + 255) / 256;
    const uint64_t cap = static_cast<uint64_t>(num_cus) * 32;
    if (blocks > cap) blocks = cap;
    return blocks ? static_cast<int>(blocks) : 1;
}

}  // namespace

int launch_gather(const void* in, const uint32_t* index, void* out, uint32_t n, uint32_t record_bytes, int num_cus,
                  hipStream_t s)
{
    const dim3 grid(grid_for(n, num_cus)), block(256);
    switch (record_bytes) {
        case 4:
            hipLaunchKernelGGL(gather_kernel<uint32_t>, grid, block, 0, s, static_cast<const uint32_t*>(in), index,
                               static_cast<uint32_t*>(out), n);
            break;
        case 8:
            hipLaunchKernelGGL(gather_kernel<uint2>, grid, block, 0, s, static_cast<const uint2*>(in), index,
                               static_cast<uint2*>(out), n);
            break;
        case 16:
            hipLaunchKernelGGL(gather_kernel<uint4>, grid, block, 0, s, static_cast<const uint4*>(in), index,
                               static_cast<uint4*>(out), n);
            break;
        default:
            if (record_bytes == 0 || record_bytes % 4 != 0)
                return fail(MRS_ERR_INVALID_ARGUMENT, "gather: record size %u is not a positive multiple of 4", record_bytes);
            hipLaunchKernelGGL(gather_words_kernel, dim3(grid_for(static_cast<uint64_t>(n) * (record_bytes / 4), num_cus)),
                               block, 0, s, static_cast<const uint32_t*>(in), index, static_cast<uint32_t*>(out), n,
                               record_bytes / 4);
    }
    MRS_CHECK_LAUNCH("gather_kernel");
    return MRS_OK;
}

}  // namespace mrs
