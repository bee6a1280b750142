// Achievable HBM bandwidth on this part: read-only and copy sweeps over 4 GiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int UNROLL, bool NT>
__global__ void read_k(const uint4* __restrict__ p, size_t nvec, unsigned* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; v + (UNROLL - 1) * stride < nvec; v += UNROLL * stride) {
        uint4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) { v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p + v + u * stride)); x[u] = make_uint4(t.x, t.y, t.z, t.w); } else x[u] = p[v + u * stride]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// each block streams a contiguous chunk (tile-like access)
template <int UNROLL>
__global__ void read_tile(const uint4* __restrict__ p, size_t nvec, unsigned* out) {
    const size_t per_block = (size_t)blockDim.x * UNROLL;
    unsigned acc = 0;
    for (size_t base = (size_t)blockIdx.x * per_block; base + per_block <= nvec; base += (size_t)gridDim.x * per_block) {
        uint4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = p[base + u * blockDim.x + threadIdx.x];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int UNROLL, bool NT>
__global__ void copy_k(const uint4* __restrict__ p, uint4* __restrict__ q, size_t nvec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; v + (UNROLL - 1) * stride < nvec; v += UNROLL * stride) {
        uint4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = p[v + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) { v4u t = {x[u].x, x[u].y, x[u].z, x[u].w}; __builtin_nontemporal_store(t, reinterpret_cast<v4u*>(q + v + u * stride)); } else q[v + u * stride] = x[u]; }
    }
}
template <typename F> float time_it(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize()); float best = 1e9;
    for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}
int main() {
    const size_t bytes = 4ull << 30; const size_t nvec = bytes / 16;
    uint4 *p, *q; unsigned* o; CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&o, 16));
    CK(hipMemset(p, 1, bytes)); CK(hipMemset(q, 2, bytes));
    for (int threads : {256, 512, 1024}) for (int grid : {1024, 2048, 4096, 16384}) {
        float a = time_it([&] { read_k<4, false><<<grid, threads>>>(p, nvec, o); });
        float b = time_it([&] { read_k<8, false><<<grid, threads>>>(p, nvec, o); });
        float c = time_it([&] { read_k<8, true><<<grid, threads>>>(p, nvec, o); });
        float d = time_it([&] { read_tile<8><<<grid, threads>>>(p, nvec, o); });
        printf("read  thr %4d grid %5d: u4 %.0f  u8 %.0f  u8nt %.0f  tile8 %.0f GB/s\n", threads, grid, bytes / a / 1e6, bytes / b / 1e6, bytes / c / 1e6, bytes / d / 1e6);
    }
    for (int threads : {256, 1024}) for (int grid : {2048, 8192}) {
        float a = time_it([&] { copy_k<4, false><<<grid, threads>>>(p, q, nvec); });
        float b = time_it([&] { copy_k<4, true><<<grid, threads>>>(p, q, nvec); });
        printf("copy  thr %4d grid %5d: u4 %.0f  u4nt %.0f GB/s (read+write)\n", threads, grid, 2.0 * bytes / a / 1e6, 2.0 * bytes / b / 1e6);
    }
    float m = time_it([&] { CK(hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, 0)); });
    printf("hipMemcpy D2D: %.0f GB/s (read+write)\n", 2.0 * bytes / m / 1e6);
    float s = time_it([&] { CK(hipMemsetAsync(q, 0, bytes, 0)); });
    printf("hipMemset: %.0f GB/s\n", bytes / s / 1e6);
    return 0;
}
