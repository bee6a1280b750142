// Does HBM copy bandwidth depend on whether the compute units read and write in phase?
// 256 persistent workgroups copy tiles of 256 KiB (registers as the staging area):
//   free : every workgroup reads a tile, writes it, takes the next
//   sync : all workgroups read, grid barrier, all write, grid barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int THREADS = 1024, VPT = 16;  // 16 x uint4 per thread = 256 KiB per tile

__device__ void grid_barrier(unsigned* counter, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}

template <bool SYNC>
__global__ __launch_bounds__(THREADS) void copy_tiles(const uint4* __restrict__ p, uint4* __restrict__ q, size_t ntiles, unsigned* counter)
{
    unsigned phase = 0;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t base = t * (size_t)(THREADS * VPT);
        uint4 x[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) x[i] = p[base + i * THREADS + threadIdx.x];
        if (SYNC) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); grid_barrier(counter, ++phase * gridDim.x); }
#pragma unroll
        for (int i = 0; i < VPT; ++i) q[base + i * THREADS + threadIdx.x] = x[i];
        if (SYNC) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); grid_barrier(counter, ++phase * gridDim.x); }
    }
}

int main() {
    const size_t bytes = 4ull << 30, nvec = bytes / 16, ntiles = nvec / (THREADS * VPT);
    uint4 *p, *q; unsigned* c; CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&c, 4));
    CK(hipMemset(p, 1, bytes)); CK(hipMemset(q, 2, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int sync = 0; sync < 2; ++sync) for (int grid : {256, 512}) {
        if (sync && grid != 256) continue;  // the barrier needs every workgroup resident
        float best = 1e9;
        for (int r = 0; r < 4; ++r) {
            CK(hipMemset(c, 0, 4));
            CK(hipEventRecord(a));
            if (sync) copy_tiles<true><<<grid, THREADS>>>(p, q, ntiles, c); else copy_tiles<false><<<grid, THREADS>>>(p, q, ntiles, c);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("%s grid %d: %.3f ms, %.0f GB/s read+write\n", sync ? "in phase " : "free     ", grid, best, 2.0 * bytes / best / 1e6);
    }
    return 0;
}
