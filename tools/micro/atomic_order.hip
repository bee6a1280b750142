// Probe: do LDS atomic adds with return value hand out ranks in ascending lane order
// when several lanes of one instruction hit the same address?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int BITS>
__device__ __forceinline__ void match_digit(uint32_t d, uint32_t* out_lo, uint32_t* out_hi)
{
    uint32_t lo = ~0u, hi = ~0u;
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const uint32_t x = static_cast<uint32_t>(__builtin_amdgcn_sbfe(d, b, 1));
        const unsigned long long bal = __builtin_amdgcn_uicmp(x, 0u, 33);
        lo = __builtin_amdgcn_bitop3_b32(lo, static_cast<uint32_t>(bal), x, 0x90);
        hi = __builtin_amdgcn_bitop3_b32(hi, static_cast<uint32_t>(bal >> 32), x, 0x90);
    }
    *out_lo = lo; *out_hi = hi;
}

__global__ __launch_bounds__(1024) void probe(unsigned long long* bad, unsigned long long* descending, int iters, uint32_t modulus, uint32_t seed)
{
    __shared__ uint32_t cnt[16][256];
    __shared__ uint32_t ref[16][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = threadIdx.x; j < 16 * 256; j += blockDim.x) { (&cnt[0][0])[j] = 0; (&ref[0][0])[j] = 0; }
    __syncthreads();
    unsigned long long nbad = 0, ndesc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t d = hash32(seed + (blockIdx.x * 1024u + threadIdx.x) * 977u + it * 0x9E3779B9u) % modulus;
        uint32_t mlo, mhi;
        match_digit<8>(d, &mlo, &mhi);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        const uint32_t total = __popc(mlo) + __popc(mhi);
        const uint32_t prev = ref[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (below == 0) ref[wave][d] = prev + total;
        const uint32_t got = atomicAdd(&cnt[wave][d], 1u);
        if (got != prev + below) ++nbad;
        if (got == prev + (total - 1 - below)) ++ndesc;
        __builtin_amdgcn_wave_barrier();
    }
    if (nbad) atomicAdd(bad, nbad);
    atomicAdd(descending, ndesc);
}

int main() {
    unsigned long long *d; CK(hipMalloc(&d, 16));
    for (uint32_t modulus : {256u, 64u, 32u, 7u, 2u, 1u, 33u, 255u}) {
        for (int threads : {64, 256, 1024}) {
            CK(hipMemset(d, 0, 16));
            probe<<<2048, threads>>>(d, d + 1, 2000, modulus, 12345u + modulus);
            CK(hipDeviceSynchronize());
            unsigned long long h[2]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            printf("modulus %3u threads %4d : %llu mismatches of %llu lane-ops (descending-consistent %llu)\n", modulus, threads, h[0],
                   2048ull * threads * 2000ull, h[1]);
        }
    }
    return 0;
}
