// Device check behind kRankAtomic: when several lanes of ONE returning LDS atomic add hit
// the same word, are they served in ascending lane order?  The ISA does not promise it;
// gfx950 does it, and a stable rank from one atomic per key depends on it.  The probe
// ranks pseudo-random digits both ways (atomics against the ballot-based match) over
// full and partial lane masks and many conflict patterns, and counts disagreements.
#include "mrs_radix_pass.hpp"

namespace mrs {

namespace {

__device__ __forceinline__ uint32_t probe_hash(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

constexpr int kProbeThreads = 1024;
constexpr int kProbeWaves = kProbeThreads / 64;

__global__ __launch_bounds__(kProbeThreads) void lds_atomic_order_probe_kernel(uint32_t* mismatches, int iters,
                                                                               uint32_t modulus, uint32_t keep_mask)
{
    __shared__ uint32_t by_atomic[kProbeWaves][256];
    __shared__ uint32_t by_match[kProbeWaves][256];
    const int wave = threadIdx.x >> 6;
    for (int j = threadIdx.x; j < kProbeWaves * 256; j += kProbeThreads) {
        (&by_atomic[0][0])[j] = 0;
        (&by_match[0][0])[j] = 0;
    }
    __syncthreads();
    uint32_t bad = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t h = probe_hash((blockIdx.x * kProbeThreads + threadIdx.x) * 977u + it * 0x9E3779B9u + modulus);
        const uint32_t d = h % modulus;
        // keep_mask = 0: all lanes take part; otherwise a pseudo-random subset
        const bool mine = keep_mask == 0u || ((h >> 24) & keep_mask) != 0u;
        uint32_t mlo, mhi;
        match_digit<8>(d, __ballot(mine), &mlo, &mhi);
        const uint32_t below = lanes_below(mlo, mhi);
        const uint32_t prev = by_match[wave][d];
        if (mine && below == 0) by_match[wave][d] = prev + __popc(mlo) + __popc(mhi);
        __builtin_amdgcn_wave_barrier();
        if (mine) {
            const uint32_t got = atomicAdd(&by_atomic[wave][d], 1u);
            bad += got != prev + below;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Which die (XCD) does a workgroup run on?  The die-aware ticket lists of the digit pass need
// workgroups on every one of the 8 dies; the probe counts what a large grid reads.
__global__ void die_id_probe_kernel(uint32_t* hist)
{
    if (threadIdx.x == 0) {
        const uint32_t die = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));  // HW_REG_XCC_ID[3:0]
        atomicAdd(&hist[die & 15u], 1u);
    }
}

}  // namespace

// *all_dies = a grid of many workgroups lands on dies 0..7 in comparable shares and on no other id.
int probe_die_ids(bool* all_dies)
{
    *all_dies = false;
    constexpr uint32_t kGrid = 8192;
    uint32_t* d_hist = nullptr;
    MRS_HIP_TRY(hipMalloc(&d_hist, 16 * sizeof(uint32_t)));
    hipError_t e = hipMemset(d_hist, 0, 16 * sizeof(uint32_t));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(die_id_probe_kernel, dim3(kGrid), dim3(64), 0, 0, d_hist);
        e = hipGetLastError();
    }
    uint32_t hist[16] = {};
    if (e == hipSuccess) e = hipMemcpy(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost);
    (void)hipFree(d_hist);
    if (e != hipSuccess) return fail(MRS_ERR_HIP, "die id probe failed: %s", hipGetErrorString(e));
    bool ok = true;
    for (int i = 0; i < 16; ++i) {
        if (i < kTicketLists) ok = ok && hist[i] >= kGrid / kTicketLists / 2;
        else ok = ok && hist[i] == 0;
    }
    *all_dies = ok;
    return MRS_OK;
}

// *in_order = the device passed; the current device must be the one to test.
int probe_lds_atomic_order(bool* in_order)
{
    *in_order = false;
    uint32_t* d_bad = nullptr;
    MRS_HIP_TRY(hipMalloc(&d_bad, sizeof(uint32_t)));
    hipError_t e = hipMemset(d_bad, 0, sizeof(uint32_t));
    if (e == hipSuccess) {
        const uint32_t moduli[] = {256u, 255u, 33u, 32u, 7u, 2u, 1u};
        for (uint32_t m : moduli)
            for (uint32_t keep : {0u, 1u, 3u})
                hipLaunchKernelGGL(lds_atomic_order_probe_kernel, dim3(64), dim3(kProbeThreads), 0, 0, d_bad, 64, m, keep);
        e = hipGetLastError();
    }
    uint32_t bad = 1;
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return fail(MRS_ERR_HIP, "LDS atomic order probe failed: %s", hipGetErrorString(e));
    *in_order = bad == 0;
    return MRS_OK;
}

}  // namespace mrs
