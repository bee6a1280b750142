// Whole LSD radix sort of a small input in one launch of one workgroup.
//
// The reference sorts 3000 triangles with 8 passes of three dispatches each
// (src/main.rs:594-665); at that size every launch is pure latency.  Here all passes run
// inside one workgroup: (key, source index) pairs ping-pong between two LDS images, each
// pass ranks with the same wave-order scheme as the digit pass (mrs_radix_pass.hpp), and
// the payload, whatever its size, moves once at the end as a gather by source index.
// A pass whose digit is the same for every key (zero high bytes of a short code) is skipped.
#include "mrs_radix_pass.hpp"

namespace mrs {

constexpr int kSmallThreads = 1024;
constexpr int kSmallKpt = 4;
constexpr int kSmallCapacity = kSmallThreads * kSmallKpt;

template <int BITS, typename KeyT>
__global__ __launch_bounds__(kSmallThreads) void small_sort_kernel(const SmallSortArgs a)
{
    constexpr int RADIX = 1 << BITS;
    constexpr int WAVES = kSmallThreads / 64;
    constexpr int SCAN_WAVES = (RADIX + 63) / 64;
    __shared__ KeyT s_key[2][kSmallCapacity];
    __shared__ uint16_t s_src[2][kSmallCapacity];
    __shared__ uint32_t wave_hist[WAVES][RADIX];
    __shared__ uint32_t wsum[SCAN_WAVES];
    __shared__ uint32_t s_skip[2];  // one per pass parity: a flag is reset a full pass after its last read

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t n = a.n;
    // stability order: wave w owns elements [w*256, w*256+256), item i of lane l is w*256 + i*64 + l
    const uint32_t my_first = wave * 64 * kSmallKpt + lane;

    const KeyT* __restrict__ keys_in = static_cast<const KeyT*>(a.keys_in);
#pragma unroll
    for (int i = 0; i < kSmallKpt; ++i) {
        const uint32_t e = my_first + i * 64;
        if (e < n) {
            s_key[0][e] = keys_in[e];
            s_src[0][e] = static_cast<uint16_t>(e);
        }
    }
    int cur = 0;
    for (uint32_t pass = 0; pass < a.num_passes; ++pass) {
        const uint32_t shift = a.begin_bit + pass * BITS;
        const uint32_t width = a.end_bit - shift < static_cast<uint32_t>(BITS) ? a.end_bit - shift : BITS;
        const uint32_t mask = (1u << width) - 1u;
        for (int j = tid; j < WAVES * RADIX; j += kSmallThreads) (&wave_hist[0][0])[j] = 0;
        if (tid == 0) s_skip[pass & 1] = 0;
        __syncthreads();  // also: the image written by the previous pass (or the load) is complete

        KeyT key[kSmallKpt];
        uint32_t pos[kSmallKpt];
#pragma unroll
        for (int i = 0; i < kSmallKpt; ++i) {
            const uint32_t e = my_first + i * 64;
            const bool mine = e < n;
            key[i] = mine ? s_key[cur][e] : static_cast<KeyT>(0);
            const uint32_t d = static_cast<uint32_t>(key[i] >> shift) & mask;
            uint32_t mlo, mhi;
            match_digit<BITS>(d, __ballot(mine), &mlo, &mhi);
            const uint32_t below = lanes_below(mlo, mhi);
            const uint32_t prev = wave_hist[wave][d];
            if (mine && below == 0) wave_hist[wave][d] = prev + __popc(mlo) + __popc(mhi);
            pos[i] = prev + below;
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();

        uint32_t count = 0, incl = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t c = wave_hist[w][tid];
                wave_hist[w][tid] = count;
                count += c;
            }
            if (count == n) s_skip[pass & 1] = 1;  // one digit holds everything: the pass is the identity
            incl = count;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(incl, o);
                if (lane >= o) incl += y;
            }
            if (lane == 63 || tid == RADIX - 1) wsum[wave] = incl;
        }
        __syncthreads();
        if (s_skip[pass & 1]) continue;  // uniform over the workgroup
        if (tid < RADIX) {
            uint32_t off = incl - count;
#pragma unroll
            for (int w = 0; w < SCAN_WAVES; ++w)
                if (w < wave) off += wsum[w];
#pragma unroll
            for (int w = 0; w < WAVES; ++w) wave_hist[w][tid] += off;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kSmallKpt; ++i) {
            const uint32_t e = my_first + i * 64;
            if (e < n) {
                const uint32_t d = static_cast<uint32_t>(key[i] >> shift) & mask;
                const uint32_t dst = pos[i] + wave_hist[wave][d];
                s_key[cur ^ 1][dst] = key[i];
                s_src[cur ^ 1][dst] = s_src[cur][e];
            }
        }
        cur ^= 1;
        __syncthreads();  // the per-wave bases are read until here; the next pass clears them
    }
    __syncthreads();

    // ---- keys out in order, payload gathered by source index ---------------------------------
    KeyT* __restrict__ keys_out = static_cast<KeyT*>(a.keys_out);
    for (uint32_t j = tid; j < n; j += kSmallThreads) keys_out[j] = s_key[cur][j];
    if (a.payload_bytes == 4) {
        const uint32_t* __restrict__ vin = static_cast<const uint32_t*>(a.vals_in);
        uint32_t* __restrict__ vout = static_cast<uint32_t*>(a.vals_out);
        if (a.index_payload) {
            for (uint32_t j = tid; j < n; j += kSmallThreads) vout[j] = s_src[cur][j];
        } else {
            for (uint32_t j = tid; j < n; j += kSmallThreads) vout[j] = vin[s_src[cur][j]];
        }
    } else if (a.payload_bytes == 16) {
        const uint4* __restrict__ vin = static_cast<const uint4*>(a.vals_in);
        uint4* __restrict__ vout = static_cast<uint4*>(a.vals_out);
        for (uint32_t j = tid; j < n; j += kSmallThreads) vout[j] = vin[s_src[cur][j]];
    }
    if (a.vals_final && a.payload_bytes) {
        // the result belongs in the buffer the payload came from: every read of it is done
        // once the whole workgroup is past this barrier (one workgroup, one compute unit)
        __threadfence();
        __syncthreads();
        const uint32_t words = n * (a.payload_bytes / 4);
        const uint32_t* vout = static_cast<const uint32_t*>(a.vals_out);
        uint32_t* vfin = static_cast<uint32_t*>(a.vals_final);
        for (uint32_t j = tid; j < words; j += kSmallThreads)
            vfin[j] = __hip_atomic_load(vout + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

uint32_t small_sort_capacity() { return kSmallCapacity; }

int launch_small_sort(int key_words, int bits, const SmallSortArgs& a, hipStream_t s)
{
    if (a.n == 0 || a.n > static_cast<uint32_t>(kSmallCapacity))
        return fail(MRS_ERR_INVALID_ARGUMENT, "small sort: %u elements, capacity %d", a.n, kSmallCapacity);
    if (key_words == 1 && bits == 8)
        hipLaunchKernelGGL((small_sort_kernel<8, uint32_t>), dim3(1), dim3(kSmallThreads), 0, s, a);
    else if (key_words == 1 && bits == 4)
        hipLaunchKernelGGL((small_sort_kernel<4, uint32_t>), dim3(1), dim3(kSmallThreads), 0, s, a);
    else if (key_words == 2 && bits == 8)
        hipLaunchKernelGGL((small_sort_kernel<8, unsigned long long>), dim3(1), dim3(kSmallThreads), 0, s, a);
    else if (key_words == 2 && bits == 4)
        hipLaunchKernelGGL((small_sort_kernel<4, unsigned long long>), dim3(1), dim3(kSmallThreads), 0, s, a);
    else
        return fail(MRS_ERR_UNSUPPORTED, "small sort: key_words %d, digit_bits %d", key_words, bits);
    MRS_CHECK_LAUNCH("small_sort_kernel");
    return MRS_OK;
}

}  // namespace mrs
