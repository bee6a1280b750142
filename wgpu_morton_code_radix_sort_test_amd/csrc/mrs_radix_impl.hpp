// Histogram kernels of the radix sort for gfx950: the per-tile digit-major block
// histogram of the three-kernel pass and the all-passes global histogram that the
// onesweep passes start from.  The digit pass itself is in mrs_radix_pass.hpp.
#ifndef MRS_RADIX_IMPL_HPP
#define MRS_RADIX_IMPL_HPP

#include <type_traits>

#include "mrs_internal.h"
#include "mrs_radix_pass.hpp"

namespace mrs {

// ---- per-tile digit histogram, digit-major (three-kernel pass) ------------------
// radix_sort_compute_histogram, src/shaders/radix_sort_histogram.wgsl:42-60, for
// a tile of `tile_items` keys instead of 256.
// Every counter is kept in R copies, copy = lane % R, so that the lanes of one LDS atomic spread
// over R banks whatever their digits are (see BankedHistogram below): R = 16 or 32 for tiles of
// 8192 keys or more, fewer for small tiles, whose time would go into clearing and adding up the
// copies.  A tile whose keys start on a 16-byte boundary is read with nontemporal 16-byte loads,
// four in flight per thread.
template <int BITS, typename KeyT>
__global__ __launch_bounds__(256) void block_histogram_kernel(const KeyT* __restrict__ keys, uint32_t n,
                                                              uint32_t tile_items, uint32_t num_tiles,
                                                              uint32_t shift, uint32_t mask,
                                                              uint32_t* __restrict__ hist)
{
    constexpr int RADIX = 1 << BITS;
    constexpr int RMAX = RADIX >= 256 ? 16 : 32;
    constexpr int VEC = 16 / sizeof(KeyT);
    constexpr int UNROLL = 4;
    __shared__ uint32_t s_hist[RADIX * RMAX];
    const uint32_t tile = blockIdx.x;
    const uint32_t R = tile_items >= 8192 ? RMAX : (tile_items >= 2048 ? 4 : 1);  // a power of two
    for (uint32_t j = threadIdx.x; j < RADIX * R; j += 256) s_hist[j] = 0;
    __syncthreads();
    const uint32_t copy = threadIdx.x & (R - 1);
    auto count = [&](KeyT k) { atomicAdd(&s_hist[(static_cast<uint32_t>(k >> shift) & mask) * R + copy], 1u); };
    const unsigned long long base = static_cast<unsigned long long>(tile) * tile_items;
    const unsigned long long end = base + tile_items < n ? base + tile_items : n;
    unsigned long long i = base;
    if ((reinterpret_cast<uintptr_t>(keys + base) & 15u) == 0) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* __restrict__ vkeys = reinterpret_cast<const u32x4*>(keys + base);
        const unsigned long long nvec = (end - base) / VEC;
        auto count_vec = [&](const u32x4& x) {
            if constexpr (sizeof(KeyT) == 4) {
                count(x.x); count(x.y); count(x.z); count(x.w);
            } else {
                count(static_cast<KeyT>(x.x) | (static_cast<KeyT>(x.y) << 32));
                count(static_cast<KeyT>(x.z) | (static_cast<KeyT>(x.w) << 32));
            }
        };
        unsigned long long v = threadIdx.x;
        for (; v + (UNROLL - 1) * 256 < nvec; v += UNROLL * 256) {
            u32x4 x[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) x[u] = __builtin_nontemporal_load(vkeys + v + u * 256);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) count_vec(x[u]);
        }
        for (; v < nvec; v += 256) count_vec(__builtin_nontemporal_load(vkeys + v));
        i = base + nvec * VEC;
    }
    for (i += threadIdx.x; i < end; i += 256) count(keys[i]);
    __syncthreads();
    for (int d = threadIdx.x; d < RADIX; d += 256) {
        uint32_t c = 0;
        for (uint32_t r = 0; r < R; ++r) c += s_hist[d * R + ((r + d) & (R - 1))];  // rotated: threads on distinct banks
        hist[static_cast<size_t>(d) * num_tiles + tile] = c;
    }
}

// ---- global digit histograms of all passes in one read of the keys ---------------
struct GlobalHistArgs {
    const void* keys;
    uint32_t n;
    uint32_t begin_bit;
    uint32_t end_bit;
    uint32_t num_passes;
    const uint8_t* digit_map;  // partition mode: one pass, digit = map[(key >> begin_bit) & map_mask]
    uint32_t map_mask;
    uint32_t* ghist;           // [num_passes][RADIX], must be zero
    // optional: the kernel runs only if *gate == gate_value (a word an earlier kernel of the same
    // stream wrote: the device's choice between two pass schedules, see mrs_hybrid.hip)
    const uint32_t* gate;
    uint32_t gate_value;
};

// One LDS counter set per workgroup, one LDS atomic per key and pass; the pass loop is
// unrolled at compile time (passes beyond a.num_passes are skipped by a scalar branch).
// A wave whose lanes all hold one digit (constant high bytes, sorted or all-equal
// input) would serialise 64 ways on one LDS word: it adds 64 from one lane instead.
template <int BITS, typename KeyT>
__global__ __launch_bounds__(256) void global_histogram_kernel(const GlobalHistArgs a)
{
    constexpr int RADIX = 1 << BITS;
    constexpr int MAXP = (sizeof(KeyT) * 8 + BITS - 1) / BITS;
    constexpr int VEC = 16 / sizeof(KeyT);
    constexpr int UNROLL = 2;  // vectors in flight per thread
    __shared__ uint32_t s_hist[MAXP * RADIX];
    if (a.gate && *a.gate != a.gate_value) return;  // uniform
    for (int j = threadIdx.x; j < MAXP * RADIX; j += 256) s_hist[j] = 0;
    __syncthreads();

    const KeyT* __restrict__ keys = static_cast<const KeyT*>(a.keys);
    const int lane = threadIdx.x & 63;
    const bool mapped = a.digit_map != nullptr;

    // full waves only
    auto count = [&](KeyT k) {
        if (mapped) {
            const uint32_t d = a.digit_map[static_cast<uint32_t>(k >> a.begin_bit) & a.map_mask];
            atomicAdd(&s_hist[d], 1u);
            return;
        }
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            if (p < static_cast<int>(a.num_passes)) {
                const uint32_t shift = a.begin_bit + p * BITS;
                const uint32_t width = a.end_bit - shift < static_cast<uint32_t>(BITS) ? a.end_bit - shift : BITS;
                const uint32_t d = static_cast<uint32_t>(k >> shift) & ((1u << width) - 1u);
                const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
                if (__builtin_amdgcn_uicmp(d, d0, 32 /* == */) == ~0ull) {
                    if (lane == 0) atomicAdd(&s_hist[p * RADIX + d0], 64u);
                } else {
                    atomicAdd(&s_hist[p * RADIX + d], 1u);
                }
            }
        }
    };
    // any subset of lanes
    auto count_some = [&](KeyT k) {
        if (mapped) {
            atomicAdd(&s_hist[a.digit_map[static_cast<uint32_t>(k >> a.begin_bit) & a.map_mask]], 1u);
            return;
        }
        for (uint32_t p = 0; p < a.num_passes; ++p) {
            const uint32_t shift = a.begin_bit + p * BITS;
            const uint32_t width = a.end_bit - shift < static_cast<uint32_t>(BITS) ? a.end_bit - shift : BITS;
            atomicAdd(&s_hist[p * RADIX + (static_cast<uint32_t>(k >> shift) & ((1u << width) - 1u))], 1u);
        }
    };
    using VecT = typename std::conditional<sizeof(KeyT) == 4, uint4, ulonglong2>::type;
    auto count_vec = [&](const VecT& x) {
        count(x.x);
        count(x.y);
        if constexpr (sizeof(KeyT) == 4) {
            count(x.z);
            count(x.w);
        }
    };

    const unsigned long long nvec = a.n / VEC;
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * 256;
    const VecT* __restrict__ vkeys = reinterpret_cast<const VecT*>(keys);
    // The wave-wide comparison in `count` needs whole waves: each wave covers 64 consecutive
    // vectors per step, and steps that are not entirely inside the input go to count_some.
    unsigned long long v = static_cast<unsigned long long>(blockIdx.x) * 256 + threadIdx.x;
    const unsigned long long wave_first = v - lane;
    unsigned long long done = 0;  // steps of this wave already counted
    while (wave_first + (done + UNROLL - 1) * stride + 64 <= nvec) {
        VecT x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = vkeys[v + (done + u) * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) count_vec(x[u]);
        done += UNROLL;
    }
    while (wave_first + done * stride + 64 <= nvec) {
        count_vec(vkeys[v + done * stride]);
        ++done;
    }
    if (v + done * stride < nvec) {  // the one partial step
        const VecT x = vkeys[v + done * stride];
        count_some(x.x);
        count_some(x.y);
        if constexpr (sizeof(KeyT) == 4) {
            count_some(x.z);
            count_some(x.w);
        }
    }
    // tail (n not a multiple of the vector width): first lanes of block 0
    if (blockIdx.x == 0 && nvec * VEC + threadIdx.x < a.n) count_some(keys[nvec * VEC + threadIdx.x]);
    __syncthreads();
    for (int j = threadIdx.x; j < static_cast<int>(a.num_passes) * RADIX; j += 256) {
        const uint32_t c = s_hist[j];
        if (c) atomicAdd(&a.ghist[j], c);
    }
}

// The same histograms with every counter kept in R copies, copy = lane % R.  The cost of an
// LDS atomic instruction is set by its worst bank: with one counter per digit, 64 random
// digits put 5 or 6 lanes on some bank (and equal digits put all 64 on one word); with the
// copies interleaved, the lanes of one instruction spread over R consecutive banks whatever
// their digits are.  2^28 keys x 4 digits: 0.18 ms against 0.29-0.34 ms for the kernel
// above (tools/micro/hist_bench.hip), which is the rate of a read-only sweep.  One workgroup
// per compute unit.
template <int BITS, typename KeyT>
struct BankedHistogram {
    static constexpr int RADIX = 1 << BITS;
    static constexpr int MAXP = (sizeof(KeyT) * 8 + BITS - 1) / BITS;
    static constexpr int COUNTERS = MAXP * RADIX;
    static constexpr int WORDS = 16384;  // 64 KiB (twice as many copies measured the same)
    static constexpr int R = WORDS / COUNTERS > 32 ? 32 : WORDS / COUNTERS;
    static constexpr int THREADS = 1024;
    static_assert(R >= 4 && (R & (R - 1)) == 0, "copies must be a power of two");
};

// ALL: a.num_passes equals MAXP (the usual case), so no pass needs a test.
template <int BITS, typename KeyT, bool ALL>
__global__ __launch_bounds__(1024) void global_histogram_banked_kernel(const GlobalHistArgs a)
{
    using B = BankedHistogram<BITS, KeyT>;
    constexpr int RADIX = B::RADIX, MAXP = B::MAXP, R = B::R, THREADS = B::THREADS;
    constexpr int VEC = 16 / sizeof(KeyT);
    constexpr int UNROLL = 4;  // vectors in flight per thread
    __shared__ uint32_t s_hist[B::COUNTERS * R];
    __shared__ uint8_t s_map[ALL ? 1 : kMaxMapEntries];  // partition mode: the digit table
    if (a.gate && *a.gate != a.gate_value) return;  // uniform
    for (int j = threadIdx.x; j < B::COUNTERS * R; j += THREADS) s_hist[j] = 0;
    if constexpr (!ALL) {
        if (a.digit_map)
            for (uint32_t j = threadIdx.x; j <= a.map_mask && j < kMaxMapEntries; j += THREADS) s_map[j] = a.digit_map[j];
    }
    __syncthreads();

    const KeyT* __restrict__ keys = static_cast<const KeyT*>(a.keys);
    const uint32_t copy = threadIdx.x & (R - 1);
    auto count = [&](KeyT k) {
        if constexpr (!ALL) {
            // partition mode: one pass over at most 16 mapped digits, where the copies matter
            // most (a wave's 64 lanes would otherwise queue on a handful of words)
            if (a.digit_map) {
                const uint32_t d = s_map[static_cast<uint32_t>(k >> a.begin_bit) & a.map_mask];
                atomicAdd(&s_hist[d * R + copy], 1u);
                return;
            }
        }
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            if (ALL || p < static_cast<int>(a.num_passes)) {
                const uint32_t shift = a.begin_bit + p * BITS;
                const uint32_t width = a.end_bit - shift < static_cast<uint32_t>(BITS) ? a.end_bit - shift : BITS;
                uint32_t d;
                if constexpr (sizeof(KeyT) == 4)
                    d = __builtin_amdgcn_ubfe(k, shift, width);
                else
                    d = static_cast<uint32_t>(k >> shift) & ((1u << width) - 1u);
                atomicAdd(&s_hist[(p * RADIX + d) * R + copy], 1u);
            }
        }
    };
    using VecT = typename std::conditional<sizeof(KeyT) == 4, uint4, ulonglong2>::type;
    auto count_vec = [&](const VecT& x) {
        count(x.x);
        count(x.y);
        if constexpr (sizeof(KeyT) == 4) {
            count(x.z);
            count(x.w);
        }
    };
    // nontemporal: the keys are read once here (0.205 -> 0.178 ms at 2^28 32-bit keys)
    auto load_vec = [](const VecT* p) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        VecT x;
        __builtin_memcpy(&x, &r, 16);
        return x;
    };
    const unsigned long long nvec = a.n / VEC;
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * THREADS;
    const VecT* __restrict__ vkeys = reinterpret_cast<const VecT*>(keys);
    unsigned long long v = static_cast<unsigned long long>(blockIdx.x) * THREADS + threadIdx.x;
    for (; v + (UNROLL - 1) * stride < nvec; v += UNROLL * stride) {
        VecT x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = load_vec(vkeys + v + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) count_vec(x[u]);
    }
    for (; v < nvec; v += stride) count_vec(vkeys[v]);
    // tail (n not a multiple of the vector width): first lanes of block 0
    if (blockIdx.x == 0 && nvec * VEC + threadIdx.x < a.n) count(keys[nvec * VEC + threadIdx.x]);
    __syncthreads();
    for (int j = threadIdx.x; j < static_cast<int>(a.num_passes) * RADIX; j += THREADS) {
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) c += s_hist[j * R + ((r + j) & (R - 1))];  // rotated: threads on distinct banks
        if (c) atomicAdd(&a.ghist[j], c);
    }
}

// In-place exclusive scan of each pass's global histogram; one workgroup of
// RADIX threads (rounded up to a wave) per pass.
//
// With `pass_plan` the workgroup of pass p also decides whether its pass runs.  A pass whose
// digit is the same for all n keys (the zero high bytes of a short code) is a stable identity
// and can be left out, but the ping-pong schedule and the side the caller was promised
// (src/main.rs:819-823) are fixed on the host before any key is read.  So passes are left out
// in pairs: if the number of such passes is odd, the last of them runs after all (it costs what
// a copy to the promised side would), and every pass that runs learns from bit 1 whether an odd
// number of passes before it was left out, i.e. whether its input lies in its output buffers.
//
// The other workgroups scan their histograms in place while this one looks at them.  The test
// "every word is 0 or n" gives the same answer on counts, on their exclusive scan and on any
// mixture of the two: with one occupied bin both are 0 or n everywhere, and at the second
// occupied bin of any other histogram the count and the scan both lie strictly between 0 and n.
//
// `keep_passes`: the first that many passes run whatever their digits are.  A sort that makes up
// its payload in pass 0 (PassArgs::index_payload) passes 2: pass 0 has to run for the payload to
// exist at all, and pass 1 with it for the count of passes left out to stay even.
template <int BITS>
__global__ void global_histogram_scan_kernel(uint32_t* ghist, uint32_t* pass_plan, uint32_t n, uint32_t num_passes,
                                             uint32_t keep_passes)
{
    constexpr int RADIX = 1 << BITS;
    __shared__ uint32_t s[RADIX];
    __shared__ uint32_t s_spread[kMaxPasses];  // pass q has keys in more than one bin
    uint32_t* h = ghist + static_cast<size_t>(blockIdx.x) * RADIX;
    const int t = threadIdx.x;
    const uint32_t own = t < RADIX ? h[t] : 0;
    if (t < RADIX) s[t] = own;
    if (t < kMaxPasses) s_spread[t] = static_cast<uint32_t>(t) < keep_passes ? 1u : 0u;
    __syncthreads();
    if (pass_plan && t < RADIX) {
        for (uint32_t q = 0; q < num_passes; ++q) {
            const uint32_t v = q == blockIdx.x ? own
                                               : __hip_atomic_load(ghist + q * RADIX + t, __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_AGENT);
            if (v != 0 && v != n) s_spread[q] = 1;  // benign race: every writer stores 1
        }
    }
    for (int o = 1; o < RADIX; o <<= 1) {
        uint32_t y = 0;
        if (t < RADIX && t >= o) y = s[t - o];
        __syncthreads();
        if (t < RADIX) s[t] += y;
        __syncthreads();
    }
    if (t < RADIX) h[t] = s[t] - own;
    if (pass_plan && t == 0) {
        const uint32_t p = blockIdx.x;
        uint32_t identities = 0, last = 0;
        for (uint32_t q = 0; q < num_passes; ++q)
            if (!s_spread[q]) { ++identities; last = q; }
        uint32_t left_out_before = 0;
        for (uint32_t q = 0; q < p; ++q)
            if (!s_spread[q] && !((identities & 1u) && q == last)) ++left_out_before;
        const uint32_t leave_out = !s_spread[p] && !((identities & 1u) && p == last);
        pass_plan[p] = leave_out | ((left_out_before & 1u) << 1);
    }
}

}  // namespace mrs

#endif
