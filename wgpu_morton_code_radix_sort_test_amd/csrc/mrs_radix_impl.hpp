// Device code of one LSD radix digit pass for gfx950 (wave64).
//
// A workgroup of THREADS threads owns a tile of THREADS*KPT consecutive
// elements.  Element order inside the tile (it defines stability): wave w owns
// the contiguous slice [w*64*KPT, (w+1)*64*KPT); item i of lane l is element
// w*64*KPT + i*64 + l, so every wave-wide load is one contiguous 256-byte
// (u32 key) or 512-byte (u64 key) run.
//
// The pass does what the reference's radix_sort_index does per 256-key block
// (src/shaders/radix_sort_index.wgsl:255-308) -- stable rank inside the block,
// LDS-staged local sort, slot-ordered global write -- but ranks with wave-wide
// ballots (one match per key, no LDS atomics, no 40-barrier bitmap build) and
// with many keys per thread.  Two ways of obtaining the global offset of
// (tile, digit):
//   SCATTER  : read it from the scanned digit-major block histogram, exactly
//              the reference's `storage_histograms[num_workgroups*digit+block]`
//              (radix_sort_index.wgsl:282-290);
//   ONESWEEP : chained scan with decoupled look-back over per-tile status
//              words, so the pass reads and writes every element once and no
//              block histogram is materialised.
#ifndef MRS_RADIX_IMPL_HPP
#define MRS_RADIX_IMPL_HPP

#include <type_traits>

#include "mrs_internal.h"

namespace mrs {

struct NoVal {};

enum PassMode { kModeScatter = 0, kModeOnesweep = 1 };

// status word: [63:34] tag (pass id, never 0) | [33:32] flag | [31:0] count
constexpr unsigned long long kFlagAggregate = 1ull << 32;
constexpr unsigned long long kFlagInclusive = 2ull << 32;
// ticks of the 100 MHz constant clock a look-back may wait before giving up
constexpr unsigned long long kSpinTimeoutTicks = 400000000ull;  // 4 s

template <typename KeyT>
__device__ __forceinline__ uint32_t key_digit(KeyT k, uint32_t shift, uint32_t mask,
                                              const uint8_t* __restrict__ map)
{
    uint32_t d = static_cast<uint32_t>(k >> shift) & mask;
    if (map) d = map[d];
    return d;
}

// Number of set bits of m in lanes below the calling lane.
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}

// Mask of the valid lanes of this wave that hold the same digit as the caller.
template <int BITS>
__device__ __forceinline__ unsigned long long match_digit(uint32_t d, bool valid)
{
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

__device__ __forceinline__ unsigned long long status_load(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void status_store(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE, int BITS, int THREADS, int KPT, typename KeyT, typename ValT>
__global__ __launch_bounds__(THREADS) void digit_pass_kernel(const PassArgs a)
{
    constexpr int RADIX = 1 << BITS;
    constexpr int WAVES = THREADS / 64;
    constexpr int TILE = THREADS * KPT;
    constexpr bool HAS_VAL = !std::is_same<ValT, NoVal>::value;
    constexpr int VAL_BYTES = HAS_VAL ? sizeof(ValT) : 0;
    constexpr int STAGE_ELEM = sizeof(KeyT) > VAL_BYTES ? sizeof(KeyT) : VAL_BYTES;
    constexpr int SCAN_WAVES = (RADIX + 63) / 64;
    static_assert(THREADS % 64 == 0 && THREADS >= RADIX, "one thread per digit is required");

    __shared__ __attribute__((aligned(16))) unsigned char s_stage[TILE * STAGE_ELEM];
    __shared__ uint32_t s_wave_hist[WAVES][RADIX];
    __shared__ uint32_t s_gbase[RADIX];
    __shared__ uint32_t s_wsum[SCAN_WAVES];
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_abort;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // ---- tile id -----------------------------------------------------------
    // ONESWEEP takes a ticket so that every tile a workgroup can wait for is
    // owned by a workgroup that is already running.
    if (MODE == kModeOnesweep) {
        if (tid == 0) {
            s_tile = atomicAdd(a.tile_counter, 1u);
            s_abort = 0;
        }
    }
    for (int j = tid; j < WAVES * RADIX; j += THREADS) (&s_wave_hist[0][0])[j] = 0;
    __syncthreads();
    const uint32_t tile = MODE == kModeOnesweep ? s_tile : blockIdx.x;

    const unsigned long long tile_base = static_cast<unsigned long long>(tile) * TILE;
    const uint32_t tile_valid =
        (a.n - tile_base) < static_cast<unsigned long long>(TILE) ? static_cast<uint32_t>(a.n - tile_base) : TILE;
    const bool full_tile = tile_valid == TILE;

    // ---- load keys -----------------------------------------------------------
    const KeyT* __restrict__ keys_in = static_cast<const KeyT*>(a.keys_in);
    const uint32_t my_first = wave * 64 * KPT + lane;  // element of item 0 inside the tile
    KeyT key[KPT];
    if (full_tile) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) key[i] = keys_in[tile_base + my_first + i * 64];
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t e = my_first + i * 64;
            key[i] = e < tile_valid ? keys_in[tile_base + e] : static_cast<KeyT>(~static_cast<KeyT>(0));
        }
    }
    // payload loads are issued early so that their latency hides behind ranking
    ValT val[KPT];
    if constexpr (HAS_VAL) {
        const ValT* __restrict__ vals_in = static_cast<const ValT*>(a.vals_in);
        if (full_tile) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) val[i] = vals_in[tile_base + my_first + i * 64];
        } else {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t e = my_first + i * 64;
                if (e < tile_valid) val[i] = vals_in[tile_base + e];
            }
        }
    }

    // ---- rank inside the wave ------------------------------------------------
    // rank[i] = number of earlier elements of this wave with the same digit.
    // s_wave_hist[wave][d] carries the running count; the lowest lane of each
    // digit group advances it.  LDS operations of one wave execute in order, so
    // the read of item i+1 sees the update of item i.
    uint32_t pos[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const bool valid = full_tile || (my_first + i * 64 < tile_valid);
        const uint32_t d = key_digit(key[i], a.shift, a.mask, a.digit_map);
        const unsigned long long m = match_digit<BITS>(d, valid);
        const uint32_t below = lanes_below(m);
        const uint32_t prev = s_wave_hist[wave][d];
        if (valid && below == 0) s_wave_hist[wave][d] = prev + __popcll(m);
        pos[i] = prev + below;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // ---- tile histogram, per-wave bases, digit offsets ------------------------
    uint32_t tile_count = 0;   // keys of this tile with digit == tid
    uint32_t incl = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = s_wave_hist[w][tid];
            s_wave_hist[w][tid] = tile_count;
            tile_count += c;
        }
        if (MODE == kModeOnesweep) {
            // publish the aggregate as early as possible
            unsigned long long* mine = a.status + static_cast<size_t>(tile) * RADIX + tid;
            const unsigned long long tagbits = static_cast<unsigned long long>(a.tag) << 34;
            status_store(mine, tagbits | (tile == 0 ? kFlagInclusive : kFlagAggregate) | tile_count);
        }
        incl = tile_count;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63 || tid == RADIX - 1) s_wsum[wave] = incl;
    }
    __syncthreads();
    uint32_t tile_off = 0;     // keys of this tile with a smaller digit
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < SCAN_WAVES; ++w)
            if (w < wave) tile_off += s_wsum[w];
        tile_off += incl - tile_count;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s_wave_hist[w][tid] += tile_off;
    }
    __syncthreads();

    // ---- local sort of the keys through LDS -----------------------------------
    KeyT* stage_keys = reinterpret_cast<KeyT*>(s_stage);
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const bool valid = full_tile || (my_first + i * 64 < tile_valid);
        const uint32_t d = key_digit(key[i], a.shift, a.mask, a.digit_map);
        pos[i] += s_wave_hist[wave][d];
        if (valid) stage_keys[pos[i]] = key[i];
    }

    // ---- global base of every digit of this tile ------------------------------
    if (tid < RADIX) {
        uint32_t base;
        if (MODE == kModeScatter) {
            base = a.scanned[static_cast<size_t>(tid) * a.num_tiles + tile];
        } else {
            uint32_t excl = 0;
            if (tile != 0) {
                const unsigned long long tagbits = static_cast<unsigned long long>(a.tag) << 34;
                long long t = static_cast<long long>(tile) - 1;
                unsigned long long t0 = 0;
                uint32_t spins = 0;
                bool failed = false;
                while (true) {
                    const unsigned long long v = status_load(a.status + static_cast<size_t>(t) * RADIX + tid);
                    if ((v >> 34) == a.tag) {
                        excl += static_cast<uint32_t>(v);
                        if (v & kFlagInclusive) break;
                        --t;
                        spins = 0;
                        continue;
                    }
                    // predecessor has not published yet
                    ++spins;
                    if ((spins & 63u) == 0u) {
                        const unsigned long long now = wall_clock64();
                        if (t0 == 0) t0 = now;
                        if (now - t0 > kSpinTimeoutTicks ||
                            __hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                            failed = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (failed) {
                    atomicExch(a.error_flag, 1u);
                    s_abort = 1;
                }
                unsigned long long* mine = a.status + static_cast<size_t>(tile) * RADIX + tid;
                status_store(mine, tagbits | kFlagInclusive | static_cast<uint32_t>(excl + tile_count));
            }
            base = a.digit_base[tid] + excl;
        }
        s_gbase[tid] = base - tile_off;
    }
    __syncthreads();
    if (MODE == kModeOnesweep) {
        if (s_abort) return;  // never write through an offset that was not resolved
    }

    // ---- write the keys back in slot order -------------------------------------
    KeyT* __restrict__ keys_out = static_cast<KeyT*>(a.keys_out);
    uint32_t gpos[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t slot = i * THREADS + tid;
        if (full_tile || slot < tile_valid) {
            const KeyT k = stage_keys[slot];
            const uint32_t d = key_digit(k, a.shift, a.mask, a.digit_map);
            gpos[i] = s_gbase[d] + slot;
            keys_out[gpos[i]] = k;
        }
    }

    // ---- payload: same permutation ----------------------------------------------
    if constexpr (HAS_VAL) {
        ValT* stage_vals = reinterpret_cast<ValT*>(s_stage);
        ValT* __restrict__ vals_out = static_cast<ValT*>(a.vals_out);
        __syncthreads();  // the key image is dead
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const bool valid = full_tile || (my_first + i * 64 < tile_valid);
            if (valid) stage_vals[pos[i]] = val[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t slot = i * THREADS + tid;
            if (full_tile || slot < tile_valid) vals_out[gpos[i]] = stage_vals[slot];
        }
    }
}

// ---- per-tile digit histogram, digit-major (three-kernel pass) ------------------
// radix_sort_compute_histogram, src/shaders/radix_sort_histogram.wgsl:42-60, for
// a tile of `tile_items` keys instead of 256.
template <int BITS, typename KeyT>
__global__ __launch_bounds__(256) void block_histogram_kernel(const KeyT* __restrict__ keys, uint32_t n,
                                                              uint32_t tile_items, uint32_t num_tiles,
                                                              uint32_t shift, uint32_t mask,
                                                              uint32_t* __restrict__ hist)
{
    constexpr int RADIX = 1 << BITS;
    __shared__ uint32_t s_hist[RADIX];
    const uint32_t tile = blockIdx.x;
    for (int j = threadIdx.x; j < RADIX; j += 256) s_hist[j] = 0;
    __syncthreads();
    const unsigned long long base = static_cast<unsigned long long>(tile) * tile_items;
    const unsigned long long end = base + tile_items < n ? base + tile_items : n;
    for (unsigned long long i = base + threadIdx.x; i < end; i += 256) {
        const uint32_t d = static_cast<uint32_t>(keys[i] >> shift) & mask;
        atomicAdd(&s_hist[d], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < RADIX; d += 256) hist[static_cast<size_t>(d) * num_tiles + tile] = s_hist[d];
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
};

template <int BITS, typename KeyT>
__global__ __launch_bounds__(256) void global_histogram_kernel(const GlobalHistArgs a)
{
    constexpr int RADIX = 1 << BITS;
    constexpr int MAXP = (sizeof(KeyT) * 8 + BITS - 1) / BITS;
    constexpr int VEC = 16 / sizeof(KeyT);
    __shared__ uint32_t s_hist[MAXP * RADIX];
    for (int j = threadIdx.x; j < MAXP * RADIX; j += 256) s_hist[j] = 0;
    __syncthreads();

    const KeyT* __restrict__ keys = static_cast<const KeyT*>(a.keys);
    const int lane = threadIdx.x & 63;

    auto count = [&](KeyT k, bool valid) {
        for (uint32_t p = 0; p < a.num_passes; ++p) {
            uint32_t d;
            if (a.digit_map) {
                d = a.digit_map[static_cast<uint32_t>(k >> a.begin_bit) & a.map_mask];
            } else {
                const uint32_t shift = a.begin_bit + p * BITS;
                const uint32_t width = a.end_bit - shift < static_cast<uint32_t>(BITS) ? a.end_bit - shift : BITS;
                d = static_cast<uint32_t>(k >> shift) & ((1u << width) - 1u);
            }
            // a wave whose lanes all hold one digit (constant high bytes, sorted
            // or all-equal input) would serialise 64 ways on one LDS word
            const unsigned long long act = __ballot(valid);
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
            const bool uniform = __ballot(valid && d == d0) == act && act == ~0ull;
            if (uniform) {
                if (lane == 0) atomicAdd(&s_hist[p * RADIX + d0], 64u);
            } else if (valid) {
                atomicAdd(&s_hist[p * RADIX + d], 1u);
            }
        }
    };

    const unsigned long long nvec = a.n / VEC;
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * 256;
    using VecT = typename std::conditional<sizeof(KeyT) == 4, uint4, ulonglong2>::type;
    const VecT* __restrict__ vkeys = reinterpret_cast<const VecT*>(keys);
    // every lane of a wave runs the same number of iterations (inactive lanes
    // carry valid = false) because `count` uses wave-wide ballots
    const unsigned long long first = static_cast<unsigned long long>(blockIdx.x) * 256 + threadIdx.x;
    for (unsigned long long base = first - lane; base < nvec; base += stride) {
        const unsigned long long v = base + lane;
        const bool valid = v < nvec;
        VecT x;
        if (valid) x = vkeys[v];
        if constexpr (sizeof(KeyT) == 4) {
            count(valid ? x.x : 0, valid);
            count(valid ? x.y : 0, valid);
            count(valid ? x.z : 0, valid);
            count(valid ? x.w : 0, valid);
        } else {
            count(valid ? x.x : 0, valid);
            count(valid ? x.y : 0, valid);
        }
    }
    // tail (n not a multiple of the vector width): first wave of block 0
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        const unsigned long long i = nvec * VEC + threadIdx.x;
        const bool valid = i < a.n;
        count(valid ? keys[i] : 0, valid);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < static_cast<int>(a.num_passes) * RADIX; j += 256) {
        const uint32_t c = s_hist[j];
        if (c) atomicAdd(&a.ghist[j], c);
    }
}

// In-place exclusive scan of each pass's global histogram; one workgroup of
// RADIX threads (rounded up to a wave) per pass.
template <int BITS>
__global__ void global_histogram_scan_kernel(uint32_t* ghist)
{
    constexpr int RADIX = 1 << BITS;
    __shared__ uint32_t s[RADIX];
    uint32_t* h = ghist + static_cast<size_t>(blockIdx.x) * RADIX;
    const int t = threadIdx.x;
    const uint32_t own = t < RADIX ? h[t] : 0;
    if (t < RADIX) s[t] = own;
    __syncthreads();
    for (int o = 1; o < RADIX; o <<= 1) {
        uint32_t y = 0;
        if (t < RADIX && t >= o) y = s[t - o];
        __syncthreads();
        if (t < RADIX) s[t] += y;
        __syncthreads();
    }
    if (t < RADIX) h[t] = s[t] - own;
}

// ---- launchers ---------------------------------------------------------------------
template <int MODE, int BITS, int THREADS, int KPT, typename KeyT, typename ValT>
int launch_pass_cfg(const PassArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL((digit_pass_kernel<MODE, BITS, THREADS, KPT, KeyT, ValT>), dim3(a.num_tiles), dim3(THREADS),
                       0, s, a);
    MRS_CHECK_LAUNCH("digit_pass_kernel");
    return MRS_OK;
}

}  // namespace mrs

#endif
