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
#ifndef MRS_RADIX_PASS_HPP
#define MRS_RADIX_PASS_HPP

#include <type_traits>

#include "mrs_internal.h"

namespace mrs {

struct NoVal {};

enum PassMode { kModeScatter = 0, kModeOnesweep = 1 };

// Tile status word of the chained scan: the top two bits are the state, the rest a
// digit count.  0 = not published (the words are zeroed before every sort call),
// 1 = count of this tile alone, 2 = count of this tile and every tile before it.
// 32-bit words serve inputs below 2^30 elements; 64-bit words anything larger.
template <typename StatusT>
struct StatusBits {
    static constexpr int kValueBits = sizeof(StatusT) * 8 - 2;
    static constexpr StatusT kAggregate = static_cast<StatusT>(1) << kValueBits;
    static constexpr StatusT kInclusive = static_cast<StatusT>(2) << kValueBits;
    static constexpr StatusT kValueMask = kAggregate - 1;
    // predecessors fetched per look-back round trip (one register pair or register each)
    static constexpr int kWidth = sizeof(StatusT) == 4 ? 16 : 8;
};

// Timing-only ablations for diagnostic builds (results are wrong when non-zero):
// 1 = no look-back, 2 = no ranking, 4 = no payload, 8 = no global stores.
#ifndef MRS_ABLATE
#define MRS_ABLATE 0
#endif
constexpr int kAblate = MRS_ABLATE;

// Keeps a value live for the ablation builds that drop its store.
template <typename T>
__device__ __forceinline__ void keep_alive(const T& x)
{
    uint32_t w[(sizeof(T) + 3) / 4];
    __builtin_memcpy(w, &x, sizeof(T));
#pragma unroll
    for (unsigned i = 0; i < (sizeof(T) + 3) / 4; ++i) asm volatile("" ::"v"(w[i]));
}

// ticks of the 100 MHz constant clock a look-back may wait before giving up
constexpr unsigned long long kSpinTimeoutTicks = 400000000ull;  // 4 s

template <bool MAP, typename KeyT>
__device__ __forceinline__ uint32_t key_digit(KeyT k, uint32_t shift, uint32_t mask,
                                              const uint8_t* __restrict__ map)
{
    uint32_t d = static_cast<uint32_t>(k >> shift) & mask;
    if constexpr (MAP) d = map[d];
    return d;
}

// Number of set bits of (hi:lo) in lanes below the calling lane.
__device__ __forceinline__ uint32_t lanes_below(uint32_t lo, uint32_t hi)
{
    return __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
}

// Mask (as two halves) of the lanes of this wave, among those in `valid`, that hold
// the same digit as the caller.  Per digit bit: one ballot, then "lanes whose bit
// equals mine" = ~(ballot ^ x) with x = all-ones when my bit is set.
template <int BITS>
__device__ __forceinline__ void match_digit(uint32_t d, unsigned long long valid, uint32_t* out_lo, uint32_t* out_hi)
{
    uint32_t lo = static_cast<uint32_t>(valid), hi = static_cast<uint32_t>(valid >> 32);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const uint32_t x = 0u - ((d >> b) & 1u);
        const unsigned long long bal = __ballot(x != 0u);
        lo &= ~(static_cast<uint32_t>(bal) ^ x);
        hi &= ~(static_cast<uint32_t>(bal >> 32) ^ x);
    }
    *out_lo = lo;
    *out_hi = hi;
}

// Status words are single naturally aligned words that carry their own validity, so
// relaxed device-scope accesses are all the hand-off needs: no flag, no fence.
template <typename StatusT>
__device__ __forceinline__ StatusT status_load(const StatusT* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename StatusT>
__device__ __forceinline__ void status_store(StatusT* p, StatusT v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Decoupled look-back for one digit: sum of the digit counts of all tiles before
// `tile`.  Walks back over the predecessors' status words, adding aggregates until
// an inclusive prefix is found.  J predecessors are fetched per round trip: a
// dependent read costs microseconds on a busy chip, and the nearest inclusive
// prefix is as many tiles back as enter look-back during one such read.
template <int RADIX, typename StatusT>
__device__ __forceinline__ uint32_t look_back(const PassArgs& a, uint32_t tile, uint32_t digit, bool* failed)
{
    using SB = StatusBits<StatusT>;
    constexpr int J = SB::kWidth;
    const StatusT* status = static_cast<const StatusT*>(a.status);
    uint32_t excl = 0;
    long long t = static_cast<long long>(tile) - 1;  // nearest predecessor not yet accounted for
    unsigned long long t0 = 0;
    uint32_t spins = 0;
    while (true) {
        StatusT v[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const long long tj = t - j;
            // tile 0 always publishes an inclusive prefix, so the walk ends before it
            // would need a tile below 0
            v[j] = tj >= 0 ? status_load(status + static_cast<size_t>(tj) * RADIX + digit) : static_cast<StatusT>(0);
        }
        bool open = true, found = false;
        int consumed = 0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (open) {
                if (v[j] >> SB::kValueBits) {
                    excl += static_cast<uint32_t>(v[j] & SB::kValueMask);
                    ++consumed;
                    if (v[j] & SB::kInclusive) { found = true; open = false; }
                } else {
                    open = false;  // not published yet: everything further back waits
                }
            }
        }
#ifdef MRS_STAMPS
        if (a.stamps && digit == 0) {
            a.stamps[static_cast<size_t>(tile) * 16 + 8] += 1;               // round trips
            a.stamps[static_cast<size_t>(tile) * 16 + 9] += consumed;        // tiles walked
            a.stamps[static_cast<size_t>(tile) * 16 + 10] += consumed == 0;  // rounds that found nothing
        }
#endif
        if (found) return excl;
        t -= consumed;
        if (consumed == 0) {
            ++spins;
            if ((spins & 63u) == 0u) {
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > kSpinTimeoutTicks ||
                    __hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    *failed = true;
                    return excl;
                }
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

// Diagnostic builds (-DMRS_STAMPS) record the 100 MHz clock at phase boundaries of
// every tile into a.stamps[tile*16 + k]; no stamp exists in a normal build.
#ifdef MRS_STAMPS
#define MRS_STAMP(k)                                                                               \
    do {                                                                                           \
        if (a.stamps && tid == 0) a.stamps[static_cast<size_t>(tile) * 16 + (k)] = wall_clock64(); \
    } while (0)
#else
#define MRS_STAMP(k) do { } while (0)
#endif

template <int BITS, int THREADS, int KPT, typename KeyT, typename ValT>
struct PassShared {
    static constexpr int RADIX = 1 << BITS;
    static constexpr int WAVES = THREADS / 64;
    static constexpr int TILE = THREADS * KPT;
    static constexpr bool HAS_VAL = !std::is_same<ValT, NoVal>::value;
    static constexpr int VAL_BYTES = HAS_VAL ? sizeof(ValT) : 0;
    static constexpr int STAGE_ELEM = sizeof(KeyT) > VAL_BYTES ? sizeof(KeyT) : VAL_BYTES;
    static constexpr int SCAN_WAVES = (RADIX + 63) / 64;

    __attribute__((aligned(16))) unsigned char stage[TILE * STAGE_ELEM];
    uint32_t wave_hist[WAVES][RADIX];  // per-wave digit counts, later per-wave bases
    uint32_t gbase[RADIX];             // global position of slot 0 of each digit, minus its tile offset
    uint32_t wsum[SCAN_WAVES];
    uint32_t tile;
    uint32_t abort;
};

// Everything after the tile id is known.  FULL = the tile has TILE valid elements.
template <int MODE, bool MAP, bool FULL, typename StatusT, int BITS, int THREADS, int KPT, typename KeyT,
          typename ValT>
__device__ __forceinline__ void process_tile(const PassArgs& a, PassShared<BITS, THREADS, KPT, KeyT, ValT>& sh,
                                             const uint32_t tile, const uint32_t tile_valid)
{
    using S = PassShared<BITS, THREADS, KPT, KeyT, ValT>;
    using SB = StatusBits<StatusT>;
    constexpr int RADIX = S::RADIX;
    constexpr int WAVES = S::WAVES;
    constexpr int TILE = S::TILE;
    constexpr bool HAS_VAL = S::HAS_VAL;
    constexpr int SCAN_WAVES = S::SCAN_WAVES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const unsigned long long tile_base = static_cast<unsigned long long>(tile) * TILE;
    const uint32_t my_first = wave * 64 * KPT + lane;  // element of item 0 inside the tile

    // ---- load keys -------------------------------------------------------------------
    const KeyT* __restrict__ keys_in = static_cast<const KeyT*>(a.keys_in) + tile_base + my_first;
    KeyT key[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if (FULL || my_first + i * 64 < tile_valid) key[i] = keys_in[i * 64];
        else key[i] = static_cast<KeyT>(~static_cast<KeyT>(0));
    }
#ifdef MRS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MRS_STAMP(2);
#endif

    // ---- rank inside the wave ------------------------------------------------------
    // pos[i] = number of earlier elements of this wave with the same digit.
    // wave_hist[wave][d] carries the running count; the lowest lane of each digit
    // group advances it.  LDS operations of one wave execute in order, so the read
    // of item i+1 sees the update of item i.
    uint32_t pos[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if (kAblate & 2) { pos[i] = 0; continue; }
        const uint32_t d = key_digit<MAP>(key[i], a.shift, a.mask, a.digit_map);
        const unsigned long long valid = FULL ? ~0ull : __ballot(my_first + i * 64 < tile_valid);
        uint32_t mlo, mhi;
        match_digit<BITS>(d, valid, &mlo, &mhi);
        const uint32_t below = lanes_below(mlo, mhi);
        const uint32_t prev = sh.wave_hist[wave][d];
        const bool mine = FULL || (my_first + i * 64 < tile_valid);
        if (mine && below == 0) sh.wave_hist[wave][d] = prev + __popc(mlo) + __popc(mhi);
        pos[i] = prev + below;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    MRS_STAMP(3);

    // ---- tile histogram, per-wave bases, digit offsets ---------------------------------
    uint32_t tile_count = 0;  // keys of this tile with digit == tid
    uint32_t incl = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = sh.wave_hist[w][tid];
            sh.wave_hist[w][tid] = tile_count;
            tile_count += c;
        }
        if (MODE == kModeOnesweep) {
            // publish the aggregate as early as possible
            StatusT* mine = static_cast<StatusT*>(a.status) + static_cast<size_t>(tile) * RADIX + tid;
            status_store<StatusT>(mine, (tile == 0 ? SB::kInclusive : SB::kAggregate) | tile_count);
        }
        incl = tile_count;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63 || tid == RADIX - 1) sh.wsum[wave] = incl;
    }
    __syncthreads();
    uint32_t tile_off = 0;  // keys of this tile with a smaller digit
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < SCAN_WAVES; ++w)
            if (w < wave) tile_off += sh.wsum[w];
        tile_off += incl - tile_count;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sh.wave_hist[w][tid] += tile_off;
    }
    __syncthreads();
    MRS_STAMP(4);

    // ---- local sort of the keys through LDS ---------------------------------------
    KeyT* stage_keys = reinterpret_cast<KeyT*>(sh.stage);
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t d = key_digit<MAP>(key[i], a.shift, a.mask, a.digit_map);
        pos[i] += sh.wave_hist[wave][d];
        if (kAblate & 2) pos[i] = my_first + i * 64;
        if (FULL || my_first + i * 64 < tile_valid) stage_keys[pos[i]] = key[i];
    }
    // The payload is fetched now: its latency hides behind the look-back below, and its
    // registers take over from the keys, which live in LDS from here on.
    ValT val[KPT];
    if constexpr (HAS_VAL) {
        if (!(kAblate & 4)) {
            const ValT* __restrict__ vals_in = static_cast<const ValT*>(a.vals_in) + tile_base + my_first;
#pragma unroll
            for (int i = 0; i < KPT; ++i)
                if (FULL || my_first + i * 64 < tile_valid) val[i] = vals_in[i * 64];
        }
    }
    MRS_STAMP(5);

    // ---- global base of every digit of this tile ------------------------------------
    if (tid < RADIX) {
        uint32_t base;
        if (MODE == kModeScatter) {
            base = a.scanned[static_cast<size_t>(tid) * a.num_tiles + tile];
        } else {
            uint32_t excl = 0;
            if (tile != 0) {
                bool failed = false;
                if (!(kAblate & 1)) excl = look_back<RADIX, StatusT>(a, tile, tid, &failed);
                if (failed) {
                    atomicExch(a.error_flag, 1u);
                    sh.abort = 1;
                }
                StatusT* mine = static_cast<StatusT*>(a.status) + static_cast<size_t>(tile) * RADIX + tid;
                status_store<StatusT>(mine, SB::kInclusive | static_cast<StatusT>(excl + tile_count));
            }
            base = a.digit_base[tid] + excl;
        }
        sh.gbase[tid] = base - tile_off;
    }
    __syncthreads();
    if (MODE == kModeOnesweep) {
        if (sh.abort) return;  // never write through an offset that was not resolved
    }
    MRS_STAMP(6);

    // ---- write the keys back in slot order ---------------------------------------------
    KeyT* __restrict__ keys_out = static_cast<KeyT*>(a.keys_out);
    uint32_t gpos[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t slot = i * THREADS + tid;
        if (FULL || slot < tile_valid) {
            const KeyT k = stage_keys[slot];
            const uint32_t d = key_digit<MAP>(k, a.shift, a.mask, a.digit_map);
            gpos[i] = sh.gbase[d] + slot;
            if (kAblate & 3) gpos[i] = static_cast<uint32_t>(tile_base) + slot;
            if (kAblate & 8) { keep_alive(k); keep_alive(gpos[i]); }
            else keys_out[gpos[i]] = k;
        }
    }

    // ---- payload: same permutation ------------------------------------------------------
    if constexpr (HAS_VAL && !(kAblate & 4)) {
        ValT* stage_vals = reinterpret_cast<ValT*>(sh.stage);
        ValT* __restrict__ vals_out = static_cast<ValT*>(a.vals_out);
        __syncthreads();  // the key image is dead
#pragma unroll
        for (int i = 0; i < KPT; ++i)
            if (FULL || my_first + i * 64 < tile_valid) stage_vals[pos[i]] = val[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t slot = i * THREADS + tid;
            if (FULL || slot < tile_valid) {
                const ValT v = stage_vals[slot];
                if (kAblate & 8) keep_alive(v);
                else vals_out[gpos[i]] = v;
            }
        }
    }
#ifdef MRS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MRS_STAMP(7);
#endif
}

template <int MODE, bool MAP, typename StatusT, int BITS, int THREADS, int KPT, typename KeyT, typename ValT>
__global__ __launch_bounds__(THREADS) void digit_pass_kernel(const PassArgs a)
{
    using S = PassShared<BITS, THREADS, KPT, KeyT, ValT>;
    static_assert(THREADS % 64 == 0 && THREADS >= S::RADIX, "one thread per digit is required");
    __shared__ S sh;

    const int tid = threadIdx.x;
#ifdef MRS_STAMPS
    const unsigned long long t_begin = wall_clock64();
#endif
    // ---- tile id ---------------------------------------------------------------------
    // ONESWEEP takes a ticket so that every tile a workgroup can wait for is owned by
    // a workgroup that is already running.
    if (MODE == kModeOnesweep) {
        if (tid == 0) {
            sh.tile = atomicAdd(a.tile_counter, 1u);
            sh.abort = 0;
        }
    }
    for (int j = tid; j < S::WAVES * S::RADIX; j += THREADS) (&sh.wave_hist[0][0])[j] = 0;
    __syncthreads();
    const uint32_t tile = MODE == kModeOnesweep ? sh.tile : blockIdx.x;
#ifdef MRS_STAMPS
    if (a.stamps && tid == 0) a.stamps[static_cast<size_t>(tile) * 16 + 0] = t_begin;
#endif
    MRS_STAMP(1);

    const unsigned long long tile_base = static_cast<unsigned long long>(tile) * S::TILE;
    const uint32_t tile_valid = (a.n - tile_base) < static_cast<unsigned long long>(S::TILE)
                                    ? static_cast<uint32_t>(a.n - tile_base)
                                    : S::TILE;
    if (tile_valid == S::TILE)
        process_tile<MODE, MAP, true, StatusT>(a, sh, tile, tile_valid);
    else
        process_tile<MODE, MAP, false, StatusT>(a, sh, tile, tile_valid);
}

template <int MODE, bool MAP, int BITS, int THREADS, int KPT, typename KeyT, typename ValT>
int launch_pass_cfg(const PassArgs& a, hipStream_t s)
{
    // 32-bit status words hold counts below 2^30; the scatter mode has no status words
    if (MODE == kModeOnesweep && a.wide_status)
        hipLaunchKernelGGL((digit_pass_kernel<MODE, MAP, unsigned long long, BITS, THREADS, KPT, KeyT, ValT>),
                           dim3(a.num_tiles), dim3(THREADS), 0, s, a);
    else
        hipLaunchKernelGGL((digit_pass_kernel<MODE, MAP, uint32_t, BITS, THREADS, KPT, KeyT, ValT>),
                           dim3(a.num_tiles), dim3(THREADS), 0, s, a);
    MRS_CHECK_LAUNCH("digit_pass_kernel");
    return MRS_OK;
}

}  // namespace mrs

#endif
