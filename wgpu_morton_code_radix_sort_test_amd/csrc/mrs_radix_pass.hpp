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

// How a wave ranks its keys (position of a key among the keys of its wave with the same
// digit, earlier elements first).
//   kRankMatch : every lane finds the lanes holding its digit with one ballot per digit bit;
//                the first of them advances the wave's counter in LDS.  About 45 vector
//                instructions per key for 8-bit digits.
//   kRankAtomic: one returning LDS atomic add per key on the wave's counter (none when the
//                whole wave holds one digit).  The order in
//                which the lanes of one instruction that hit the same counter are served is
//                not part of the ISA contract; on gfx950 it is ascending lane order, which
//                is what stability needs.  The host verifies that on the device at context
//                creation and uses kRankMatch if the check ever fails.
enum RankMode { kRankMatch = 0, kRankAtomic = 1 };

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
// 1 = no look-back, 2 = no ranking, 4 = no payload, 8 = no global stores,
// 16 = no look-back, but the tile's digit runs go where tiles of equal digit counts would put
//      them (the scatter pattern without the chain),
// 32 = everything computed, but the output addresses are in tile order (the chain without the
//      scatter pattern),
// 64 = with 32: every address one element further (sequential, but no store is line-aligned),
// 128 = with 32: aligned 256-byte segments, each sent to a pseudo-random place of the array,
// 256 = with 32: the digit's base is still read from LDS and added, multiplied by a zero the
//       compiler cannot see (the address keeps its dependency on two chained LDS reads).
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

// Makes the compiler forget how a value was computed (no instruction is emitted).
__device__ __forceinline__ void launder(uint32_t& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void launder(unsigned long long& x)
{
    uint32_t lo = static_cast<uint32_t>(x), hi = static_cast<uint32_t>(x >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    x = (static_cast<unsigned long long>(hi) << 32) | lo;
}

// ticks of the 100 MHz constant clock a look-back may wait before giving up
constexpr unsigned long long kSpinTimeoutTicks = 400000000ull;  // 4 s

// Class of a key among `count` ascending splitter values: twice the number of splitters the key
// has reached, minus one for every splitter marked in `exact_mask` that it equals, so that the
// keys equal to such a splitter form a class of their own between their neighbours (the sharded
// sort cuts inside that class by element order).
template <typename KeyT>
__device__ __forceinline__ uint32_t splitter_class(KeyT k, const KeyT* split, uint32_t count, uint32_t exact_mask)
{
    uint32_t c = 0;
    for (uint32_t r = 0; r < count; ++r) {
        const KeyT v = split[r];
        c += k >= v ? 2u : 0u;
        c -= (k == v && ((exact_mask >> r) & 1u)) ? 1u : 0u;
    }
    return c;
}

// MAP: 0 = the digit is the key's bit field; kMapTable = the bit field indexes a table of
// digits; kMapTableAndSplitters = table entries equal to kMapBySplitters mark buckets that hold
// a splitter, whose keys are compared one by one.  The last form puts a data-dependent branch
// behind every table lookup (the lookups of a thread's keys no longer overlap), so the host
// selects it only for tables that do contain the mark: skewed inputs of the sharded sort.
// kMapWindow: the digit is a bit field of (key >> shift) - window_base, see PassArgs.
// kMapRank: the digit is the number of marked fine cells at or before the key's, see PassArgs.
enum MapMode { kMapNone = 0, kMapTable = 1, kMapTableAndSplitters = 2, kMapWindow = 3, kMapRank = 4 };

template <int MAP, typename KeyT>
__device__ __forceinline__ uint32_t key_digit(KeyT k, const PassArgs& a)
{
    if constexpr (MAP == kMapWindow) return ((static_cast<uint32_t>(k >> a.shift) - a.window_base) >> a.window_low) & a.mask;
    if constexpr (MAP == kMapRank) {
        const uint32_t k32 = static_cast<uint32_t>(k);
        const uint32_t e = a.cell_entry[((k32 >> a.shift) & a.mask) - a.cell_first];
        const uint32_t f = (e & 0xFFFFu) + (__builtin_amdgcn_ubfe(k32, 0u, a.shift) >> (e >> 16));
        const unsigned long long w = a.rank_words[f >> 5];
        return static_cast<uint32_t>(w >> 32) + __popc(static_cast<uint32_t>(w) & (0xFFFFFFFFu >> (31u - (f & 31u))));
    }
    uint32_t d = static_cast<uint32_t>(k >> a.shift) & a.mask;
    if constexpr (MAP == kMapTable || MAP == kMapTableAndSplitters) d = a.digit_map[d];
    if constexpr (MAP == kMapTableAndSplitters) {
        if (d == kMapBySplitters) d = splitter_class<KeyT>(k, static_cast<const KeyT*>(a.splitters), a.num_splitters, a.exact_mask);
    }
    return d;
}

// Number of set bits of (hi:lo) in lanes below the calling lane.
__device__ __forceinline__ uint32_t lanes_below(uint32_t lo, uint32_t hi)
{
    return __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
}

// Mask (as two halves) of the lanes of this wave, among those in `valid`, that hold
// the same digit as the caller.  Per digit bit: x = all-ones where my bit is set, one
// compare yields the ballot of x, and "lanes whose bit equals mine" = ~(ballot ^ x) is
// folded into the running mask by one three-input bit operation per half
// (truth table A & ~(B ^ C) = 0xF0 & ~(0xCC ^ 0xAA) = 0x90): four instructions per bit.
template <int BITS>
__device__ __forceinline__ void match_digit(uint32_t d, unsigned long long valid, uint32_t* out_lo, uint32_t* out_hi)
{
    uint32_t lo = static_cast<uint32_t>(valid), hi = static_cast<uint32_t>(valid >> 32);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const uint32_t x = static_cast<uint32_t>(__builtin_amdgcn_sbfe(d, b, 1));
        const unsigned long long bal = __builtin_amdgcn_uicmp(x, 0u, 33 /* != */);
        lo = __builtin_amdgcn_bitop3_b32(lo, static_cast<uint32_t>(bal), x, 0x90);
        hi = __builtin_amdgcn_bitop3_b32(hi, static_cast<uint32_t>(bal >> 32), x, 0x90);
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

// Decoupled look-back, one window: the status words of one digit of the J tiles
// first, first-1, ... are requested.  The first tile of a chain (`floor`: tile 0, or the first
// tile of a segment when the pass runs one chain per segment) always publishes an inclusive
// prefix, so a walk ends at it: slots that would fall below it re-read it and are never used.
template <int RADIX, typename StatusT, int J>
__device__ __forceinline__ void window_issue(const StatusT* status, int first, uint32_t digit, StatusT (&v)[J],
                                             int floor = 0)
{
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int tj = max(first - j, floor);
        if constexpr (sizeof(StatusT) == 4) {
            // at most 2^22 tiles of 256 words: the byte offset fits 32 bits
            const uint32_t off = (static_cast<uint32_t>(tj) * RADIX + digit) * 4u;
            v[j] = status_load(reinterpret_cast<const StatusT*>(reinterpret_cast<const char*>(status) + off));
        } else {
            v[j] = status_load(status + static_cast<size_t>(tj) * RADIX + digit);
        }
    }
}

// Sum of the counts of the leading published words of a window, up to and including the
// first inclusive prefix; *info = words consumed | (inclusive prefix found) << 16.
template <typename StatusT, int J>
__device__ __forceinline__ uint32_t window_reduce(const StatusT (&v)[J], uint32_t* info)
{
    using SB = StatusBits<StatusT>;
    uint32_t sum = 0, consumed = 0;
    bool open = true, found = false;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        if (open) {
            if (v[j] >> SB::kValueBits) {
                sum += static_cast<uint32_t>(v[j] & SB::kValueMask);
                ++consumed;
                if (v[j] & SB::kInclusive) { found = true; open = false; }
            } else {
                open = false;  // not published yet: everything further back waits
            }
        }
    }
    *info = consumed | (found ? 1u << 16 : 0u);
    return sum;
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

// base[index] = v.  With OFF32 the byte offset is known to fit 32 bits, which lets the
// store take a scalar base and a 32-bit register offset instead of a 64-bit address.
// Cache policy of the scatter's stores and of the tile loads in diagnostic builds
// (-DMRS_STORE_POLICY=1: non-temporal stores, 2: write-through (sc1) stores of 4-byte elements;
// -DMRS_LOAD_NT: non-temporal loads).  The product uses the default policy for both.
#ifndef MRS_STORE_POLICY
#define MRS_STORE_POLICY 0
#endif
// -DMRS_SCALAR_WRITE_OUT=1: diagnostic builds write four-byte elements back one per lane and
// instruction, as every other element type is written.
#ifndef MRS_SCALAR_WRITE_OUT
#define MRS_SCALAR_WRITE_OUT 0
#endif
template <typename T>
__device__ __forceinline__ void policy_store(T* p, const T& v)
{
#if MRS_STORE_POLICY == 1
    if constexpr (sizeof(T) <= 8) __builtin_nontemporal_store(v, p);
    else *p = v;
#elif MRS_STORE_POLICY == 2
    if constexpr (sizeof(T) == 4)
        __hip_atomic_store(reinterpret_cast<uint32_t*>(p), *reinterpret_cast<const uint32_t*>(&v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
#else
    *p = v;
#endif
}
template <typename T>
__device__ __forceinline__ T policy_load(const T* p)
{
#ifdef MRS_LOAD_NT
    if constexpr (sizeof(T) <= 8) return __builtin_nontemporal_load(p);
    else return *p;
#else
    return *p;
#endif
}

template <bool OFF32, typename T>
__device__ __forceinline__ void store_element(T* base, uint32_t index, const T& v)
{
    if constexpr (OFF32)
        policy_store(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + index * static_cast<uint32_t>(sizeof(T))), v);
    else
        policy_store(base + index, v);
}
// The same for a value that spans several elements of `elem_bytes` each: `index` counts elements,
// so the address need only have the element's alignment (gfx950 serves the 16-byte store at any
// 4-byte boundary).
template <bool OFF32, typename T>
__device__ __forceinline__ void store_element(T* base, uint32_t index, const T& v, uint32_t elem_bytes)
{
    if constexpr (OFF32)
        policy_store(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + index * elem_bytes), v);
    else
        policy_store(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + static_cast<size_t>(index) * elem_bytes), v);
}

// Tile-local positions are below 2^16 (no tile shape exceeds 65536 elements), so two
// share a register.
template <int N>
struct PackedPos {
    uint32_t w[(N + 1) / 2];
    __device__ __forceinline__ uint32_t get(int i) const { return (i & 1) ? (w[i / 2] >> 16) : (w[i / 2] & 0xFFFFu); }
    // first write of element i; the elements are written in ascending order
    __device__ __forceinline__ void set_fresh(int i, uint32_t v)
    {
        if (i & 1) {
            w[i / 2] |= v << 16;
            asm volatile("" : "+v"(w[i / 2]));  // as in set_pair
        } else {
            w[i / 2] = v;
        }
    }
    // The word is hidden from the optimiser, which would otherwise keep both halves in
    // registers of their own and undo the packing.
    __device__ __forceinline__ void set_pair(int j, uint32_t lo, uint32_t hi)
    {
        w[j] = lo | (hi << 16);
        asm volatile("" : "+v"(w[j]));
    }
    // overwrite of element i, the other half kept
    __device__ __forceinline__ void set(int i, uint32_t v)
    {
        if (i & 1) w[i / 2] = (w[i / 2] & 0xFFFFu) | (v << 16);
        else w[i / 2] = (w[i / 2] & 0xFFFF0000u) | v;
    }
};

// Digits (at most 8 bits) of N elements, four to a register.
template <int N>
struct PackedDigits {
    uint32_t w[(N + 3) / 4];
    __device__ __forceinline__ uint32_t get(int i) const { return (w[i / 4] >> (8 * (i & 3))) & 0xFFu; }
    // first write of element i; the elements are written in ascending order
    __device__ __forceinline__ void set_fresh(int i, uint32_t d)
    {
        if (i & 3) w[i / 4] |= d << (8 * (i & 3));
        else w[i / 4] = d;
        // a complete word is hidden from the optimiser, which would otherwise keep the
        // digits in registers of their own
        if ((i & 3) == 3 || i == N - 1) asm volatile("" : "+v"(w[i / 4]));
    }
};

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
// `tile_base`: index of the tile's first element in the arrays; `chain_first`: the first tile of
// the chained scan this tile belongs to (0 when the whole input is one chain; the first tile of
// the tile's segment when every segment of the input is scanned on its own, in which case
// a.digit_base is the segment's).  `tid`: threadIdx.x (a workgroup that loops over tiles passes a
// copy whose origin the compiler cannot see, so that nothing derived from it is kept in registers
// from one tile to the next).
template <int MODE, int RANK, int MAP, bool FULL, typename StatusT, int BITS, int THREADS, int KPT, typename KeyT,
          typename ValT>
__device__ __forceinline__ void process_tile(const PassArgs& a, PassShared<BITS, THREADS, KPT, KeyT, ValT>& sh,
                                             const uint32_t tile, const uint32_t tile_valid,
                                             const unsigned long long tile_base, const uint32_t chain_first,
                                             const int tid)
{
    using S = PassShared<BITS, THREADS, KPT, KeyT, ValT>;
    using SB = StatusBits<StatusT>;
    constexpr int RADIX = S::RADIX;
    constexpr int WAVES = S::WAVES;
    constexpr int TILE = S::TILE;
    constexpr bool HAS_VAL = S::HAS_VAL;
    constexpr int SCAN_WAVES = S::SCAN_WAVES;
    static_assert(TILE <= 65536 && BITS <= 8, "positions are packed in 16 bits, digits in 8");
    // The host picks 32-bit status words only when every key and payload array is at most
    // 4 GiB, so byte offsets into them fit 32 bits there.
    constexpr bool OFF32 = MODE == kModeOnesweep && sizeof(StatusT) == 4;

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t my_first = wave * 64 * KPT + lane;  // element of item 0 inside the tile

    // ---- load keys -------------------------------------------------------------------
    const KeyT* __restrict__ keys_in = static_cast<const KeyT*>(a.keys_in) + tile_base + my_first;
    KeyT key[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if (FULL || my_first + i * 64 < tile_valid) key[i] = policy_load(keys_in + i * 64);
        else key[i] = static_cast<KeyT>(~static_cast<KeyT>(0));
    }
#ifdef MRS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MRS_STAMP(2);
#endif

    // ---- rank inside the wave ------------------------------------------------------
    // pos[i] = number of earlier elements of this wave with the same digit.
    // wave_hist[wave][d] carries the running count.  LDS operations of one wave execute
    // in order, so item i+1 sees the update of item i.
    PackedPos<KPT> pos;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if (kAblate & 2) { pos.set_fresh(i, 0); continue; }
        const uint32_t d = key_digit<MAP>(key[i], a);
        const bool mine = FULL || (my_first + i * 64 < tile_valid);
        if constexpr (RANK == kRankAtomic) {
            uint32_t r = 0;
            // 64 lanes on one counter would be served one after the other (constant digits:
            // equal keys, zero high bytes, a one-part partition); that case needs no atomic
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
            if (FULL && __builtin_amdgcn_uicmp(d, d0, 32 /* == */) == ~0ull) {
                r = sh.wave_hist[wave][d0] + lane;
                __builtin_amdgcn_wave_barrier();
                if (lane == 63) sh.wave_hist[wave][d0] = r + 1;
            } else if (mine) {
                r = atomicAdd(&sh.wave_hist[wave][d], 1u);
            }
            pos.set_fresh(i, r);
        } else {
            // the lowest lane of each digit group advances the counter
            const unsigned long long valid = FULL ? ~0ull : __ballot(mine);
            uint32_t mlo, mhi;
            match_digit<BITS>(d, valid, &mlo, &mhi);
            const uint32_t below = lanes_below(mlo, mhi);
            const uint32_t prev = sh.wave_hist[wave][d];
            if (mine && below == 0) sh.wave_hist[wave][d] = prev + __popc(mlo) + __popc(mhi);
            pos.set_fresh(i, prev + below);
            __builtin_amdgcn_wave_barrier();
        }
    }
    // The digits and counter addresses of the ranking are cheap to recompute and costly to
    // keep: hide the keys' provenance so that the staging below does not keep them alive.
#pragma unroll
    for (int i = 0; i < KPT; ++i) launder(key[i]);
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
            status_store<StatusT>(mine, (tile == chain_first ? SB::kInclusive : SB::kAggregate) | tile_count);
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

    // ---- look-back, first round trip ------------------------------------------------------
    // Thread d fetches the status words of digit d of the J tiles before this one.  These
    // waves resolve the look-back and publish the inclusive prefix BEFORE they stage their
    // keys: every microsecond between the arrival of the status words and the publication
    // adds tiles to the walk of every successor.  The other waves stage their keys
    // meanwhile, so the LDS stays busy, and no wave has asked for the payload yet (a status
    // read that queues behind the payload loads of the workgroup comes back microseconds
    // later).  A dependent device-scope read costs microseconds on a busy chip and as many
    // tiles enter look-back meanwhile, hence a window of J predecessors per round trip.
#ifdef MRS_LB_WIDTH
    constexpr int J = MRS_LB_WIDTH;
#else
    // tiles of 32 keys per thread leave fewer registers for the window
    constexpr int J = KPT >= 32 ? SB::kWidth * 3 / 2 : SB::kWidth * 2;
#endif
    const bool in_walk = MODE == kModeOnesweep && tid < RADIX;
    const uint32_t lb_digit = tid;
    const StatusT* const status = static_cast<const StatusT*>(a.status);
    const bool walker = in_walk && !(kAblate & (1 | 16)) && tile != chain_first;
    // of the walked digit: first output position in the whole array, count and offset in this tile
    uint32_t digit_base = 0;
    const uint32_t walk_count = tile_count, walk_off = tile_off;
    if (in_walk) digit_base = a.digit_base[lb_digit];
    StatusT lbv[J];
    if (walker) window_issue<RADIX, StatusT, J>(status, static_cast<int>(tile) - 1, lb_digit, lbv, static_cast<int>(chain_first));

    // ---- global base of every digit of this tile ------------------------------------
    if (MODE == kModeScatter) {
        if (tid < RADIX) sh.gbase[tid] = a.scanned[static_cast<size_t>(tid) * a.num_tiles + tile] - tile_off;
    } else {
        // Every wave resolves its own digits without waiting for the other waves: the sooner
        // an inclusive prefix is published, the shorter the walks of the tiles behind.
        uint32_t excl = 0;
        bool done = !walker;
        int next = static_cast<int>(tile) - 1;  // nearest predecessor not yet accounted for
        unsigned long long t0 = 0;
        uint32_t spins = 0;
        // The window held in lbv is reduced; then publish, or move the walk further back.
        auto resolve_round = [&]() {
            if (done) return;
            uint32_t info = 0;
            excl += window_reduce<StatusT, J>(lbv, &info);
            const uint32_t consumed = info & 0xFFFFu;
            const bool found = (info >> 16) != 0u;
#ifdef MRS_STAMPS
            if (a.stamps && tid == 0) {
                a.stamps[static_cast<size_t>(tile) * 16 + 8] += 1;               // round trips
                a.stamps[static_cast<size_t>(tile) * 16 + 9] += consumed;        // tiles walked
                a.stamps[static_cast<size_t>(tile) * 16 + 10] += consumed == 0;  // rounds that found nothing
            }
#endif
            if (found) {
                done = true;
                status_store<StatusT>(static_cast<StatusT*>(a.status) + static_cast<size_t>(tile) * RADIX + lb_digit,
                                      SB::kInclusive | static_cast<StatusT>(excl + walk_count));
                sh.gbase[lb_digit] = digit_base + excl - walk_off;
                return;
            }
            next -= static_cast<int>(consumed);
            if (consumed == 0) {
                ++spins;
                if ((spins & 63u) == 0u) {
                    const unsigned long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    if (now - t0 > kSpinTimeoutTicks ||
                        __hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        atomicExch(a.error_flag, 1u);
                        sh.abort = 1;
                        done = true;  // gives up; nothing is written through the unresolved offset
                    }
                }
            }
        };
        resolve_round();
        MRS_STAMP(11);
        while (__ballot(!done) != 0ull) {
            if (!done) window_issue<RADIX, StatusT, J>(status, next, lb_digit, lbv, static_cast<int>(chain_first));
            resolve_round();
        }
        // the first tile of a chain has nothing before it
        if (in_walk && !walker) sh.gbase[lb_digit] = digit_base - walk_off + ((kAblate & 16) ? tile * walk_count : 0u);
        MRS_STAMP(12);
    }
    // ---- local sort of the keys through LDS ---------------------------------------
    KeyT* stage_keys = reinterpret_cast<KeyT*>(sh.stage);
#pragma unroll
    for (int i = 0; i < KPT; i += 2) {
        uint32_t p[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2 && i + h < KPT; ++h) {
            const uint32_t d = key_digit<MAP>(key[i + h], a);
            p[h] = pos.get(i + h) + sh.wave_hist[wave][d];
            if (kAblate & 2) p[h] = my_first + (i + h) * 64;
            if (FULL || my_first + (i + h) * 64 < tile_valid) stage_keys[p[h]] = key[i + h];
        }
        pos.set_pair(i / 2, p[0], p[1]);
        if ((i & 7) == 6) __builtin_amdgcn_sched_barrier(0);  // bounds the registers in flight
    }
    // The payload is fetched now: its latency hides behind the key write-out, and its
    // registers take over from the keys, which live in LDS from here on.
    ValT val[KPT];
    if constexpr (HAS_VAL) {
        if (!(kAblate & 4)) {
            bool fetched = false;
            if constexpr (std::is_same<ValT, uint32_t>::value) {
                // a sort of (key, source index) pairs: the first pass makes the indices up
                if (a.index_payload) {  // uniform
                    const uint32_t first_index = static_cast<uint32_t>(tile_base) + my_first;
#pragma unroll
                    for (int i = 0; i < KPT; ++i) val[i] = first_index + i * 64;
                    fetched = true;
                }
            }
            if (!fetched) {
                const ValT* __restrict__ vals_in = static_cast<const ValT*>(a.vals_in) + tile_base + my_first;
#pragma unroll
                for (int i = 0; i < KPT; ++i)
                    if (FULL || my_first + i * 64 < tile_valid) val[i] = policy_load(vals_in + i * 64);
            }
        }
    }

    MRS_STAMP(5);
    __syncthreads();
    if (MODE == kModeOnesweep) {
        if (sh.abort) return;  // never write through an offset that was not resolved
    }
    MRS_STAMP(6);

    // ---- write the keys back in slot order ---------------------------------------------
    KeyT* __restrict__ keys_out = static_cast<KeyT*>(a.keys_out);
    // The payload in the same slot goes to the same place; the digit of the slot (a
    // quarter of a register) is what is remembered of it.
    PackedDigits<KPT> slot_digit;
    // Four-byte elements leave in groups of four consecutive slots per lane: one 16-byte LDS read,
    // one look-up of the run's base and one 16-byte store (a wave writes 1 KiB per instruction,
    // so fewer lines are touched twice) wherever the four slots lie in one digit's run, which the
    // first and the last digit tell because slots are sorted by digit.  The few lanes whose group
    // straddles a run boundary (about two per wave) store element by element.
    constexpr bool VEC4 = sizeof(KeyT) == 4 && KPT % 4 == 0 && !(kAblate & ~(1 | 2 | 4 | 8 | 32)) && !MRS_SCALAR_WRITE_OUT;
    if constexpr (VEC4) {
#pragma unroll
        for (int j = 0; j < KPT / 4; ++j) {
            const uint32_t s0 = (j * THREADS + tid) * 4u;
            uint32_t d[4] = {0, 0, 0, 0};
            if (FULL || s0 + 3 < tile_valid) {
                const uint4 kk = *reinterpret_cast<const uint4*>(stage_keys + s0);
                const KeyT k[4] = {kk.x, kk.y, kk.z, kk.w};
                d[0] = key_digit<MAP>(k[0], a);
                d[3] = key_digit<MAP>(k[3], a);
                if (d[0] == d[3]) {
                    d[1] = d[2] = d[0];
                    uint32_t g = sh.gbase[d[0]] + s0;
                    if (kAblate & (3 | 32)) g = static_cast<uint32_t>(tile_base) + s0;
                    if (kAblate & 8) { keep_alive(kk); keep_alive(g); }
                    else store_element<OFF32>(reinterpret_cast<uint4*>(keys_out), g, kk, 4u);
                } else {
                    d[1] = key_digit<MAP>(k[1], a);
                    d[2] = key_digit<MAP>(k[2], a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t g = sh.gbase[d[e]] + s0 + e;
                        if (kAblate & (3 | 32)) g = static_cast<uint32_t>(tile_base) + s0 + e;
                        if (kAblate & 8) { keep_alive(k[e]); keep_alive(g); }
                        else store_element<OFF32>(keys_out, g, k[e]);
                    }
                }
            } else if (!FULL) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (s0 + e < tile_valid) {
                        const KeyT k = stage_keys[s0 + e];
                        d[e] = key_digit<MAP>(k, a);
                        store_element<OFF32>(keys_out, sh.gbase[d[e]] + s0 + e, k);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) slot_digit.set_fresh(4 * j + e, d[e]);
            if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // bounds the registers in flight
        }
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t slot = i * THREADS + tid;
            uint32_t d = 0;
            if (FULL || slot < tile_valid) {
                const KeyT k = stage_keys[slot];
                d = key_digit<MAP>(k, a);
                uint32_t g = sh.gbase[d] + slot;
                if (kAblate & (3 | 32)) g = static_cast<uint32_t>(tile_base) + slot + ((kAblate & 256) ? g * (a.num_tiles >> 31) : 0u);
                if (kAblate & 16) g = g < a.n ? g : slot;  // the estimate may overshoot the last run
                if (kAblate & 64) g = g + 1 < a.n ? g + 1 : 0;
                if (kAblate & 128) {
                    const uint32_t chunks = a.n >> 6;  // n is a multiple of 64 in the timing runs
                    g = ((g >> 6) * 2654435761u % chunks) * 64u + (g & 63u);
                }
                if (kAblate & 8) { keep_alive(k); keep_alive(g); }
                else store_element<OFF32>(keys_out, g, k);
            }
            slot_digit.set_fresh(i, d);
            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // bounds the registers in flight
        }
    }

    // ---- payload: same permutation ------------------------------------------------------
    if constexpr (HAS_VAL && !(kAblate & 4)) {
        ValT* stage_vals = reinterpret_cast<ValT*>(sh.stage);
        ValT* __restrict__ vals_out = static_cast<ValT*>(a.vals_out);
        MRS_STAMP(13);
        __syncthreads();  // the key image is dead
#pragma unroll
        for (int i = 0; i < KPT; ++i)
            if (FULL || my_first + i * 64 < tile_valid) stage_vals[pos.get(i)] = val[i];
        __syncthreads();
        MRS_STAMP(14);
        if constexpr (VEC4 && sizeof(ValT) == 4) {
#pragma unroll
            for (int j = 0; j < KPT / 4; ++j) {
                const uint32_t s0 = (j * THREADS + tid) * 4u;
                if (FULL || s0 + 3 < tile_valid) {
                    const uint4 vv = *reinterpret_cast<const uint4*>(stage_vals + s0);
                    const uint32_t d0 = slot_digit.get(4 * j), d3 = slot_digit.get(4 * j + 3);
                    if (d0 == d3) {
                        uint32_t g = sh.gbase[d0] + s0;
                        if (kAblate & (3 | 32)) g = static_cast<uint32_t>(tile_base) + s0;
                        if (kAblate & 8) keep_alive(vv);
                        else store_element<OFF32>(reinterpret_cast<uint4*>(vals_out), g, vv, 4u);
                    } else {
                        const uint32_t v[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            uint32_t g = sh.gbase[slot_digit.get(4 * j + e)] + s0 + e;
                            if (kAblate & (3 | 32)) g = static_cast<uint32_t>(tile_base) + s0 + e;
                            if (kAblate & 8) keep_alive(v[e]);
                            else store_element<OFF32>(reinterpret_cast<uint32_t*>(vals_out), g, v[e]);
                        }
                    }
                } else if (!FULL) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (s0 + e < tile_valid)
                            store_element<OFF32>(vals_out, sh.gbase[slot_digit.get(4 * j + e)] + s0 + e, stage_vals[s0 + e]);
                }
                if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                // the slot numbering of the key loop above
                const uint32_t slot = VEC4 ? ((i / 4) * THREADS + tid) * 4u + (i & 3) : i * THREADS + tid;
                if (FULL || slot < tile_valid) {
                    const ValT v = stage_vals[slot];
                    uint32_t g = sh.gbase[slot_digit.get(i)] + slot;
                    if (kAblate & (3 | 32)) g = static_cast<uint32_t>(tile_base) + slot + ((kAblate & 256) ? g * (a.num_tiles >> 31) : 0u);
                    if (kAblate & 16) g = g < a.n ? g : slot;
                    if (kAblate & 64) g = g + 1 < a.n ? g + 1 : 0;
                    if (kAblate & 128) {
                        const uint32_t chunks = a.n >> 6;
                        g = ((g >> 6) * 2654435761u % chunks) * 64u + (g & 63u);
                    }
                    if (kAblate & 8) keep_alive(v);
                    else store_element<OFF32>(vals_out, g, v);
                }
                if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#ifdef MRS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MRS_STAMP(7);
#endif
}

// Tile id of a workgroup.  With xcd_block = B > 0 the tiles are dealt in blocks of B
// consecutive tiles, block g to die g % 8: a workgroup draws from the list of the die it runs
// on (hardware register XCC_ID), and from the other lists once its own is exhausted.  Every
// list is drawn in ascending order and there are at least as many workgroups as tiles, so the
// lowest unfinished tile is either held by a running workgroup or the next to be drawn ON ITS
// DIE: the look-back makes progress as long as this kernel keeps getting workgroup slots on
// every die, which holds when it has the device to itself or shares it with kernels that end on
// their own (a single-list pass, a collective).  Two die-aware passes at once could each hold the
// slots the other waits for; the host never launches that combination (begin_ticketed_call in
// mrs_api.hip) and a wait that does starve ends in the time-out of the look-back, not in a hang.
// With B = 0 there is one list: drawn tiles are always a prefix, the lowest unfinished tile is
// always running, and one free slot anywhere is enough.
//
// The draw in two steps, for callers that have something to wait for in between: ticket_issue
// sends the atomic of the first list (the die's own, or the only one), ticket_tile turns what it
// returned into a tile and goes on to the other dies' lists if that list is exhausted.
// `num_tiles`: tiles of this pass (a.num_tiles, or a count only the device knows).
struct Ticket {
    uint32_t k;     // what the first list's counter held
    uint32_t list;  // which list that was
};
__device__ __forceinline__ Ticket ticket_issue(const PassArgs& a)
{
    Ticket t;
    t.list = 0;
    if (a.xcd_block != 0) {
        // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
        t.list = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & (kTicketLists - 1);
    }
    t.k = atomicAdd(a.tile_counter + t.list, 1u);
    return t;
}
__device__ __forceinline__ uint32_t ticket_tile(const PassArgs& a, const Ticket first, const uint32_t num_tiles)
{
    const uint32_t B = a.xcd_block;
    if (B == 0) return first.k;
    uint32_t k = first.k, list = first.list;
    for (uint32_t i = 0;;) {
        const unsigned long long tile = (static_cast<unsigned long long>(k / B) * kTicketLists + list) * B + k % B;
        if (tile < num_tiles) return static_cast<uint32_t>(tile);
        if (++i == kTicketLists) break;
        list = (first.list + i) & (kTicketLists - 1);
        k = atomicAdd(a.tile_counter + list, 1u);
    }
    return num_tiles;  // only with more workgroups than tiles
}
__device__ __forceinline__ uint32_t draw_tile(const PassArgs& a) { return ticket_tile(a, ticket_issue(a), a.num_tiles); }

// The scatter of the three-kernel pass has no chain, so any one-to-one map from workgroup to
// tile will do.  Workgroups are dealt to the 8 dies round robin by index; this map gives the 8
// workgroups that land on one die out of every 64 a block of 8 consecutive tiles, for the same
// reason as the ticket lists above.  The last incomplete group of 64 keeps the identity.
__device__ __forceinline__ uint32_t die_local_tile(uint32_t wg, uint32_t num_tiles)
{
    if (wg >= (num_tiles & ~63u)) return wg;
    const uint32_t r = wg & 63u;
    return (wg & ~63u) + (r & 7u) * 8u + (r >> 3);
}

// Waves per SIMD the register allocation must leave room for.  A 512 x 32 tile holds
// 64 KiB in LDS, so two workgroups fit a compute unit only if each wave stays within
// 128 registers; the other shapes are left to the compiler.
constexpr int pass_min_waves_per_simd(int threads, int kpt) { return threads == 512 && kpt == 32 ? 4 : 1; }

template <int MODE, int RANK, int MAP, typename StatusT, int BITS, int THREADS, int KPT, typename KeyT, typename ValT>
__device__ __forceinline__ void run_digit_pass(const PassArgs& a, PassShared<BITS, THREADS, KPT, KeyT, ValT>& sh)
{
    using S = PassShared<BITS, THREADS, KPT, KeyT, ValT>;
    const int tid = threadIdx.x;
#ifdef MRS_STAMPS
    const unsigned long long t_begin = wall_clock64();
#endif
    // ---- tile id ---------------------------------------------------------------------
    // ONESWEEP takes a ticket so that every tile a workgroup can wait for is owned by
    // a workgroup that is already running.
    if (MODE == kModeOnesweep) {
        if (tid == 0) {
            sh.tile = draw_tile(a);
            sh.abort = 0;
        }
    }
    for (int j = tid; j < S::WAVES * S::RADIX; j += THREADS) (&sh.wave_hist[0][0])[j] = 0;
    __syncthreads();
    const uint32_t tile = MODE == kModeOnesweep ? sh.tile : die_local_tile(blockIdx.x, a.num_tiles);
    if (tile >= a.num_tiles) return;  // uniform; cannot happen with one workgroup per tile
#ifdef MRS_STAMPS
    if (a.stamps && tid == 0) a.stamps[static_cast<size_t>(tile) * 16 + 0] = t_begin;
#endif
    MRS_STAMP(1);

    // tile < num_tiles, so tile * TILE < n < 2^32.  Kept in 32 bits and without a select on
    // purpose: ROCm 7.2's compiler turned `(u64 difference < TILE) ? difference : TILE` into a
    // vector compare followed by s_cselect_b32 on a stale SCC (tools/check_scc.py finds these).
    const uint32_t remaining = a.n - tile * static_cast<uint32_t>(S::TILE);
    if (remaining >= static_cast<uint32_t>(S::TILE))
        process_tile<MODE, RANK, MAP, true, StatusT>(a, sh, tile, S::TILE, static_cast<unsigned long long>(tile) * S::TILE, 0u, tid);
    else
        process_tile<MODE, RANK, MAP, false, StatusT>(a, sh, tile, remaining, static_cast<unsigned long long>(tile) * S::TILE, 0u, tid);
}

template <int MODE, int RANK, int MAP, typename StatusT, int BITS, int THREADS, int KPT, typename KeyT, typename ValT>
__global__ __launch_bounds__(THREADS, pass_min_waves_per_simd(THREADS, KPT)) void digit_pass_kernel(const PassArgs a)
{
    using S = PassShared<BITS, THREADS, KPT, KeyT, ValT>;
    static_assert(THREADS % 64 == 0 && THREADS >= S::RADIX, "one thread per digit is required");
    __shared__ S sh;
    if constexpr (MODE == kModeOnesweep && MAP == kMapNone && sizeof(KeyT) == 8) {
        // Sorts of 64-bit keys: the device decided after the histograms which passes run
        // and on which side their input lies (PassArgs::pass_plan).  The word is uniform.
        const uint32_t plan = a.pass_plan ? *a.pass_plan : 0u;
        if (plan & 1u) return;  // nothing to move; no ticket is drawn, no status word written
        const bool other_side = (plan & 2u) != 0u;
        PassArgs b = a;
        b.keys_in = other_side ? a.keys_out : a.keys_in;
        b.vals_in = other_side ? a.vals_out : a.vals_in;
        b.keys_out = other_side ? const_cast<void*>(a.keys_in) : a.keys_out;
        b.vals_out = other_side ? const_cast<void*>(a.vals_in) : a.vals_out;
        run_digit_pass<MODE, RANK, MAP, StatusT>(b, sh);
    } else if constexpr (MAP != kMapNone) {
        // The digit table (at most kMaxMapEntries bytes) is read three times per key: keep it in
        // LDS.  The first barrier of run_digit_pass orders these stores before any lookup.
        __shared__ uint8_t s_map[kMaxMapEntries];
        __shared__ KeyT s_split[kMaxSplitters + 1];
        for (uint32_t j = threadIdx.x; j <= a.mask && j < kMaxMapEntries; j += THREADS) s_map[j] = a.digit_map[j];
        if constexpr (MAP == kMapTableAndSplitters) {
            if (threadIdx.x < a.num_splitters && threadIdx.x < kMaxSplitters)
                s_split[threadIdx.x] = static_cast<const KeyT*>(a.splitters)[threadIdx.x];
        }
        PassArgs b = a;
        b.digit_map = s_map;
        b.splitters = s_split;
        run_digit_pass<MODE, RANK, MAP, StatusT>(b, sh);
    } else {
        run_digit_pass<MODE, RANK, MAP, StatusT>(a, sh);
    }
}

template <int MODE, int RANK, int MAP, int BITS, int THREADS, int KPT, typename KeyT, typename ValT>
int launch_pass_cfg(const PassArgs& a, hipStream_t s)
{
    // 32-bit status words hold counts below 2^30; the scatter mode has no status words
    auto wide = digit_pass_kernel<MODE, RANK, MAP, unsigned long long, BITS, THREADS, KPT, KeyT, ValT>;
    auto narrow = digit_pass_kernel<MODE, RANK, MAP, uint32_t, BITS, THREADS, KPT, KeyT, ValT>;
    const bool use_wide = MODE == kModeOnesweep && a.wide_status;
    if (use_wide)
        hipLaunchKernelGGL(wide, dim3(a.num_tiles), dim3(THREADS), 0, s, a);
    else
        hipLaunchKernelGGL(narrow, dim3(a.num_tiles), dim3(THREADS), 0, s, a);
    MRS_CHECK_LAUNCH("digit_pass_kernel");
    return MRS_OK;
}

}  // namespace mrs

#endif
