// Kernels of the most-significant-digit-first schedule (see mrs_hybrid.hpp) for gfx950.
#include "mrs_hybrid.hpp"

#include "mrs_radix_pass.hpp"

namespace mrs {

// Inclusive prefix sum over the 64 lanes of a wave whose lanes are all active: inside rows of 16
// lanes by DPP row shifts (a lane without a source lane adds 0), then the totals of the rows
// before it (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  Six dependent adds,
// against six round trips through the LDS crossbar for the same scan by __shfl_up.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x)
{
    x += __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xF, 0xF, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xF, 0xF, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xF, 0xF, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0u, x, 0x118, 0xF, 0xF, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0u, x, 0x142, 0xA, 0xF, false);  // row_bcast:15
    x += __builtin_amdgcn_update_dpp(0u, x, 0x143, 0xC, 0xF, false);  // row_bcast:31
    return x;
}

// ---- exact 16-bit histogram -----------------------------------------------------------------
// One LDS atomic per key into 65536 counters of 16 bits, two to a word (128 KiB: one workgroup
// of 1024 threads per compute unit), 16-byte nontemporal loads, eight in flight per thread (four: 6 % slower).  A
// workgroup writes its table to `packed`; hybrid_reduce_kernel adds the tables up.
//
// The counts are exact whatever the keys are.  A field never wraps: the one add that takes a
// field from below 32768 to 32768 or more (exactly one add does, fields only grow) takes 32768
// out again and puts it on the sub-bucket's 32-bit spill counter in global memory; until it has
// done so the field can grow by what the workgroup has in flight, far below another 32768.  A
// wave that finds a field already holding 512 of its workgroup's keys (evenly spread keys leave
// about 16 at 2^28) counts differently from then on: the lanes of one instruction that hold the
// first lane's sub-bucket are counted by ONE add of their number (equal or clustered keys would
// otherwise queue 64 lanes on one word: 0.9 ms instead of 0.17 at 2^28 equal keys).  As a last
// check a workgroup whose fields and spills do not add up to the keys it read raises
// hdr->overflow, and the decision kernel then chooses the other schedule.
//
// The same read of the keys also counts the two lowest digits of the plan's ordinary passes (bits
// [begin_bit, begin_bit + 8) and the 8 above), into 8 interleaved copies of 256 counters each
// (copy = lane % 8: the lanes of one instruction spread over the LDS banks whatever their digits
// are), added to `low_hist` at the end.  Together with the exact table (which holds the digits
// above those 16 bits: the local sort takes at most 16) that is every histogram the ordinary
// passes need, so a call the device refuses does not read the keys a second time.
constexpr int kLowCopies = 8;
__global__ __launch_bounds__(1024) void hybrid_histogram_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                 uint32_t vectors, uint32_t shift, uint32_t bits,
                                                                 uint32_t* __restrict__ packed, HybridCleared* hc,
                                                                 uint32_t window_base, uint32_t windowed, uint32_t end_bit,
                                                                 uint32_t begin_bit, uint32_t* __restrict__ low_hist)
{
#ifndef MRS_HYB_HIST_UNROLL
#define MRS_HYB_HIST_UNROLL 8
#endif
    constexpr int THREADS = 1024, UNROLL = MRS_HYB_HIST_UNROLL;
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[kHybPackedWords];
    __shared__ uint32_t s_low[2][256 * kLowCopies];
    __shared__ int s_diff;
    __shared__ uint32_t s_flags;  // bit 0: a key outside the window, bit 1: bits at or above end_bit differ
    const uint32_t tid = threadIdx.x;
    // counters beyond 2^bits are never touched: they are neither cleared, nor written out, nor read by
    // the kernels that follow
    const uint32_t words4 = (1u << bits) / 8u;  // 16-byte groups of packed words in use (bits >= 14)
    {
        uint4* z = reinterpret_cast<uint4*>(s_cnt);
        for (uint32_t j = tid; j < words4; j += THREADS) z[j] = make_uint4(0, 0, 0, 0);
    }
    for (uint32_t j = tid; j < 2 * 256 * kLowCopies; j += THREADS) (&s_low[0][0])[j] = 0;
    if (tid == 0) {
        s_diff = 0;
        s_flags = 0;
    }
    __syncthreads();
    const uint32_t low_copy = tid & (kLowCopies - 1);
    // the two lowest digits of the ordinary passes; no result is needed, so no returning atomic
    auto count_low = [&](uint32_t k, uint32_t amount) __attribute__((always_inline)) {
#ifdef MRS_HYB_NO_LOW  // timing builds only: the ordinary passes of a refused call get no histograms
        return;
#endif
        atomicAdd(&s_low[0][__builtin_amdgcn_ubfe(k, begin_bit, 8) * kLowCopies + low_copy], amount);
        atomicAdd(&s_low[1][__builtin_amdgcn_ubfe(k, begin_bit + 8u, 8) * kLowCopies + low_copy], amount);
    };
    uint32_t mine = 0;     // keys this lane read
    uint32_t spilled = 0;  // times this lane moved 32768 counts out of a field
    uint32_t differ = 0;   // bits in which a key of this lane differs from the first key of the array
    uint32_t outside = 0;  // a key of this lane lies outside the promised window
    const uint32_t first_key = keys[0];
    auto add_field = [&](uint32_t b, uint32_t amount) __attribute__((always_inline)) {
        const uint32_t half = (b & 1u) * 16u;
        const uint32_t old = atomicAdd(&s_cnt[b >> 1], amount << half);
        const uint32_t before = (old >> half) & 0xFFFFu;
        if (before < 0x8000u && before + amount >= 0x8000u) {
            atomicSub(&s_cnt[b >> 1], 0x8000u << half);
            atomicAdd(&hc->spill[b], 0x8000u);
            ++spilled;
        }
    };
    uint32_t most = 0;  // largest count this lane has found in a field
    auto sub_bucket = [&](uint32_t k) __attribute__((always_inline)) {
        differ |= k ^ first_key;
        uint32_t b = __builtin_amdgcn_ubfe(k, shift, bits);  // bits <= 16: counters beyond 2^bits stay zero
        if (windowed) {  // uniform
            // sub-bucket = (key >> shift) - window_base; a key outside the window is counted somewhere
            // and reported (the other schedule sorts any keys)
            b = (k >> shift) - window_base;
            if (b >> bits) {
                outside = 1;
                b = 0;
            }
        }
        return b;
    };
    // evenly spread keys: one add per key and nothing else
    auto count_plain = [&](uint32_t k) __attribute__((always_inline)) {
        count_low(k, 1u);
        const uint32_t b = sub_bucket(k);
        const uint32_t half = (b & 1u) * 16u;
        const uint32_t old = atomicAdd(&s_cnt[b >> 1], 1u << half);
        const uint32_t before = (old >> half) & 0xFFFFu;
        most = before > most ? before : most;
        if (before == 0x7FFFu) {
            atomicSub(&s_cnt[b >> 1], 0x8000u << half);
            atomicAdd(&hc->spill[b], 0x8000u);
            ++spilled;
        }
    };
    // keys that crowd on few sub-buckets: the lanes that hold the first lane's sub-bucket share one add
    auto count_grouped = [&](uint32_t k) __attribute__((always_inline)) {
        const uint32_t b = sub_bucket(k);
        const uint32_t b0 = __builtin_amdgcn_readfirstlane(b);
        const unsigned long long same = __builtin_amdgcn_uicmp(b, b0, 32 /* == */);  // among the active lanes
        const uint32_t group = static_cast<uint32_t>(__popcll(same));
        if (group < 8u) {  // uniform: this instruction's keys are spread after all
            count_low(k, 1u);
            add_field(b, 1u);
            return;
        }
        {
            // equal keys share their low digits too: one lane counts for the lanes that hold the
            // first lane's 16 low bits
            const uint32_t low = __builtin_amdgcn_ubfe(k, begin_bit, 16);
            const uint32_t low0 = __builtin_amdgcn_readfirstlane(low);
            const unsigned long long same_low = __builtin_amdgcn_uicmp(low, low0, 32 /* == */);
            const uint32_t group_low = static_cast<uint32_t>(__popcll(same_low));
            if (group_low >= 8u && low == low0) {
                if (lanes_below(static_cast<uint32_t>(same_low), static_cast<uint32_t>(same_low >> 32)) == 0u) count_low(k, group_low);
            } else {
                count_low(k, 1u);
            }
        }
        if (b == b0) {
            if (lanes_below(static_cast<uint32_t>(same), static_cast<uint32_t>(same >> 32)) == 0u) add_field(b0, group);
        } else {
            add_field(b, 1u);
        }
    };
    // A wave counts the plain way until one of its lanes finds a field that already holds 512 keys
    // of this workgroup (evenly spread keys leave a field about n / (65536 * workgroups) of them:
    // 16 at 2^28), and the grouped way from then on.
    bool crowded = false;  // uniform
    auto count = [&](uint32_t k) __attribute__((always_inline)) {
        if (crowded) count_grouped(k);
        else count_plain(k);
    };
    auto look = [&]() __attribute__((always_inline)) {
        if (!crowded && __builtin_amdgcn_uicmp(most, 512u, 35 /* >= */) != 0ull) crowded = true;
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* __restrict__ vkeys = reinterpret_cast<const u32x4*>(keys);
    const uint32_t stride = gridDim.x * THREADS;
    uint32_t v = blockIdx.x * THREADS + tid;
    // v + 3 * stride < 2^28 + 3 * 2^18: no wrap for any n < 2^32
    for (; v + (UNROLL - 1) * stride < vectors; v += UNROLL * stride) {
        u32x4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = __builtin_nontemporal_load(vkeys + v + u * stride);
        if (crowded) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                count_grouped(x[u].x); count_grouped(x[u].y); count_grouped(x[u].z); count_grouped(x[u].w);
            }
        } else {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                count_plain(x[u].x); count_plain(x[u].y); count_plain(x[u].z); count_plain(x[u].w);
            }
            look();
        }
        mine += 4 * UNROLL;
    }
    for (; v < vectors; v += stride) {
        const u32x4 x = __builtin_nontemporal_load(vkeys + v);
        count(x.x); count(x.y); count(x.z); count(x.w);
        look();
        mine += 4;
    }
    // what the vectors do not cover: everything if the array is not 16-byte aligned, else < 4 keys
    for (unsigned long long i = static_cast<unsigned long long>(vectors) * 4 + blockIdx.x * THREADS + tid; i < n; i += stride) {
        count(keys[i]);
        mine += 1;
    }
    __syncthreads();
    uint32_t fields = 0;
    {
        const uint4* c = reinterpret_cast<const uint4*>(s_cnt);
        uint4* out = reinterpret_cast<uint4*>(packed + static_cast<size_t>(blockIdx.x) * kHybPackedWords);
        for (uint32_t j = tid; j < words4; j += THREADS) {
            const uint4 w = c[j];
            out[j] = w;
            fields += (w.x & 0xFFFFu) + (w.x >> 16) + (w.y & 0xFFFFu) + (w.y >> 16) + (w.z & 0xFFFFu) + (w.z >> 16) +
                      (w.w & 0xFFFFu) + (w.w >> 16);
        }
    }
    // the low digits: the copies added up, one global add per counter that is not zero
    for (uint32_t j = tid; j < 2 * 256; j += THREADS) {
        uint32_t sum = 0;
#pragma unroll
        for (int cc = 0; cc < kLowCopies; ++cc) sum += (&s_low[0][0])[j * kLowCopies + cc];
        if (sum) atomicAdd(&low_hist[j], sum);
    }
    int diff = static_cast<int>(mine) - static_cast<int>(fields) - static_cast<int>(spilled * 0x8000u);
    uint32_t flags = outside | ((end_bit < 32u && (differ >> end_bit) != 0u) ? 2u : 0u);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        diff += __shfl_down(diff, o);
        flags |= __shfl_down(flags, o);
    }
    if ((tid & 63u) == 0) {
        if (diff != 0) atomicAdd(&s_diff, diff);
        if (flags != 0) atomicOr(&s_flags, flags);
    }
    __syncthreads();
    if (tid == 0) {
        if (s_diff != 0 || (s_flags & 1u)) atomicOr(&hc->hdr.overflow, 1u);
        if (s_flags & 2u) atomicOr(&hc->hdr.high_vary, 1u);
    }
}

// counts[i] = spill[i] + sum over the workgroups' tables of field i.  A workgroup owns 64
// consecutive packed words; its four waves each add up a quarter of the tables (256 bytes per
// table and wave), and the quarters meet in LDS.
__global__ __launch_bounds__(256) void hybrid_reduce_kernel(const uint32_t* __restrict__ packed, int grid,
                                                             const uint32_t* __restrict__ spill,
                                                             uint32_t* __restrict__ counts)
{
    __shared__ uint32_t s_part[4][64][2];
    const uint32_t lane = threadIdx.x & 63u, part = threadIdx.x >> 6;
    const uint32_t j = blockIdx.x * 64u + lane;  // < kHybPackedWords: the grid covers the entries in use, exactly
    uint32_t lo = 0, hi = 0;
    int g = static_cast<int>(part);
    for (; g + 12 < grid; g += 16) {
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = packed[static_cast<size_t>(g + 4 * u) * kHybPackedWords + j];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lo += w[u] & 0xFFFFu;
            hi += w[u] >> 16;
        }
    }
    for (; g < grid; g += 4) {
        const uint32_t w = packed[static_cast<size_t>(g) * kHybPackedWords + j];
        lo += w & 0xFFFFu;
        hi += w >> 16;
    }
    s_part[part][lane][0] = lo;
    s_part[part][lane][1] = hi;
    __syncthreads();
    if (part == 0) {
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            lo += s_part[q][lane][0];
            hi += s_part[q][lane][1];
        }
        const uint2 sp = reinterpret_cast<const uint2*>(spill)[j];
        reinterpret_cast<uint2*>(counts)[j] = make_uint2(lo + sp.x, hi + sp.y);
    }
}

// One workgroup: exclusive scan of the 65536 counts (thread t owns entries 64t .. 64t+63; a bucket
// has 2^sub_bits >= 64 sub-buckets, so it starts with a thread), bucket starts, the tile table of
// pass B, the list of oversized sub-buckets with their tile table, and the decision.  Buckets and
// sub-buckets the split does not use are empty and start at n.
// If the decision is against the schedule, the digit histograms of the plan's four ordinary passes
// are finished here: `ghist` holds the counts of the two lowest digits (the histogram kernel's), the
// two above them are sums over the exact table (digit p of the keys of sub-bucket i is a bit field
// of i + window_base, because the table starts at or below bit begin_bit + 16), and all four are
// turned into exclusive prefix sums, which is what the passes read.
struct OrdinaryDigits {
    uint32_t* ghist;        // [4][256]: counts of passes 0 and 1 on entry
    uint32_t shift2, mask2; // digit of pass 2 = ((i + window_base) >> shift2) & mask2
    uint32_t shift3, mask3;
    uint32_t window_base;
};

__global__ __launch_bounds__(1024) void hybrid_scan_kernel(HybridTables* tb, HybridCleared* hc, uint32_t n,
                                                            uint32_t tile_items, uint32_t force_mode, uint32_t sub_bits,
                                                            uint32_t compact_ok, const uint32_t cap, const AdaptArgs adapt,
                                                            const OrdinaryDigits od)
{
    // the adaptive plan (below): pairs per level-A cell, their running sums, and per candidate bucket
    // size the cell at which the bucket after the one starting at cell i starts
    constexpr int kCandidates = 4;
    __shared__ uint32_t s_cell[kAdaptCells];
    __shared__ uint32_t s_run[kAdaptCells];
    __shared__ unsigned short s_nxt[kCandidates][kAdaptCells];
    __shared__ unsigned short s_bfirst[kCandidates][kHybBuckets + 2];
    __shared__ uint32_t s_groups[kCandidates], s_maxb[kCandidates];
    __shared__ uint32_t s_wsum[16];
    __shared__ uint32_t s_bigw[16];
    __shared__ uint32_t s_bstart[kHybBuckets + 1];
    __shared__ uint32_t s_tsum[4];
    __shared__ uint32_t s_bigtsum[4];
    __shared__ uint32_t s_bigcnt[kHybMaxBig];
    __shared__ uint32_t s_high[2][256];
    __shared__ uint32_t s_psum[16];
    __shared__ uint32_t s_max, s_bigpairs, s_mode;
    HybridHeader* hdr = &hc->hdr;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t c[64];
    // the split uses the first 2^msd_bits entries of the table; the rest is zero and is neither read
    // nor (below) given a start, except the one entry behind the last sub-bucket
    const uint32_t used = 1u << adapt.msd_bits;
    {
        const uint4* c4 = reinterpret_cast<const uint4*>(tb->counts) + tid * 16;
        const bool live = tid * 64u < used;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint4 x = live ? c4[q] : make_uint4(0u, 0u, 0u, 0u);
            c[4 * q] = x.x; c[4 * q + 1] = x.y; c[4 * q + 2] = x.z; c[4 * q + 3] = x.w;
        }
    }
    uint32_t sum = 0, mx = 0, nbig = 0, bigp = 0;
#pragma unroll
    for (int q = 0; q < 64; ++q) {
        sum += c[q];
        mx = c[q] > mx ? c[q] : mx;
        nbig += c[q] > cap ? 1u : 0u;
        bigp += c[q] > cap ? c[q] : 0u;
    }
    uint32_t incl = sum, bincl = nbig;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        const uint32_t z = __shfl_up(bincl, o);
        if (lane >= static_cast<uint32_t>(o)) {
            incl += y;
            bincl += z;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t y = __shfl_down(mx, o);
        mx = y > mx ? y : mx;
        bigp += __shfl_down(bigp, o);
    }
    if (lane == 63) {
        s_wsum[wave] = incl;
        s_bigw[wave] = bincl;
    }
    if (tid == 0) {
        s_max = 0;
        s_bigpairs = 0;
    }
    if (tid < kHybMaxBig) s_bigcnt[tid] = 0;
    __syncthreads();
    if (lane == 0) {
        atomicMax(&s_max, mx);
        if (bigp) atomicAdd(&s_bigpairs, bigp);
    }
    uint32_t run = incl - sum, bigoff = bincl - nbig, total_big = 0;
    for (uint32_t w = 0; w < 16; ++w) {
        if (w < wave) {
            run += s_wsum[w];
            bigoff += s_bigw[w];
        }
        total_big += s_bigw[w];
    }
    // thread t starts bucket t >> (sub_bits - 6) if it starts one; the threads past the last bucket
    // the split uses hold zeros, so the unused buckets start at n
    const uint32_t tpb = sub_bits - 6u;  // log2(threads per bucket)
    if ((tid & ((1u << tpb) - 1u)) == 0 && (tid >> tpb) <= kHybBuckets) s_bstart[tid >> tpb] = run;
    if (tid == 1023 && (1024u >> tpb) <= kHybBuckets) s_bstart[1024u >> tpb] = run + sum;  // sub_bits == 8: no thread starts "bucket 256"
    if (tid * 64u <= used) {
        uint4* o4 = reinterpret_cast<uint4*>(tb->sub_start) + tid * 16;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            uint4 o;
            o.x = run; run += c[4 * q];
            o.y = run; run += c[4 * q + 1];
            o.z = run; run += c[4 * q + 2];
            o.w = run; run += c[4 * q + 3];
            o4[q] = o;
        }
    } else {
        run += sum;  // (zero: the thread holds no pairs)
    }
    if (tid == 1023) tb->sub_start[kHybSubs] = run;
    // the oversized sub-buckets, in key order (rare: a second walk over the thread's counts)
    if (nbig != 0u) {
        uint32_t at = run - sum;
#pragma unroll  // (a loop that indexes c[] at run time would move the whole array to scratch memory, for every access)
        for (int q = 0; q < 64; ++q) {
            if (c[q] > cap) {
                if (bigoff < kHybMaxBig) {
                    tb->big_start[bigoff] = at;
                    tb->big_end[bigoff] = at + c[q];
                    s_bigcnt[bigoff] = c[q];
                }
                ++bigoff;
            }
            at += c[q];
        }
    }
    __syncthreads();
    uint32_t tiles = 0, tincl = 0, bigtiles = 0, bigtincl = 0;
    if (tid < kHybBuckets) {
        const uint32_t bs = s_bstart[tid];
        const uint32_t cnt = s_bstart[tid + 1] - bs;
        tiles = (cnt + tile_items - 1) / tile_items;
        tb->bucket_start[tid] = bs;
        tincl = tiles;
        bigtiles = (s_bigcnt[tid] + tile_items - 1) / tile_items;  // kHybMaxBig == kHybBuckets: one thread per entry of either table
        bigtincl = bigtiles;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(tincl, o);
            const uint32_t z = __shfl_up(bigtincl, o);
            if (lane >= static_cast<uint32_t>(o)) {
                tincl += y;
                bigtincl += z;
            }
        }
        if (lane == 63) {
            s_tsum[wave] = tincl;
            s_bigtsum[wave] = bigtincl;
        }
    }
    if (tid == kHybBuckets) tb->bucket_start[kHybBuckets] = s_bstart[kHybBuckets];
    __syncthreads();
    if (tid < kHybBuckets) {
        uint32_t off = tincl - tiles, bigoff_t = bigtincl - bigtiles;
        for (uint32_t w = 0; w < 4; ++w)
            if (w < wave) {
                off += s_tsum[w];
                bigoff_t += s_bigtsum[w];
            }
        tb->first_tile[tid] = off;
        tb->big_first_tile[tid] = bigoff_t;  // entries past the last oversized sub-bucket: the total
        if (tid == kHybBuckets - 1) {
            const uint32_t total_tiles = off + tiles;
            const uint32_t total_big_tiles = bigoff_t + bigtiles;
            // Counts that do not add up to n are never acted on.  More oversized sub-buckets than the
            // table holds, or more than a quarter of the pairs in them (those pairs are moved four
            // times and counted twice): the ordinary passes are the better schedule.
            // Nor is a call whose local sort was set up to stage 6 bytes per pair when the keys turn
            // out to differ above the sorted range (the ordinary passes carry any bits along).
            const bool sane = hdr->overflow == 0 && s_bstart[kHybBuckets] == n && force_mode != kHybFallback &&
                              !(compact_ok != 0u && hdr->high_vary != 0u);
            const bool ok = sane && adapt.force == 0u && total_big <= kHybMaxBig && s_bigpairs <= n / 4u;
            // refused for the sizes of the sub-buckets only: buckets and groups cut to the keys' density may do
            const bool try_adapt = sane && !ok && adapt.offered != 0u;
            tb->first_tile[kHybBuckets] = total_tiles;
            tb->big_first_tile[kHybMaxBig] = total_big_tiles;
            hdr->num_tiles_b = ok ? total_tiles : 0u;
            hdr->num_big = ok ? total_big : 0u;
            hdr->num_tiles_big = ok ? total_big_tiles : 0u;
            hdr->big_pairs = s_bigpairs;
            hdr->max_sub = s_max;
            hdr->mode = ok ? kHybRun : kHybFallback;
            s_mode = ok ? kHybRun : (try_adapt ? kHybAdapt : kHybFallback);
        }
    }
    if (tid < 512) (&s_high[0][0])[tid] = 0;
    __syncthreads();
    if (s_mode == kHybRun) return;  // uniform
    // ---- the ordinary passes run: their digit histograms ----
    if (sum != 0u) {
        // consecutive sub-buckets share their digits in runs (one run per thread for pass 3 unless a
        // window base shifts the thread's 64 sub-buckets across a boundary): one add per run
        const uint32_t first = tid * 64u + od.window_base;
        uint32_t run2 = (first >> od.shift2) & od.mask2, sum2 = 0;
        uint32_t run3 = (first >> od.shift3) & od.mask3, sum3 = 0;
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            const uint32_t at = first + static_cast<uint32_t>(q);
            const uint32_t d2 = (at >> od.shift2) & od.mask2, d3 = (at >> od.shift3) & od.mask3;
            if (d2 != run2) {
                if (sum2) atomicAdd(&s_high[0][run2], sum2);
                run2 = d2;
                sum2 = 0;
            }
            if (d3 != run3) {
                if (sum3) atomicAdd(&s_high[1][run3], sum3);
                run3 = d3;
                sum3 = 0;
            }
            sum2 += c[q];
            sum3 += c[q];
        }
        if (sum2) atomicAdd(&s_high[0][run2], sum2);
        if (sum3) atomicAdd(&s_high[1][run3], sum3);
    }
    __syncthreads();
    {
        // thread t: digit t & 255 of pass t >> 8; four waves per pass
        const uint32_t pass = tid >> 8, digit = tid & 255u;
        const uint32_t own = pass < 2 ? od.ghist[tid] : s_high[pass - 2][digit];
        const uint32_t inc = wave_inclusive_sum(own);
        if (lane == 63) s_psum[wave] = inc;
        __syncthreads();
        uint32_t off = inc - own;
        for (uint32_t w = pass * 4u; w < wave; ++w) off += s_psum[w];
        od.ghist[tid] = off;
    }
    if (s_mode != kHybAdapt) return;  // uniform
    // ---- the adaptive plan: at most 256 buckets of whole level-A cells --------------------------
    // pairs per cell (a cell is 4, 8 or 16 consecutive entries of the exact table)
    {
        const uint32_t per_cell = (1u << (adapt.msd_bits - kAdaptCellBits)) - 1u;
        uint32_t acc = 0;
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            acc += c[q];
            if ((static_cast<uint32_t>(q + 1) & per_cell) == 0u) {
                const uint32_t cell = (tid * 64u + static_cast<uint32_t>(q)) >> (adapt.msd_bits - kAdaptCellBits);
                if (cell < kAdaptCells) s_cell[cell] = acc;
                acc = 0;
            }
        }
    }
    __syncthreads();
    {
        // s_run[i] = pairs of cells 0 .. i
        uint32_t v[4], sum4 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = s_cell[tid * 4u + q];
            sum4 += v[q];
        }
        const uint32_t inc = wave_inclusive_sum(sum4);
        if (lane == 63) s_psum[wave] = inc;
        __syncthreads();
        uint32_t at = inc - sum4;
        for (uint32_t w = 0; w < wave; ++w) at += s_psum[w];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            at += v[q];
            s_run[tid * 4u + q] = at;
            tb->cell_count[tid * 4u + q] = v[q];
        }
    }
    __syncthreads();
    // first cell whose running sum exceeds x (4096 if none): the first occupied cell behind the cells summed in x
    auto first_above = [&](uint32_t x) __attribute__((always_inline)) {
        uint32_t lo = 0, hi = kAdaptCells;
#pragma unroll 1
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_run[mid] > x) hi = mid;
            else lo = mid + 1;
        }
        return lo;
    };
    // Greedy cut for four bucket sizes at once: a bucket that starts at cell i takes cells while it
    // stays within the size and within kAdaptMaxSpan cells (one cell at least), and the next one
    // starts at the first occupied cell after it.
    const uint32_t total = s_run[kAdaptCells - 1];
    const uint32_t limits[kCandidates] = {total / 224u + 1u, total / 192u + 1u, total / 160u + 1u, total / 128u + 1u};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t i = tid * 4u + q;
        const uint32_t before = i ? s_run[i - 1] : 0u;
#pragma unroll
        for (int k = 0; k < kCandidates; ++k) {
            uint32_t j = first_above(before + limits[k]);  // cells [i, j) hold at most the limit
            j = j > i + 1u ? j : i + 1u;
            j = j < i + kAdaptMaxSpan ? j : i + kAdaptMaxSpan;
            j = j < kAdaptCells ? j : kAdaptCells;
            s_nxt[k][i] = static_cast<unsigned short>(first_above(s_run[j - 1]));
        }
    }
    __syncthreads();
    if (tid < kCandidates) {
        uint32_t i = first_above(0u), g = 0, mb = 0;
#pragma unroll 1
        while (i < kAdaptCells && g <= kHybBuckets) {
            if (g < kHybBuckets) s_bfirst[tid][g] = static_cast<unsigned short>(i);
            const uint32_t nx = s_nxt[tid][i];
            const uint32_t cnt = s_run[nx - 1] - (i ? s_run[i - 1] : 0u);
            mb = cnt > mb ? cnt : mb;
            ++g;
            i = nx;
        }
        s_groups[tid] = g;
        s_maxb[tid] = mb;
    }
    __syncthreads();
    int pick = -1;
#pragma unroll
    for (int k = kCandidates - 1; k >= 0; --k)
        if (s_groups[k] <= kHybBuckets && s_maxb[k] <= adapt.bucket_max) pick = k;
    if (pick < 0) return;  // uniform: the ordinary passes run (hdr->mode says so already)
    const uint32_t buckets = s_groups[pick];
    uint32_t atiles = 0, atincl = 0;
    if (tid < kHybBuckets) {
        const bool live = tid < buckets;
        const uint32_t first = live ? s_bfirst[pick][tid] : kAdaptCells;
        const uint32_t behind = tid + 1u < buckets ? s_bfirst[pick][tid + 1] : kAdaptCells;  // first cell of the next bucket
        const uint32_t before = live && first ? s_run[first - 1] : 0u;
        const uint32_t cnt = live ? s_run[behind - 1] - before : 0u;
        // last occupied cell of the bucket: the first cell whose running sum reaches the bucket's end
        uint32_t last = 0;
        if (live) {
            const uint32_t upto = s_run[behind - 1];
            uint32_t lo = first, hi = behind - 1u;
#pragma unroll 1
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_run[mid] >= upto) hi = mid;
                else lo = mid + 1;
            }
            last = lo;
        }
        tb->bucket_start[tid] = live ? before : n;
        tb->cell_first[tid] = first;
        tb->cell_last[tid] = last;
        const uint32_t from = tid == 0u ? 0u : first;  // the empty cells before the first bucket count as its own
#pragma unroll 1
        for (uint32_t cell = from; live && cell < behind; ++cell) tb->map_a[cell] = static_cast<uint8_t>(tid);
        atiles = (cnt + tile_items - 1) / tile_items;
        atincl = wave_inclusive_sum(atiles);
        if (lane == 63) s_tsum[wave] = atincl;
    }
    __syncthreads();
    if (tid < kHybBuckets) {
        uint32_t off = atincl - atiles;
        for (uint32_t w = 0; w < wave; ++w) off += s_tsum[w];
        tb->first_tile[tid] = off;
        if (tid == kHybBuckets - 1) {
            tb->first_tile[kHybBuckets] = off + atiles;
            tb->bucket_start[kHybBuckets] = n;
            hdr->num_tiles_b = off + atiles;
            hdr->adapt_buckets = buckets;
            hdr->max_sub = 0;    // the counting kernel's workgroups enter the groups' sizes
            hdr->big_pairs = 0;  // ... and the fine cells that are too large for a workgroup
            hdr->big_limit = n / 4u;
            hdr->mode = kHybAdapt;
        }
    }
}

// ---- the adaptive schedule: fine cells and groups of one bucket ------------------------------
// Workgroup b, after pass A has gathered bucket b in `keys`:
//   1. divides every occupied level-A cell of the bucket into a power of two of fine cells, about one
//      per q pairs, with q the smallest of cap/512, cap/256, ... that keeps the bucket within 2^14 fine
//      cells; a fine cell never straddles a multiple of 1 << low16_shift (the local sort takes 16 bits);
//   2. counts the bucket's keys per fine cell (one LDS atomic per key; one add for a whole wave
//      instruction whose lanes hold one fine cell);
//   3. cuts the run of fine cells into groups greedily: a group takes fine cells while it stays
//      within `cap` pairs and inside one 16-bit cell, one fine cell at least; where group g of the
//      bucket starts is sub_start[256 b + g], which fine cells start a group is rank[b];
//   4. enters fine cells of more than `cap` pairs in the list of oversized sub-buckets.
// A bucket of more than 256 groups raises hdr->adapt_fail; that, or too many or too full oversized
// cells, hands the call to the ordinary passes (hybrid_mode).  The tile table of the passes over the
// oversized cells is made by their histogram kernel.
__global__ __launch_bounds__(1024) void hybrid_adapt_count_kernel(const uint32_t* __restrict__ keys, HybridTables* tb,
                                                                   HybridCleared* hc, const uint32_t n,
                                                                   const uint32_t tile_items, const uint32_t cap,
                                                                   const AdaptArgs adapt)
{
    constexpr uint32_t THREADS = 1024;
    __shared__ uint32_t s_fine[kAdaptFine];              // pairs per fine cell, then their running sums
    __shared__ unsigned short s_next[kAdaptFine];        // fine cell at which the group after the one starting here starts
    __shared__ unsigned short s_bpos[kAdaptFine / 4 + 1];  // fine cells that begin a 16-bit cell, ascending
    __shared__ uint32_t s_bound[kAdaptRankWords];        // the same as a bit map
    __shared__ uint32_t s_bpre[kAdaptRankWords];         // set bits of s_bound in the words before
    __shared__ uint32_t s_start[kAdaptRankWords];        // fine cells that start a group
    __shared__ uint32_t s_entry[kAdaptMaxSpan];
    __shared__ unsigned short s_gstart[kHybBuckets + 1];
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_total, s_ngroups, s_listed, s_nocc;
    __shared__ uint32_t s_occ[kAdaptMaxSpan];  // occupied level-A cells of the bucket, ascending: first fine cell | cell << 16
    HybridHeader* hdr = &hc->hdr;
    if (hdr->mode != kHybAdapt) return;  // uniform
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t b = blockIdx.x;
    const uint32_t start = tb->bucket_start[b], end = tb->bucket_start[b + 1];
    const bool live = b < hdr->adapt_buckets && end > start;  // uniform
    uint32_t ngroups = 0;
    if (live) {
        const uint32_t first_cell = tb->cell_first[b], span = tb->cell_last[b] - first_cell + 1u;  // <= kAdaptMaxSpan
        const uint32_t cell_shift = adapt.cell_shift;
        const uint32_t min_log = cell_shift - adapt.low16_shift;  // a cell has at least this many fine cells (log2)
        // ... and at most so many that a fine cell's number ends where the sorted bits begin (keys that
        // differ below that are equal to a stable sort)
        const uint32_t max_log = min(cell_shift - (adapt.low16_shift - 16u), kAdaptFineBits);
        // ---- 1. fine cells per level-A cell ----
        const uint32_t mine = tid < span ? tb->cell_count[first_cell + tid] : 0u;
        uint32_t want_log = 0, want = 0;
        for (uint32_t q = cap / 512u > 0u ? cap / 512u : 1u;; q *= 2u) {
            want = 0;
            if (mine != 0u) {
                const uint32_t parts = (mine + q - 1u) / q;  // >= 1
                want_log = 32u - static_cast<uint32_t>(__clz(parts - 1u | 0u));
                if (parts <= 1u) want_log = 0;
                want_log = want_log > min_log ? want_log : min_log;
                want_log = want_log < max_log ? want_log : max_log;
                want = 1u << want_log;
            }
            const uint32_t inc = wave_inclusive_sum(want);
            __syncthreads();  // s_w is read below by every thread of the previous round
            if (lane == 63) s_w[wave] = inc;
            __syncthreads();
            uint32_t all = 0, at = inc - want;
            for (uint32_t w = 0; w < 16; ++w) {
                all += s_w[w];
                if (w < wave) at += s_w[w];
            }
            if (all <= kAdaptFine) {  // uniform
                if (tid < span) {
                    const uint32_t fine_shift = cell_shift - want_log;
                    const uint32_t e = mine != 0u ? (at | (fine_shift << 16)) : 0u;
                    s_entry[tid] = e;
                    tb->cell_entry[first_cell + tid] = e;
                }
                if (tid == 0) s_total = all;
                break;
            }
        }
        // the occupied cells as a list (for the way back from a fine cell to its keys)
        {
            const uint32_t flag = tid < span && mine != 0u ? 1u : 0u;
            const uint32_t inc = wave_inclusive_sum(flag);
            __syncthreads();  // every thread has read s_w in the last round above
            if (lane == 63) s_w[wave] = inc;
            __syncthreads();
            uint32_t at = inc - flag;
            for (uint32_t w = 0; w < wave; ++w) at += s_w[w];
            if (flag) s_occ[at] = (s_entry[tid] & 0xFFFFu) | (tid << 16);
            if (tid == THREADS - 1u) s_nocc = at + flag;
        }
        for (uint32_t j = tid; j < kAdaptFine; j += THREADS) s_fine[j] = 0;
        for (uint32_t j = tid; j < kAdaptRankWords; j += THREADS) {
            s_bound[j] = 0;
            s_start[j] = 0;
        }
        __syncthreads();
        // fine cells that begin a 16-bit cell
        if (tid < span && mine != 0u) {
            const uint32_t base = s_entry[tid] & 0xFFFFu;
            const uint32_t step = want >> min_log;  // fine cells per 16-bit cell
            for (uint32_t j = 0; j < (1u << min_log); ++j) {
                const uint32_t f = base + j * step;
                atomicOr(&s_bound[f >> 5], 1u << (f & 31u));
            }
        }
        // ---- 2. count ----
        auto count = [&](uint32_t k) __attribute__((always_inline)) {
            const uint32_t e = s_entry[((k >> cell_shift) & (kAdaptCells - 1u)) - first_cell];
            const uint32_t f = (e & 0xFFFFu) + (__builtin_amdgcn_ubfe(k, 0u, cell_shift) >> (e >> 16));
            const uint32_t f0 = __builtin_amdgcn_readfirstlane(f);
            const unsigned long long same = __builtin_amdgcn_uicmp(f, f0, 32 /* == */);
            const uint32_t group = static_cast<uint32_t>(__popcll(same));
            if (group >= 8u && f == f0) {
                if (lanes_below(static_cast<uint32_t>(same), static_cast<uint32_t>(same >> 32)) == 0u) atomicAdd(&s_fine[f0], group);
            } else {
                atomicAdd(&s_fine[f], 1u);
            }
        };
        {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            uint32_t head_end = start, body_end = start;
            if ((reinterpret_cast<uintptr_t>(keys) & 15u) == 0u) {
                head_end = (start + 3u) & ~3u;
                head_end = head_end < end ? head_end : end;
                body_end = head_end + ((end - head_end) & ~3u);
            } else {
                head_end = end;
                body_end = end;
            }
            for (uint32_t i = start + tid; i < head_end; i += THREADS) count(keys[i]);
            const u32x4* __restrict__ vk = reinterpret_cast<const u32x4*>(keys);
            const uint32_t v_end = body_end / 4u;
            uint32_t v = head_end / 4u + tid;
            for (; v + 7u * THREADS < v_end; v += 8u * THREADS) {
                u32x4 x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = __builtin_nontemporal_load(vk + v + u * THREADS);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    count(x[u].x); count(x[u].y); count(x[u].z); count(x[u].w);
                }
            }
            for (; v < v_end; v += THREADS) {
                const u32x4 x = __builtin_nontemporal_load(vk + v);
                count(x.x); count(x.y); count(x.z); count(x.w);
            }
            for (uint32_t i = body_end + tid; i < end; i += THREADS) count(keys[i]);
        }
        __syncthreads();
        // ---- 3. groups ----
        // running sums in place: thread t owns fine cells 16 t .. 16 t + 15
        {
            uint32_t v[16], sum = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                v[q] = s_fine[tid * 16u + q];
                sum += v[q];
            }
            const uint32_t inc = wave_inclusive_sum(sum);
            if (lane == 63) s_w[wave] = inc;
            __syncthreads();
            uint32_t at = inc - sum;
            for (uint32_t w = 0; w < wave; ++w) at += s_w[w];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                at += v[q];
                s_fine[tid * 16u + q] = at;
            }
        }
        __syncthreads();
        // the 16-bit cells' first fine cells as a sorted list
        if (tid < kAdaptRankWords) {
            const uint32_t bits = s_bound[tid];
            const uint32_t cnt = __popc(bits);
            const uint32_t inc = wave_inclusive_sum(cnt);  // waves 0 to 7, all lanes
            if (lane == 63) s_w[wave] = inc;
            s_bpre[tid] = inc - cnt;
        }
        __syncthreads();
        if (tid < kAdaptRankWords) {
            uint32_t at = s_bpre[tid];
            for (uint32_t w = 0; w < wave; ++w) at += s_w[w];
            s_bpre[tid] = at;
            uint32_t bits = s_bound[tid];
            while (bits) {
                const uint32_t bit = __builtin_ctz(bits);
                bits &= bits - 1u;
                s_bpos[at++] = static_cast<unsigned short>(tid * 32u + bit);
            }
        }
        __syncthreads();
        const uint32_t total_fine = s_total;
        // first fine cell whose running sum exceeds x (kAdaptFine if none)
        auto first_above = [&](uint32_t x) __attribute__((always_inline)) {
            uint32_t lo = 0, hi = kAdaptFine;
#pragma unroll 1
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_fine[mid] > x) hi = mid;
                else lo = mid + 1;
            }
            return lo;
        };
        const uint32_t boundaries = s_bpre[kAdaptRankWords - 1] + __popc(s_bound[kAdaptRankWords - 1]);
#pragma unroll 1
        for (uint32_t q = 0; q < 16; ++q) {
            const uint32_t f = tid * 16u + q;
            if (f >= total_fine) {  // no pairs there: never visited
                s_next[f] = static_cast<unsigned short>(kAdaptFine);
                continue;
            }
            const uint32_t before = f ? s_fine[f - 1] : 0u;
            uint32_t e = first_above(before + cap);  // fine cells [f, e) hold at most cap pairs
            e = e > f + 1u ? e : f + 1u;
            // ... and lie in one 16-bit cell: the next of those begins at the first listed fine cell after f
            const uint32_t listed = s_bpre[f >> 5] + __popc(s_bound[f >> 5] & (0xFFFFFFFFu >> (31u - (f & 31u))));
            const uint32_t wall = listed < boundaries ? s_bpos[listed] : kAdaptFine;
            e = e < wall ? e : wall;
            const uint32_t nx = first_above(s_fine[e - 1]);  // the first occupied fine cell behind the group
            s_next[f] = static_cast<unsigned short>(nx);  // kAdaptFine: the group takes the bucket's last pair
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t f = first_above(0u), g = 0;
#pragma unroll 1
            while (f < kAdaptFine && g <= kHybBuckets) {
                if (g < kHybBuckets) s_gstart[g] = static_cast<unsigned short>(f);
                ++g;
                f = s_next[f];
            }
            s_ngroups = f < kAdaptFine ? kHybBuckets + 1u : g;
        }
        __syncthreads();
        ngroups = s_ngroups;
    }
    // ---- 4. tables ----
    const bool fits = ngroups <= kHybBuckets;
    // this bucket's part of the list of all groups
    if (tid == 0) s_listed = live && fits ? atomicAdd(&hdr->adapt_groups, ngroups) : 0u;
    __syncthreads();
    if (tid < kHybBuckets) {
        uint32_t at = end, cnt = 0;
        if (live && fits && tid < ngroups) {
            const uint32_t f = s_gstart[tid];
            const uint32_t before = f ? s_fine[f - 1] : 0u;
            const uint32_t upto = tid + 1u < ngroups ? s_fine[s_gstart[tid + 1] - 1u] : end - start;
            at = start + before;
            cnt = upto - before;
            atomicOr(&s_start[f >> 5], 1u << (f & 31u));
            if (cnt > cap) {
                const uint32_t slot = atomicAdd(&hdr->num_big, 1u);
                if (slot < kHybMaxBig) {
                    tb->big_start[slot] = at;
                    tb->big_end[slot] = at + cnt;
                }
                atomicAdd(&hdr->big_pairs, cnt);
            }
            atomicMax(&hdr->max_sub, cnt);
            // The low key bits in which keys of the group can differ: from the first key of its first
            // fine cell and the last key of its last occupied one (both in one 16-bit cell).
            uint32_t span_bits = 16;
            {
                const uint32_t first_cell = tb->cell_first[b], cell_shift = adapt.cell_shift;
                // last occupied fine cell: the first whose running sum reaches the group's end
                uint32_t lo = f, hi = kAdaptFine - 1u;
#pragma unroll 1
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_fine[mid] >= upto) hi = mid;
                    else lo = mid + 1;
                }
                const uint32_t f_last = lo;
                auto key_of = [&](uint32_t fine, bool last_key) __attribute__((always_inline)) {
                    // the occupied cell with the largest first fine cell <= fine
                    uint32_t a = 0, z = s_nocc - 1u;
#pragma unroll 1
                    while (a < z) {
                        const uint32_t mid = (a + z + 1u) >> 1;
                        if ((s_occ[mid] & 0xFFFFu) <= fine) a = mid;
                        else z = mid - 1u;
                    }
                    const uint32_t cell = s_occ[a] >> 16, base = s_occ[a] & 0xFFFFu;
                    const uint32_t fine_shift = s_entry[cell] >> 16;
                    uint32_t k = ((first_cell + cell) << cell_shift) | ((fine - base) << fine_shift);
                    if (last_key) k |= (1u << fine_shift) - 1u;
                    return k;
                };
                const uint32_t differ = (key_of(f, false) ^ key_of(f_last, true)) & 0xFFFFu;
                span_bits = 32u - static_cast<uint32_t>(__clz(differ));
            }
            tb->group_first[s_listed + tid] = at;  // at most 65536 in all: 256 buckets of at most 256
            tb->group_pairs[s_listed + tid] = cnt | (span_bits << 27);  // cnt <= n / 4 < 2^27 wherever the schedule is offered
        }
        tb->sub_start[b * kHybBuckets + tid] = at;
        tb->counts[b * kHybBuckets + tid] = cnt;
        if (b == gridDim.x - 1u && tid == 0) tb->sub_start[kHybSubs] = n;
    }
    if (live && !fits && tid == 0) atomicOr(&hdr->adapt_fail, 1u);
    __syncthreads();
    if (live && tid < kAdaptRankWords) {
        const uint32_t bits = s_start[tid];
        const uint32_t cnt = __popc(bits);
        const uint32_t inc = wave_inclusive_sum(cnt);
        if (lane == 63) s_w[wave] = inc;
        s_bpre[tid] = inc - cnt;
    }
    __syncthreads();
    if (live && tid < kAdaptRankWords) {
        uint32_t at = s_bpre[tid];
        for (uint32_t w = 0; w < wave; ++w) at += s_w[w];
        // digit of fine cell f = group starts at or before f, minus one
        tb->rank[b][tid] = static_cast<unsigned long long>(s_start[tid]) | (static_cast<unsigned long long>(at - 1u) << 32);
    }
}

// ---- the oversized sub-buckets: digit counts and run positions -------------------------------
// counts of the two local digits of every oversized sub-bucket, tile by tile of the passes that
// will move them (a tile never straddles two of them)
struct BigHistogramArgs {
    const uint32_t* keys;
    HybridTables* tb;
    HybridCleared* hc;
    uint32_t tile_items, shift, bits1, bits2;
};
struct BigHistogramShared {
    uint32_t h[2][256];
    uint32_t seg[3];
    uint32_t ft[4];
};
// (a device function: the launch it runs in is also an ordinary pass of the other schedule, see
// hybrid_big_histogram_kernel)
template <int THREADS>
__device__ __forceinline__ void big_histogram(const BigHistogramArgs& g, BigHistogramShared& sm, const uint32_t mode)
{
    static_assert(THREADS >= 512, "a thread per counter of the two digits");
    const uint32_t* __restrict__ keys = g.keys;
    HybridTables* tb = g.tb;
    HybridCleared* hc = g.hc;
    const uint32_t tile_items = g.tile_items, shift = g.shift, bits1 = g.bits1, bits2 = g.bits2;
    uint32_t (&s_h)[2][256] = sm.h;
    uint32_t (&s_seg)[3] = sm.seg;
    uint32_t (&s_ft)[4] = sm.ft;
    const uint32_t tid = threadIdx.x;
    uint32_t num_tiles = hc->hdr.num_tiles_big;
    uint32_t ft0 = 0, ft1 = 0, bs0 = 0, bs1 = 0;
    if (mode == kHybAdapt) {
        // the adaptive schedule's list is in no order and has no tile table yet: every workgroup makes
        // it for itself, the first one also for the two passes that follow
        const uint32_t nbig = hc->hdr.num_big;  // <= kHybMaxBig in this mode
        if (nbig == 0u) return;
        uint32_t tiles = 0, incl = 0;
        if (tid < kHybMaxBig) {  // waves 0 to 3, all lanes
            if (tid < nbig) {
                bs0 = tb->big_start[tid];
                bs1 = tb->big_end[tid];
                tiles = (bs1 - bs0 + tile_items - 1u) / tile_items;
            }
            incl = wave_inclusive_sum(tiles);
            if ((tid & 63u) == 63u) s_ft[tid >> 6] = incl;
        }
        __syncthreads();
        if (tid < kHybMaxBig) {
            ft0 = incl - tiles;
            for (uint32_t w = 0; w < (tid >> 6); ++w) ft0 += s_ft[w];
            ft1 = ft0 + tiles;
            if (blockIdx.x == 0) tb->big_first_tile[tid] = ft0;
        }
        num_tiles = s_ft[0] + s_ft[1] + s_ft[2] + s_ft[3];
        if (blockIdx.x == 0 && tid == 0) {
            tb->big_first_tile[kHybMaxBig] = num_tiles;
            hc->hdr.num_tiles_big = num_tiles;
        }
    } else {
        if (num_tiles == 0u) return;  // uniform: the usual case, and nothing below is needed
        if (tid < kHybMaxBig) {
            ft0 = tb->big_first_tile[tid];
            ft1 = tb->big_first_tile[tid + 1];
            bs0 = tb->big_start[tid];
            bs1 = tb->big_end[tid];
        }
    }
    for (uint32_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        if (tid < 512) (&s_h[0][0])[tid] = 0;
        if (tid < kHybMaxBig && ft0 <= tile && tile < ft1) {  // exactly one thread
            s_seg[0] = tid;
            s_seg[1] = bs0 + (tile - ft0) * tile_items;
            s_seg[2] = bs1;
        }
        __syncthreads();
        const uint32_t seg = s_seg[0], base = s_seg[1];
        const uint32_t left = s_seg[2] - base;
        const uint32_t cnt = left < tile_items ? left : tile_items;
        for (uint32_t i = tid; i < cnt; i += THREADS) {
            const uint32_t k = keys[base + i];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t d = h == 0 ? __builtin_amdgcn_ubfe(k, shift, bits1) : __builtin_amdgcn_ubfe(k, shift + bits1, bits2);
                // the keys of an oversized sub-bucket are often few distinct values: one add for
                // the lanes that hold the first lane's digit
                const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
                const unsigned long long same = __builtin_amdgcn_uicmp(d, d0, 32 /* == */);
                const uint32_t group = static_cast<uint32_t>(__popcll(same));
                if (group >= 8u && d == d0) {
                    if (lanes_below(static_cast<uint32_t>(same), static_cast<uint32_t>(same >> 32)) == 0u) atomicAdd(&s_h[h][d0], group);
                } else {
                    atomicAdd(&s_h[h][d], 1u);
                }
            }
        }
        __syncthreads();
        if (tid < 512) {
            const uint32_t cc = (&s_h[0][0])[tid];
            if (cc) atomicAdd(&hc->big_hist[seg][tid >> 8][tid & 255u], cc);
        }
        __syncthreads();  // s_h and s_seg are rewritten for the next tile
    }
}

// ---- chained-scan passes that the device switches on or off --------------------------------
// The digit pass of mrs_radix_pass.hpp behind a gate.  One launch serves both schedules, which
// never run in the same call: if hdr->mode == kHybRun the pass `run` describes, over the whole
// array (pass A) or, with a segment table, over segments of it (pass B: the buckets pass A made;
// the two passes over the oversized sub-buckets): tile t belongs to the segment s with
// first_tile[s] <= t < first_tile[s+1], covers TILE elements from start[s] + (t - first_tile[s]) *
// TILE (fewer in the segment's last tile), and its chained scan starts at first_tile[s] with the
// segment's own digit bases.  If hdr->mode == kHybFallback (and the launch is DUAL) the pass `other`
// describes over the whole array: one of the plan's ordinary passes.  A launch that has
// nothing to do in the mode the device chose costs its workgroups' start and end, so the fewer
// launches the better: four launches carry A, B and the two oversized passes of one schedule and
// the four ordinary passes of the other.
// PERSISTENT: the workgroup draws tiles until none is left (same time per pass as a workgroup
// per tile within 1 %, also when the next ticket is drawn during the write-out: the hand-over from
// tile to tile is not what bounds a pass; but a launch with nothing to do costs 0.005 ms instead of
// 0.1, and one that runs holds every compute unit until the pass ends, which a collective waiting
// on another stream feels).
// One schedule's pass: every access to its description is an access to kernel arguments known at
// compile time (a description picked at run time from two made every pass 15 to 30 % slower).
// MAP == kMapRank (pass B of the adaptive schedule): the digit of a key is its group inside the
// tile's bucket; the bucket's tables (`tables`) are copied to `s_entry` and `s_rank` tile by tile.
// COUNTS: the segment table's digit_base holds the digit COUNTS of each segment, not yet the places
// of the digits' runs (the passes over oversized sub-buckets: their histogram kernel leaves counts);
// every tile turns its segment's 256 counts into places, in `s_bases`.
template <bool SEG, bool PERSISTENT, int MAP, int THREADS, int KPT, bool COUNTS = false, typename StatusT = uint32_t>
__device__ __forceinline__ void gated_pass(const PassArgs& a, const SegmentTable& seg,
                                           PassShared<8, THREADS, KPT, uint32_t, uint32_t>& sh, uint32_t (&s_seg)[4],
                                           const HybridTables* __restrict__ tables = nullptr, uint32_t* s_entry = nullptr,
                                           unsigned long long* s_rank = nullptr, uint32_t* s_bases = nullptr)
{
    using S = PassShared<8, THREADS, KPT, uint32_t, uint32_t>;
    uint32_t tid = threadIdx.x;
    const uint32_t num_tiles = SEG ? *seg.num_tiles : a.num_tiles;
    // no ticket is drawn for a pass without tiles (the usual state of the passes over oversized
    // sub-buckets: with die-aware tickets a workgroup would try all eight lists in turn)
    if (num_tiles == 0u) return;  // uniform
    uint32_t ft0 = 0, ft1 = 0, bs0 = 0, bs1 = 0;
    if constexpr (SEG) {
        if (tid < kHybBuckets) {  // a segment table has at most 256 segments
            ft0 = seg.first_tile[tid];
            ft1 = seg.first_tile[tid + 1];
            bs0 = seg.start[tid];
            bs1 = seg.end[tid];
        }
    }
    for (;;) {
        if constexpr (PERSISTENT) {
            // nothing derived from the thread's number is to be carried over from tile to tile (the
            // registers are needed inside the tile: 19 spilled values and 9 % of the time otherwise)
            asm volatile("" : "+v"(tid));
        }
        if (tid == 0) {
            sh.tile = ticket_tile(a, ticket_issue(a), num_tiles);
            sh.abort = 0;
        }
        __syncthreads();
        const uint32_t tile = sh.tile;
        if (tile >= num_tiles) return;  // uniform
        // Ranking touches only the wave's own counters, so each wave clears its own row and no
        // barrier is needed before the ranking (the rows are read across waves after its barrier);
        // a workgroup without a tile has left by now.
        for (int j = tid & 63; j < S::RADIX; j += 64) sh.wave_hist[tid >> 6][j] = 0;
        if constexpr (SEG) {
            if (tid < kHybBuckets && ft0 <= tile && tile < ft1) {  // exactly one thread: empty segments have ft0 == ft1
                s_seg[0] = tid;
                s_seg[1] = ft0;
                s_seg[2] = bs0;
                s_seg[3] = bs1;
            }
            __syncthreads();
            const uint32_t which = s_seg[0], first = s_seg[1], start = s_seg[2], end = s_seg[3];
            const uint32_t base = start + (tile - first) * static_cast<uint32_t>(S::TILE);
            const uint32_t remaining = end - base;
            PassArgs b = a;
            b.digit_base = seg.digit_base + which * seg.digit_stride;
            if constexpr (MAP == kMapRank) {
                static_assert(SEG, "groups are numbered inside a bucket");
                // the end-of-tile barrier below is behind every look-up of the previous tile's tables
                const uint32_t first_cell = tables->cell_first[which];
                const uint32_t cells = tables->cell_last[which] - first_cell + 1u;  // <= kAdaptMaxSpan
                for (uint32_t j = tid; j < cells; j += THREADS) s_entry[j] = tables->cell_entry[first_cell + j];
                for (uint32_t j = tid; j < kAdaptRankWords; j += THREADS) s_rank[j] = tables->rank[which][j];
                __syncthreads();
                b.digit_base = seg.digit_base + which * kHybBuckets;  // group g of bucket b is table entry 256 b + g
                b.cell_entry = s_entry;
                b.rank_words = s_rank;
                b.cell_first = first_cell;
            }
            if constexpr (COUNTS) {
                static_assert(SEG && THREADS >= 256, "a thread per digit");
                // s_bases[d] = start of the segment + counts of the digits below d; s_bases[256 ..] carry the waves' sums
                uint32_t cnt = 0, inc = 0;
                if (tid < 256u) {  // waves 0 to 3, all lanes
                    cnt = b.digit_base[tid];
                    inc = wave_inclusive_sum(cnt);
                    if ((tid & 63u) == 63u) s_bases[256u + (tid >> 6)] = inc;
                }
                __syncthreads();
                if (tid < 256u) {
                    uint32_t off = start + inc - cnt;
                    for (uint32_t w = 0; w < (tid >> 6); ++w) off += s_bases[256u + w];
                    s_bases[tid] = off;
                }
                __syncthreads();
                b.digit_base = s_bases;
            }
            if (remaining >= static_cast<uint32_t>(S::TILE))
                process_tile<kModeOnesweep, kRankAtomic, MAP, true, StatusT>(b, sh, tile, S::TILE, base, first, tid);
            else
                process_tile<kModeOnesweep, kRankAtomic, MAP, false, StatusT>(b, sh, tile, remaining, base, first, tid);
        } else {
            const uint32_t remaining = a.n - tile * static_cast<uint32_t>(S::TILE);
            const unsigned long long base = static_cast<unsigned long long>(tile) * S::TILE;
            if (remaining >= static_cast<uint32_t>(S::TILE))
                process_tile<kModeOnesweep, kRankAtomic, MAP, true, StatusT>(a, sh, tile, S::TILE, base, 0u, tid);
            else
                process_tile<kModeOnesweep, kRankAtomic, MAP, false, StatusT>(a, sh, tile, remaining, base, 0u, tid);
        }
        if (!PERSISTENT) return;
        // A look-back that gave up has raised the plan's error flag: the output is lost, the ticket of
        // the next tile may never have been drawn, and every other workgroup gives up on that flag.
        if (sh.abort) return;  // uniform: read after the barrier behind which it was written
        __syncthreads();  // the tile's LDS image is dead
    }
}

// SEG: `run` is a pass over the segments of `seg`.  DUAL: `other` exists (persistent launches only).
// ROLE names the launch for profilers and changes nothing else (kPassA, kPassB, kPassOversized,
// kPassOrdinary): without it pass B and the usually idle passes over oversized sub-buckets would be
// one kernel name, and a profiler's average duration of it would describe neither.
// ADAPT: the launch also holds the same pass of the adaptive schedule (`adapt`; pass A by the byte
// table adapt.digit_map, pass B over the same segment table by the groups of `tables`).
enum PassRole { kPassA = 0, kPassB = 1, kPassOversized = 2, kPassOrdinary = 3 };
// OTHER_WIDE: `other` uses 64-bit tile status words and 64-bit addressing (plans of 2^30 pairs, whose
// ordinary passes can count 2^30 keys in one digit; `run` and `adapt` never do).
template <bool SEG, bool DUAL, bool PERSISTENT, int MAP, int THREADS, int KPT, int ROLE, bool ADAPT, bool OTHER_WIDE = false>
__global__ __launch_bounds__(THREADS, 1) void hybrid_pass_kernel(const PassArgs run, const SegmentTable seg,
                                                                  const PassArgs other, const HybridHeader* __restrict__ hdr,
                                                                  const PassArgs adapt, const HybridTables* __restrict__ tables)
{
    using S = PassShared<8, THREADS, KPT, uint32_t, uint32_t>;
    __shared__ S sh;
    __shared__ uint32_t s_seg[4];
    const uint32_t mode = hybrid_mode(*hdr);  // uniform
    // (the passes over oversized sub-buckets are the same in the adaptive schedule: its list holds fine cells)
    if constexpr (ROLE == kPassOversized) {
        // (the same in the adaptive schedule: its list holds fine cells)
        if (mode == kHybRun || mode == kHybAdapt) {
            __shared__ uint32_t s_bases[256 + 4];
            gated_pass<SEG, PERSISTENT, MAP, THREADS, KPT, true>(run, seg, sh, s_seg, nullptr, nullptr, nullptr, s_bases);
            return;
        }
    } else if (mode == kHybRun) {
        gated_pass<SEG, PERSISTENT, MAP, THREADS, KPT>(run, seg, sh, s_seg);
        return;
    }
    if constexpr (ADAPT) {
        if (mode == kHybAdapt) {
            if constexpr (SEG) {
                __shared__ uint32_t s_entry[kAdaptMaxSpan];
                __shared__ unsigned long long s_rank[kAdaptRankWords];
                gated_pass<true, PERSISTENT, kMapRank, THREADS, KPT>(adapt, seg, sh, s_seg, tables, s_entry, s_rank);
            } else {
                // the first barrier of the pass orders these stores before any look-up
                __shared__ __attribute__((aligned(4))) uint8_t s_map[kAdaptCells];
                for (uint32_t j = threadIdx.x; j < kAdaptCells / 4; j += THREADS)
                    reinterpret_cast<uint32_t*>(s_map)[j] = reinterpret_cast<const uint32_t*>(adapt.digit_map)[j];
                PassArgs b = adapt;
                b.digit_map = s_map;
                gated_pass<false, PERSISTENT, kMapTable, THREADS, KPT>(b, seg, sh, s_seg);
            }
            return;
        }
    }
    if constexpr (DUAL) {
        if (mode == kHybFallback) {
            if constexpr (OTHER_WIDE) gated_pass<false, PERSISTENT, kMapNone, THREADS, KPT, false, unsigned long long>(other, seg, sh, s_seg);
            else gated_pass<false, PERSISTENT, kMapNone, THREADS, KPT>(other, seg, sh, s_seg);
        }
    }
}

// The digit counts of the oversized sub-buckets (three-move schedules) or, if the device chose the
// ordinary passes, one of those: the two never run in the same call, and a launch of its own for
// either would have nothing to do in most calls.
template <int THREADS, int KPT, bool OTHER_WIDE>
__global__ __launch_bounds__(THREADS, 1) void hybrid_big_histogram_kernel(const BigHistogramArgs g, const PassArgs other,
                                                                           const HybridHeader* __restrict__ hdr)
{
    using S = PassShared<8, THREADS, KPT, uint32_t, uint32_t>;
    __shared__ S sh;
    __shared__ uint32_t s_seg[4];
    __shared__ BigHistogramShared sm;
    const uint32_t mode = hybrid_mode(*hdr);  // uniform
    if (mode == kHybRun || mode == kHybAdapt) {
        big_histogram<THREADS>(g, sm, mode);
    } else if (mode == kHybFallback && other.keys_in != nullptr) {
        const SegmentTable none{};
        if constexpr (OTHER_WIDE) gated_pass<false, true, kMapNone, THREADS, KPT, false, unsigned long long>(other, none, sh, s_seg);
        else gated_pass<false, true, kMapNone, THREADS, KPT>(other, none, sh, s_seg);
    }
}

// ---- local sort -----------------------------------------------------------------------------
// A workgroup takes sub-buckets blockIdx.x, blockIdx.x + gridDim.x, ...: those of at most
// THREADS * KPT pairs (larger ones are the segmented passes'), loaded in the element order of the
// digit pass (wave w owns a contiguous slice, item i of lane l is element w*64*ku + i*64 + l,
// ku = items per thread of THIS sub-bucket), sorted by two stable counting passes through LDS on
// bits [shift, shift + bits1) and [shift + bits1, shift + bits1 + bits2), each at most 8 bits wide,
// ranked like the digit pass (returning LDS atomics on per-wave counters, lane order), and
// written back over the sub-bucket, which nobody else touches.
// COMPACT: a pair is staged as its low 16 key bits and its payload, 6 bytes instead of 8, so the
// same 60 KiB of LDS hold 10240 pairs instead of 7680; the rest of the key is the same for every
// key of the sub-bucket (key_shift <= 16, and the histogram kernel found no differing bits above
// the sorted range: the decision kernel refuses the schedule otherwise) and is put back at the
// write-out.
constexpr uint32_t kLocalMinGrid = 256;  // so that a workgroup has at most this many sub-buckets
constexpr uint32_t kLocalMaxPerGroup = kHybSubs / kLocalMinGrid;
constexpr int local_stage_bytes(int threads, int kpt, bool compact) { return threads * kpt * (compact ? 6 : 8); }
static_assert(local_stage_bytes(kHybLocalThreads, kHybLocalKptFull, false) == local_stage_bytes(kHybLocalThreads, kHybLocalKptCompact, true),
              "both stagings of the 512-thread workgroup take the same LDS");

template <int THREADS, int KPT, bool COMPACT>
__global__ __launch_bounds__(THREADS, 4) void hybrid_local_sort_kernel(uint32_t* __restrict__ keys,
                                                                     uint32_t* __restrict__ vals,
                                                                     const HybridHeader* __restrict__ hdr,
                                                                     const HybridTables* __restrict__ tb,
                                                                     const uint32_t run_shift, const uint32_t run_bits1,
                                                                     const uint32_t run_bits2, const uint32_t run_subs,
                                                                     const uint32_t run_key_shift, const uint32_t adapt_low16,
                                                                     const uint32_t stream_out)
{
    constexpr int WAVES = THREADS / 64;
    constexpr int RADIX = 256;
    constexpr uint32_t CAP = THREADS * KPT;
    static_assert(THREADS >= RADIX, "one thread per digit in the scan");
    // 60 KiB with 512 threads: two workgroups fit a compute unit's 160 KiB (120 KiB with 1024: one).
    // Full: (key, payload) per slot.  Compact: CAP payloads, then CAP 16-bit keys.
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[local_stage_bytes(THREADS, KPT, COMPACT)];
    __shared__ uint32_t s_hist[2][WAVES][RADIX];
    __shared__ uint32_t s_wsum[2][4];
    __shared__ uint32_t s_span[kLocalMaxPerGroup];  // adaptive groups: the low key bits in which a group's keys can differ
    __shared__ uint32_t s_first[kLocalMaxPerGroup], s_count[kLocalMaxPerGroup];
    // The adaptive schedule's groups: every entry of the table may be one, each lies in one 16-bit
    // cell and is sorted on the two 8-bit digits below adapt_low16.  All of these are uniform.
    const uint32_t mode = hybrid_mode(*hdr);
    const bool adaptive = mode == kHybAdapt && adapt_low16 != 0u;
    if (mode != kHybRun && !adaptive) return;
    const uint32_t shift = adaptive ? adapt_low16 - 16u : run_shift;
    const uint32_t bits1 = adaptive ? 8u : run_bits1, bits2 = adaptive ? 8u : run_bits2;
    const uint32_t key_shift = adaptive ? adapt_low16 : run_key_shift;
    const uint32_t num_subs = adaptive ? hdr->adapt_groups : run_subs;
    if (blockIdx.x >= num_subs) return;
    uint2* const s_pair = reinterpret_cast<uint2*>(s_raw);
    uint32_t* const s_val = reinterpret_cast<uint32_t*>(s_raw);
    unsigned short* const s_key16 = reinterpret_cast<unsigned short*>(s_raw + CAP * 4);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t shifts[2] = {shift, shift + bits1};
    const uint32_t widths[2] = {bits1, bits2};

    // element i of this thread in a sub-bucket of `count` pairs, `per` to a thread
    auto element = [&](uint32_t per, int i) __attribute__((always_inline)) { return wave * 64u * per + lane + i * 64u; };
    // KU: items per thread the code is unrolled for (several copies for sub-buckets of different
    // sizes, chosen per sub-bucket, cost 30 spilled registers at the joins: there is one copy).
    // Straight-line on purpose: every item is loaded, items past the sub-bucket's end from its
    // first element (one cache line for the whole wave; the value is never used).  Conditional
    // loads into the arrays made the compiler copy whole arrays at every join.
    auto load_bucket = [&](auto ku_tag, uint32_t (&key)[KPT], uint32_t (&val)[KPT], uint32_t first, uint32_t count) __attribute__((always_inline)) {
        constexpr int KU = decltype(ku_tag)::value;
        const uint32_t per = (count + THREADS - 1) / THREADS;
        const uint32_t* __restrict__ kin = keys + first;
        const uint32_t* __restrict__ vin = vals + first;
#pragma unroll
        for (int i = 0; i < KU; ++i) {
            const uint32_t e = element(per, i);
            const uint32_t ec = (static_cast<uint32_t>(i) < per && e < count) ? e : 0u;
#ifdef MRS_LOCAL_NT_LOAD  // A/B builds
            key[i] = __builtin_nontemporal_load(kin + ec);
            val[i] = __builtin_nontemporal_load(vin + ec);
#else
            key[i] = kin[ec];  // nontemporal loads measured 7 % slower
            val[i] = vin[ec];
#endif
        }
    };
    // sorts the `cnt` pairs held in (key, val) into the stage; returns the key bits every key of the
    // sub-bucket has (COMPACT), which the write-out puts back
    auto sort_in_stage = [&](auto ku_tag, uint32_t (&key)[KPT], uint32_t (&val)[KPT], const uint32_t cnt, const uint32_t span) __attribute__((always_inline)) -> uint32_t {
        constexpr int KU = decltype(ku_tag)::value;
        const uint32_t ku = (cnt + THREADS - 1) / THREADS;
        // item 0 of thread 0 always exists
        uint32_t high = 0;
        if constexpr (COMPACT) high = __builtin_amdgcn_readfirstlane(key[0]) >> key_shift << key_shift;
        PackedPos<KU> pos;  // ranks and slots are below 2^14: two to a register
        // A group of the adaptive schedule spans whatever key range its fine cells do, often far less
        // than 16 bits; counted on two fixed 8-bit digits the upper one would take few values, and the
        // lanes of a wave would queue on few counters.  The bits in which the group's keys differ are
        // divided between the two digits instead (the bits above them are the same for all keys):
        // `span` low bits, from the counting kernel.
        uint32_t dshift[2] = {shifts[0], shifts[1]}, dwidth[2] = {widths[0], widths[1]};
        if (adaptive) {  // uniform; span: what the counting kernel found (uniform too)
            dwidth[0] = span > 1u ? (span + 1u) / 2u : 1u;
            dwidth[1] = span > 1u ? span - dwidth[0] : 1u;
            dshift[1] = dshift[0] + dwidth[0];
        }
        // The counters are zero: cleared before the first sub-bucket and then, each set, right
        // after its last use below; between that and the next use lie the barriers of the other
        // digit's phase.  The stage is not written before three barriers from here, behind which
        // every wave has left the previous write-out.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t sh = dshift[h], wd = dwidth[h];
            if (h == 1) {
                // the order pass 1 left in LDS, read back slice by slice
#pragma unroll
                for (int i = 0; i < KU; ++i) {
                    const uint32_t e = element(ku, i);
                    const uint32_t ec = (static_cast<uint32_t>(i) < ku && e < cnt) ? e : 0u;  // straight-line, as the loads
                    if constexpr (COMPACT) {
                        key[i] = s_key16[ec];
                        val[i] = s_val[ec];
                    } else {
                        const uint2 kv = s_pair[ec];
                        key[i] = kv.x;
                        val[i] = kv.y;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < KU; ++i) {
                uint32_t r = 0;
#ifdef MRS_LOCAL_RANK_MATCH  // A/B builds: ballots instead of returning atomics
                {
                    const bool mine = static_cast<uint32_t>(i) < ku && element(ku, i) < cnt;
                    const uint32_t d = __builtin_amdgcn_ubfe(key[i], sh, wd);
                    const unsigned long long valid = __ballot(mine);
                    if (valid != 0ull) {  // uniform
                        uint32_t mlo, mhi;
                        match_digit<8>(d, valid, &mlo, &mhi);
                        const uint32_t below = lanes_below(mlo, mhi);
                        const uint32_t prev = s_hist[h][wave][d];
                        if (mine && below == 0u) s_hist[h][wave][d] = prev + __popc(mlo) + __popc(mhi);
                        r = prev + below;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
#else
                if (static_cast<uint32_t>(i) < ku && element(ku, i) < cnt)
                    r = atomicAdd(&s_hist[h][wave][__builtin_amdgcn_ubfe(key[i], sh, wd)], 1u);
#endif
                pos.set_fresh(i, r);
            }
            __syncthreads();  // every key is counted (and, h == 1, every slot of the stage has been read)
            uint32_t tot = 0, incl = 0;
            uint32_t c[WAVES];
            if (tid < RADIX) {
#pragma unroll
                for (int w = 0; w < WAVES; ++w) c[w] = s_hist[h][w][tid];
#pragma unroll
                for (int w = 0; w < WAVES; ++w) {
                    const uint32_t x = c[w];
                    c[w] = tot;
                    tot += x;
                }
                incl = wave_inclusive_sum(tot);  // waves 0 to 3, all lanes
                if (lane == 63) s_wsum[h][wave] = incl;
            }
            __syncthreads();  // the per-wave parts wait in registers for the digit's offset
            if (tid < RADIX) {
                uint32_t off = incl - tot;
                for (uint32_t w = 0; w < 4; ++w)
                    if (w < wave) off += s_wsum[h][w];
#pragma unroll
                for (int w = 0; w < WAVES; ++w) s_hist[h][w][tid] = c[w] + off;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < KU; ++i) {
                if (static_cast<uint32_t>(i) < ku && element(ku, i) < cnt) {
                    const uint32_t slot = pos.get(i) + s_hist[h][wave][__builtin_amdgcn_ubfe(key[i], sh, wd)];
                    if constexpr (COMPACT) {
                        s_key16[slot] = static_cast<unsigned short>(key[i]);
                        s_val[slot] = val[i];
                    } else {
                        s_pair[slot] = make_uint2(key[i], val[i]);
                    }
                }
            }
            __syncthreads();
            for (int j = tid; j < WAVES * RADIX; j += THREADS) (&s_hist[h][0][0])[j] = 0;  // for the next sub-bucket
        }
        return high;
    };
    // the sorted stage over [start, start + cnt), two slots per lane: one 8-byte store each for keys
    // and payload (a sub-bucket starts at any 4-byte boundary; gfx950 serves the wider store there).
    // Four slots and 16-byte stores measured 2 % slower.  stream_out (uniform; the host sets it for
    // inputs well beyond what the last-level cache holds): the stores are nontemporal -- nobody reads
    // the sorted sub-bucket soon, and as ordinary stores they push the next sub-buckets' pairs out of
    // the caches: 0.918 against 0.97-0.997 ms at 2^28 pairs, 1.85 against 1.95 at 2^29; at 2^26 the
    // ordinary stores are the faster ones, 0.305 against 0.327 ms.
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    auto store2 = [&](uint32_t* p, uint32_t x, uint32_t y) __attribute__((always_inline)) {
        if (stream_out) {
            const u32x2 v = {x, y};
            __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
        } else {
            *reinterpret_cast<uint2*>(p) = make_uint2(x, y);
        }
    };
    auto store1 = [&](uint32_t* p, uint32_t x) __attribute__((always_inline)) {
        if (stream_out) __builtin_nontemporal_store(x, p);
        else *p = x;
    };
    auto write_out = [&](const uint32_t start, const uint32_t cnt, const uint32_t high) __attribute__((always_inline)) {
        uint32_t* __restrict__ kout = keys + start;
        uint32_t* __restrict__ vout = vals + start;
        for (uint32_t s2 = 2u * tid; s2 < cnt; s2 += 2u * THREADS) {
            if constexpr (COMPACT) {
                const uint32_t kk = *reinterpret_cast<const uint32_t*>(&s_key16[s2]);  // s2 is even: two keys in one word
                if (s2 + 1 < cnt) {
                    const uint2 vv = *reinterpret_cast<const uint2*>(&s_val[s2]);
                    store2(kout + s2, high | (kk & 0xFFFFu), high | (kk >> 16));
                    store2(vout + s2, vv.x, vv.y);
                } else {
                    store1(kout + s2, high | (kk & 0xFFFFu));
                    store1(vout + s2, s_val[s2]);
                }
            } else {
                if (s2 + 1 < cnt) {
                    const uint4 q = *reinterpret_cast<const uint4*>(&s_pair[s2]);
                    store2(kout + s2, q.x, q.z);
                    store2(vout + s2, q.y, q.w);
                } else {
                    const uint2 kv = s_pair[s2];
                    store1(kout + s2, kv.x);
                    store1(vout + s2, kv.y);
                }
            }
        }
    };

    // This workgroup's sub-buckets are blockIdx.x, + gridDim.x, ...: their bounds go to LDS once
    // (a scalar load inside the loop would be waited for at the next LDS access); one that does not
    // fit is entered as empty (the segmented passes sort it).  The pairs of the next sub-bucket are
    // requested as soon as the second scatter has emptied the registers, so they travel while the
    // current one is written out (two register sets, with the request before the sort, measured the
    // same time and cost 60 to 80 registers more).
    const uint32_t mine = (num_subs - 1u - blockIdx.x) / gridDim.x + 1u;  // blockIdx.x < num_subs <= kHybSubs
    for (uint32_t j = tid; j < mine; j += THREADS) {
        const uint32_t which = blockIdx.x + j * gridDim.x;
        // (the adaptive schedule's groups: from the list of all of them; the table's entries 256 b + g
        // are used for about half of the g, which would leave half of the workgroups without work)
        const uint32_t first = adaptive ? tb->group_first[which] : tb->sub_start[which];
        // (pairs in the low 27 bits, the span above them)
        const uint32_t packed_pairs = adaptive ? tb->group_pairs[which] : 0u;
        const uint32_t count = adaptive ? packed_pairs & 0x7FFFFFFu : tb->sub_start[which + 1] - first;
        s_span[j] = packed_pairs >> 27;
        s_first[j] = first;
        s_count[j] = count <= CAP ? count : 0u;
    }
    for (int j = tid; j < 2 * WAVES * RADIX; j += THREADS) (&s_hist[0][0][0])[j] = 0;
    __syncthreads();
    const std::integral_constant<int, KPT> all_items{};
    uint32_t key[KPT], val[KPT];
    if (s_count[0] != 0u) load_bucket(all_items, key, val, s_first[0], s_count[0]);
    for (uint32_t j = 0; j < mine; ++j) {
        const uint32_t first = s_first[j], count = s_count[j];
        uint32_t high = 0;
        if (count != 0u) high = sort_in_stage(all_items, key, val, count, __builtin_amdgcn_readfirstlane(s_span[j]));
        if (j + 1 < mine && s_count[j + 1] != 0u) load_bucket(all_items, key, val, s_first[j + 1], s_count[j + 1]);
        if (count != 0u) write_out(first, count, high);
    }
}

// ---- host side --------------------------------------------------------------------------------
size_t hybrid_tables_bytes() { return (sizeof(HybridTables) + 255) / 256 * 256; }
size_t hybrid_packed_bytes(int grid) { return static_cast<size_t>(grid) * kHybPackedWords * sizeof(uint32_t); }

int hybrid_histogram_grid(int num_cus, uint64_t n)
{
    // at least 16 vectors per thread: every workgroup ends with a 128 KiB table to write
    uint64_t g = (n / 4 + 1024 * 16 - 1) / (1024 * 16);
    if (g > static_cast<uint64_t>(num_cus)) g = num_cus;
    return g == 0 ? 1 : static_cast<int>(g);
}

int hybrid_local_sort_grid(int num_cus, bool big)
{
    // 78 KiB of LDS each: two per compute unit (154 KiB: one)
    const int g = big ? num_cus : num_cus * 2;
    return g < static_cast<int>(kLocalMinGrid) ? static_cast<int>(kLocalMinGrid) : g;
}

int hybrid_launch_histogram(const uint32_t* keys, uint32_t n, uint32_t shift, uint32_t bits, bool windowed, uint32_t window_base,
                            uint32_t begin_bit, uint32_t end_bit, int grid, uint32_t* packed, HybridCleared* hc,
                            uint32_t* low_hist, hipStream_t s)
{
    if (shift > begin_bit + 16u) return fail(MRS_ERR_UNSUPPORTED, "the exact table must start within 16 bits of begin_bit");
    if (bits < 14u || bits > 16u) return fail(MRS_ERR_INVALID_ARGUMENT, "the exact table has 2^14 to 2^16 entries");
    const ProfilerRange range("Histogram calculation");  // src/main.rs:597
    const uint32_t vectors = (reinterpret_cast<uintptr_t>(keys) & 15u) ? 0u : n / 4u;
    hipLaunchKernelGGL(hybrid_histogram_kernel, dim3(grid), dim3(1024), 0, s, keys, n, vectors, shift, bits, packed, hc,
                       window_base, windowed ? 1u : 0u, end_bit, begin_bit, low_hist);
    MRS_CHECK_LAUNCH("hybrid_histogram_kernel");
    return MRS_OK;
}

int hybrid_launch_scan(const uint32_t* packed, int grid, HybridTables* tables, HybridCleared* hc, uint32_t n,
                       uint32_t tile_items, uint32_t force_mode, uint32_t sub_bits, bool compact_ok, uint32_t cap,
                       const AdaptArgs& adapt, uint32_t* ghist, uint32_t begin_bit, uint32_t end_bit, uint32_t shift,
                       uint32_t window_base, hipStream_t s)
{
    if (cap == 0 || cap > (compact_ok ? kHybLocalCapBig : kHybLocalCapFull))
        return fail(MRS_ERR_INVALID_ARGUMENT, "a workgroup of the local sort holds at most %u pairs", compact_ok ? kHybLocalCapBig : kHybLocalCapFull);
    if (adapt.offered && (adapt.msd_bits < kAdaptCellBits || adapt.msd_bits > 16u || adapt.cell_shift < adapt.low16_shift ||
                          adapt.cell_shift - adapt.low16_shift > 4u || adapt.low16_shift < 16u))
        return fail(MRS_ERR_UNSUPPORTED, "the adaptive schedule needs 28 to 32 sorted key bits");
    if (sub_bits < 6 || sub_bits > 8) return fail(MRS_ERR_UNSUPPORTED, "a bucket has 64 to 256 sub-buckets");
    if (shift > begin_bit + 16u || end_bit <= begin_bit + 24u || end_bit > begin_bit + 32u)
        return fail(MRS_ERR_UNSUPPORTED, "the ordinary passes are four digits of 8 bits, the exact table starts within the lowest two");
    OrdinaryDigits od{};
    od.ghist = ghist;
    od.shift2 = begin_bit + 16u - shift;
    od.mask2 = 255u;
    od.shift3 = begin_bit + 24u - shift;
    od.mask3 = (1u << (end_bit - begin_bit - 24u)) - 1u;
    od.window_base = window_base;
    const ProfilerRange range("Small Prefix Sum");  // src/main.rs:625
    // (only the 2^msd_bits entries the split uses: a workgroup adds up 128 of them)
    if (adapt.msd_bits < 14u || adapt.msd_bits > 16u) return fail(MRS_ERR_INVALID_ARGUMENT, "the exact table has 2^14 to 2^16 entries");
    hipLaunchKernelGGL(hybrid_reduce_kernel, dim3((1u << adapt.msd_bits) / 128u), dim3(256), 0, s, packed, grid, hc->spill, tables->counts);
    MRS_CHECK_LAUNCH("hybrid_reduce_kernel");
    hipLaunchKernelGGL(hybrid_scan_kernel, dim3(1), dim3(1024), 0, s, tables, hc, n, tile_items, force_mode, sub_bits,
                       compact_ok ? 1u : 0u, cap, adapt, od);
    MRS_CHECK_LAUNCH("hybrid_scan_kernel");
    return MRS_OK;
}

int hybrid_launch_adapt_count(const uint32_t* keys, HybridTables* tables, HybridCleared* hc, uint32_t n, uint32_t tile_items,
                              uint32_t cap, const AdaptArgs& adapt, hipStream_t s)
{
    const ProfilerRange range("Histogram calculation");  // src/main.rs:597
    hipLaunchKernelGGL(hybrid_adapt_count_kernel, dim3(kHybBuckets), dim3(1024), 0, s, keys, tables, hc, n, tile_items, cap, adapt);
    MRS_CHECK_LAUNCH("hybrid_adapt_count_kernel");
    return MRS_OK;
}

template <bool SEG, bool DUAL, bool PERSISTENT, int MAP, int ROLE, bool ADAPT = false>
static int launch_pass_shape(int threads, int kpt, const PassArgs& run, const SegmentTable& seg, const PassArgs& other,
                             const HybridHeader* hdr, uint32_t grid, hipStream_t s, const PassArgs& adapt = PassArgs{},
                             const HybridTables* tables = nullptr)
{
    if (other.wide_status) {
        if constexpr (DUAL && PERSISTENT) {
            if (threads != 1024 || kpt != 32) return fail(MRS_ERR_UNSUPPORTED, "plans of 2^30 pairs use tiles of 1024 x 32 keys");
            hipLaunchKernelGGL((hybrid_pass_kernel<SEG, DUAL, PERSISTENT, MAP, 1024, 32, ROLE, ADAPT, true>), dim3(grid), dim3(1024), 0, s, run,
                               seg, other, hdr, adapt, tables);
            MRS_CHECK_LAUNCH("hybrid_pass_kernel");
            return MRS_OK;
        } else {
            return fail(MRS_ERR_UNSUPPORTED, "only persistent launches carry an ordinary pass with 64-bit status words");
        }
    }
    if (threads == 1024 && kpt == 32)
        hipLaunchKernelGGL((hybrid_pass_kernel<SEG, DUAL, PERSISTENT, MAP, 1024, 32, ROLE, ADAPT>), dim3(grid), dim3(1024), 0, s, run, seg,
                           other, hdr, adapt, tables);
    else if (threads == 512 && kpt == 8)
        hipLaunchKernelGGL((hybrid_pass_kernel<SEG, DUAL, PERSISTENT, MAP, 512, 8, ROLE, ADAPT>), dim3(grid), dim3(512), 0, s, run, seg,
                           other, hdr, adapt, tables);
    else
        return fail(MRS_ERR_UNSUPPORTED, "no gated pass kernel for %d threads x %d keys per thread", threads, kpt);
    MRS_CHECK_LAUNCH("hybrid_pass_kernel");
    return MRS_OK;
}

int hybrid_launch_pass(int role, const PassArgs* run, const SegmentTable* seg, const PassArgs* other, bool persistent, bool windowed,
                       int threads, int kpt, const HybridHeader* hdr, uint32_t grid, hipStream_t s, const PassArgs* adapt,
                       const HybridTables* tables)
{
    if (!run && !other) return fail(MRS_ERR_INVALID_ARGUMENT, "a gated pass needs at least one description");
    if ((run && run->wide_status) || (adapt && adapt->wide_status))
        return fail(MRS_ERR_UNSUPPORTED, "the passes of the three-move schedule use 32-bit tile status words");
    if (adapt && (!tables || !persistent || windowed || role > kPassB))
        return fail(MRS_ERR_UNSUPPORTED, "the adaptive schedule's passes are persistent launches of pass A and pass B");
    const PassArgs& ad = adapt ? *adapt : PassArgs{};
    const ProfilerRange range("Indexing");  // src/main.rs:658
    if (grid == 0) grid = 1;
    // a `run` that is not given has no tiles, so it does nothing in its mode
    const PassArgs none{};
    const PassArgs& r = run ? *run : none;
    const PassArgs& o = other ? *other : none;
    const bool segmented = run && seg;
    const SegmentTable table = segmented ? *seg : SegmentTable{};
    switch (role) {
        case kPassA:  // over the whole array
            if (segmented) break;
            if (persistent) {
                // the ordinary passes start one launch later (the adaptive schedule may hand over to them after its pass A)
                if (windowed || other) break;
                return launch_pass_shape<false, false, true, kMapNone, kPassA, true>(threads, kpt, r, table, o, hdr, grid, s, ad, tables);
            }
            if (other) break;
            return windowed ? launch_pass_shape<false, false, false, kMapWindow, kPassA>(threads, kpt, r, table, o, hdr, grid, s)
                            : launch_pass_shape<false, false, false, kMapNone, kPassA>(threads, kpt, r, table, o, hdr, grid, s);
        case kPassB:  // bucket by bucket
            if (!segmented) break;
            if (persistent) {
                if (windowed || !other) break;
                return launch_pass_shape<true, true, true, kMapNone, kPassB, true>(threads, kpt, r, table, o, hdr, grid, s, ad, tables);
            }
            if (other) break;
            return windowed ? launch_pass_shape<true, false, false, kMapWindow, kPassB>(threads, kpt, r, table, o, hdr, grid, s)
                            : launch_pass_shape<true, false, false, kMapNone, kPassB>(threads, kpt, r, table, o, hdr, grid, s);
        case kPassOversized:  // plain digits, persistent, always beside an ordinary pass
            if (!segmented || !persistent || windowed || !other) break;
            return launch_pass_shape<true, true, true, kMapNone, kPassOversized>(threads, kpt, r, table, o, hdr, grid, s);
        case kPassOrdinary:  // an ordinary pass with a launch of its own
            if (run || !persistent || windowed) break;
            return launch_pass_shape<false, true, true, kMapNone, kPassOrdinary>(threads, kpt, r, table, o, hdr, grid, s);
        default: break;
    }
    return fail(MRS_ERR_UNSUPPORTED, "no gated pass kernel for this combination (role %d)", role);
}

int hybrid_launch_big_histogram(const uint32_t* keys, HybridTables* tables, HybridCleared* hc, uint32_t tile_items,
                                uint32_t shift, uint32_t bits1, uint32_t bits2, const PassArgs* other, int threads, int kpt,
                                int grid, hipStream_t s)
{
    if (bits1 == 0 || bits1 > 8 || bits2 == 0 || bits2 > 8)
        return fail(MRS_ERR_UNSUPPORTED, "oversized sub-buckets are sorted on two digits of 1 to 8 bits");
    const ProfilerRange range("Histogram calculation");  // src/main.rs:597
    const BigHistogramArgs g{keys, tables, hc, tile_items, shift, bits1, bits2};
    const PassArgs o = other ? *other : PassArgs{};
    const dim3 blocks(grid < 1 ? 1 : grid);
    if (threads == 1024 && kpt == 32) {
        if (o.wide_status)
            hipLaunchKernelGGL((hybrid_big_histogram_kernel<1024, 32, true>), blocks, dim3(1024), 0, s, g, o, &hc->hdr);
        else
            hipLaunchKernelGGL((hybrid_big_histogram_kernel<1024, 32, false>), blocks, dim3(1024), 0, s, g, o, &hc->hdr);
    } else if (threads == 512 && kpt == 8 && !o.wide_status) {
        hipLaunchKernelGGL((hybrid_big_histogram_kernel<512, 8, false>), blocks, dim3(512), 0, s, g, o, &hc->hdr);
    } else {
        return fail(MRS_ERR_UNSUPPORTED, "no kernel for %d threads x %d keys per thread", threads, kpt);
    }
    MRS_CHECK_LAUNCH("hybrid_big_histogram_kernel");
    return MRS_OK;
}

int hybrid_launch_local_sort(uint32_t* keys, uint32_t* vals, const HybridHeader* hdr, const HybridTables* tables,
                             uint32_t shift, uint32_t bits1, uint32_t bits2, uint32_t num_subs, uint32_t key_shift,
                             bool compact_ok, uint32_t adapt_low16, int grid, hipStream_t s, bool big, bool stream_out)
{
    const uint32_t so = stream_out ? 1u : 0u;
    if (big && (!compact_ok || adapt_low16 != 0u))
        return fail(MRS_ERR_UNSUPPORTED, "the 1024-thread local sort stages 6 bytes per pair and serves the fixed split only");
    if (adapt_low16 != 0u && (adapt_low16 < 16u || (compact_ok && adapt_low16 != 16u)))
        return fail(MRS_ERR_UNSUPPORTED, "the adaptive schedule's groups are sorted on 16 bits, staged in 6 bytes only if those are the low 16");
    if (num_subs == 0 || num_subs > kHybSubs) return fail(MRS_ERR_UNSUPPORTED, "at most 65536 sub-buckets");
    if (bits1 == 0 || bits1 > 8 || bits2 == 0 || bits2 > 8)
        return fail(MRS_ERR_UNSUPPORTED, "the local sort takes two digits of 1 to 8 bits");
    if (compact_ok && key_shift > 16) return fail(MRS_ERR_INVALID_ARGUMENT, "6-byte staging needs the sub-bucket's bits to start at bit 16 or below");
    const ProfilerRange range("Local sort");
    if (big)
        hipLaunchKernelGGL((hybrid_local_sort_kernel<kHybLocalThreadsBig, kHybLocalKptCompact, true>), dim3(grid), dim3(kHybLocalThreadsBig),
                           0, s, keys, vals, hdr, tables, shift, bits1, bits2, num_subs, key_shift, adapt_low16, so);
    else if (compact_ok)
        hipLaunchKernelGGL((hybrid_local_sort_kernel<kHybLocalThreads, kHybLocalKptCompact, true>), dim3(grid), dim3(kHybLocalThreads),
                           0, s, keys, vals, hdr, tables, shift, bits1, bits2, num_subs, key_shift, adapt_low16, so);
    else
        hipLaunchKernelGGL((hybrid_local_sort_kernel<kHybLocalThreads, kHybLocalKptFull, false>), dim3(grid), dim3(kHybLocalThreads),
                           0, s, keys, vals, hdr, tables, shift, bits1, bits2, num_subs, key_shift, adapt_low16, so);
    MRS_CHECK_LAUNCH("hybrid_local_sort_kernel");
    return MRS_OK;
}

}  // namespace mrs
