// Kernels of the most-significant-digit-first schedule (see mrs_hybrid.hpp) for gfx950.
#include "mrs_hybrid.hpp"

#include "mrs_radix_pass.hpp"

namespace mrs {

// ---- 16-bit histogram ---------------------------------------------------------------------
// One LDS atomic per key into 65536 counters of 16 bits, two to a word (128 KiB: one workgroup
// of 1024 threads per compute unit), 16-byte nontemporal loads, four in flight per thread.  A
// workgroup writes its table to `packed`; hybrid_reduce_kernel adds the tables up.  A counter that
// wraps (65536 keys of one workgroup in one sub-bucket) loses 65535 or 65536 from the sum of the
// fields, never gains: a workgroup whose fields do not add up to the keys it counted raises
// hdr->overflow, and the decision kernel then chooses the other schedule, which such an input
// needs anyway.  The atomics return the old word: a lane that finds more than `cap` keys already
// counted in its field knows that the sub-bucket is too large whatever the other workgroups hold,
// raises the flag and lets the workgroup stop (equal or few distinct keys would otherwise queue
// 64 lanes on one word for the whole input: 0.9 ms instead of 0.17 at 2^28).
__global__ __launch_bounds__(1024) void hybrid_histogram_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                 uint32_t vectors, uint32_t shift, uint32_t bits,
                                                                 uint32_t* __restrict__ packed, HybridHeader* hdr,
                                                                 uint32_t cap, uint32_t window_base, uint32_t windowed)
{
    constexpr int THREADS = 1024, UNROLL = 4;
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[kHybPackedWords];
    __shared__ int s_diff;
    __shared__ volatile uint32_t s_stop;  // volatile: polled in the loops with no barrier in between
    const uint32_t tid = threadIdx.x;
    {
        uint4* z = reinterpret_cast<uint4*>(s_cnt);
        for (uint32_t j = tid; j < kHybPackedWords / 4; j += THREADS) z[j] = make_uint4(0, 0, 0, 0);
    }
    if (tid == 0) {
        s_diff = 0;
        s_stop = 0;
    }
    __syncthreads();
    uint32_t mine = 0;
    uint32_t most = 0;  // largest count this lane has seen in a field before its own increment
    auto count = [&](uint32_t k) __attribute__((always_inline)) {
        uint32_t b = __builtin_amdgcn_ubfe(k, shift, bits);  // bits <= 16: counters beyond 2^bits stay zero
        if (windowed) {  // uniform
            // sub-bucket = (key >> shift) - window_base; a key outside the window ends the count
            // (the other schedule sorts any keys)
            b = (k >> shift) - window_base;
            if (b >> bits) {
                most = cap;
                b = 0;
            }
        }
        const uint32_t half = (b & 1u) * 16u;
        const uint32_t old = atomicAdd(&s_cnt[b >> 1], 1u << half);
        const uint32_t seen = (old >> half) & 0xFFFFu;
        most = seen > most ? seen : most;
    };
    // uniform for the wave; the other waves learn of it from LDS at their next step
    auto too_many = [&]() {
        if (__builtin_amdgcn_uicmp(most, cap, 35 /* >= */) != 0ull) s_stop = 1;
        return s_stop != 0;
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
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            count(x[u].x); count(x[u].y); count(x[u].z); count(x[u].w);
        }
        mine += 4 * UNROLL;
        if (too_many()) break;
    }
    for (; v < vectors && s_stop == 0; v += stride) {
        const u32x4 x = __builtin_nontemporal_load(vkeys + v);
        count(x.x); count(x.y); count(x.z); count(x.w);
        mine += 4;
    }
    // what the vectors do not cover: everything if the array is not 16-byte aligned, else < 4 keys
    for (unsigned long long i = static_cast<unsigned long long>(vectors) * 4 + blockIdx.x * THREADS + tid; i < n; i += stride) {
        count(keys[i]);
        mine += 1;
    }
    if (most >= cap) s_stop = 1;  // seen in the last steps, which do not look
    __syncthreads();
    if (s_stop != 0) {  // uniform: the counts are incomplete and nobody will read them
        if (tid == 0) atomicOr(&hdr->overflow, 1u);
        return;
    }
    uint32_t fields = 0;
    {
        const uint4* c = reinterpret_cast<const uint4*>(s_cnt);
        uint4* out = reinterpret_cast<uint4*>(packed + static_cast<size_t>(blockIdx.x) * kHybPackedWords);
        for (uint32_t j = tid; j < kHybPackedWords / 4; j += THREADS) {
            const uint4 w = c[j];
            out[j] = w;
            fields += (w.x & 0xFFFFu) + (w.x >> 16) + (w.y & 0xFFFFu) + (w.y >> 16) + (w.z & 0xFFFFu) + (w.z >> 16) +
                      (w.w & 0xFFFFu) + (w.w >> 16);
        }
    }
    int diff = static_cast<int>(mine) - static_cast<int>(fields);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) diff += __shfl_down(diff, o);
    if ((tid & 63u) == 0 && diff != 0) atomicAdd(&s_diff, diff);
    __syncthreads();
    if (tid == 0 && s_diff != 0) atomicOr(&hdr->overflow, 1u);
}

// counts[i] = sum over the workgroups' tables of field i.  A workgroup owns 64 consecutive packed
// words; its four waves each add up a quarter of the tables (256 bytes per table and wave), and
// the quarters meet in LDS.
__global__ __launch_bounds__(256) void hybrid_reduce_kernel(const uint32_t* __restrict__ packed, int grid,
                                                             uint32_t* __restrict__ counts)
{
    __shared__ uint32_t s_part[4][64][2];
    const uint32_t lane = threadIdx.x & 63u, part = threadIdx.x >> 6;
    const uint32_t j = blockIdx.x * 64u + lane;  // < kHybPackedWords: the grid is exact
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
        reinterpret_cast<uint2*>(counts)[j] = make_uint2(lo, hi);
    }
}

// One workgroup: exclusive scan of the 65536 counts (thread t owns entries 64t .. 64t+63; a bucket
// has 2^sub_bits >= 64 sub-buckets, so it starts with a thread), bucket starts, the tile table of
// pass B and the decision.  Buckets and sub-buckets the split does not use are empty and start at n.
__global__ __launch_bounds__(1024) void hybrid_scan_kernel(HybridTables* tb, HybridHeader* hdr, uint32_t n,
                                                            uint32_t tile_items, uint32_t cap, uint32_t force_mode,
                                                            uint32_t sub_bits)
{
    __shared__ uint32_t s_wsum[16];
    __shared__ uint32_t s_bstart[kHybBuckets + 1];
    __shared__ uint32_t s_tsum[4];
    __shared__ uint32_t s_max;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t c[64];
    {
        const uint4* c4 = reinterpret_cast<const uint4*>(tb->counts) + tid * 16;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint4 x = c4[q];
            c[4 * q] = x.x; c[4 * q + 1] = x.y; c[4 * q + 2] = x.z; c[4 * q + 3] = x.w;
        }
    }
    uint32_t sum = 0, mx = 0;
#pragma unroll
    for (int q = 0; q < 64; ++q) {
        sum += c[q];
        mx = c[q] > mx ? c[q] : mx;
    }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t y = __shfl_down(mx, o);
        mx = y > mx ? y : mx;
    }
    if (lane == 63) s_wsum[wave] = incl;
    if (tid == 0) s_max = 0;
    __syncthreads();
    if (lane == 0) atomicMax(&s_max, mx);
    uint32_t run = incl - sum;
    for (uint32_t w = 0; w < 16; ++w)
        if (w < wave) run += s_wsum[w];
    // thread t starts bucket t >> (sub_bits - 6) if it starts one; the threads past the last bucket
    // the split uses hold zeros, so the unused buckets start at n
    const uint32_t tpb = sub_bits - 6u;  // log2(threads per bucket)
    if ((tid & ((1u << tpb) - 1u)) == 0 && (tid >> tpb) <= kHybBuckets) s_bstart[tid >> tpb] = run;
    if (tid == 1023 && (1024u >> tpb) <= kHybBuckets) s_bstart[1024u >> tpb] = run + sum;  // sub_bits == 8: no thread starts "bucket 256"
    {
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
        if (tid == 1023) tb->sub_start[kHybSubs] = run;
    }
    __syncthreads();
    uint32_t tiles = 0, tincl = 0;
    if (tid < kHybBuckets) {
        const uint32_t bs = s_bstart[tid];
        const uint32_t cnt = s_bstart[tid + 1] - bs;
        tiles = (cnt + tile_items - 1) / tile_items;
        tb->bucket_start[tid] = bs;
        tincl = tiles;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(tincl, o);
            if (lane >= static_cast<uint32_t>(o)) tincl += y;
        }
        if (lane == 63) s_tsum[wave] = tincl;
    }
    if (tid == kHybBuckets) tb->bucket_start[kHybBuckets] = s_bstart[kHybBuckets];
    __syncthreads();
    if (tid < kHybBuckets) {
        uint32_t off = tincl - tiles;
        for (uint32_t w = 0; w < 4; ++w)
            if (w < wave) off += s_tsum[w];
        tb->first_tile[tid] = off;
        if (tid == kHybBuckets - 1) {
            const uint32_t total_tiles = off + tiles;
            // counts that do not add up to n (a wrapped counter) are never acted on
            const bool ok = s_max <= cap && hdr->overflow == 0 && s_bstart[kHybBuckets] == n && force_mode != kHybFallback;
            tb->first_tile[kHybBuckets] = total_tiles;
            hdr->num_tiles_b = ok ? total_tiles : 0u;
            hdr->max_sub = hdr->overflow ? 0xFFFFFFFFu : s_max;  // a histogram that stopped early has no counts to report
            hdr->mode = ok ? kHybRun : kHybFallback;
        }
    }
}

// ---- chained-scan passes that the device switches on or off --------------------------------
// The digit pass of mrs_radix_pass.hpp behind a gate (hdr->mode == run_if_mode), over the whole
// array (pass A, or one of the plan's ordinary passes when the device chose those) or, SEG, over
// the buckets pass A made: tile t belongs to the bucket b with first_tile[b] <= t < first_tile[b+1],
// covers TILE elements from bucket_start[b] + (t - first_tile[b]) * TILE (fewer in the bucket's
// last tile), and its chained scan starts at first_tile[b] with the sub-bucket starts of b as
// digit bases.  PERSISTENT: the workgroup draws tiles until none is left (same time per pass as a
// workgroup per tile, 0.86-0.88 ms at 2^28, also when the next ticket is drawn during the write-out:
// the hand-over from tile to tile is not what bounds a pass; but a switched-off launch costs 0.005 ms
// instead of 0.1, and one that runs holds every compute unit until the pass ends, which a collective
// waiting on another stream feels).
template <bool SEG, bool PERSISTENT, int MAP, int THREADS, int KPT>
__global__ __launch_bounds__(THREADS, 1) void hybrid_pass_kernel(const PassArgs a, const HybridHeader* __restrict__ hdr,
                                                                  const HybridTables* __restrict__ tb,
                                                                  const uint32_t run_if_mode)
{
    using S = PassShared<8, THREADS, KPT, uint32_t, uint32_t>;
    __shared__ S sh;
    __shared__ uint32_t s_seg[4];
    uint32_t tid = threadIdx.x;
    const uint32_t mode = hdr->mode;  // uniform
    if (PERSISTENT && mode != run_if_mode) return;
    const uint32_t num_tiles = SEG ? hdr->num_tiles_b : a.num_tiles;
    uint32_t ft0 = 0, ft1 = 0, bs0 = 0, bs1 = 0;
    if constexpr (SEG) {
        if (tid < kHybBuckets) {
            ft0 = tb->first_tile[tid];
            ft1 = tb->first_tile[tid + 1];
            bs0 = tb->bucket_start[tid];
            bs1 = tb->bucket_start[tid + 1];
        }
    }
    for (;;) {
        if constexpr (PERSISTENT) {
            // nothing derived from the thread's number is to be carried over from tile to tile (the
            // registers are needed inside the tile: 19 spilled values and 9 % of the time otherwise)
            asm volatile("" : "+v"(tid));
        }
        // The gate word and the first ticket travel together: a workgroup of a launch that runs
        // does not wait for one after the other, and one of a launch that is switched off leaves
        // as soon as the word is there (having wasted a ticket of a counter nobody else reads).
        Ticket ticket{0, 0};
        if (tid == 0) ticket = ticket_issue(a);
        if (mode != run_if_mode) return;  // uniform
        if (tid == 0) {
            sh.tile = ticket_tile(a, ticket, num_tiles);
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
            if (tid < kHybBuckets && ft0 <= tile && tile < ft1) {  // exactly one thread: empty buckets have ft0 == ft1
                s_seg[0] = tid;
                s_seg[1] = ft0;
                s_seg[2] = bs0;
                s_seg[3] = bs1;
            }
            __syncthreads();
            const uint32_t seg = s_seg[0], first = s_seg[1], start = s_seg[2], end = s_seg[3];
            const uint32_t base = start + (tile - first) * static_cast<uint32_t>(S::TILE);
            const uint32_t remaining = end - base;
            PassArgs b = a;
            b.digit_base = tb->sub_start + seg * (a.mask + 1u);  // the bucket's sub-buckets, a.mask + 1 of them
            if (remaining >= static_cast<uint32_t>(S::TILE))
                process_tile<kModeOnesweep, kRankAtomic, MAP, true, uint32_t>(b, sh, tile, S::TILE, base, first, tid);
            else
                process_tile<kModeOnesweep, kRankAtomic, MAP, false, uint32_t>(b, sh, tile, remaining, base, first, tid);
        } else {
            const uint32_t remaining = a.n - tile * static_cast<uint32_t>(S::TILE);
            const unsigned long long base = static_cast<unsigned long long>(tile) * S::TILE;
            if (remaining >= static_cast<uint32_t>(S::TILE))
                process_tile<kModeOnesweep, kRankAtomic, MAP, true, uint32_t>(a, sh, tile, S::TILE, base, 0u, tid);
            else
                process_tile<kModeOnesweep, kRankAtomic, MAP, false, uint32_t>(a, sh, tile, remaining, base, 0u, tid);
        }
        if (!PERSISTENT) return;
        // A look-back that gave up has raised the plan's error flag: the output is lost, the ticket of
        // the next tile may never have been drawn, and every other workgroup gives up on that flag.
        if (sh.abort) return;  // uniform: read after the barrier behind which it was written
        __syncthreads();  // the tile's LDS image is dead
    }
}

// ---- local sort -----------------------------------------------------------------------------
// A workgroup takes sub-buckets blockIdx.x, blockIdx.x + gridDim.x, ...: at most THREADS * KPT
// pairs each (the decision kernel saw to that), loaded in the element order of the digit pass
// (wave w owns a contiguous slice, item i of lane l is element w*64*ku + i*64 + l, ku = items per
// thread of THIS sub-bucket), sorted by two stable counting passes through LDS on bits
// [shift, shift + bits1) and [shift + bits1, shift + bits1 + bits2), each at most 8 bits wide,
// ranked like the digit pass (returning LDS atomics on per-wave counters, lane order), and
// written back over the sub-bucket, which nobody else touches.
constexpr uint32_t kLocalMinGrid = 256;  // so that a workgroup has at most this many sub-buckets
constexpr uint32_t kLocalMaxPerGroup = kHybSubs / kLocalMinGrid;

template <int THREADS, int KPT>
__global__ __launch_bounds__(THREADS, 4) void hybrid_local_sort_kernel(uint32_t* __restrict__ keys,
                                                                     uint32_t* __restrict__ vals,
                                                                     const HybridHeader* __restrict__ hdr,
                                                                     const HybridTables* __restrict__ tb,
                                                                     const uint32_t shift, const uint32_t bits1,
                                                                     const uint32_t bits2, const uint32_t num_subs)
{
    constexpr int WAVES = THREADS / 64;
    constexpr int RADIX = 256;
    static_assert(THREADS >= RADIX, "one thread per digit in the scan");
    __shared__ __attribute__((aligned(16))) uint2 s_stage[THREADS * KPT];  // 60 KiB at 512 x 15: two workgroups fit a compute unit's 160 KiB
    __shared__ uint32_t s_hist[2][WAVES][RADIX];
    __shared__ uint32_t s_wsum[2][4];
    __shared__ uint32_t s_first[kLocalMaxPerGroup], s_count[kLocalMaxPerGroup];
    if (hdr->mode != kHybRun || blockIdx.x >= num_subs) return;  // uniform
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t shifts[2] = {shift, shift + bits1};
    const uint32_t widths[2] = {bits1, bits2};

    // element i of this thread in a sub-bucket of `count` pairs, `per` to a thread
    auto element = [&](uint32_t per, int i) __attribute__((always_inline)) { return wave * 64u * per + lane + i * 64u; };
    // Straight-line on purpose: every item is loaded, items past the sub-bucket's end from its
    // first element (one cache line for the whole wave; the value is never used).  Conditional
    // loads into the arrays made the compiler copy whole arrays at every join.
    auto load_bucket = [&](uint32_t (&key)[KPT], uint32_t (&val)[KPT], uint32_t first, uint32_t count) __attribute__((always_inline)) {
        if (count == 0) return;  // uniform; sort_bucket does not look at the registers either
        const uint32_t per = (count + THREADS - 1) / THREADS;
        const uint32_t* __restrict__ kin = keys + first;
        const uint32_t* __restrict__ vin = vals + first;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t e = element(per, i);
            const uint32_t ec = (static_cast<uint32_t>(i) < per && e < count) ? e : 0u;
            key[i] = kin[ec];  // nontemporal loads measured 7 % slower
            val[i] = vin[ec];
        }
    };
    // sorts the `cnt` pairs held in (key, val) and writes them over [start, start + cnt)
    auto sort_bucket = [&](uint32_t (&key)[KPT], uint32_t (&val)[KPT], const uint32_t start, const uint32_t cnt) __attribute__((always_inline)) {
        if (cnt == 0) return;  // uniform
        const uint32_t ku = (cnt + THREADS - 1) / THREADS;
        PackedPos<KPT> pos;  // ranks and slots are below 2^13: two to a register
        for (int j = tid; j < 2 * WAVES * RADIX; j += THREADS) (&s_hist[0][0][0])[j] = 0;
        __syncthreads();  // the counters are zero (and every wave has left the previous write-out)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t sh = shifts[h], wd = widths[h];
            if (h == 1) {
                // the order pass 1 left in LDS, read back slice by slice
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    const uint32_t e = element(ku, i);
                    const uint2 kv = s_stage[(static_cast<uint32_t>(i) < ku && e < cnt) ? e : 0u];  // straight-line, as the loads
                    key[i] = kv.x;
                    val[i] = kv.y;
                }
            }
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                uint32_t r = 0;
                if (static_cast<uint32_t>(i) < ku && element(ku, i) < cnt)
                    r = atomicAdd(&s_hist[h][wave][__builtin_amdgcn_ubfe(key[i], sh, wd)], 1u);
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
                incl = tot;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t y = __shfl_up(incl, o);
                    if (lane >= static_cast<uint32_t>(o)) incl += y;
                }
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
            for (int i = 0; i < KPT; ++i) {
                if (static_cast<uint32_t>(i) < ku && element(ku, i) < cnt)
                    s_stage[pos.get(i) + s_hist[h][wave][__builtin_amdgcn_ubfe(key[i], sh, wd)]] = make_uint2(key[i], val[i]);
            }
            __syncthreads();
        }
        // write-out, two slots per lane: one 16-byte LDS read, one 8-byte store each for keys and
        // payload (a sub-bucket starts at any 4-byte boundary; gfx950 serves the wider store there).
        // Four slots and 16-byte stores measured 2 % slower.
        uint32_t* __restrict__ kout = keys + start;
        uint32_t* __restrict__ vout = vals + start;
        for (uint32_t s2 = 2u * tid; s2 < cnt; s2 += 2u * THREADS) {
            if (s2 + 1 < cnt) {
                const uint4 q = *reinterpret_cast<const uint4*>(&s_stage[s2]);
                *reinterpret_cast<uint2*>(kout + s2) = make_uint2(q.x, q.z);
                *reinterpret_cast<uint2*>(vout + s2) = make_uint2(q.y, q.w);
            } else {
                const uint2 kv = s_stage[s2];
                kout[s2] = kv.x;
                vout[s2] = kv.y;
            }
        }
    };

    // This workgroup's sub-buckets are blockIdx.x, + gridDim.x, ...: their bounds go to LDS once
    // (a scalar load inside the loop would be waited for at the next LDS access).  Two register
    // sets take turns: while one sub-bucket is sorted and written out, the pairs of the next are
    // already on their way into the other set, so the workgroup always has loads in flight.
    const uint32_t mine = (num_subs - 1u - blockIdx.x) / gridDim.x + 1u;  // blockIdx.x < num_subs <= kHybSubs
    for (uint32_t j = tid; j < mine; j += THREADS) {
        const uint32_t which = blockIdx.x + j * gridDim.x;
        s_first[j] = tb->sub_start[which];
        s_count[j] = tb->sub_start[which + 1] - tb->sub_start[which];
    }
    __syncthreads();
    uint32_t key_a[KPT], val_a[KPT], key_b[KPT], val_b[KPT];
    load_bucket(key_a, val_a, s_first[0], s_count[0]);
    for (uint32_t j = 0;;) {
        if (j + 1 < mine) load_bucket(key_b, val_b, s_first[j + 1], s_count[j + 1]);
        sort_bucket(key_a, val_a, s_first[j], s_count[j]);
        if (++j >= mine) break;
        if (j + 1 < mine) load_bucket(key_a, val_a, s_first[j + 1], s_count[j + 1]);
        sort_bucket(key_b, val_b, s_first[j], s_count[j]);
        if (++j >= mine) break;
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

int hybrid_local_sort_grid(int num_cus)
{
    // 78 KiB of LDS each: two per compute unit
    const int g = num_cus * 2;
    return g < static_cast<int>(kLocalMinGrid) ? static_cast<int>(kLocalMinGrid) : g;
}  // 80 KiB of LDS each: two per compute unit

int hybrid_launch_histogram(const uint32_t* keys, uint32_t n, uint32_t shift, uint32_t bits, bool windowed, uint32_t window_base,
                            int grid, uint32_t* packed, HybridHeader* hdr, hipStream_t s)
{
    const ProfilerRange range("Histogram calculation");  // src/main.rs:597
    const uint32_t vectors = (reinterpret_cast<uintptr_t>(keys) & 15u) ? 0u : n / 4u;
    hipLaunchKernelGGL(hybrid_histogram_kernel, dim3(grid), dim3(1024), 0, s, keys, n, vectors, shift, bits, packed, hdr,
                       kHybLocalCap, window_base, windowed ? 1u : 0u);
    MRS_CHECK_LAUNCH("hybrid_histogram_kernel");
    return MRS_OK;
}

int hybrid_launch_scan(const uint32_t* packed, int grid, HybridTables* tables, HybridHeader* hdr, uint32_t n,
                       uint32_t tile_items, uint32_t force_mode, uint32_t sub_bits, hipStream_t s)
{
    if (sub_bits < 6 || sub_bits > 8) return fail(MRS_ERR_UNSUPPORTED, "a bucket has 64 to 256 sub-buckets");
    const ProfilerRange range("Small Prefix Sum");  // src/main.rs:625
    hipLaunchKernelGGL(hybrid_reduce_kernel, dim3(kHybPackedWords / 64), dim3(256), 0, s, packed, grid, tables->counts);
    MRS_CHECK_LAUNCH("hybrid_reduce_kernel");
    hipLaunchKernelGGL(hybrid_scan_kernel, dim3(1), dim3(1024), 0, s, tables, hdr, n, tile_items, kHybLocalCap, force_mode, sub_bits);
    MRS_CHECK_LAUNCH("hybrid_scan_kernel");
    return MRS_OK;
}

template <bool SEG, bool PERSISTENT, int MAP>
static int launch_pass_shape(int threads, int kpt, const PassArgs& a, const HybridHeader* hdr, const HybridTables* tables,
                             uint32_t run_if_mode, uint32_t grid, hipStream_t s)
{
    if (threads == 1024 && kpt == 32)
        hipLaunchKernelGGL((hybrid_pass_kernel<SEG, PERSISTENT, MAP, 1024, 32>), dim3(grid), dim3(1024), 0, s, a, hdr, tables, run_if_mode);
    else if (threads == 512 && kpt == 8)
        hipLaunchKernelGGL((hybrid_pass_kernel<SEG, PERSISTENT, MAP, 512, 8>), dim3(grid), dim3(512), 0, s, a, hdr, tables, run_if_mode);
    else
        return fail(MRS_ERR_UNSUPPORTED, "no gated pass kernel for %d threads x %d keys per thread", threads, kpt);
    MRS_CHECK_LAUNCH("hybrid_pass_kernel");
    return MRS_OK;
}

int hybrid_launch_pass(bool segmented, bool persistent, bool windowed, int threads, int kpt, const PassArgs& a,
                       const HybridHeader* hdr, const HybridTables* tables, uint32_t run_if_mode, uint32_t grid, hipStream_t s)
{
    if (a.wide_status) return fail(MRS_ERR_UNSUPPORTED, "gated passes use 32-bit tile status words");
    const ProfilerRange range("Indexing");  // src/main.rs:658
    if (grid == 0) grid = 1;
    if (persistent) {  // the plan's ordinary passes, or pass B of a sort that shares the device with nothing: plain digits
        if (windowed) return fail(MRS_ERR_UNSUPPORTED, "persistent gated passes take plain bit-field digits");
        return segmented ? launch_pass_shape<true, true, kMapNone>(threads, kpt, a, hdr, tables, run_if_mode, grid, s)
                         : launch_pass_shape<false, true, kMapNone>(threads, kpt, a, hdr, tables, run_if_mode, grid, s);
    }
    if (windowed)
        return segmented ? launch_pass_shape<true, false, kMapWindow>(threads, kpt, a, hdr, tables, run_if_mode, grid, s)
                         : launch_pass_shape<false, false, kMapWindow>(threads, kpt, a, hdr, tables, run_if_mode, grid, s);
    return segmented ? launch_pass_shape<true, false, kMapNone>(threads, kpt, a, hdr, tables, run_if_mode, grid, s)
                     : launch_pass_shape<false, false, kMapNone>(threads, kpt, a, hdr, tables, run_if_mode, grid, s);
}

int hybrid_launch_local_sort(uint32_t* keys, uint32_t* vals, const HybridHeader* hdr, const HybridTables* tables,
                             uint32_t shift, uint32_t bits1, uint32_t bits2, uint32_t num_subs, int grid, hipStream_t s)
{
    if (num_subs == 0 || num_subs > kHybSubs) return fail(MRS_ERR_UNSUPPORTED, "at most 65536 sub-buckets");
    if (bits1 == 0 || bits1 > 8 || bits2 == 0 || bits2 > 8)
        return fail(MRS_ERR_UNSUPPORTED, "the local sort takes two digits of 1 to 8 bits");
    const ProfilerRange range("Local sort");
    hipLaunchKernelGGL((hybrid_local_sort_kernel<kHybLocalThreads, kHybLocalKpt>), dim3(grid), dim3(kHybLocalThreads), 0, s,
                       keys, vals, hdr, tables, shift, bits1, bits2, num_subs);
    MRS_CHECK_LAUNCH("hybrid_local_sort_kernel");
    return MRS_OK;
}

}  // namespace mrs
