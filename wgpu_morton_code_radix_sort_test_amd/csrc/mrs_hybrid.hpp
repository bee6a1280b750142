// Most-significant-digit-first schedule for (u32 key, u32 payload) pairs: declarations shared by
// mrs_hybrid.hip (kernels) and mrs_api.hip (the launch schedule).
//
// The reference sorts least significant digit first, NUM_PASSES scatters of the whole array
// (src/main.rs:594-665).  A stable sort has one answer, so any schedule that is stable gives the
// same bytes; this one moves every pair three times instead of four:
//   pass A  one chained-scan scatter by the top 7 or 8 of the sorted bits (128 or 256 buckets);
//   pass B  one chained-scan scatter by the next 7 or 8 bits, every bucket a chain of its own
//           (2^14 to 2^16 sub-buckets; tiles never straddle a bucket);
//   local   every sub-bucket is loaded into LDS by one workgroup, sorted there by the remaining
//           2..16 bits in two stable counting passes, and written back in place, sequentially.
// The split (14, 15 or 16 bits for A and B together) is the host's, from n: sub-buckets of a few
// thousand pairs keep the local sort's workgroups busy, and the local sort takes at most 16 bits.
//
// A sub-bucket must fit a workgroup, which only the device can know (the host never reads a key),
// so an exact 16-bit histogram taken before pass A decides on the device, sub-bucket by sub-bucket:
//   * a sub-bucket of at most `cap` pairs is sorted inside LDS (cap: 10240 when a pair can be staged
//     in 6 bytes -- its low 16 key bits and its payload, the rest of the key being the sub-bucket's
//     -- else 7680);
//   * a larger one (equal keys, a cluster) is sorted where it lies by two more chained-scan
//     scatters on the remaining bits, least significant digit first, a -> b -> a inside its own
//     range, one chain per oversized sub-bucket, after a histogram of just those pairs;
//   * if more than 256 sub-buckets are oversized, or they hold more than a quarter of the pairs
//     (four moves of most pairs: no better than the ordinary schedule), every kernel above returns
//     at once and the plan's four least-significant-digit-first passes run instead; otherwise
//     those return at once.
// Both schedules end on the same side (A: a -> b, B: b -> a, local and oversized: in place; four
// passes: a again).
//
// Keys that crowd into few of the fixed sub-buckets (points on surfaces: most of the 65536 cells of
// the leading 16 bits are empty and the occupied ones hold several workgroups' worth) can still be
// sorted in three moves if the two scatters follow the keys' density instead of fixed bit fields
// (kHybAdapt):
//   plan    the exact table, added up over the 4096 cells of the leading 12 bits, is cut into at
//           most 256 buckets of about n / 224 pairs each (whole cells, at most 256 cells from a
//           bucket's first to its last): a 4096-entry byte table;
//   pass A  scatters by that table;
//   count   one workgroup per bucket reads its bucket and counts it on up to 2^14 "fine cells" (every
//           occupied 12-bit cell of the bucket is divided into a power of two of them, more where
//           it holds more pairs, never across a multiple of 2^16 key values), then cuts the run
//           of fine cells into at most 256 groups of at most `cap` pairs: a bit per fine cell
//           that starts a group, with a running count per word;
//   pass B  scatters by the number of group starts at or before the key's fine cell (two LDS reads);
//   local   a group lies inside one 16-bit cell, so the local sort takes it on its low 16 bits as
//           it takes a sub-bucket; a fine cell that holds more than `cap` pairs alone is an
//           oversized sub-bucket as above.
// Tried only when the fixed sub-buckets were refused for their sizes.  If the count finds a
// bucket that needs more than 256 groups (or too many oversized fine cells), the ordinary passes
// run after all: they start one launch later than pass A for that reason.
#ifndef MRS_HYBRID_HPP
#define MRS_HYBRID_HPP

#include "mrs_internal.h"

namespace mrs {

constexpr uint32_t kHybBuckets = 256;   // first level, at most: the top 8 of the sorted bits
constexpr uint32_t kHybSubs = 65536;    // second level, at most: the top 16
constexpr uint32_t kHybRun = 1;         // HybridHeader::mode: the three-move schedule runs
constexpr uint32_t kHybFallback = 2;    // the plan's least-significant-digit-first passes run
constexpr uint32_t kHybAdapt = 3;       // the three-move schedule runs on buckets and groups cut to the keys' density
constexpr uint32_t kAdaptCellBits = 12; // level-A cells: the leading 12 sorted bits
constexpr uint32_t kAdaptCells = 1u << kAdaptCellBits;
constexpr uint32_t kAdaptFineBits = 14; // fine cells per bucket, at most
constexpr uint32_t kAdaptFine = 1u << kAdaptFineBits;
constexpr uint32_t kAdaptRankWords = kAdaptFine / 32;
constexpr uint32_t kAdaptMaxSpan = 256; // level-A cells from a bucket's first to its last, at most
constexpr int kHybLocalThreads = 512;   // workgroup of the local sort
constexpr int kHybLocalKptFull = 15;    // pairs per thread at most when a pair takes 8 bytes of LDS
#ifndef MRS_LOCAL_KPT_COMPACT
#define MRS_LOCAL_KPT_COMPACT 20
#endif
constexpr int kHybLocalKptCompact = MRS_LOCAL_KPT_COMPACT; // ... when it takes 6
constexpr uint32_t kHybLocalCapFull = kHybLocalThreads * kHybLocalKptFull;        // 7680
constexpr uint32_t kHybLocalCapCompact = kHybLocalThreads * kHybLocalKptCompact;  // 10240
// Inputs whose evenly spread sub-buckets no longer fit that (2^29.1 to 2^30 pairs): one workgroup of
// 1024 threads per compute unit, 6-byte staging only.
constexpr int kHybLocalThreadsBig = 1024;
constexpr uint32_t kHybLocalCapBig = kHybLocalThreadsBig * kHybLocalKptCompact;   // 20480
constexpr uint32_t kHybPackedWords = kHybSubs / 2;  // one workgroup's 16-bit counters, two to a word
constexpr uint32_t kHybMaxBig = 256;    // oversized sub-buckets one call sorts where they lie

// At the start of the block that every sort call clears before its first kernel.
struct HybridHeader {
    uint32_t mode;           // 0 until hybrid_scan_kernel has decided
    uint32_t num_tiles_b;    // tiles of pass B (0 unless mode == kHybRun)
    uint32_t overflow;       // the histogram cannot be trusted (a key outside the promised window, counts that do not add up)
    uint32_t max_sub;        // largest sub-bucket (diagnostic)
    uint32_t num_big;        // oversized sub-buckets (0 unless mode == kHybRun)
    uint32_t num_tiles_big;  // tiles of the two passes over them
    uint32_t big_pairs;      // pairs they hold
    uint32_t high_vary;      // key bits at and above the sorted range differ between keys: no 6-byte staging
    uint32_t adapt_buckets;  // kHybAdapt: buckets pass A makes
    uint32_t big_limit;      // kHybAdapt: pairs the oversized fine cells may hold together (n / 4)
    uint32_t adapt_fail;     // kHybAdapt: a bucket needs more than 256 groups
    uint32_t adapt_groups;   // kHybAdapt: groups in all
    uint32_t pad[52];
};

// The schedule a call runs.  The decision kernel writes `mode`; the adaptive schedule can still
// turn out not to work while its buckets are counted (a bucket that cannot be cut, too many or too
// full oversized fine cells), which every later kernel, and the host, reads from what the counting
// workgroups left in the header: no workgroup has to wait for the last of them.
__host__ __device__ inline uint32_t hybrid_mode(const HybridHeader& h)
{
    if (h.mode == kHybAdapt && (h.adapt_fail != 0u || h.num_big > kHybMaxBig || h.big_pairs > h.big_limit)) return kHybFallback;
    return h.mode;
}
static_assert(sizeof(HybridHeader) == 256, "the header takes one 256-byte slot of the cleared block");

// Cleared by every sort call together with the header (it follows it in memory).
struct HybridCleared {
    HybridHeader hdr;
    // Counts the histogram workgroups moved out of their 16-bit counters, 32768 at a time, before
    // those could wrap: the exact count of sub-bucket i is spill[i] + the sum of the fields.
    uint32_t spill[kHybSubs];
    // Oversized sub-bucket j: counts of its two local digits.
    uint32_t big_hist[kHybMaxBig][2][256];
};

// Written in full by the reduce and scan kernels of every call; never cleared.
struct HybridTables {
    uint32_t bucket_start[kHybBuckets + 4];  // first element of bucket b; [256] = n
    uint32_t first_tile[kHybBuckets + 4];    // first pass-B tile of bucket b; [256] = tiles in all
    uint32_t sub_start[kHybSubs + 4];        // first element of sub-bucket i; [65536] = n
    uint32_t counts[kHybSubs];               // elements of sub-bucket i
    uint32_t big_first_tile[kHybMaxBig + 4]; // first tile of oversized sub-bucket j; from [num_big] on: tiles in all
    uint32_t big_start[kHybMaxBig + 4];      // its first element
    uint32_t big_end[kHybMaxBig + 4];        // one past its last
    // kHybAdapt only
    uint8_t map_a[kAdaptCells];              // level-A cell -> bucket
    uint32_t cell_count[kAdaptCells];        // pairs of level-A cell c
    uint32_t cell_entry[kAdaptCells];        // first fine cell of level-A cell c in its bucket | (first key bit of a fine cell's number) << 16
    uint32_t cell_first[kHybBuckets + 4];    // first occupied level-A cell of bucket b
    uint32_t cell_last[kHybBuckets + 4];     // last one
    // bucket b, word w: bit f % 32 = fine cell f = 32 w + f % 32 starts a group | (group starts in words before w, minus one) << 32
    unsigned long long rank[kHybBuckets][kAdaptRankWords];
    // the groups of all buckets in one list, in no particular order, for the local sort: first pair
    // and pairs of group i < HybridHeader::adapt_groups
    uint32_t group_first[kHybSubs];
    uint32_t group_pairs[kHybSubs];
};

// What the adaptive schedule adds to a call (null plan pointers: not offered).
struct AdaptArgs {
    uint32_t offered;       // the plan can run it (host's conditions)
    uint32_t force;         // test aid: skip the fixed sub-buckets
    uint32_t cell_shift;    // first key bit of a level-A cell's number (end_bit - 12)
    uint32_t low16_shift;   // groups never straddle a multiple of 1 << low16_shift (begin_bit + 16)
    uint32_t bucket_max;    // pairs a bucket may hold
    uint32_t msd_bits;      // bits of the exact table
};

// The segments a segmented chained-scan pass runs over (at most 256): segment s holds the
// elements [start[s], end[s]) and the tiles [first_tile[s], first_tile[s + 1]); the runs of its
// digits begin at digit_base[s * digit_stride + d].
struct SegmentTable {
    const uint32_t* first_tile;
    const uint32_t* start;
    const uint32_t* end;
    const uint32_t* digit_base;
    uint32_t digit_stride;
    const uint32_t* num_tiles;  // device word: tiles of the pass
};

// Device bytes behind the status words of a plan that can run this schedule.
size_t hybrid_tables_bytes();
size_t hybrid_packed_bytes(int grid);
int hybrid_histogram_grid(int num_cus, uint64_t n);
int hybrid_local_sort_grid(int num_cus, bool big = false);

// Exact histogram of the `bits` <= 16 key bits from `shift` up: one packed table per workgroup
// plus the spill counts in `hc`.  windowed: of (key >> shift) - window_base instead, which must be
// below 2^bits for every key (a key outside that window makes the device refuse the schedule).
// `end_bit`: bits at and above it are not sorted on; hc->hdr.high_vary records whether they differ.
// low_hist (u32[2][256], zeroed): receives the counts of the digits [begin_bit, begin_bit + 8) and
// [begin_bit + 8, begin_bit + 16) of every key, the two lowest digits of the plan's ordinary passes.
int hybrid_launch_histogram(const uint32_t* keys, uint32_t n, uint32_t shift, uint32_t bits, bool windowed, uint32_t window_base,
                            uint32_t begin_bit, uint32_t end_bit, int grid, uint32_t* packed, HybridCleared* hc,
                            uint32_t* low_hist, hipStream_t s);
// Sum of the `grid` packed tables and the spill counts, its exclusive scan, the tile tables of
// pass B and of the passes over oversized sub-buckets, and the decision.  `compact_ok`: the host
// allows 6-byte staging (the local sort's bits and everything below them fit 16 bits).  If the
// decision is against the schedule, ghist (u32[4][256]; [0] and [1] as hybrid_launch_histogram left
// them) receives the exclusive digit prefix sums of the plan's four ordinary passes over
// [begin_bit, end_bit): the two upper digits are sums over the exact table, which starts at `shift`
// (within 16 bits of begin_bit) and, windowed, at sub-bucket `window_base`.
// `cap`: pairs a workgroup of the local sort takes (test aid: less than it really holds).
int hybrid_launch_scan(const uint32_t* packed, int grid, HybridTables* tables, HybridCleared* hc, uint32_t n,
                       uint32_t tile_items, uint32_t force_mode, uint32_t sub_bits, bool compact_ok, uint32_t cap,
                       const AdaptArgs& adapt, uint32_t* ghist, uint32_t begin_bit, uint32_t end_bit, uint32_t shift,
                       uint32_t window_base, hipStream_t s);
// kHybAdapt: one workgroup per bucket counts the bucket (in `keys`, where pass A left it) on its
// fine cells and cuts it into groups, enters oversized fine cells in their list and notes in the
// header if its bucket cannot be cut (hybrid_mode then reads kHybFallback).
int hybrid_launch_adapt_count(const uint32_t* keys, HybridTables* tables, HybridCleared* hc, uint32_t n, uint32_t tile_items,
                              uint32_t cap, const AdaptArgs& adapt, hipStream_t s);
// One launch of the chained-scan pass for both schedules (either description may be null): `run`
// is the pass of the three-move schedule (with `seg`: tiles and chains per segment, run->digit_base
// and run->num_tiles ignored), taken if the device chose that schedule; `other` one of the plan's
// ordinary passes, taken if it chose those.
// persistent: `grid` workgroups draw tiles until none is left (a launch with nothing to do then
// costs `grid` workgroups, not one per tile); a launch with a workgroup per tile serves `run` only.
// windowed: the digits are fields of (key >> shift) - window_base (PassArgs).
// role: which launch of the schedule this is (0 pass A, 1 pass B, 2 a pass over the oversized
// sub-buckets, 3 an ordinary pass on its own); it selects the kernel's name, nothing else.
// An `other` with 64-bit tile status words (plans of 2^30 pairs) is served by launches of their own
// kind; `run` and `adapt` always use 32-bit words (no digit of theirs counts 2^30 keys).
// `adapt`: null, or (roles 0 and 1, persistent) the same pass of the adaptive schedule, taken if the
// device chose that one: pass A by the table adapt->digit_map over the bits from adapt->shift up,
// pass B over the same segment table by the group numbers of `tables` (adapt->shift: first bit of a
// level-A cell's number).
int hybrid_launch_pass(int role, const PassArgs* run, const SegmentTable* seg, const PassArgs* other, bool persistent, bool windowed,
                       int threads, int kpt, const HybridHeader* hdr, uint32_t grid, hipStream_t s,
                       const PassArgs* adapt = nullptr, const HybridTables* tables = nullptr);
// Digit counts of the oversized sub-buckets on bits [shift, shift + bits1) and the bits2 above
// them (the passes over them turn the counts into run positions tile by tile).  `other`: null, or the
// ordinary pass the same launch runs if the device chose those (persistent, `grid` workgroups of
// `threads`, the plan's tile shape).
int hybrid_launch_big_histogram(const uint32_t* keys, HybridTables* tables, HybridCleared* hc, uint32_t tile_items,
                                uint32_t shift, uint32_t bits1, uint32_t bits2, const PassArgs* other, int threads, int kpt,
                                int grid, hipStream_t s);
// Sorts every sub-bucket that fits a workgroup in place on bits [shift, shift + bits1 + bits2).
// key_shift: key bits from here up are the same for all keys of a sub-bucket; compact_ok as for
// hybrid_launch_scan (6-byte staging; the decision kernel has refused the schedule if the bits
// above the sorted range vary).  adapt_low16: 0, or the adaptive schedule is offered and its groups
// (65536 table entries) are sorted on the two 8-bit digits below bit adapt_low16.
// big: the workgroups of 1024 threads that hold kHybLocalCapBig pairs (6-byte staging only).
// stream_out: the sorted sub-buckets are written with nontemporal stores (inputs far larger than the
// last-level cache).
int hybrid_launch_local_sort(uint32_t* keys, uint32_t* vals, const HybridHeader* hdr, const HybridTables* tables,
                             uint32_t shift, uint32_t bits1, uint32_t bits2, uint32_t num_subs, uint32_t key_shift,
                             bool compact_ok, uint32_t adapt_low16, int grid, hipStream_t s, bool big = false,
                             bool stream_out = false);

}  // namespace mrs

#endif
