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
//           9..16 bits in two stable counting passes, and written back in place, sequentially.
// The split (14, 15 or 16 bits for A and B together) is the host's, from n: sub-buckets of about
// 4096 pairs keep the local sort's workgroups busy, and the local sort takes at most 16 bits.
// It needs sub-buckets that fit a workgroup, which only the device can know (the host never reads
// a key), so a 16-bit histogram taken before pass A decides on the device: if any sub-bucket
// exceeds the capacity, the three kernels above return at once and the four least-significant-
// digit-first passes of the plan run instead; otherwise those return at once.  Both schedules end
// on the same side (A: a -> b, B: b -> a, local: in place; four passes: a again).
#ifndef MRS_HYBRID_HPP
#define MRS_HYBRID_HPP

#include "mrs_internal.h"

namespace mrs {

constexpr uint32_t kHybBuckets = 256;   // first level, at most: the top 8 of the sorted bits
constexpr uint32_t kHybSubs = 65536;    // second level, at most: the top 16
constexpr uint32_t kHybRun = 1;         // HybridHeader::mode: the three-move schedule runs
constexpr uint32_t kHybFallback = 2;    // the plan's least-significant-digit-first passes run
constexpr int kHybLocalThreads = 512;   // workgroup of the local sort
constexpr int kHybLocalKpt = 15;        // pairs per thread at most: sub-buckets of up to 7680 pairs
constexpr uint32_t kHybLocalCap = kHybLocalThreads * kHybLocalKpt;
constexpr uint32_t kHybPackedWords = kHybSubs / 2;  // one workgroup's 16-bit counters, two to a word

// Inside the block that every sort call clears before its first kernel.
struct HybridHeader {
    uint32_t mode;         // 0 until hybrid_scan_kernel has decided
    uint32_t num_tiles_b;  // tiles of pass B (0 unless mode == kHybRun)
    uint32_t overflow;     // a 16-bit counter of the histogram wrapped: counts are not to be trusted
    uint32_t max_sub;      // largest sub-bucket (diagnostic); 0xFFFFFFFF if the histogram stopped early or wrapped
    uint32_t pad[60];
};
static_assert(sizeof(HybridHeader) == 256, "the header takes one 256-byte slot of the cleared block");

// Written in full by the reduce and scan kernels of every call; never cleared.
struct HybridTables {
    uint32_t bucket_start[kHybBuckets + 4];  // first element of bucket b; [256] = n
    uint32_t first_tile[kHybBuckets + 4];    // first pass-B tile of bucket b; [256] = tiles in all
    uint32_t sub_start[kHybSubs + 4];        // first element of sub-bucket i; [65536] = n
    uint32_t counts[kHybSubs];               // elements of sub-bucket i
};

// Device bytes behind the status words of a plan that can run this schedule.
size_t hybrid_tables_bytes();
size_t hybrid_packed_bytes(int grid);
int hybrid_histogram_grid(int num_cus, uint64_t n);
int hybrid_local_sort_grid(int num_cus);

// Histogram of the `bits` <= 16 key bits from `shift` up: one packed table per workgroup.
// windowed: of (key >> shift) - window_base instead, which must be below 2^bits for every key (a
// key outside that window makes the device refuse the schedule).
int hybrid_launch_histogram(const uint32_t* keys, uint32_t n, uint32_t shift, uint32_t bits, bool windowed, uint32_t window_base,
                            int grid, uint32_t* packed, HybridHeader* hdr, hipStream_t s);
// Sum of the `grid` packed tables, its exclusive scan, the tile table of pass B and the decision.
int hybrid_launch_scan(const uint32_t* packed, int grid, HybridTables* tables, HybridHeader* hdr, uint32_t n,
                       uint32_t tile_items, uint32_t force_mode, uint32_t sub_bits, hipStream_t s);
// One chained-scan pass that runs only if hdr->mode == run_if_mode.  segmented: pass B (tiles and
// chains per bucket, a.digit_base and a.num_tiles ignored); otherwise a pass over the whole array.
// persistent: `grid` workgroups draw tiles until none is left (an idle launch then costs `grid`
// workgroups, not one per tile).
// windowed: the digits are fields of (key >> a.shift) - a.window_base (PassArgs).
int hybrid_launch_pass(bool segmented, bool persistent, bool windowed, int threads, int kpt, const PassArgs& a,
                       const HybridHeader* hdr, const HybridTables* tables, uint32_t run_if_mode, uint32_t grid,
                       hipStream_t s);
// Sorts every sub-bucket in place on bits [shift, shift + bits1 + bits2).
int hybrid_launch_local_sort(uint32_t* keys, uint32_t* vals, const HybridHeader* hdr, const HybridTables* tables,
                             uint32_t shift, uint32_t bits1, uint32_t bits2, uint32_t num_subs, int grid, hipStream_t s);

}  // namespace mrs

#endif
