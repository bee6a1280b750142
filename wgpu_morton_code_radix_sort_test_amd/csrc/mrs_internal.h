// Internal declarations shared by the translation units of libmrs.so.
#ifndef MRS_INTERNAL_H
#define MRS_INTERNAL_H

#include <hip/hip_runtime.h>

#include <pthread.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "mrs.h"

namespace mrs {

// Records the calling thread's error message and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();
// Zeroes `bytes` of device memory (address and size multiples of 4) behind the work already on `s`.
int clear_async(void* p, size_t bytes, hipStream_t s);

#define MRS_HIP_TRY(expr)                                                              \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return ::mrs::fail(MRS_ERR_HIP, "%s failed: %s (%s:%d)", #expr,            \
                               hipGetErrorString(_e), __FILE__, __LINE__);             \
    } while (0)

// The launch itself can fail (bad configuration); report it instead of
// leaving a sticky error behind.
#define MRS_CHECK_LAUNCH(name)                                                         \
    do {                                                                               \
        hipError_t _e = hipGetLastError();                                             \
        if (_e != hipSuccess)                                                          \
            return ::mrs::fail(MRS_ERR_HIP, "launch of %s failed: %s", name,           \
                               hipGetErrorString(_e));                                 \
    } while (0)

// Diagnostic builds (-DMRS_DESTROY_TRACE) print every runtime call of the teardown paths to stderr
// with the calling thread, before and after, and do not serialise the teardown of sharded-sort
// handles: a process that dies inside one of the calls leaves the names of the calls in flight.
#ifdef MRS_DESTROY_TRACE
#define MRS_TEARDOWN(expr)                                                                              \
    do {                                                                                               \
        std::fprintf(stderr, "[teardown %lx] > %s\n", (unsigned long)pthread_self(), #expr);           \
        const int _rc = static_cast<int>(expr);                                                        \
        std::fprintf(stderr, "[teardown %lx] < %s = %d\n", (unsigned long)pthread_self(), #expr, _rc); \
    } while (0)
#else
#define MRS_TEARDOWN(expr) (void)(expr)
#endif

constexpr int kMaxPasses = 16;   // 64-bit keys with 4-bit digits
constexpr int kMaxRadix = 256;
constexpr int kTicketLists = 8;  // one per accelerator die (XCD) of the MI355X
constexpr int kMaxMapEntries = 4096;  // entries of a partition's digit table (12 key bits)
constexpr int kMaxSplitters = 15;     // key values a partition can split at (16 destinations)
constexpr uint32_t kMapBySplitters = 0xFFu;  // digit table entry: this bucket holds a splitter

// Per-launch arguments of a digit pass.
struct PassArgs {
    const void* keys_in;
    const void* vals_in;
    void* keys_out;
    void* vals_out;
    uint32_t n;
    uint32_t num_tiles;
    uint32_t shift;            // first bit of the digit
    uint32_t mask;             // (1 << digit bits of this pass) - 1
    const uint8_t* digit_map;  // optional: digit = digit_map[(key >> shift) & mask]
    // optional, with digit_map: table entries equal to kMapBySplitters send the key to a comparison
    // with `num_splitters` ascending key values (u32 or u64 like the keys) instead:
    // digit = 2 * #{r : key >= splitter[r]} - #{r : key == splitter[r] and bit r of exact_mask}
    const void* splitters;
    uint32_t num_splitters;
    uint32_t exact_mask;
    // kMapWindow passes: digit = (((key >> shift) - window_base) >> window_low) & mask, for keys known
    // to lie in a window of the key space (the most-significant-digit-first schedule on one slice
    // of a sharded sort)
    uint32_t window_base;
    uint32_t window_low;
    // kMapRank passes (the adaptive most-significant-digit-first schedule, mrs_hybrid.hpp): `shift` is
    // the first bit of a level-A cell's number, cell_entry[cell - cell_first] = first fine cell of
    // that cell | (first key bit of a fine cell's number) << 16, and the digit is the number of group
    // starts at or before the key's fine cell, from rank_words (bit map with running counts)
    const uint32_t* cell_entry;
    const unsigned long long* rank_words;
    uint32_t cell_first;
    // three-kernel: scanned block histogram, digit-major [radix][num_tiles]
    const uint32_t* scanned;
    // onesweep
    const uint32_t* digit_base;  // exclusive scan of the global digit histogram of this pass
    void* status;                // [num_tiles][radix] tile status words of this pass, zeroed
    uint32_t wide_status;        // status words are 64-bit (inputs of 2^30 elements and more)
    uint32_t* tile_counter;      // ticket dispensers of this pass: kTicketLists counters
    uint32_t xcd_block;          // tiles per ticket block (see draw_tile in mrs_radix_pass.hpp); 0 = one list
    uint32_t* error_flag;
    // optional, 64-bit keys: one word written by global_histogram_scan_kernel for this pass.
    // Bit 0: every key has the same digit, the pass would move nothing: do not run.  Bit 1: an
    // odd number of earlier passes did not run, so the input is where the output was expected:
    // exchange the roles of the two buffer pairs.
    const uint32_t* pass_plan;
    uint32_t atomic_rank;        // host side only: rank with returning LDS atomics (kRankAtomic)
    // 4-byte payloads: the payload of input element i is i itself (the first pass of a sort of
    // (key, source index) pairs); vals_in is not read
    uint32_t index_payload;
#ifdef MRS_STAMPS
    unsigned long long* stamps;  // diagnostic builds only: [num_tiles][8] phase clocks
#endif
};

// Control block at the start of a onesweep plan's temporary memory; zeroed by
// one memset per sort call together with the status words that follow it.
struct OnesweepControl {
    uint32_t ghist[kMaxPasses * kMaxRadix];  // global digit histograms, then their scans
    uint32_t tile_counter[kMaxPasses][kTicketLists];
    uint32_t pass_plan[kMaxPasses];  // see PassArgs::pass_plan; 0 = run as scheduled
};
static_assert(sizeof(OnesweepControl) % 16 == 0, "memset block must be a multiple of 16 bytes");

// Arguments of the one-workgroup sort of small inputs (mrs_small_sort.hip).
struct SmallSortArgs {
    const void* keys_in;
    const void* vals_in;
    void* keys_out;     // where the sorted keys go
    void* vals_out;     // where the gathered payload goes; never the buffer of vals_in
    void* vals_final;   // if not null: vals_out is copied here afterwards (may be the buffer of vals_in)
    uint32_t n;
    uint32_t begin_bit, end_bit;
    uint32_t num_passes;
    uint32_t payload_bytes;  // 0, 4 or 16
    uint32_t index_payload;  // 4-byte payloads: the payload of input element i is i; vals_in is not read
};

struct KernelConfig {
    int threads;
    int kpt;  // keys per thread
};

// What tests and A/B runs can ask of a plan (mrs_context_set_attribute, MRS_ATTR_TEST_*).
struct TestAids {
    uint64_t hybrid_min_n = 0;         // 0 = the library's thresholds
    uint32_t hybrid_leading_bits = 0;  // 0 = chosen from n
    bool hybrid_force_fallback = false;
    bool hybrid_force_adaptive = false;  // the fixed sub-buckets are passed over: buckets and groups cut to the keys
    bool hybrid_no_adaptive = false;     // that schedule is not offered
    uint32_t hybrid_capacity = 0;        // 0 = what a workgroup of the local sort holds; less: more groups, more oversized ones
    bool hybrid_big_local = false;       // the local sort's workgroups of 1024 threads at any size (where 6-byte staging applies)
    bool no_small_sort = false;
    bool no_pass_skip = false;
    bool wide_status = false;
};

// Tuning aids of profiling scripts, read from the environment once, when a context is created
// (never inside a sort call): MRS_XCD_BLOCK, MRS_HIST_BLOCKS_PER_CU, MRS_HIST_PLAIN,
// MRS_HIST_BANKED_PER_CU.
struct Tuning {
    int xcd_block = -1;                // tiles per die-aware ticket block; -1 = the default (8)
    uint32_t hist_blocks_per_cu = 16;  // workgroups per compute unit of the plain digit histogram
    bool hist_plain = false;           // the digit histogram without bank-spread counters (A/B aid)
    uint32_t hist_banked_per_cu = 1;   // workgroups per compute unit of the bank-spread one
};

// A roctx range around the launches enqueued while the object lives, if ranges are switched on
// (MRS_ATTR_ROCTX_RANGES); otherwise one relaxed load.
class ProfilerRange {
public:
    explicit ProfilerRange(const char* name);
    ~ProfilerRange();
    ProfilerRange(const ProfilerRange&) = delete;
    ProfilerRange& operator=(const ProfilerRange&) = delete;

private:
    bool pushed_;
};

}  // namespace mrs

struct mrs_context {
    int device;
    int num_cus;
    // Returning LDS atomics of one instruction that hit the same word are served in
    // ascending lane order on this device (checked at creation): the digit passes may rank
    // with them.
    bool lds_atomics_in_lane_order;
    bool lds_atomic_probe_passed;  // what the creation-time check found (the flag above can be cleared)
    // A large grid lands on all 8 dies in comparable shares (checked at creation): the digit
    // passes may hand consecutive tiles to workgroups of one die.
    bool workgroups_on_all_dies;
    bool die_aware_tickets;  // workgroups_on_all_dies, unless switched off (mrs_context_set_attribute)
    int live_plans;          // plans created on this context and not yet destroyed
    // Address, on this device, of the library's static scratch array (mrs_sort_pairs_noalloc).
    void* static_scratch;
    // test and A/B aids (MRS_ATTR_TEST_*): copied into every plan at its creation
    mrs::TestAids aids;
    mrs::Tuning tuning;
};

namespace mrs {
struct HybridCleared;
struct HybridTables;
}

struct mrs_plan {
    mrs_context* ctx;
    mrs_plan_desc desc;
    // MRS_ALGO_HYBRID: desc.algo reads MRS_ALGO_ONESWEEP and the plan carries the tables of
    // mrs_hybrid.hpp behind its status words
    bool hybrid;
    mrs::HybridCleared* hyb_cleared;  // header, spill counts, digit counts of oversized sub-buckets: inside the block cleared per call, before the status words
    mrs::HybridTables* hyb_tables;
    uint32_t* hyb_packed;           // one packed 16-bit histogram per workgroup of the histogram kernel
    bool hyb_last;                  // the last sort call enqueued both schedules (hyb_cleared->hdr is its record)
    bool hyb_compact_ok;            // ... and its local sort could stage a pair in 6 bytes if the device agreed
    uint32_t hyb_cap;               // ... and took a sub-bucket of at most this many pairs
    uint32_t radix_bits, radix, num_passes;
    uint32_t threads, kpt, tile_items;
    uint64_t max_tiles;
    uint64_t histogram_len;
    // device memory owned by the plan
    void* temp;
    size_t temp_bytes;
    bool owns_temp;  // temp came from hipMalloc in mrs_plan_create (false: the library's static scratch)
    // Set by the sharded sort on its own plans: their calls are ordered by events across its two
    // streams, so the guard against concurrent die-aware passes treats them as one stream.
    const void* lineage;
    // three-kernel
    uint32_t* hist;          // [radix][max_tiles]
    uint32_t* scan_partials; // chunk sums of the scan
    uint64_t scan_partials_len;
    // onesweep
    mrs::OnesweepControl* control;
    char* status;                // num_passes regions of [num_tiles][radix] status words
    uint32_t status_word_bytes;  // 4 below 2^30 elements, else 8
    char* zero_begin;            // control block + status regions: cleared by one memset per call
    uint32_t* error_flag;        // sticky device word
    // the last mrs_sort_pairs call let the device decide which passes run (control->pass_plan is its record)
    bool pass_plan_live;
    // mrs_morton_encode30_for_sort left the digit histograms of these keys in `control`
    const void* primed_keys;
    uint64_t primed_n;
    mrs::TestAids aids;  // the context's, as they were when the plan was created
    // optional per-launch timing (mrs_plan_set_timing): events recorded on the
    // caller's stream around every launch of the last mrs_sort_pairs call
    bool timing;
    int num_marks;
    hipEvent_t marks[mrs::kMaxPasses * 4 + 8];
    const char* mark_labels[mrs::kMaxPasses * 4 + 8];
};

namespace mrs {

// mrs_api.hip, for the sharded sort (mrs_dist.hip)
int partition_internal(mrs_plan* plan, uint64_t n, const void* keys_in, const void* vals_in, void* keys_out,
                       void* vals_out, uint32_t shift, uint32_t bits, const uint8_t* map_dev, const void* splitters_dev,
                       uint32_t num_splitters, uint32_t exact_mask, const uint32_t* class_starts_dev, hipStream_t s);
int partition_pass(mrs_plan* plan, uint64_t n, const void* keys_in, const void* vals_in, void* keys_out, void* vals_out,
                   uint32_t shift, uint32_t bits, const uint8_t* map_dev, const void* splitters_dev, uint32_t num_splitters,
                   uint32_t exact_mask, bool scan_histogram, hipStream_t s);
// `window` (may be null): two key values every key lies between, inclusive; lets an MRS_ALGO_HYBRID
// plan split the window, not the whole key space, into sub-buckets.
void forget_stream_activity(int device, hipStream_t s);
int sort_pairs_internal(mrs_plan* plan, uint64_t n, void* keys_a, void* vals_a, void* keys_b, void* vals_b, hipStream_t s,
                        int* result_in_b, const uint64_t* window = nullptr);

// scan (mrs_scan.hip)
uint64_t scan_partials_len(uint64_t len);
int launch_exclusive_scan(uint32_t* vals, uint64_t len, uint32_t* partials, hipStream_t s);

}  // namespace mrs

#endif
