/*
 * mrs.h -- C ABI of the MI355X (gfx950) Morton-key radix sort.
 *
 * This is the boundary a host program binds (Rust `extern "C"`, ctypes, cgo ...)
 * in place of the reference's wgpu compute path.  The reference has exactly one
 * entry point for that path,
 *
 *     pub fn run_compute_shaders(device, encoder) -> Vec<Buffer>   src/main.rs:376
 *
 * which records one Morton-encode dispatch and NUM_PASSES x {histogram, prefix
 * sum, scatter} dispatches into a command encoder that the caller submits.
 * The functions below are that recording split into its parts.  Like the
 * recording, every function only ENQUEUES work on the caller's HIP stream and
 * never synchronises with the host; unlike wgpu (which panics through
 * on_uncaptured_error, src/main.rs:711-717) errors come back as a negative
 * status and a message from mrs_last_error_string().
 *
 * Conventions
 *   - all pointers named *_dev / keys* / vals* are DEVICE pointers owned by the
 *     caller; the library never frees or reallocates them;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - the device of the context must be the calling thread's current device;
 *   - a plan may be used by one host thread and one stream at a time; distinct
 *     plans are independent.
 */
#ifndef MRS_H
#define MRS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRS_VERSION_MAJOR 0
#define MRS_VERSION_MINOR 1

/* status codes */
#define MRS_OK 0
#define MRS_ERR_INVALID_ARGUMENT (-1)
#define MRS_ERR_HIP (-2)
#define MRS_ERR_UNSUPPORTED (-3)
#define MRS_ERR_OUT_OF_MEMORY (-4)
#define MRS_ERR_DEVICE_TIMEOUT (-5)
#define MRS_ERR_PEER_FAILED (-6) /* in-process transport: another rank of the group returned an error
                                    from (or before) a collective; the group is dead from then on */

/* pass algorithm */
#define MRS_ALGO_THREE_KERNEL 0 /* histogram -> prefix sum -> scatter, as src/main.rs:594-665 */
#define MRS_ALGO_ONESWEEP 1     /* one histogram kernel + one chained-scan kernel per digit   */
#define MRS_ALGO_HYBRID 2       /* (u32 key, 4-byte payload) pairs on 25..32 key bits, n <= 2^30:
                                   three moves of every pair instead of the four of
                                   src/main.rs:594 -- two chained-scan scatters by the top two
                                   digits (the second one bucket by bucket), then every one of
                                   the 65536 sub-buckets sorted inside a workgroup's LDS and
                                   written back sequentially.  An exact 16-bit histogram taken
                                   first lets the DEVICE decide sub-bucket by sub-bucket: one
                                   that does not fit a workgroup (equal keys, a cluster) is
                                   sorted where it lies by two more chained-scan passes on the
                                   remaining bits.  When more than 256 sub-buckets are oversized,
                                   or they hold over a quarter of the pairs (codes of points on
                                   surfaces, which is what triangle centroids are,
                                   src/main.rs:114-173), the same three moves follow tables the
                                   device cuts to the keys' density instead of fixed bit fields
                                   (28..32 sorted bits from bit 0, up to about 2^28.5 pairs); only
                                   keys that cannot be cut that way either (nearly all equal)
                                   fall back to the four MRS_ALGO_ONESWEEP passes (also more than
                                   2^29.1 pairs where over 16 key bits lie below the leading 16).  The result, its side and its stability are
                                   those of MRS_ALGO_ONESWEEP every time.  Costs about 45 MiB
                                   more temporary memory                                        */

/* byte sizes of the reference buffer elements */
#define MRS_MORTON_UNIFORMS_BYTES 51248 /* MortonCodeConstants, morton_code.wgsl:19-26 */
#define MRS_VERTEX_STRIDE 48            /* Vertex, morton_code.wgsl:1-9, src/types.rs:12-17 */
#define MRS_TRIANGLE_STRIDE 16          /* TriangleIndices, src/types.rs:30-36            */

typedef struct mrs_context mrs_context;
typedef struct mrs_plan mrs_plan;

/* Replaces the compile-time constants of src/constants.rs:1-21 and the buffer
 * sizing of create_all_buffers / create_prefix_buffers (src/main.rs:31-85,
 * 236-253). */
typedef struct mrs_plan_desc {
    uint64_t n;            /* capacity in elements (NUM_TRIANGLES); 0 < n < 2^32             */
    int32_t key_words;     /* 1: u32 key.  2: u64 key stored (lo, hi), src/main.rs:60-61   */
    int32_t payload_bytes; /* 0, 4 (u32 index) or 16 (TriangleIndices)                      */
    int32_t digit_bits;    /* BITS_PER_PASS: 4 or 8                                         */
    int32_t begin_bit;     /* first key bit sorted on (0)                                   */
    int32_t end_bit;       /* one past the last key bit (32*key_words = NUM_PASSES*8 bits)  */
    int32_t algo;          /* MRS_ALGO_*                                                    */
    int32_t tile_items;    /* keys per workgroup: 0 = default chosen from n and the element
                              sizes, or 256, 1024, 4096, 8192 (u32 keys with payload <= 4
                              bytes, or 16-byte payload), 16384 (payload <= 4 bytes), 32768
                              (u32 keys with payload <= 4 bytes).  256 with
                              MRS_ALGO_THREE_KERNEL gives the reference's histogram
                              layout [256][ceil(n/256)] exactly                             */
} mrs_plan_desc;

/* Sizes a plan reports (mrs_plan_get_info). */
typedef struct mrs_plan_info {
    uint32_t num_passes;     /* NUM_PASSES                                                  */
    uint32_t radix;          /* HISTOGRAM_SIZE = 1 << digit_bits                            */
    uint32_t tile_items;     /* resolved keys per workgroup                                 */
    uint32_t threads;        /* workgroup size                                              */
    uint64_t max_tiles;      /* ceil(n / tile_items) = number_of_workgroups                 */
    uint64_t histogram_len;  /* u32 entries of the block histogram (three-kernel), radix*W  */
    uint64_t temp_bytes;     /* device bytes the plan owns                                  */
} mrs_plan_info;

/* ---- library ------------------------------------------------------------ */
/* Message of the calling thread's last failing call ("" if none). */
const char* mrs_last_error_string(void);
/* MRS_VERSION_MAJOR << 16 | MRS_VERSION_MINOR */
int mrs_version(void);

/* ---- context: replaces Instance/adapter/device acquisition, src/main.rs:692-707 */
int mrs_context_create(int device_id, mrs_context** out);
int mrs_context_destroy(mrs_context* ctx);
/* What the context found on its device (the counterpart of reading adapter info and
 * limits, src/main.rs:697-707).  attribute: MRS_ATTR_*. */
#define MRS_ATTR_COMPUTE_UNITS 0 /* compute units of the device                               */
#define MRS_ATTR_RANK_MODE 1     /* how digit passes rank keys inside a wave: 0 = ballot
                                    matching, 1 = returning LDS atomics (chosen by a device
                                    check at creation; can be set to 0, and back to 1 where
                                    the check passed)                                           */
#define MRS_ATTR_DIE_AWARE_TICKETS 2 /* 1 if a large grid was seen to land on all 8 dies (XCDs) in
                                    comparable shares: onesweep passes then hand blocks of
                                    consecutive tiles to workgroups of one die               */
#define MRS_ATTR_NOALLOC_TIMEOUT 3 /* synchronises the device; 1 if a chained-scan workgroup of an
                                    mrs_sort_pairs_noalloc call gave up waiting since the last
                                    query (that call's output is invalid); reading clears it     */
/* Range markers for an external profiler, the counterpart of the reference's named compute passes
 * (src/main.rs:583-662).  1: every launch a sort or encode call of THIS PROCESS enqueues is
 * bracketed by a roctx range (roctxRangePush/Pop, bound at run time from
 * librocprofiler-sdk-roctx.so or libroctx64.so) named as the reference names the pass it stands
 * for: "Morton Coding", "Histogram calculation", "Large Prefix Sum part 1", "Small Prefix Sum",
 * "Large Prefix Sum part 2", "Indexing"; launches without a counterpart there are named for what
 * they do ("Clear", "Local sort", "Gather").  `rocprofv3 --marker-trace --kernel-trace` then reads
 * like the reference's NSight capture.  Setting it fails with MRS_ERR_UNSUPPORTED if no roctx
 * library can be loaded. */
#define MRS_ATTR_ROCTX_RANGES 4
/* Aids for tests and A/B runs.  A plan copies them from its context WHEN IT IS CREATED; sort calls
 * never look at the context or the environment again.  All default to 0. */
#define MRS_ATTR_TEST_HYBRID_MIN_N 5          /* MRS_ALGO_HYBRID plans offer the three-move schedule from
                                                 this many pairs on (default: 2^26 .. 2^27 by key width),
                                                 so that it runs on inputs a checker can follow          */
#define MRS_ATTR_TEST_HYBRID_LEADING_BITS 6   /* 14..16: bits the two scatters of that schedule take     */
#define MRS_ATTR_TEST_HYBRID_FORCE_FALLBACK 7 /* 1: the device takes the plan's ordinary passes whatever
                                                 the keys are                                            */
#define MRS_ATTR_TEST_NO_SMALL_SORT 8         /* 1: inputs of at most 4096 elements take the general
                                                 multi-launch path, not the one-workgroup sort          */
#define MRS_ATTR_TEST_NO_PASS_SKIP 9          /* 1: 64-bit keys run every pass (src/main.rs:594-665)     */
#define MRS_ATTR_TEST_WIDE_STATUS 10          /* 1: chained-scan plans use the 64-bit tile status words
                                                 of inputs above 2^30 elements at any size              */
#define MRS_ATTR_TEST_HYBRID_FORCE_ADAPTIVE 11 /* 1: the device passes over the fixed sub-buckets and cuts
                                                 buckets and groups to the keys (MRS_SCHEDULE_MSD_ADAPTIVE)
                                                 wherever that schedule is offered                       */
#define MRS_ATTR_TEST_HYBRID_NO_ADAPTIVE 12   /* 1: that schedule is not offered                          */
#define MRS_ATTR_TEST_HYBRID_BIG_LOCAL 14     /* 1: the local sort runs its workgroups of 1024 threads
                                                 (20480 pairs each, what inputs of 2^29.1 to 2^30 pairs
                                                 take) at any size where a pair is staged in 6 bytes     */
#define MRS_ATTR_TEST_HYBRID_CAPACITY 13      /* pairs a workgroup of the local sort is taken to hold
                                                 (0: what it holds, 7680 or 10240): smaller values make
                                                 oversized sub-buckets and many groups out of small inputs */
int mrs_context_get_attribute(const mrs_context* ctx, int attribute, int64_t* value);
/* MRS_ATTR_DIE_AWARE_TICKETS can be cleared (value 0) and, where the creation-time check passed,
 * set again.  With die-aware tickets a pass makes progress only if it keeps getting workgroup
 * slots on every die.  The library itself never runs two such passes at once on a device: a sort
 * call that finds the previous die-aware call of the process on that device still running on
 * ANOTHER stream uses a single ticket list for its own passes (which needs one free slot
 * anywhere), so sorts on several streams or contexts can overlap freely.  What the library cannot
 * see is a foreign kernel that occupies every compute unit of a die for seconds, or sorts replayed
 * concurrently from captured graphs: clear the attribute for such uses (a starved pass ends in
 * MRS_ERR_DEVICE_TIMEOUT after 4 s instead of completing). */
int mrs_context_set_attribute(mrs_context* ctx, int attribute, int64_t value);

/* ---- plan: replaces create_all_buffers' temporaries (histogram + prefix_sums,
 * src/main.rs:45-51,236-253) and the per-pass uniform buffers (src/main.rs:361-374) */
int mrs_plan_create(mrs_context* ctx, const mrs_plan_desc* desc, mrs_plan** out);
int mrs_plan_destroy(mrs_plan* plan);
int mrs_plan_get_info(const mrs_plan* plan, mrs_plan_info* info);
/* Synchronises `stream` and reports MRS_ERR_DEVICE_TIMEOUT if a chained-scan
 * kernel gave up waiting for a predecessor tile since the last check (the
 * output of that call is invalid, and later passes of the plan give up early
 * until the flag has been read).  The sort calls only enqueue work and cannot
 * report it themselves: a production caller checks once per batch, at a point
 * where it synchronises anyway. */
int mrs_plan_check(mrs_plan* plan, void* stream);
/* Synchronises `stream` and reports, for the most recent mrs_sort_pairs call on
 * an MRS_ALGO_ONESWEEP plan, which passes the device left out: bit p of
 * *left_out_mask is set when pass p did not run (64-bit keys whose digit p is
 * the same for every key; see mrs_sort_pairs).  0 for other plans, for 32-bit
 * keys and for inputs sorted by the one-workgroup path.  For tests and
 * harnesses (the reference has no counterpart: it always runs every pass,
 * src/main.rs:594-665). */
int mrs_plan_get_passes_left_out(mrs_plan* plan, void* stream, uint32_t* left_out_mask);
/* Synchronises `stream` and reports which pass schedule the most recent sort call on `plan`
 * ran: MRS_SCHEDULE_PASSES (every plan but MRS_ALGO_HYBRID ones, and inputs too small for the
 * other schedule), MRS_SCHEDULE_MSD (an MRS_ALGO_HYBRID plan whose keys the device found spread
 * well enough), MRS_SCHEDULE_MSD_ADAPTIVE (keys that crowd into few of the fixed sub-buckets, e.g.
 * points on surfaces: the same three moves, with the two scatters following tables the device cut
 * to the keys' density -- at most 256 buckets of whole 12-bit cells, each cut into at most 256
 * groups that fit a workgroup; offered for 28 to 32 sorted bits from bit 0) or
 * MRS_SCHEDULE_MSD_REFUSED (the device found too many or too full oversized sub-buckets, and no such
 * cut either, and ran the ordinary passes).  *largest_sub_bucket (may be NULL) receives the size of
 * the largest of the sub-buckets the device counted, 0 when it counted none.  For tests and
 * harnesses; the reference always runs its one schedule (src/main.rs:594-665). */
/* Host only, no GPU needed: how an MRS_ALGO_HYBRID plan for key bits [begin_bit, end_bit) would
 * go about a sort of n pairs.  `window`: NULL, or two key values every key lies between (what the
 * sharded sort knows of a rank's piece).  offered = 0: the plan's ordinary passes run on the
 * host's decision (n too small or too large for sub-buckets of a workgroup's size, or no split of
 * the window leaves the local sort 16 bits or fewer).  Otherwise the two scatters take
 * leading_bits (14..16) bits from bit `shift` up -- of the key, or, windowed, of
 * (key >> shift) - base -- and the sort inside LDS the local_bits bits [begin_bit, shift), in
 * workgroups that hold local_capacity pairs (7680 staged in 8 bytes each; 10240, or 20480 in the
 * workgroups of 1024 threads that the largest inputs take, staged in 6).  The
 * device still has the last word at run time (see mrs_plan_get_schedule, mrs_plan_get_oversized). */
typedef struct mrs_hybrid_split {
    int32_t offered;
    int32_t windowed;
    uint32_t leading_bits;
    uint32_t shift;
    uint32_t base;
    uint32_t local_bits;
    uint32_t local_capacity; /* pairs a workgroup of the local sort holds: 7680, 10240 or 20480 */
} mrs_hybrid_split;
int mrs_hybrid_split_for(uint64_t n, int32_t begin_bit, int32_t end_bit, const uint64_t* window,
                         mrs_hybrid_split* out);
#define MRS_SCHEDULE_PASSES 0
#define MRS_SCHEDULE_MSD 1
#define MRS_SCHEDULE_MSD_REFUSED 2
#define MRS_SCHEDULE_MSD_ADAPTIVE 3
int mrs_plan_get_schedule(mrs_plan* plan, void* stream, int32_t* schedule, uint32_t* largest_sub_bucket);
/* Synchronises `stream`.  For the most recent sort call on an MRS_ALGO_HYBRID plan that ran
 * MRS_SCHEDULE_MSD or MRS_SCHEDULE_MSD_ADAPTIVE: how many sub-buckets (adaptive: single fine cells
 * of a bucket, that is runs of nearly equal keys) did not fit a workgroup of the local sort and were sorted
 * where they lay by the two extra chained-scan passes (*sub_buckets, at most 256), how many pairs
 * they held (*pairs, at most n / 4), and what a workgroup held in that call (*capacity: 10240 pairs
 * when a pair could be staged in 6 bytes, its low 16 key bits and its payload, else 7680).  Zeros
 * otherwise.  Any pointer may be NULL.  For tests and harnesses; no counterpart in the reference. */
int mrs_plan_get_oversized(mrs_plan* plan, void* stream, uint32_t* sub_buckets, uint64_t* pairs,
                           uint32_t* capacity);
/* Synchronises `stream`.  For the most recent sort call on an MRS_ALGO_HYBRID plan that ran
 * MRS_SCHEDULE_MSD_ADAPTIVE: into how many buckets the first scatter divided the keys (*buckets, at
 * most 256) and into how many groups the second one (*groups, at most 65536; each was sorted by one
 * visit of a workgroup of the local sort, or, if it was a single oversized cell, by the two extra
 * passes).  Zeros otherwise.  Either pointer may be NULL.  For tests and harnesses; no counterpart
 * in the reference. */
int mrs_plan_get_adaptive(mrs_plan* plan, void* stream, uint32_t* buckets, uint32_t* groups);

/* Measurement hook (the reference has no timers; its passes are named for an
 * external profiler, src/main.rs:583-662).  With timing enabled, mrs_sort_pairs
 * records a HIP event on the caller's stream before its first launch and after
 * every launch.  mrs_plan_get_timings synchronises on the last event and
 * returns, for each launch of the most recent mrs_sort_pairs call, its
 * duration in milliseconds and a label of at most 31 characters
 * ("memset", "global_histogram", "histogram_scan", "onesweep_pass", "block_histogram", "scan",
 * "scatter", "small_sort", "gather"; MRS_ALGO_HYBRID plans: "msd_histogram", "msd_scan",
 * "fallback_histogram", "msd_pass_a", "msd_adaptive_count", "msd_pass_b", "msd_local_sort",
 * "msd_oversized_histogram", "msd_oversized_pass", "fallback_pass"; one launch carries the same
 * pass of every schedule the device can choose, so e.g. "msd_pass_b" is also the first ordinary
 * pass of a refused call).  labels_out holds cap * 32 bytes. */
int mrs_plan_set_timing(mrs_plan* plan, int enable);
int mrs_plan_get_timings(mrs_plan* plan, int cap, float* ms_out, char* labels_out, int* count_out);

/* ---- Morton encode ------------------------------------------------------- */
/* Replaces the "Morton Coding" dispatch of morton_code_unrolled
 * (src/main.rs:581-592, src/shaders/morton_code.wgsl:90-335).  Buffers have the
 * reference's byte layouts: `uniforms_dev` = MortonCodeConstants blob
 * (MRS_MORTON_UNIFORMS_BYTES), `vertices_dev` stride 48 B with position at byte
 * 0, `triangles_dev` stride 16 B; writes codes_dev[2t] = lo, [2t+1] = hi. */
int mrs_morton_encode_lut(mrs_context* ctx, const void* uniforms_dev,
                          const void* vertices_dev, uint64_t num_vertices,
                          const void* triangles_dev, uint64_t n,
                          uint32_t* codes_dev, void* stream);
/* 30-bit Morton key of a point cloud (xyz f32, 12 B per point), payload = the
 * point's index.  No counterpart in the reference: this is the key shape the
 * throughput metric is quoted on.  mn/mx are HOST pointers to 3 floats (the
 * scene bounds, as create_scene computes them on the CPU, src/main.rs:148-154).
 * idx_dev may be NULL: only the keys are written (8 of the 20 bytes per point
 * less), for a sort that makes the indices up itself (mrs_sort_pairs_indexed). */
int mrs_morton_encode30(mrs_context* ctx, const float* xyz_dev, uint64_t n,
                        const float* mn, const float* mx, uint32_t* keys_dev,
                        uint32_t* idx_dev, void* stream);

/* ---- the sort ------------------------------------------------------------- */
/* Replaces the NUM_PASSES loop of src/main.rs:594-665 with its ping-pong bind
 * groups (src/main.rs:515-576): pass 0 reads (keys_a, vals_a) and writes
 * (keys_b, vals_b), pass 1 the other way round, and so on.  *result_in_b is set
 * (on the host, immediately) to 1 when the sorted data ends in the B buffers,
 * i.e. when the number of passes is odd (src/main.rs:819-823).  n <= plan n.
 * vals_* may be NULL when payload_bytes == 0.  The pair of buffers that does
 * not hold the result is scratch.  Buffers need the alignment of their element
 * (4 or 8 bytes for keys, payload_bytes for the payload), nothing more: the
 * kernels' 16-byte loads run in gfx950's unaligned access mode.  MRS_ALGO_ONESWEEP plans for 64-bit keys
 * leave out, on the device, passes whose digit is the same for every key (the
 * constant high bytes of a short code); they are left out in pairs, so the
 * result side stays the one reported here.
 * The call only enqueues work (one clearing kernel, the histogram, its scan, the
 * passes), so it may be recorded by a stream capture and replayed as a HIP graph
 * on new buffer contents; the same holds for mrs_sort_pairs_indexed,
 * mrs_sort_pairs_noalloc and mrs_sort_gather.  Replays of one plan must not
 * overlap each other, and with die-aware tickets (mrs_context_set_attribute)
 * not another sort of the device either. */
int mrs_sort_pairs(mrs_plan* plan, uint64_t n, void* keys_a, void* vals_a,
                   void* keys_b, void* vals_b, void* stream, int* result_in_b);

/* mrs_sort_pairs of (key, source index) pairs whose input payload is not stored anywhere: the
 * payload of element i IS i, and the first pass makes it up instead of reading it (4 of the 16
 * bytes per pair of that pass; no iota has to be written before the sort either).  The result
 * is what mrs_sort_pairs gives for vals_a = 0, 1, ..., n-1.  `plan` must have payload_bytes == 4;
 * vals_a is ping-pong space like vals_b and is never read before the library has written it.
 * No counterpart in the reference, which carries the 16-byte TriangleIndices through every pass
 * (src/shaders/radix_sort_index.wgsl:281-307); mrs_sort_gather is built on this call. */
int mrs_sort_pairs_indexed(mrs_plan* plan, uint64_t n, void* keys_a, uint32_t* vals_a,
                           void* keys_b, uint32_t* vals_b, void* stream, int* result_in_b);

/* ---- single stages of one pass (MRS_ALGO_THREE_KERNEL plans) --------------- */
/* radix_sort_compute_histogram (src/shaders/radix_sort_histogram.wgsl:42-60):
 * hist_dev[W*d + b] = number of keys of block b whose digit `pass` equals d. */
int mrs_histogram(mrs_plan* plan, uint64_t n, const void* keys_dev, uint32_t pass,
                  uint32_t* hist_dev, void* stream);
/* The whole prefix-sum stage (radix_sort_block_sum_large / _small / _after,
 * src/main.rs:604-655): in-place exclusive scan of hist_dev[0 .. radix*W). */
int mrs_scan(mrs_plan* plan, uint64_t n, uint32_t* hist_dev, void* stream);
/* radix_sort_index (src/shaders/radix_sort_index.wgsl:255-308): stable scatter
 * by digit `pass` using the scanned histogram. */
int mrs_scatter(mrs_plan* plan, uint64_t n, const void* keys_in, const void* vals_in,
                void* keys_out, void* vals_out, const uint32_t* scanned_hist_dev,
                uint32_t pass, void* stream);

/* mrs_morton_encode30 that also leaves the digit histograms of the keys it writes in `plan` (an
 * MRS_ALGO_ONESWEEP plan for 32-bit keys), so that mrs_sort_pairs_primed(plan, n, keys_dev, ...)
 * can skip its own read of the keys (the histogram launch before the passes, src/main.rs:594-603
 * being the reference's per-pass counterpart).  The plan's memory is used, so the call must be
 * ordered before the sort on one stream.  Any other call that takes the plan drops the histograms.
 * On an MRS_ALGO_HYBRID plan and an n that takes the three-move schedule the call is
 * mrs_morton_encode30: that schedule's own single read of the keys counts everything both of its
 * pass sequences need, and histograms left here would only cost the encoder time. */
int mrs_morton_encode30_for_sort(mrs_plan* plan, const float* xyz_dev, uint64_t n, const float* mn,
                                 const float* mx, uint32_t* keys_dev, uint32_t* idx_dev, void* stream);
/* mrs_sort_pairs that consumes the histograms mrs_morton_encode30_for_sort left in the plan.
 * Fails with MRS_ERR_INVALID_ARGUMENT unless it is the first call on the plan after that encoder
 * call and keys_a and n are the encoder's keys_dev and n; the caller vouches that the keys have
 * not been written since.  mrs_sort_pairs itself never uses left-over histograms.  On an
 * MRS_ALGO_HYBRID plan the left-over histograms only serve inputs that the plan sorts with its
 * ordinary passes by size; otherwise the call is mrs_sort_pairs. */
int mrs_sort_pairs_primed(mrs_plan* plan, uint64_t n, void* keys_a, void* vals_a, void* keys_b,
                          void* vals_b, void* stream, int* result_in_b);

/* mrs_sort_pairs for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Sorts n (u32 key, u32 payload) pairs on
 * key bits [begin_bit, end_bit) with 8-bit digits and chained-scan passes, exactly as a plan
 * created for that n would, but takes no plan: the temporary memory (tile status words, digit
 * histograms, ticket counters) is a static array of the library's code object, about 8 MiB that
 * every GPU the library runs on holds from the load of the library until the process ends.  The
 * call makes no hipMalloc/hipFree and no other allocating call; n may be at most 2^26
 * (MRS_NOALLOC_MAX_N), larger inputs are rejected with MRS_ERR_INVALID_ARGUMENT.  All calls on
 * one GPU share that GPU's array: calls that overlap in time on one GPU (two streams, two
 * threads, two contexts of one device) race and give wrong results; calls that follow each
 * other on one stream are safe, each clears what it needs.  A look-back that timed out is
 * reported through mrs_context_get_attribute(ctx, MRS_ATTR_NOALLOC_TIMEOUT, ...). */
#define MRS_NOALLOC_MAX_N (1ull << 26)
int mrs_sort_pairs_noalloc(mrs_context* ctx, uint64_t n, uint32_t* keys_a, uint32_t* vals_a,
                           uint32_t* keys_b, uint32_t* vals_b, int32_t begin_bit, int32_t end_bit,
                           void* stream, int* result_in_b);

/* Sort by key, moving each wide record once: the variant the reference's own comments
 * anticipate for TriangleIndices (src/types.rs:26-36, src/shaders/radix_sort_index.wgsl:3-6)
 * instead of carrying 16 bytes through every pass (radix_sort_index.wgsl:281-307).  `plan` must
 * have payload_bytes == 4: (key, source index) pairs are sorted exactly as by mrs_sort_pairs with
 * index_a = 0..n-1 (mrs_sort_pairs_indexed: the indices are made up by the first pass, not
 * written and read back; index_a, index_b: u32[n] scratch), then
 * records_out[j] = records_in[index[j]] for records of record_bytes (a positive multiple of 4).
 * Keys end in keys_a or keys_b as reported by result_in_b; records_out must not alias records_in.
 * The sorted source indices stay in index_a or index_b (same side as the keys).
 *
 * This is also the layout a BVH builder reads (the consumer the reference's TriangleIndices
 * anticipates with its commented-out `node_index`, src/types.rs:33,
 * src/shaders/radix_sort_index.wgsl:11-17).  With S the side result_in_b names, for leaf i in [0, n):
 *   keys_S[i]      the i-th smallest code, equal codes in source order (the sort is stable);
 *   index_S[i]     the source position of that element: the leaf -> primitive map;
 *   records_out[i] its record (e.g. the triangle's vertex indices), so leaf i needs no indirection.
 * A Karras-style builder derives the tree from keys_S alone: delta(i, j) = clz(keys_S[i] ^ keys_S[j]),
 * and for equal codes KEYBITS + clz(i ^ j), which is well defined because leaf numbers are unique and
 * stability makes the order of equal codes reproducible.  Internal-node arrays (n - 1 nodes) are the
 * builder's own; nothing here allocates them. */
int mrs_sort_gather(mrs_plan* plan, uint64_t n, void* keys_a, void* keys_b, uint32_t* index_a,
                    uint32_t* index_b, const void* records_in, void* records_out,
                    uint32_t record_bytes, void* stream, int* result_in_b);

/* ---- multi-GPU building blocks (sort sharded by top-digit bucket) ---------- */
/* counts_dev[b] (u64, 1 << bits entries) = number of keys whose bits
 * [shift, shift+bits) equal b.  Input to the all-reduce of the global digit
 * histogram. */
int mrs_bucket_histogram(mrs_context* ctx, const void* keys_dev, uint64_t n,
                         int32_t key_words, uint32_t shift, uint32_t bits,
                         uint64_t* counts_dev, void* stream);
/* Stable partition of (key, payload) pairs into `num_parts` <= 16 contiguous
 * groups: an element goes to group map_dev[(key >> shift) & ((1<<bits)-1)]
 * (u8 table, 1 << bits entries, bits <= 12).  Uses the plan's onesweep machinery (the plan
 * must be MRS_ALGO_ONESWEEP).  part_counts_dev (u32[16], may be NULL) receives
 * the group sizes. */
int mrs_partition(mrs_plan* plan, uint64_t n, const void* keys_in, const void* vals_in,
                  void* keys_out, void* vals_out, uint32_t shift, uint32_t bits,
                  const uint8_t* map_dev, uint32_t num_parts,
                  uint32_t* part_counts_dev, void* stream);

/* ---- the sort sharded over the GPUs of a node ------------------------------------------
 * No counterpart in the reference, which records everything for one device (src/main.rs:376).
 * One rank per GPU; rank r holds a slice of the (key, payload) pairs, in rank order.  A call to
 * mrs_dist_sort_pairs on every rank leaves rank r with the r-th slice of the globally sorted
 * order (stable: equal keys keep the order rank 0's slice, rank 1's slice, ...):
 *   1. histogram of the top 12 key bits of the local slice (mrs_bucket_histogram's kernel);
 *   2. ncclAllReduce(sum) of the bucket counts -> global digit histogram, and ncclAllGather of the
 *      per-rank rows in the same group, then ONE host round trip for both;
 *   3. mrs_splitter (host): the key values at which the global order is cut into `world` slices.
 *      A cut that would fall too far inside a bucket makes the splitter ask for the next digit of
 *      that bucket's keys (one more histogram + all-gather + round trip per level; only skewed
 *      inputs get there), and a cut inside a run of equal keys is placed by element order, so no
 *      key distribution can overload a rank: every rank receives total/world +- 2*tolerance pairs;
 *   4. one stable partition pass by destination (the chained-scan pass with a bucket table);
 *   5. grouped ncclSend/ncclRecv of keys and payload (all-to-all-v over xGMI), received in source
 *      rank order;
 *   6. local sort of what arrived (mrs_sort_pairs; for (u32 key, u32 payload) pairs an
 *      MRS_ALGO_HYBRID plan whose sub-buckets divide the slice's own window of the key space, the
 *      cut keys on either side of it, once a piece holds 2^26 pairs).
 * All device work is enqueued on the caller's stream. */
typedef struct mrs_comm mrs_comm;
typedef struct mrs_dist mrs_dist;

#define MRS_UNIQUE_ID_BYTES 128
/* ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to every rank (any side channel). */
int mrs_comm_get_unique_id(void* id_out);
/* ncclCommInitRank on the calling thread's current device; collective over all ranks. */
int mrs_comm_create_rccl(const void* unique_id, int32_t rank, int32_t world, mrs_comm** out);
/* Wraps a communicator the host program already owns (an ncclComm_t); not destroyed with the handle. */
int mrs_comm_wrap_rccl(void* nccl_comm, int32_t rank, int32_t world, mrs_comm** out);
/* `world` ranks inside one process, one host thread per rank: collectives are device copies behind
 * a host barrier.  Ranks may sit on different devices or share one (how the single-GPU tests run
 * the multi-rank path).  out[0 .. world) receives one handle per rank.  A rank whose
 * mrs_dist_sort_pairs fails (bad arguments, a HIP error, a count mismatch) marks the group: the
 * ranks waiting in a collective, and every later collective of the group, return
 * MRS_ERR_PEER_FAILED instead of waiting for a rank that will not come.  (Over RCCL a rank that
 * fails alone leaves its peers in the collective, as in any RCCL program.) */
int mrs_comm_create_local(int32_t world, mrs_comm** out);
int mrs_comm_destroy(mrs_comm* comm);

typedef struct mrs_dist_desc {
    uint64_t n_local_max;   /* most pairs any rank passes in one call                               */
    int32_t key_words;      /* 1 or 2, as mrs_plan_desc                                            */
    int32_t payload_bytes;  /* 0 or 4                                                              */
    int32_t end_bit;        /* keys are compared on bits [0, end_bit); higher bits must be zero    */
    int32_t exchange_chunks; /* pieces a rank's share arrives in: 1 = exchange, then sort; 2 = the
                               share is cut in two consecutive key ranges and the sort of the first
                               runs beside the exchange of the second (needs 2 * world <= 16);
                               0 = chosen by the library (2 for 2..8 ranks)                         */
    uint64_t tolerance;     /* elements a cut may miss its ideal position by before the splitter
                               refines it; 0 = n_local_max / 256                                   */
} mrs_dist_desc;

typedef struct mrs_dist_info {
    uint64_t recv_capacity;   /* most pairs a rank can receive: n_local_max + 2 * tolerance + 2    */
    uint64_t device_bytes;    /* device memory the handle owns (partition, receive and sort buffers) */
    uint32_t last_levels;     /* histograms the splitter needed in the last call (1 = top digit only) */
    uint32_t last_exact_cuts; /* cuts of the last call placed inside a run of equal keys           */
    uint64_t last_n_out;
    float last_ms[4];         /* last call with timing on: histogram + collectives + splitter,
                                 partition, exchange (all pieces; with 2 pieces the sort of the
                                 first overlaps it), local sort after the last piece arrived       */
    float last_sort_pass_ms;  /* last call with timing on: mean duration of one digit pass of the
                                 local sort (HIP events on the launch stream)                      */
    uint32_t last_sort_passes;
    uint32_t last_sort_schedule[2]; /* MRS_SCHEDULE_* of the local sort of each piece of the last call
                                       (mrs_dist_get_info waits for the device to read them)            */
    uint64_t last_sort_piece_pairs; /* pairs of the piece whose digit passes last_sort_pass_ms times: the
                                       LAST piece of the rank's share (about last_n_out / exchange_chunks),
                                       so bytes per launch = 16 * this, not 16 * last_n_out             */
} mrs_dist_info;

int mrs_dist_create(mrs_context* ctx, mrs_comm* comm, const mrs_dist_desc* desc, mrs_dist** out);
int mrs_dist_destroy(mrs_dist* dist);
int mrs_dist_set_timing(mrs_dist* dist, int enable);
int mrs_dist_get_info(mrs_dist* dist, mrs_dist_info* info);
/* Collective.  keys_in / vals_in: this rank's n_local pairs (left untouched).  On return
 * *keys_out / *vals_out point at this rank's slice of the sorted order inside memory the handle
 * owns (valid until the next call on the handle) and *n_out is its length; the data is complete
 * once `stream` has caught up.  The host blocks once (or once per splitter level) inside. */
int mrs_dist_sort_pairs(mrs_dist* dist, uint64_t n_local, const void* keys_in, const void* vals_in,
                        void** keys_out, void** vals_out, uint64_t* n_out, void* stream);
/* Synchronises `stream`; MRS_ERR_DEVICE_TIMEOUT as mrs_plan_check, for both passes the call ran. */
int mrs_dist_check(mrs_dist* dist, void* stream);

/* The splitter of step 3 on its own (host only, no GPU needed): what mrs_dist_sort_pairs uses, and
 * what a host that brings its own histograms and exchange can use. */
typedef struct mrs_splitter mrs_splitter;
typedef struct mrs_split_result {
    int32_t world;
    int32_t levels;             /* histograms consumed                                            */
    uint64_t total;             /* pairs over all ranks                                           */
    uint64_t rank_size[16];
    uint64_t cut_key[15];       /* cut r lies between destinations r and r + 1: keys below
                                   cut_key[r] go to destinations <= r                              */
    uint32_t exact_mask;        /* bit r: keys EQUAL to cut_key[r] are split by element order     */
    uint32_t reserved;
    uint64_t send_count[16][16]; /* [source][destination], in the order of the source's slice
                                    after a stable partition by destination                        */
} mrs_split_result;
int mrs_splitter_create(int32_t world, int32_t key_bits, uint64_t tolerance, mrs_splitter** out);
int mrs_splitter_destroy(mrs_splitter* sp);
/* The histogram wanted next: counts of digit (key >> *shift) & ((1 << *bits) - 1) over the keys
 * with (key >> (*shift + *bits)) == prefixes[j], for each j < *num_prefixes (at most 15; one
 * prefix 0 at the first level, which covers every key).  *bits == 0: all cuts are placed. */
int mrs_splitter_next(mrs_splitter* sp, uint32_t* shift, uint32_t* bits, uint32_t* num_prefixes,
                      uint64_t* prefixes);
/* counts[(rank * num_prefixes + j) << bits | digit] for the histogram mrs_splitter_next described. */
int mrs_splitter_feed(mrs_splitter* sp, const uint64_t* counts);
int mrs_splitter_get(const mrs_splitter* sp, mrs_split_result* out);

/* ---- input generator (stands in for create_scene's RNG, src/main.rs:114-161) */
/* xyz_dev[3i+c] = top 24 bits of hash(seed, c, i) * 2^-24, uniform in [0,1). */
int mrs_generate_points(mrs_context* ctx, uint64_t seed, uint64_t n, float* xyz_dev,
                        void* stream);
/* out_dev[i] = hash(seed, stream_id, i), a uniform u32. */
int mrs_generate_u32(mrs_context* ctx, uint64_t seed, uint64_t stream_id, uint64_t n,
                     uint32_t* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRS_H */
