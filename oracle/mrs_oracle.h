/*
 * CPU oracle for the Morton-encode + LSD radix sort path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm, used as the checker by tests/, by __graft_entry__.smoke() and by
 * bench.py's cpu_baseline leg.  Nothing in the shipped library links, loads or
 * calls it.
 *
 * Parity pin: the reference (Rust + WGSL) cannot be built or run here, and it
 * ships no golden vectors.  The oracle is pinned by (1) the known-answer and
 * property tests of the reference's own src/tests.rs, restated in
 * orc_selftest_* below, and (2) agreement with an independent stable sort
 * (orc_stable_sort_pairs), which is the unique correct answer of a stable
 * key/value sort.  LUT *contents* of the extended Morton code come from a
 * crate that is not vendored (extended_morton_coder 0.1.0): parity unpinned
 * for those contents; the encoder is checked LUT-agnostically.
 *
 * All citations are relative to the reference repository root.
 */
#ifndef MRS_ORACLE_H
#define MRS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- host helpers (src/utilities.rs:10-55) -------------------------------- */
uint64_t orc_div_ceil(uint64_t l, uint64_t r);
uint64_t orc_num_workgroups(uint64_t element_count, uint64_t workgroup_size);
uint64_t orc_round_up(uint64_t l, uint64_t r);

/* Level sizes of the hierarchical prefix sum (src/main.rs:201-233).
 * Writes at most `cap` entries to `out`, returns the number of levels. */
int orc_prefix_buffer_sizes(uint64_t num_elements, uint64_t* out, int cap);

/* ---- digit selection ------------------------------------------------------- */
/* 8-bit digit of a (lo,hi) key (radix_sort_histogram.wgsl:28-36,
 * radix_sort_index.wgsl:37-45, src/tests.rs:17-24). */
uint32_t orc_select_digit8(uint32_t pass_number, uint32_t lo, uint32_t hi);
/* 6-bit digit variant across the lo/hi boundary (src/tests.rs:7-15). */
uint32_t orc_select_digit6(uint32_t pass_number, uint32_t lo, uint32_t hi);
/* Generic: `bits` bits starting at bit `shift` of a key of `key_words` u32
 * words (word 0 = least significant). */
uint32_t orc_digit(const uint32_t* key, int key_words, uint32_t shift, uint32_t bits);
/* select_bits of src/main.rs:1042-1051. */
uint64_t orc_select_bits(uint64_t value, uint32_t i);

/* ---- one radix pass, reference structure ----------------------------------- */
/* Per-block digit histogram, digit-major: hist[W*d + b]
 * (radix_sort_histogram.wgsl:42-60).  W = ceil(n/tile).  hist has
 * (1<<bits)*W entries. */
void orc_block_histogram(const uint32_t* keys, uint64_t n, int key_words,
                         uint32_t shift, uint32_t bits, uint32_t tile,
                         uint32_t* hist);
/* In-place exclusive scan of a flat u32 array; order = outer digit, inner
 * block (src/main.rs:940-946).  Wraps modulo 2^32 like the shaders. */
void orc_exclusive_scan(uint32_t* vals, uint64_t len);
/* The hierarchical scan exactly as dispatched (block_sum_large up-sweep,
 * block_sum_small, block_sum_large_after; src/main.rs:604-655).  `len` must be
 * a multiple of 256.  Result must equal orc_exclusive_scan. */
void orc_hierarchical_scan(uint32_t* vals, uint64_t len);
/* Ballot/bitmap rank of one 256-lane block (radix_sort_index.wgsl:185-253,
 * src/tests.rs:30-79): local_offset[t] = #{u <= t : digit(u)==digit(t)} - 1.
 * Lanes >= nvalid carry digit 16000 (radix_sort_index.wgsl:267). */
void orc_ballot_rank_256(const uint32_t* digits, uint32_t nvalid,
                         uint32_t* local_offset);
/* Stable scatter of one pass: rank within block + scanned[W*d + b]
 * (radix_sort_index.wgsl:280-307), payload read at the GLOBAL index.
 * payload_words may be 0. */
void orc_scatter(const uint32_t* keys_in, const uint32_t* vals_in,
                 uint32_t* keys_out, uint32_t* vals_out, uint64_t n,
                 int key_words, int payload_words, uint32_t shift,
                 uint32_t bits, uint32_t tile, const uint32_t* scanned);

/* ---- the whole sort -------------------------------------------------------- */
/* LSD radix sort with the reference's schedule and ping-pong (src/main.rs:
 * 515-576,594-665): passes over [begin_bit,end_bit) in `bits`-wide digits,
 * even passes read A write B, odd passes read B write A.  Returns 1 if the
 * result is in the B buffers, 0 if in A (src/main.rs:819-823).  `hist_tmp`
 * needs (1<<bits)*ceil(n/tile) entries. */
int orc_lsd_sort(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b,
                 uint32_t* vals_b, uint64_t n, int key_words,
                 int payload_words, uint32_t bits, uint32_t begin_bit,
                 uint32_t end_bit, uint32_t tile, uint32_t* hist_tmp);
/* Independent check: stable merge sort by the key bits [begin_bit,end_bit). */
void orc_stable_sort_pairs(const uint32_t* keys_in, const uint32_t* vals_in,
                           uint32_t* keys_out, uint32_t* vals_out, uint64_t n,
                           int key_words, int payload_words,
                           uint32_t begin_bit, uint32_t end_bit);
/* Adjacent-pair sortedness check of src/main.rs:831-864; returns the number
 * of inversions between neighbours ("times_unsorted"). */
uint64_t orc_count_unsorted(const uint32_t* keys, uint64_t n, int key_words,
                            uint32_t num_passes_of_8);

/* ---- Morton encoders ------------------------------------------------------- */
/* Byte layout of the MortonCodeConstants blob (morton_code.wgsl:19-26,
 * src/types.rs:46-54). */
#define ORC_MORTON_BLOB_BYTES 51248
#define ORC_MORTON_LUT_WORDS 4608
#define ORC_MORTON_SIZE_LUT_WORDS 8192
/* Extended Morton code per triangle (morton_code.wgsl:103-127).  vertices:
 * stride 48 B, position at byte 0; triangles: stride 16 B; codes: u32[2n].
 * Vertex indices >= num_vertices read the last vertex. */
void orc_morton_encode_lut(const void* blob, const void* vertices, uint64_t num_vertices,
                           const void* triangles, uint64_t n, uint32_t* codes);
/* 30-bit point Morton code; payload = original index.  No counterpart in the
 * reference (the metric's key shape). */
void orc_morton_encode30(const float* xyz, uint64_t n, const float* mn,
                         const float* mx, uint32_t* keys, uint32_t* idx);

/* ---- input generator (counter-based; not the reference's ChaCha8) ---------- */
uint32_t orc_rand_u32(uint64_t seed, uint64_t stream, uint64_t index);
void orc_gen_points(uint64_t seed, uint64_t n, float* xyz);
void orc_gen_u32(uint64_t seed, uint64_t stream, uint64_t n, uint32_t* out);
void orc_minmax3(const float* xyz, uint64_t n, float* mn, float* mx);

/* ---- restated unit tests of src/tests.rs; each returns 0 on success -------- */
int orc_selftest_workgroup_local_index(uint32_t num_seeds);
int orc_selftest_workgroup_calculation(void);
int orc_selftest_digit6(void);
int orc_selftest_digit8(void);

#ifdef __cplusplus
}
#endif
#endif
