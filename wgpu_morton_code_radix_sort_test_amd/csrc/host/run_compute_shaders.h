// C++ host side of the compute path: the counterpart of the reference's
// run_compute_shaders / run_headless (src/main.rs:376-688, 690-1033) on top of the
// C ABI of libmrs.so.  Scene generation, uniform packing and buffer creation
// follow create_scene / create_morton_uniforms / create_all_buffers
// (src/main.rs:114-198, 31-85).
#ifndef MRS_RUN_COMPUTE_SHADERS_H
#define MRS_RUN_COMPUTE_SHADERS_H

#include <cstdint>
#include <vector>

#include "mrs.h"

namespace mrs_host {

// src/constants.rs:1-21
struct Constants {
    uint32_t num_vertices = 1000;
    uint32_t num_triangles = 3000;
    uint32_t bits_per_pass = 8;
    uint32_t num_passes = 8;
    uint64_t rng_seed = 7;
    int algo = MRS_ALGO_ONESWEEP;
    int tile_items = 0;
    // false: the 16-byte TriangleIndices travel through every pass, as in the reference
    // (src/shaders/radix_sort_index.wgsl:281-307).  true: (code, source index) pairs are sorted and the
    // records move once at the end (mrs_sort_gather), the variant src/types.rs:26-36 anticipates.
    bool gather_payload = false;
};

// Stand-in for extended_morton_coder::MortonCodeGenerator (crate not vendored in
// the reference): tables of the same shape with a documented meaning -- table
// 3i+c spreads bits 0..2 of its index to every third bit of a 27-bit Morton code
// held above 12 size bits; size_lut[j] = j.  Not the crate's values.
struct MortonCodeGenerator {
    uint64_t lut[9][256];
    uint64_t size_lut[4096];
    float morton_index_scale;
    float offset[3];
    float size_multiplier;
    float multiplier[3];
    MortonCodeGenerator(const float mn[3], const float mx[3]);
};

struct Scene {
    std::vector<float> positions;   // [num_vertices][3]
    std::vector<uint32_t> indices;  // [num_triangles][3]
    float scene_min[3], scene_max[3];
};

uint32_t rand_u32(uint64_t seed, uint64_t stream, uint64_t index);
Scene create_scene(const Constants& c);
std::vector<uint8_t> create_morton_uniforms(const MortonCodeGenerator& g);  // 51248 bytes
std::vector<uint8_t> pack_vertices(const Scene& s);                         // stride 48
std::vector<uint8_t> pack_triangles(const Scene& s);                        // stride 16

// src/utilities.rs:10-55, src/main.rs:201-233
uint64_t div_ceil(uint64_t l, uint64_t r);
uint64_t round_up(uint64_t l, uint64_t r);
std::vector<uint64_t> calculate_num_items_prefix_buffers(uint64_t num_elements);

// src/types.rs:68-77; prefix_sums and histogram live inside the plan
struct Buffers {
    void* vertices = nullptr;
    void* indices = nullptr;
    void* indices_2 = nullptr;
    void* morton_uniforms = nullptr;
    void* morton_codes = nullptr;
    void* morton_codes_2 = nullptr;
    void* source_index = nullptr;    // gather_payload only: u32[n] scratch, ends as the sorted source indices
    void* source_index_2 = nullptr;
    mrs_plan* plan = nullptr;
    int result_in_2 = 0;      // sorted codes (and, without gather_payload, triangles) are in the _2 buffers
    bool triangles_in_2 = false;  // where the sorted triangles are
    uint32_t num_triangles = 0;
    void release();
};

// Enqueues "Morton Coding" and the NUM_PASSES radix passes on `stream`
// (a hipStream_t).  Records only; returns an MRS_* status.
int run_compute_shaders(mrs_context* ctx, void* stream, const Constants& c, Buffers* out);

// The two halves of run_compute_shaders for hosts that keep their buffers across frames.
int create_all_buffers(mrs_context* ctx, const Constants& c, Buffers* out);
int record_frame(mrs_context* ctx, void* stream, const Constants& c, const std::vector<uint8_t>& uniforms,
                 const std::vector<uint8_t>& verts, const std::vector<uint8_t>& tris, Buffers* b);
int record_frame(mrs_context* ctx, void* stream, const Constants& c, const void* uniforms, size_t uniforms_bytes,
                 const void* verts, size_t verts_bytes, const void* tris, size_t tris_bytes, Buffers* b);

// Counterpart of the reference's State (src/state.rs:14-22, 24-87, 109-148): init() once, render()
// per frame.  Unlike the reference's render(), which calls run_compute_shaders and with it
// re-creates every buffer, pipeline and bind group each frame, render() here allocates nothing.
struct State {
    mrs_context* ctx = nullptr;
    void* stream = nullptr;  // hipStream_t
    Constants c;
    Buffers buffers;
    void* staging = nullptr;  // page-locked: this frame's uploads (uniforms, vertices, triangles), alive while the stream reads them
    size_t staging_bytes = 0;
    uint64_t frames = 0;
    void* frame_start = nullptr;  // hipEvent_t, optional: recorded when the frame's scene is ready and its uploads begin
    int init(const Constants& constants);
    int render();
    void destroy();
};

struct HeadlessResult {
    bool sorted = false;
    bool all_zero = true;
    uint32_t times_unsorted = 0;
    bool matches_stable_sort = false;  // keys AND payload equal std::stable_sort of the encoded input
    uint64_t key_checksum = 0;
};

// run + synchronise + read back + the adjacent-pair check of src/main.rs:819-864.
int run_headless(const Constants& c, HeadlessResult* res, bool verbose);
// `frames` calls of State::render on one State, then the checks of run_headless on the last frame;
// *ms_per_frame = wall time per frame (scene generation on the CPU included, as in the reference's loop),
// *device_ms_per_frame = HIP-event time of the enqueued work per frame (uploads from pageable host memory,
// encode, sort), the first of several frames left out (it loads the kernels).
int run_frames(const Constants& c, uint32_t frames, HeadlessResult* res, double* ms_per_frame, double* device_ms_per_frame);

}  // namespace mrs_host
#endif
