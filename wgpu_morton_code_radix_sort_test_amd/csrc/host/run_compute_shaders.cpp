#include "run_compute_shaders.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <numeric>

namespace mrs_host {

#define HOST_HIP_TRY(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            std::fprintf(stderr, "%s failed: %s\n", #expr, hipGetErrorString(_e));           \
            return MRS_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

static uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

uint32_t rand_u32(uint64_t seed, uint64_t stream, uint64_t index)
{
    const uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ull * (stream + 1ull));
    return static_cast<uint32_t>(mix64(k + 0x9E3779B97F4A7C15ull * (index + 1ull)) >> 32);
}

uint64_t div_ceil(uint64_t l, uint64_t r) { return (l + r - 1) / r; }
uint64_t round_up(uint64_t l, uint64_t r) { return l % r == 0 ? l : l + r - l % r; }

std::vector<uint64_t> calculate_num_items_prefix_buffers(uint64_t num_elements)
{
    std::vector<uint64_t> out;
    uint64_t items = round_up(div_ceil(num_elements, 256) * 256, 256);
    if (items <= 1024) return out;
    items = round_up(items / 256, 256);
    while (items > 1024) {
        out.push_back(items);
        items = round_up(items / 256, 256);
    }
    out.push_back(items);
    return out;
}

MortonCodeGenerator::MortonCodeGenerator(const float mn[3], const float mx[3])
{
    double diag2 = 0;
    for (int c = 0; c < 3; ++c) {
        float extent = std::max(mx[c] - mn[c], 1e-30f);
        offset[c] = mn[c];
        multiplier[c] = 512.0f / extent;
        diag2 += static_cast<double>(extent) * extent;
    }
    size_multiplier = 4095.0f / static_cast<float>(std::sqrt(diag2));
    morton_index_scale = 1.0f;
    for (int i = 0; i < 3; ++i)
        for (int c = 0; c < 3; ++c)
            for (int b = 0; b < 256; ++b) {
                uint64_t code = 0;
                for (int k = 0; k < 3; ++k)
                    if ((b >> k) & 1) code |= 1ull << (3 * (3 * i + k) + (2 - c));
                lut[3 * i + c][b] = code << 12;
            }
    for (uint64_t j = 0; j < 4096; ++j) size_lut[j] = j;
}

Scene create_scene(const Constants& c)
{
    Scene s;
    s.positions.resize(static_cast<size_t>(c.num_vertices) * 3);
    for (uint32_t i = 0; i < c.num_vertices; ++i)
        for (uint32_t k = 0; k < 3; ++k)
            s.positions[3 * i + k] = static_cast<float>(rand_u32(c.rng_seed, k, i) >> 8) * (1.0f / 16777216.0f);
    s.indices.resize(static_cast<size_t>(c.num_triangles) * 3);
    for (int k = 0; k < 3; ++k) { s.scene_min[k] = INFINITY; s.scene_max[k] = -INFINITY; }
    for (uint32_t t = 0; t < c.num_triangles; ++t)
        for (uint32_t k = 0; k < 3; ++k) {
            const uint64_t draw = rand_u32(c.rng_seed, 3 + k, t);
            const uint32_t vi = static_cast<uint32_t>((draw * c.num_vertices) >> 32);
            s.indices[3 * t + k] = vi;
            // bounds over the vertices that triangles use, src/main.rs:148-154
            for (int a = 0; a < 3; ++a) {
                s.scene_min[a] = std::min(s.scene_min[a], s.positions[3 * vi + a]);
                s.scene_max[a] = std::max(s.scene_max[a], s.positions[3 * vi + a]);
            }
        }
    return s;
}

std::vector<uint8_t> create_morton_uniforms(const MortonCodeGenerator& g)
{
    std::vector<uint8_t> blob(MRS_MORTON_UNIFORMS_BYTES, 0);
    uint32_t* lut = reinterpret_cast<uint32_t*>(blob.data());
    for (int t = 0; t < 9; ++t)
        for (int b = 0; b < 256; ++b) {
            lut[(t * 256 + b) * 2] = static_cast<uint32_t>(g.lut[t][b]);
            lut[(t * 256 + b) * 2 + 1] = static_cast<uint32_t>(g.lut[t][b] >> 32);
        }
    uint32_t* size_lut = reinterpret_cast<uint32_t*>(blob.data() + 18432);
    for (int j = 0; j < 4096; ++j) {
        size_lut[2 * j] = static_cast<uint32_t>(g.size_lut[j]);
        size_lut[2 * j + 1] = static_cast<uint32_t>(g.size_lut[j] >> 32);
    }
    std::memcpy(blob.data() + 51200, &g.morton_index_scale, 4);
    std::memcpy(blob.data() + 51216, g.offset, 12);
    std::memcpy(blob.data() + 51228, &g.size_multiplier, 4);
    std::memcpy(blob.data() + 51232, g.multiplier, 12);
    return blob;
}

std::vector<uint8_t> pack_vertices(const Scene& s)
{
    const size_t n = s.positions.size() / 3;
    std::vector<uint8_t> out(n * MRS_VERTEX_STRIDE, 0);
    for (size_t i = 0; i < n; ++i) {
        float rec[12] = {s.positions[3 * i], s.positions[3 * i + 1], s.positions[3 * i + 2], 0, 0, 1, 0, 0, 0, 0.5f, 0, 0};
        std::memcpy(out.data() + i * MRS_VERTEX_STRIDE, rec, sizeof rec);
    }
    return out;
}

std::vector<uint8_t> pack_triangles(const Scene& s)
{
    const size_t n = s.indices.size() / 3;
    std::vector<uint8_t> out(n * MRS_TRIANGLE_STRIDE, 0);
    for (size_t i = 0; i < n; ++i) std::memcpy(out.data() + i * MRS_TRIANGLE_STRIDE, &s.indices[3 * i], 12);
    return out;
}

void Buffers::release()
{
    for (void* p : {vertices, indices, indices_2, morton_uniforms, morton_codes, morton_codes_2, source_index,
                    source_index_2})
        if (p) (void)hipFree(p);
    if (plan) mrs_plan_destroy(plan);
    *this = Buffers();
}

// create_all_buffers, src/main.rs:31-85, plus the plan that owns the sort's temporaries: everything
// a frame needs that does not change from frame to frame.
int create_all_buffers(mrs_context* ctx, const Constants& c, Buffers* b)
{
    const uint64_t n = c.num_triangles;
    b->num_triangles = c.num_triangles;
    HOST_HIP_TRY(hipMalloc(&b->morton_uniforms, MRS_MORTON_UNIFORMS_BYTES));
    HOST_HIP_TRY(hipMalloc(&b->vertices, static_cast<size_t>(c.num_vertices) * MRS_VERTEX_STRIDE));
    HOST_HIP_TRY(hipMalloc(&b->indices, n * MRS_TRIANGLE_STRIDE));
    HOST_HIP_TRY(hipMalloc(&b->indices_2, n * MRS_TRIANGLE_STRIDE));
    HOST_HIP_TRY(hipMalloc(&b->morton_codes, n * 8));
    HOST_HIP_TRY(hipMalloc(&b->morton_codes_2, n * 8));
    if (c.gather_payload) {
        HOST_HIP_TRY(hipMalloc(&b->source_index, n * 4));
        HOST_HIP_TRY(hipMalloc(&b->source_index_2, n * 4));
    }
    mrs_plan_desc d{};
    d.n = n;
    d.key_words = 2;
    d.payload_bytes = c.gather_payload ? 4 : 16;
    d.digit_bits = static_cast<int32_t>(c.bits_per_pass);
    d.begin_bit = 0;
    d.end_bit = static_cast<int32_t>(c.bits_per_pass * c.num_passes);
    d.algo = c.algo;
    d.tile_items = c.tile_items;
    return mrs_plan_create(ctx, &d, &b->plan);
}

// Uploads one frame's scene into existing buffers and enqueues "Morton Coding" and the pass loop
// (src/main.rs:581-665).  No allocation; the host arrays must stay alive until the stream has
// consumed them (the callers below keep or synchronise them).
int record_frame(mrs_context* ctx, void* stream, const Constants& c, const std::vector<uint8_t>& uniforms,
                 const std::vector<uint8_t>& verts, const std::vector<uint8_t>& tris, Buffers* b)
{
    return record_frame(ctx, stream, c, uniforms.data(), uniforms.size(), verts.data(), verts.size(), tris.data(), tris.size(), b);
}

int record_frame(mrs_context* ctx, void* stream, const Constants& c, const void* uniforms, size_t uniforms_bytes,
                 const void* verts, size_t verts_bytes, const void* tris, size_t tris_bytes, Buffers* b)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t n = c.num_triangles;
    HOST_HIP_TRY(hipMemcpyAsync(b->morton_uniforms, uniforms, uniforms_bytes, hipMemcpyHostToDevice, s));
    HOST_HIP_TRY(hipMemcpyAsync(b->vertices, verts, verts_bytes, hipMemcpyHostToDevice, s));
    HOST_HIP_TRY(hipMemcpyAsync(b->indices, tris, tris_bytes, hipMemcpyHostToDevice, s));
    // "Morton Coding", src/main.rs:581-592
    if (int rc = mrs_morton_encode_lut(ctx, b->morton_uniforms, b->vertices, c.num_vertices, b->indices, n,
                                       static_cast<uint32_t*>(b->morton_codes), stream))
        return rc;
    if (c.gather_payload) {
        b->triangles_in_2 = true;
        return mrs_sort_gather(b->plan, n, b->morton_codes, b->morton_codes_2, static_cast<uint32_t*>(b->source_index),
                               static_cast<uint32_t*>(b->source_index_2), b->indices, b->indices_2,
                               MRS_TRIANGLE_STRIDE, stream, &b->result_in_2);
    }
    // the pass loop, src/main.rs:594-665
    const int rc = mrs_sort_pairs(b->plan, n, b->morton_codes, b->indices, b->morton_codes_2, b->indices_2, stream,
                                  &b->result_in_2);
    b->triangles_in_2 = b->result_in_2 != 0;
    return rc;
}

// The reference's entry point as it stands: every call builds the scene, creates every buffer and
// records the passes (src/main.rs:376-688).
int run_compute_shaders(mrs_context* ctx, void* stream, const Constants& c, Buffers* b)
{
    const Scene scene = create_scene(c);
    const MortonCodeGenerator gen(scene.scene_min, scene.scene_max);
    const std::vector<uint8_t> uniforms = create_morton_uniforms(gen), verts = pack_vertices(scene),
                               tris = pack_triangles(scene);
    if (int rc = create_all_buffers(ctx, c, b)) return rc;
    if (int rc = record_frame(ctx, stream, c, uniforms, verts, tris, b)) return rc;
    HOST_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));  // the host vectors die at return
    return MRS_OK;
}

// ---- State: what the reference keeps across frames (src/state.rs:14-22), with the buffers and
// the plan added to it so that render() allocates nothing ---------------------------------------
int State::init(const Constants& constants)
{
    c = constants;
    int dev = 0;
    HOST_HIP_TRY(hipGetDevice(&dev));
    if (int rc = mrs_context_create(dev, &ctx)) return rc;  // State::new, src/state.rs:24-87
    HOST_HIP_TRY(hipStreamCreate(reinterpret_cast<hipStream_t*>(&stream)));
    // page-locked staging for a frame's uploads (the reference's queue.write_buffer stages too): the
    // copies then run at link speed and beside the host: 0.88 -> 0.20 ms of device time per frame of 65536
    // triangles, 2.5 -> 1.4 ms per frame of a million
    staging_bytes = MRS_MORTON_UNIFORMS_BYTES + static_cast<size_t>(c.num_vertices) * MRS_VERTEX_STRIDE +
                    static_cast<size_t>(c.num_triangles) * MRS_TRIANGLE_STRIDE;
    HOST_HIP_TRY(hipHostMalloc(&staging, staging_bytes, hipHostMallocDefault));
    return create_all_buffers(ctx, c, &buffers);
}

// State::render, src/state.rs:109-148: one frame = scene (here: the same seed-7 scene with the
// frame number folded into the seed, so that consecutive frames differ), encode, sort.  The
// reference re-creates buffers, pipelines and bind groups here on every frame; this one re-uses
// what init() made: no hipMalloc, no plan creation, no host synchronisation.
int State::render()
{
    Constants frame = c;
    frame.rng_seed = c.rng_seed + frames;
    // the previous frame's uploads must have left the host arrays before they are rewritten
    HOST_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    const Scene scene = create_scene(frame);
    const MortonCodeGenerator gen(scene.scene_min, scene.scene_max);
    const std::vector<uint8_t> uniforms = create_morton_uniforms(gen), verts = pack_vertices(scene), tris = pack_triangles(scene);
    if (uniforms.size() + verts.size() + tris.size() > staging_bytes) return MRS_ERR_INVALID_ARGUMENT;
    uint8_t* const up = static_cast<uint8_t*>(staging);
    uint8_t* const vp = up + uniforms.size();
    uint8_t* const tp = vp + verts.size();
    std::memcpy(up, uniforms.data(), uniforms.size());
    std::memcpy(vp, verts.data(), verts.size());
    std::memcpy(tp, tris.data(), tris.size());
    ++frames;
    if (frame_start) HOST_HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(frame_start), static_cast<hipStream_t>(stream)));
    return record_frame(ctx, stream, c, up, uniforms.size(), vp, verts.size(), tp, tris.size(), &buffers);
}

void State::destroy()
{
    buffers.release();
    if (staging) (void)hipHostFree(staging);
    staging = nullptr;
    if (stream) (void)hipStreamDestroy(static_cast<hipStream_t>(stream));
    if (ctx) mrs_context_destroy(ctx);
    ctx = nullptr;
    stream = nullptr;
}

static uint64_t select_bits(uint64_t value, uint32_t i)  // src/main.rs:1042-1051
{
    return i >= 8 ? value : value & ((1ull << (i * 8)) - 1ull);
}

// Read-back and checks of one finished frame: the adjacent-pair check of src/main.rs:831-864 and,
// beyond the reference, keys AND payload against a stable sort of a second encode of the same scene.
static int check_frame(mrs_context* ctx, hipStream_t s, const Constants& c, const Buffers& b, HeadlessResult* res,
                       bool verbose)
{
    const uint32_t n = c.num_triangles;
    std::vector<uint64_t> keys(n), keys_in(n);
    std::vector<uint32_t> tris(static_cast<size_t>(n) * 4);
    HOST_HIP_TRY(hipMemcpy(keys.data(), b.result_in_2 ? b.morton_codes_2 : b.morton_codes, n * 8ull, hipMemcpyDeviceToHost));
    HOST_HIP_TRY(hipMemcpy(tris.data(), b.triangles_in_2 ? b.indices_2 : b.indices, n * 16ull, hipMemcpyDeviceToHost));

    res->sorted = true;
    res->all_zero = true;
    res->times_unsorted = 0;
    for (uint32_t i = 0; i + 1 < n; ++i) {
        const uint64_t before = select_bits(keys[i], c.num_passes), after = select_bits(keys[i + 1], c.num_passes);
        if (before > after) { res->times_unsorted++; res->sorted = false; }
        if (before != 0 || after != 0) res->all_zero = false;
    }
    res->key_checksum = 0;
    for (uint64_t k : keys) res->key_checksum = res->key_checksum * 1099511628211ull + k;

    Buffers b2;
    int rc = MRS_OK;
    {
        const Scene scene = create_scene(c);
        const MortonCodeGenerator gen(scene.scene_min, scene.scene_max);
        const std::vector<uint8_t> t = pack_triangles(scene), v = pack_vertices(scene), u = create_morton_uniforms(gen);
        HOST_HIP_TRY(hipMalloc(&b2.morton_codes, n * 8ull));
        HOST_HIP_TRY(hipMalloc(&b2.indices, t.size()));
        HOST_HIP_TRY(hipMalloc(&b2.vertices, v.size()));
        HOST_HIP_TRY(hipMalloc(&b2.morton_uniforms, u.size()));
        HOST_HIP_TRY(hipMemcpy(b2.indices, t.data(), t.size(), hipMemcpyHostToDevice));
        HOST_HIP_TRY(hipMemcpy(b2.vertices, v.data(), v.size(), hipMemcpyHostToDevice));
        HOST_HIP_TRY(hipMemcpy(b2.morton_uniforms, u.data(), u.size(), hipMemcpyHostToDevice));
        rc = mrs_morton_encode_lut(ctx, b2.morton_uniforms, b2.vertices, c.num_vertices, b2.indices, n,
                                   static_cast<uint32_t*>(b2.morton_codes), s);
        HOST_HIP_TRY(hipStreamSynchronize(s));
        HOST_HIP_TRY(hipMemcpy(keys_in.data(), b2.morton_codes, n * 8ull, hipMemcpyDeviceToHost));
        std::vector<uint32_t> order(n);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            return select_bits(keys_in[x], c.num_passes) < select_bits(keys_in[y], c.num_passes);
        });
        bool same = true;
        for (uint32_t i = 0; i < n && same; ++i) {
            same = keys[i] == keys_in[order[i]] &&
                   std::memcmp(&tris[4ull * i], t.data() + 16ull * order[i], 12) == 0;
        }
        res->matches_stable_sort = same;
    }
    if (verbose) {
        if (!res->sorted) std::printf("Not sorted!!! %u\n", res->times_unsorted);
        else std::printf("Sorted !!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!\n");
        if (res->all_zero) std::printf("all zero!!!\n");
    }
    b2.release();
    return rc;
}

int run_headless(const Constants& c, HeadlessResult* res, bool verbose)
{
    mrs_context* ctx = nullptr;
    int dev = 0;
    HOST_HIP_TRY(hipGetDevice(&dev));
    if (int rc = mrs_context_create(dev, &ctx)) return rc;
    hipStream_t s;
    HOST_HIP_TRY(hipStreamCreate(&s));
    Buffers b;
    int rc = run_compute_shaders(ctx, s, c, &b);
    if (rc == MRS_OK) rc = mrs_plan_check(b.plan, s);
    if (rc != MRS_OK) std::fprintf(stderr, "mrs error %d: %s\n", rc, mrs_last_error_string());
    else rc = check_frame(ctx, s, c, b, res, verbose);
    b.release();
    (void)hipStreamDestroy(s);
    mrs_context_destroy(ctx);
    return rc;
}

int run_frames(const Constants& c, uint32_t frames, HeadlessResult* res, double* ms_per_frame, double* device_ms_per_frame)
{
    State st;
    int rc = st.init(c);
    if (rc != MRS_OK) {
        std::fprintf(stderr, "mrs error %d: %s\n", rc, mrs_last_error_string());
        st.destroy();
        return rc;
    }
    hipStream_t s = static_cast<hipStream_t>(st.stream);
    hipEvent_t e0, e1;
    HOST_HIP_TRY(hipEventCreate(&e0));
    HOST_HIP_TRY(hipEventCreate(&e1));
    double device_ms = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    st.frame_start = e0;  // behind the scene generation on the CPU, which the device does not wait for
    for (uint32_t f = 0; f < frames && rc == MRS_OK; ++f) {
        rc = st.render();
        if (rc != MRS_OK) break;
        HOST_HIP_TRY(hipEventRecord(e1, s));
        HOST_HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HOST_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (f > 0 || frames == 1) device_ms += ms;  // the first frame loads the kernels: not a frame's cost
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (rc == MRS_OK) rc = mrs_plan_check(st.buffers.plan, s);
    if (rc != MRS_OK) std::fprintf(stderr, "mrs error %d: %s\n", rc, mrs_last_error_string());
    if (rc == MRS_OK) {
        *ms_per_frame = std::chrono::duration<double, std::milli>(t1 - t0).count() / (frames ? frames : 1);
        *device_ms_per_frame = device_ms / (frames > 1 ? frames - 1 : 1);
        Constants last = c;
        last.rng_seed = c.rng_seed + st.frames - 1;
        rc = check_frame(st.ctx, s, last, st.buffers, res, false);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    st.destroy();
    return rc;
}

}  // namespace mrs_host
