// mrs_run: the headless driver (the role of `cargo run` with --cfg radix_sort_readback,
// src/main.rs:1035-1040): generates the scene, encodes, sorts, reads back and checks.
//
//   mrs_run [--triangles N] [--vertices N] [--passes P] [--algo onesweep|three_kernel] [--tile T] [--gather]
//           [--frames K]
//
// --gather sorts (code, source index) pairs and moves the 16-byte triangles once (mrs_sort_gather).
// --frames K plays the windowed path instead (State::render per frame, src/state.rs:109-148) on a
// State that keeps its buffers and plan: K frames, then the same checks on the last one.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "run_compute_shaders.h"

int main(int argc, char** argv)
{
    mrs_host::Constants c;
    unsigned long frames = 0;
    for (int i = 1; i < argc; ++i) {
        auto next = [&](const char* flag) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", flag); std::exit(2); }
            return argv[++i];
        };
        if (!std::strcmp(argv[i], "--triangles")) c.num_triangles = std::strtoul(next("--triangles"), nullptr, 10);
        else if (!std::strcmp(argv[i], "--vertices")) c.num_vertices = std::strtoul(next("--vertices"), nullptr, 10);
        else if (!std::strcmp(argv[i], "--passes")) c.num_passes = std::strtoul(next("--passes"), nullptr, 10);
        else if (!std::strcmp(argv[i], "--tile")) c.tile_items = std::atoi(next("--tile"));
        else if (!std::strcmp(argv[i], "--gather")) c.gather_payload = true;
        else if (!std::strcmp(argv[i], "--frames")) frames = std::strtoul(next("--frames"), nullptr, 10);
        else if (!std::strcmp(argv[i], "--algo")) {
            const char* a = next("--algo");
            c.algo = std::strcmp(a, "three_kernel") == 0 ? MRS_ALGO_THREE_KERNEL : MRS_ALGO_ONESWEEP;
        } else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (c.num_passes < 1 || c.num_passes > 8) { std::fprintf(stderr, "passes must be between 1 and 8\n"); return 2; }
    mrs_host::HeadlessResult r;
    if (frames) {
        double ms = 0.0, device_ms = 0.0;
        if (mrs_host::run_frames(c, static_cast<uint32_t>(frames), &r, &ms, &device_ms) != MRS_OK) return 1;
        std::printf("{\"triangles\": %u, \"frames\": %lu, \"ms_per_frame\": %.4f, \"device_ms_per_frame\": %.4f, "
                    "\"sorted\": %s, \"matches_stable_sort\": %s, \"key_checksum\": \"%016llx\"}\n",
                    c.num_triangles, frames, ms, device_ms, r.sorted ? "true" : "false",
                    r.matches_stable_sort ? "true" : "false", static_cast<unsigned long long>(r.key_checksum));
        return r.sorted && r.matches_stable_sort && !r.all_zero ? 0 : 1;
    }
    const int rc = mrs_host::run_headless(c, &r, true);
    if (rc != MRS_OK) return 1;
    std::printf("{\"triangles\": %u, \"sorted\": %s, \"times_unsorted\": %u, \"all_zero\": %s, "
                "\"matches_stable_sort\": %s, \"key_checksum\": \"%016llx\"}\n",
                c.num_triangles, r.sorted ? "true" : "false", r.times_unsorted, r.all_zero ? "true" : "false",
                r.matches_stable_sort ? "true" : "false", static_cast<unsigned long long>(r.key_checksum));
    return r.sorted && r.matches_stable_sort && !r.all_zero ? 0 : 1;
}
