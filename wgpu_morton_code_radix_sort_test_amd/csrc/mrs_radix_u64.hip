// Digit-pass kernels for 64-bit keys ranked by wave-wide digit matching, and the
// histogram kernels of that key width.  64-bit keys are stored as (lo, hi) u32 pairs,
// i.e. as little-endian u64 (src/main.rs:60-61, src/shaders/radix_sort_index.wgsl:268-270).
#include "mrs_radix_dispatch.hpp"

namespace mrs {
int launch_pass_u64_atomic(int mode, int bits, int threads, int kpt, int payload_bytes, const PassArgs& a,
                            hipStream_t s);

int launch_pass_u64(int mode, int bits, int threads, int kpt, int payload_bytes, const PassArgs& a, hipStream_t s)
{
    if (a.atomic_rank) return launch_pass_u64_atomic(mode, bits, threads, kpt, payload_bytes, a, s);
    return launch_pass_key<unsigned long long, kRankMatch>(mode, bits, threads, kpt, payload_bytes, a, s);
}
int launch_block_histogram_u64(int bits, uint32_t n, const void* keys, uint32_t tile_items, uint32_t num_tiles,
                               uint32_t shift, uint32_t mask, uint32_t* hist, hipStream_t s)
{
    return launch_block_histogram_key<unsigned long long>(bits, n, keys, tile_items, num_tiles, shift, mask, hist, s);
}
int launch_global_histogram_u64(int bits, int grid, int grid_banked, const GlobalHistArgs& a, hipStream_t s)
{
    return launch_global_histogram_key<unsigned long long>(bits, grid, grid_banked, a, s);
}
}  // namespace mrs
