// Digit-pass kernels for 32-bit keys ranked by wave-wide digit matching, and the
// histogram kernels of that key width.
#include "mrs_radix_dispatch.hpp"

namespace mrs {
int launch_pass_u32_atomic(int mode, int bits, int threads, int kpt, int payload_bytes, const PassArgs& a,
                            hipStream_t s);

int launch_pass_u32(int mode, int bits, int threads, int kpt, int payload_bytes, const PassArgs& a, hipStream_t s)
{
    if (a.atomic_rank) return launch_pass_u32_atomic(mode, bits, threads, kpt, payload_bytes, a, s);
    return launch_pass_key<uint32_t, kRankMatch>(mode, bits, threads, kpt, payload_bytes, a, s);
}
int launch_block_histogram_u32(int bits, uint32_t n, const void* keys, uint32_t tile_items, uint32_t num_tiles,
                               uint32_t shift, uint32_t mask, uint32_t* hist, hipStream_t s)
{
    return launch_block_histogram_key<uint32_t>(bits, n, keys, tile_items, num_tiles, shift, mask, hist, s);
}
int launch_global_histogram_u32(int bits, int grid, int grid_banked, const GlobalHistArgs& a, hipStream_t s)
{
    return launch_global_histogram_key<uint32_t>(bits, grid, grid_banked, a, s);
}
}  // namespace mrs
