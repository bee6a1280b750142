// Instantiation table of the digit-pass kernels for one key type.  Included by
// mrs_radix_u32.hip and mrs_radix_u64.hip so that the two key widths compile in
// parallel.
#ifndef MRS_RADIX_DISPATCH_HPP
#define MRS_RADIX_DISPATCH_HPP

#include "mrs_radix_impl.hpp"

namespace mrs {

// (threads, keys per thread) shapes every key/payload combination supports:
//   256 x 1  -- the reference's one-key-per-thread 256-key block
//   256 x 4  -- small tiles (16-byte payloads, small inputs)
// and the large tiles, only for payloads of at most 4 bytes:
//   512 x 8, 512 x 16 (u32 keys only)
template <int MODE, bool MAP, int BITS, int THREADS, int KPT, typename KeyT, typename ValT>
int launch_pass_any(const PassArgs& a, hipStream_t s)
{
    return launch_pass_cfg<MODE, MAP, BITS, THREADS, KPT, KeyT, ValT>(a, s);
}

template <int MODE, bool MAP, int BITS, typename KeyT, typename ValT>
int launch_pass_shape(int threads, int kpt, const PassArgs& a, hipStream_t s)
{
    if (threads == 256 && kpt == 1) return launch_pass_any<MODE, MAP, BITS, 256, 1, KeyT, ValT>(a, s);
    if (threads == 256 && kpt == 4) return launch_pass_any<MODE, MAP, BITS, 256, 4, KeyT, ValT>(a, s);
    if constexpr (sizeof(ValT) <= 4) {
        if (threads == 512 && kpt == 8) return launch_pass_any<MODE, MAP, BITS, 512, 8, KeyT, ValT>(a, s);
        if constexpr (sizeof(KeyT) == 4) {
            if (threads == 512 && kpt == 16) return launch_pass_any<MODE, MAP, BITS, 512, 16, KeyT, ValT>(a, s);
            if (threads == 256 && kpt == 16) return launch_pass_any<MODE, MAP, BITS, 256, 16, KeyT, ValT>(a, s);
            if (threads == 256 && kpt == 32) return launch_pass_any<MODE, MAP, BITS, 256, 32, KeyT, ValT>(a, s);
            if (threads == 1024 && kpt == 16) return launch_pass_any<MODE, MAP, BITS, 1024, 16, KeyT, ValT>(a, s);
            if (threads == 512 && kpt == 32) return launch_pass_any<MODE, MAP, BITS, 512, 32, KeyT, ValT>(a, s);
            if (threads == 1024 && kpt == 32) return launch_pass_any<MODE, MAP, BITS, 1024, 32, KeyT, ValT>(a, s);
            if (threads == 1024 && kpt == 8) return launch_pass_any<MODE, MAP, BITS, 1024, 8, KeyT, ValT>(a, s);
        }
    }
    return fail(MRS_ERR_UNSUPPORTED, "no kernel for %d threads x %d keys per thread", threads, kpt);
}

template <int MODE, typename KeyT, typename ValT>
int launch_pass_bits(int bits, int threads, int kpt, const PassArgs& a, hipStream_t s)
{
    // the bucket -> part table is only used by the partition pass (onesweep machinery)
    if constexpr (MODE == kModeOnesweep) {
        if (a.digit_map) {
            if (bits == 4) return launch_pass_shape<MODE, true, 4, KeyT, ValT>(threads, kpt, a, s);
            if (bits == 8) return launch_pass_shape<MODE, true, 8, KeyT, ValT>(threads, kpt, a, s);
        }
    }
    if (bits == 4) return launch_pass_shape<MODE, false, 4, KeyT, ValT>(threads, kpt, a, s);
    if (bits == 8) return launch_pass_shape<MODE, false, 8, KeyT, ValT>(threads, kpt, a, s);
    return fail(MRS_ERR_UNSUPPORTED, "digit_bits must be 4 or 8");
}

template <typename KeyT>
int launch_pass_key(int mode, int bits, int threads, int kpt, int payload_bytes, const PassArgs& a,
                    hipStream_t s)
{
#define MRS_ROUTE(MODE)                                                                              \
    switch (payload_bytes) {                                                                         \
        case 0: return launch_pass_bits<MODE, KeyT, NoVal>(bits, threads, kpt, a, s);                \
        case 4: return launch_pass_bits<MODE, KeyT, uint32_t>(bits, threads, kpt, a, s);             \
        case 16: return launch_pass_bits<MODE, KeyT, uint4>(bits, threads, kpt, a, s);               \
        default: return fail(MRS_ERR_UNSUPPORTED, "payload_bytes must be 0, 4 or 16");               \
    }
    if (mode == kModeScatter) { MRS_ROUTE(kModeScatter) }
    MRS_ROUTE(kModeOnesweep)
#undef MRS_ROUTE
}

template <typename KeyT>
int launch_block_histogram_key(int bits, uint32_t n, const void* keys, uint32_t tile_items,
                               uint32_t num_tiles, uint32_t shift, uint32_t mask, uint32_t* hist,
                               hipStream_t s)
{
    const KeyT* k = static_cast<const KeyT*>(keys);
    if (bits == 4)
        hipLaunchKernelGGL((block_histogram_kernel<4, KeyT>), dim3(num_tiles), dim3(256), 0, s, k, n, tile_items,
                           num_tiles, shift, mask, hist);
    else
        hipLaunchKernelGGL((block_histogram_kernel<8, KeyT>), dim3(num_tiles), dim3(256), 0, s, k, n, tile_items,
                           num_tiles, shift, mask, hist);
    MRS_CHECK_LAUNCH("block_histogram_kernel");
    return MRS_OK;
}

template <typename KeyT>
int launch_global_histogram_key(int bits, int grid, const GlobalHistArgs& a, hipStream_t s)
{
    if (bits == 4)
        hipLaunchKernelGGL((global_histogram_kernel<4, KeyT>), dim3(grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((global_histogram_kernel<8, KeyT>), dim3(grid), dim3(256), 0, s, a);
    MRS_CHECK_LAUNCH("global_histogram_kernel");
    return MRS_OK;
}

}  // namespace mrs
#endif
