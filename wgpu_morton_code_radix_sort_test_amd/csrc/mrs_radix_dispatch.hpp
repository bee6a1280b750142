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
//   512 x 8
// 16-byte payloads add 1024 x 8, 64-bit keys with payloads of at most 4 bytes 1024 x 16, and
// 32-bit keys with such payloads the large tiles 512 x 16 ... 1024 x 32.  RANK selects how keys are ranked inside a wave
// (kRankMatch / kRankAtomic, see mrs_radix_pass.hpp); each value is compiled in a
// translation unit of its own.
template <int MODE, int RANK, int MAP, int BITS, typename KeyT, typename ValT>
int launch_pass_shape(int threads, int kpt, const PassArgs& a, hipStream_t s)
{
#define MRS_SHAPE(T, K) \
    if (threads == T && kpt == K) return launch_pass_cfg<MODE, RANK, MAP, BITS, T, K, KeyT, ValT>(a, s)
    MRS_SHAPE(256, 1);
    MRS_SHAPE(256, 4);
    MRS_SHAPE(512, 8);
    if constexpr (sizeof(ValT) > 4) {
        MRS_SHAPE(1024, 8);
    }
    if constexpr (sizeof(ValT) <= 4 && sizeof(KeyT) == 8) {
        MRS_SHAPE(1024, 16);
    }
    if constexpr (sizeof(ValT) <= 4) {
        if constexpr (sizeof(KeyT) == 4) {
            MRS_SHAPE(512, 16);
            MRS_SHAPE(512, 32);
            MRS_SHAPE(1024, 16);
            MRS_SHAPE(1024, 32);
        }
    }
#undef MRS_SHAPE
    return fail(MRS_ERR_UNSUPPORTED, "no kernel for %d threads x %d keys per thread", threads, kpt);
}

template <int MODE, int RANK, typename KeyT, typename ValT>
int launch_pass_bits(int bits, int threads, int kpt, const PassArgs& a, hipStream_t s)
{
    // the bucket -> part table is only used by the partition pass (onesweep machinery)
    if constexpr (MODE == kModeOnesweep) {
        if (a.digit_map && a.num_splitters) {
            // only the sharded sort gets here, with 8-bit digits and payloads of at most 4 bytes
            if constexpr (sizeof(ValT) <= 4) {
                if (bits == 8) return launch_pass_shape<MODE, RANK, kMapTableAndSplitters, 8, KeyT, ValT>(threads, kpt, a, s);
            }
            return fail(MRS_ERR_UNSUPPORTED, "a partition by splitters needs 8-bit digits and a payload of at most 4 bytes");
        }
        if (a.digit_map) {
            if (bits == 4) return launch_pass_shape<MODE, RANK, kMapTable, 4, KeyT, ValT>(threads, kpt, a, s);
            if (bits == 8) return launch_pass_shape<MODE, RANK, kMapTable, 8, KeyT, ValT>(threads, kpt, a, s);
        }
    }
    if (bits == 4) return launch_pass_shape<MODE, RANK, kMapNone, 4, KeyT, ValT>(threads, kpt, a, s);
    if (bits == 8) return launch_pass_shape<MODE, RANK, kMapNone, 8, KeyT, ValT>(threads, kpt, a, s);
    return fail(MRS_ERR_UNSUPPORTED, "digit_bits must be 4 or 8");
}

template <typename KeyT, int RANK>
int launch_pass_key(int mode, int bits, int threads, int kpt, int payload_bytes, const PassArgs& a,
                    hipStream_t s)
{
#define MRS_ROUTE(MODE)                                                                              \
    switch (payload_bytes) {                                                                         \
        case 0: return launch_pass_bits<MODE, RANK, KeyT, NoVal>(bits, threads, kpt, a, s);          \
        case 4: return launch_pass_bits<MODE, RANK, KeyT, uint32_t>(bits, threads, kpt, a, s);       \
        case 16: return launch_pass_bits<MODE, RANK, KeyT, uint4>(bits, threads, kpt, a, s);         \
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

// grid_banked > 0 selects global_histogram_banked_kernel.
template <typename KeyT>
int launch_global_histogram_key(int bits, int grid, int grid_banked, const GlobalHistArgs& a, hipStream_t s)
{
    if (grid_banked > 0) {
        const bool all4 = !a.digit_map && a.num_passes == static_cast<uint32_t>(BankedHistogram<4, KeyT>::MAXP);
        const bool all8 = !a.digit_map && a.num_passes == static_cast<uint32_t>(BankedHistogram<8, KeyT>::MAXP);
        const dim3 g(grid_banked), b(1024);
        if (bits == 4 && all4)
            hipLaunchKernelGGL((global_histogram_banked_kernel<4, KeyT, true>), g, b, 0, s, a);
        else if (bits == 4)
            hipLaunchKernelGGL((global_histogram_banked_kernel<4, KeyT, false>), g, b, 0, s, a);
        else if (all8)
            hipLaunchKernelGGL((global_histogram_banked_kernel<8, KeyT, true>), g, b, 0, s, a);
        else
            hipLaunchKernelGGL((global_histogram_banked_kernel<8, KeyT, false>), g, b, 0, s, a);
        MRS_CHECK_LAUNCH("global_histogram_banked_kernel");
        return MRS_OK;
    }
    if (bits == 4)
        hipLaunchKernelGGL((global_histogram_kernel<4, KeyT>), dim3(grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((global_histogram_kernel<8, KeyT>), dim3(grid), dim3(256), 0, s, a);
    MRS_CHECK_LAUNCH("global_histogram_kernel");
    return MRS_OK;
}

}  // namespace mrs
#endif
