// Digit-pass kernels for 32-bit keys ranked by returning LDS atomics.
#include "mrs_radix_dispatch.hpp"

namespace mrs {
int launch_pass_u32_atomic(int mode, int bits, int threads, int kpt, int payload_bytes, const PassArgs& a,
                            hipStream_t s)
{
    return launch_pass_key<uint32_t, kRankAtomic>(mode, bits, threads, kpt, payload_bytes, a, s);
}
}  // namespace mrs
