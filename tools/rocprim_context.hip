// Same-box context number, NOT part of the product: rocprim::radix_sort_pairs (the vendor's
// device-wide radix sort, header-only under /opt/rocm/include/rocprim) on the benchmark's own
// inputs, as a small shared library of its own that only bench.py loads (tools/librocprim_context.so).
// Nothing here is linked into libmrs.so and no product path calls it.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>  // rocprim's headers use memset on the host without including it
#include <rocprim/device/device_radix_sort.hpp>

extern "C" {

// Temporary bytes rocprim wants for n (u32 key, u32 value) pairs; 0 on error.
size_t rocprim_context_temp_bytes(size_t n, unsigned begin_bit, unsigned end_bit)
{
    size_t bytes = 0;
    const hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, static_cast<const uint32_t*>(nullptr),
                                                   static_cast<uint32_t*>(nullptr), static_cast<const uint32_t*>(nullptr),
                                                   static_cast<uint32_t*>(nullptr), n, begin_bit, end_bit, nullptr);
    return e == hipSuccess ? bytes : 0;
}

// Enqueues one sort on `stream`; returns the hipError_t.
int rocprim_context_sort_pairs(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                               const uint32_t* vals_in, uint32_t* vals_out, size_t n, unsigned begin_bit,
                               unsigned end_bit, void* stream)
{
    return static_cast<int>(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit,
                                                      end_bit, static_cast<hipStream_t>(stream)));
}

}  // extern "C"
