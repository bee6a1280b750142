// Micro-benchmark: shapes of the 30-bit Morton encoder's memory stream (12 B read + 8 B written per
// point).  hipcc --offload-arch=gfx950 -O3 -o encode_bench encode_bench.hip && ./encode_bench
// U = groups of four points a thread has in flight; T = threads per workgroup; B = workgroups per CU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

template <int U, int T>
__global__ __launch_bounds__(T) void encode_kernel(const float* __restrict__ xyz, unsigned long long n,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ idx)
{
    const unsigned long long groups = n / 4;
    const unsigned long long sweep = static_cast<unsigned long long>(gridDim.x) * T * U;
    for (unsigned long long base = static_cast<unsigned long long>(blockIdx.x) * T * U + threadIdx.x; base < groups; base += sweep) {
        float4 a[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4* p = reinterpret_cast<const float4*>(xyz + (base + static_cast<unsigned long long>(u) * T) * 12);
            a[u][0] = p[0]; a[u][1] = p[1]; a[u][2] = p[2];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long first = (base + static_cast<unsigned long long>(u) * T) * 4;
            const float f[12] = {a[u][0].x, a[u][0].y, a[u][0].z, a[u][0].w, a[u][1].x, a[u][1].y, a[u][1].z, a[u][1].w,
                                 a[u][2].x, a[u][2].y, a[u][2].z, a[u][2].w};
            uint32_t k[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t c3[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) c3[c] = static_cast<uint32_t>(fminf(fmaxf(f[3 * q + c] * 1024.0f, 0.0f), 1023.0f));
                k[q] = (spread10(c3[0]) << 2) | (spread10(c3[1]) << 1) | spread10(c3[2]);
            }
            *reinterpret_cast<uint4*>(keys + first) = make_uint4(k[0], k[1], k[2], k[3]);
            const uint32_t i0 = static_cast<uint32_t>(first);
            *reinterpret_cast<uint4*>(idx + first) = make_uint4(i0, i0 + 1, i0 + 2, i0 + 3);
        }
    }
}

// The same bytes with coalesced 16-byte loads: a wave reads 3 KiB as three 1 KiB rows and hands
// each lane its four points through LDS.  U tiles of T groups are in flight per sweep; NT marks the
// loads and stores nontemporal.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// NT: 0 plain, 1 nontemporal 4-byte accesses, 2 nontemporal 16-byte accesses, 3 nontemporal 16-byte
// stores only, 4 nontemporal 16-byte loads only
template <int T, int U, int NT>
__global__ __launch_bounds__(T) void encode_lds_kernel(const float* __restrict__ xyz, unsigned long long n,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ idx)
{
    __shared__ float4 s[U][T * 3];
    const unsigned long long groups = n / 4;
    const unsigned long long sweep = static_cast<unsigned long long>(gridDim.x) * T * U;
    for (unsigned long long gb = static_cast<unsigned long long>(blockIdx.x) * T * U; gb < groups; gb += sweep) {
        float4 r[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4* p = reinterpret_cast<const float4*>(xyz + (gb + static_cast<unsigned long long>(u) * T) * 12);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if constexpr (NT == 2 || NT == 4) {
                    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + j * T + threadIdx.x));
                    r[u][j] = make_float4(v.x, v.y, v.z, v.w);
                } else if constexpr (NT == 1) {
                    const float* q = reinterpret_cast<const float*>(p + j * T + threadIdx.x);
                    r[u][j] = make_float4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1),
                                          __builtin_nontemporal_load(q + 2), __builtin_nontemporal_load(q + 3));
                } else {
                    r[u][j] = p[j * T + threadIdx.x];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j) s[u][j * T + threadIdx.x] = r[u][j];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 a0 = s[u][threadIdx.x * 3], a1 = s[u][threadIdx.x * 3 + 1], a2 = s[u][threadIdx.x * 3 + 2];
            const unsigned long long first = (gb + static_cast<unsigned long long>(u) * T + threadIdx.x) * 4;
            const float f[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
            uint32_t k[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t c3[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) c3[c] = static_cast<uint32_t>(fminf(fmaxf(f[3 * q + c] * 1024.0f, 0.0f), 1023.0f));
                k[q] = (spread10(c3[0]) << 2) | (spread10(c3[1]) << 1) | spread10(c3[2]);
            }
            const uint32_t i0 = static_cast<uint32_t>(first);
            if constexpr (NT == 2 || NT == 3) {
                u32x4 kv = {k[0], k[1], k[2], k[3]};
                u32x4 iv = {i0, i0 + 1, i0 + 2, i0 + 3};
                __builtin_nontemporal_store(kv, reinterpret_cast<u32x4*>(keys + first));
                __builtin_nontemporal_store(iv, reinterpret_cast<u32x4*>(idx + first));
            } else if constexpr (NT == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_nontemporal_store(k[q], keys + first + q);
                    __builtin_nontemporal_store(i0 + q, idx + first + q);
                }
            } else {
                *reinterpret_cast<uint4*>(keys + first) = make_uint4(k[0], k[1], k[2], k[3]);
                *reinterpret_cast<uint4*>(idx + first) = make_uint4(i0, i0 + 1, i0 + 2, i0 + 3);
            }
        }
        __syncthreads();
    }
}

template <typename F>
static void run(const char* name, unsigned long long n, F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    std::printf("%-28s %.4f ms  %.3f TB/s\n", name, ms, 20.0 * n / (ms * 1e-3) / 1e12);
}

int main()
{
    const unsigned long long n = 1ull << 28;
    float* xyz;
    uint32_t *keys, *idx;
    hipMalloc(&xyz, n * 12);
    hipMalloc(&keys, n * 4);
    hipMalloc(&idx, n * 4);
    hipMemset(xyz, 0x3e, n * 12);
    const int cus = 256;
#define CASE(U, T, B) run("U=" #U " T=" #T " B=" #B, n, [&] { hipLaunchKernelGGL((encode_kernel<U, T>), dim3(cus * B), dim3(T), 0, 0, xyz, n, keys, idx); })
    CASE(1, 256, 8);
    CASE(2, 256, 8);
    CASE(4, 256, 8);
    CASE(4, 256, 4);
    CASE(8, 256, 4);
    CASE(2, 1024, 2);
    CASE(4, 1024, 2);
    CASE(8, 1024, 1);
    CASE(8, 1024, 2);
    CASE(1, 256, 16);
    CASE(1, 256, 32);
#define LCASE(T, U, NT, B) run("lds T=" #T " U=" #U " NT=" #NT " B=" #B, n, [&] { hipLaunchKernelGGL((encode_lds_kernel<T, U, NT>), dim3(cus * B), dim3(T), 0, 0, xyz, n, keys, idx); })
    LCASE(1024, 1, 0, 2);
    LCASE(1024, 1, 0, 1);
    LCASE(1024, 1, 1, 2);
    LCASE(1024, 1, 1, 1);
    LCASE(1024, 1, 2, 2);
    LCASE(1024, 1, 2, 1);
    LCASE(1024, 1, 3, 2);
    LCASE(1024, 1, 3, 1);
    LCASE(1024, 1, 4, 2);
    LCASE(1024, 1, 4, 1);
    LCASE(1024, 2, 2, 1);
    LCASE(512, 1, 2, 2);
    LCASE(512, 1, 2, 4);
    LCASE(256, 1, 2, 8);
    LCASE(1024, 1, 2, 64);
    return 0;
}
