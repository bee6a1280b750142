// Micro-benchmark: what the write path alone sustains (fill), against read alone and copy.
// hipcc --offload-arch=gfx950 -O3 -o fill_bench fill_bench.hip && ./fill_bench
#include <hip/hip_runtime.h>

#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(1024) void fill_kernel(u32x4* __restrict__ out, unsigned long long nvec)
{
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * 1024;
    const u32x4 v = {1u, 2u, 3u, static_cast<uint32_t>(threadIdx.x)};
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * 1024 + threadIdx.x; i < nvec; i += stride) {
        if (NT) __builtin_nontemporal_store(v, out + i);
        else out[i] = v;
    }
}

template <bool NT>
__global__ __launch_bounds__(1024) void read_kernel(const u32x4* __restrict__ in, unsigned long long nvec, uint32_t* sink)
{
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * 1024;
    uint32_t acc = 0;
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * 1024 + threadIdx.x; i < nvec; i += stride) {
        const u32x4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <bool NT>
__global__ __launch_bounds__(1024) void copy_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, unsigned long long nvec)
{
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * 1024;
    for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * 1024 + threadIdx.x; i < nvec; i += stride) {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
        else out[i] = in[i];
    }
}

template <typename F>
static void run(const char* name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    std::printf("%-34s %.4f ms  %.3f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
}

int main()
{
    const unsigned long long bytes = 2ull << 30;
    const unsigned long long nvec = bytes / 16;
    u32x4 *a, *b;
    uint32_t* sink;
    (void)hipMalloc(&a, bytes);
    (void)hipMalloc(&b, bytes);
    (void)hipMalloc(&sink, 4);
    (void)hipMemset(a, 1, bytes);
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        const dim3 grid(256 * per_cu), block(1024);
        char name[64];
        std::snprintf(name, sizeof name, "fill 2 GiB, %d wg/CU", per_cu);
        run(name, bytes, [&] { hipLaunchKernelGGL(fill_kernel<false>, grid, block, 0, 0, b, nvec); });
        std::snprintf(name, sizeof name, "fill nt 2 GiB, %d wg/CU", per_cu);
        run(name, bytes, [&] { hipLaunchKernelGGL(fill_kernel<true>, grid, block, 0, 0, b, nvec); });
        std::snprintf(name, sizeof name, "read 2 GiB, %d wg/CU", per_cu);
        run(name, bytes, [&] { hipLaunchKernelGGL(read_kernel<false>, grid, block, 0, 0, a, nvec, sink); });
        std::snprintf(name, sizeof name, "read nt 2 GiB, %d wg/CU", per_cu);
        run(name, bytes, [&] { hipLaunchKernelGGL(read_kernel<true>, grid, block, 0, 0, a, nvec, sink); });
        std::snprintf(name, sizeof name, "copy 2+2 GiB, %d wg/CU", per_cu);
        run(name, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_kernel<false>, grid, block, 0, 0, a, b, nvec); });
        std::snprintf(name, sizeof name, "copy nt 2+2 GiB, %d wg/CU", per_cu);
        run(name, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_kernel<true>, grid, block, 0, 0, a, b, nvec); });
    }
    run("hipMemsetAsync 2 GiB", bytes, [&] { (void)hipMemsetAsync(b, 0, bytes, 0); });
    return 0;
}
