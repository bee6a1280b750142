// Does a working set that fits the 256 MB memory-side cache copy faster than one that does not?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void copy_k(const uint4* __restrict__ p, uint4* __restrict__ q, size_t nvec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) q[v] = p[v];
}
__global__ void read_k(const uint4* __restrict__ p, size_t nvec, unsigned* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) { uint4 x = p[v]; acc ^= x.x ^ x.y ^ x.z ^ x.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t max_bytes = 2ull << 30;
    uint4 *p, *q; unsigned* o; CK(hipMalloc(&p, max_bytes)); CK(hipMalloc(&q, max_bytes)); CK(hipMalloc(&o, 16));
    CK(hipMemset(p, 1, max_bytes)); CK(hipMemset(q, 2, max_bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (size_t mb : {8, 16, 32, 64, 96, 128, 192, 256, 512, 2048}) {
        const size_t bytes = mb << 20, nvec = bytes / 16;
        const int reps = (int)((8ull << 30) / bytes) + 1;
        // ping-pong copy p -> q -> p ... : working set 2 x bytes
        for (int r = 0; r < 4; ++r) copy_k<<<4096, 256>>>(r & 1 ? q : p, r & 1 ? p : q, nvec);
        CK(hipEventRecord(a));
        for (int r = 0; r < reps; ++r) copy_k<<<4096, 256>>>(r & 1 ? q : p, r & 1 ? p : q, nvec);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double copy = 2.0 * bytes * reps / ms / 1e6;
        CK(hipEventRecord(a));
        for (int r = 0; r < reps; ++r) read_k<<<4096, 256>>>(p, nvec, o);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        printf("buffer %5zu MiB: ping-pong copy %6.0f GB/s (read+write, working set %zu MiB), re-read %6.0f GB/s, %.1f us per copy launch\n",
               mb, copy, 2 * mb, 1.0 * bytes * reps / ms / 1e6, 0.0);
    }
    return 0;
}
