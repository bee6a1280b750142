// Micro-benchmark: variants of the all-passes digit histogram over 2^28 u32 keys.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void fill(uint32_t* k, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull + 12345; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        k[i] = (uint32_t)x & 0x3FFFFFFFu;
    }
}

// V1: plain LDS atomics, uint4 loads, UNROLL vectors in flight
template <int UNROLL, int WPRIV, bool CHECK>
__global__ __launch_bounds__(256) void hist_v(const uint4* __restrict__ keys, size_t nvec, uint32_t* ghist) {
    __shared__ uint32_t s[WPRIV][4 * 256];
    for (int j = threadIdx.x; j < WPRIV * 1024; j += 256) (&s[0][0])[j] = 0;
    __syncthreads();
    uint32_t* h = s[WPRIV == 1 ? 0 : (threadIdx.x >> 6) % WPRIV];
    const int lane = threadIdx.x & 63;
    auto count = [&](uint32_t k) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t d = (k >> (8 * p)) & 0xFFu;
            if (CHECK) {
                const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
                const unsigned long long eq = __builtin_amdgcn_uicmp(d, d0, 32 /* == */);
                if (eq == ~0ull) { if (lane == 0) atomicAdd(&h[p * 256 + d0], 64u); }
                else atomicAdd(&h[p * 256 + d], 1u);
            } else {
                atomicAdd(&h[p * 256 + d], 1u);
            }
        }
    };
    const size_t stride = (size_t)gridDim.x * 256;
    size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; v + (UNROLL - 1) * stride < nvec; v += UNROLL * stride) {
        uint4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = keys[v + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { count(x[u].x); count(x[u].y); count(x[u].z); count(x[u].w); }
    }
    for (; v < nvec; v += stride) { uint4 x = keys[v]; count(x.x); count(x.y); count(x.z); count(x.w); }
    __syncthreads();
    for (int j = threadIdx.x; j < 1024; j += 256) {
        uint32_t c = 0;
        for (int w = 0; w < WPRIV; ++w) c += s[w][j];
        if (c) atomicAdd(&ghist[j], c);
    }
}

// V5: R copies of every counter, copy = lane % R, so the lanes of one atomic instruction spread
// over the LDS banks whatever their digits are (R = 32: no two lanes of a half-wave share a bank)
template <int UNROLL, int R, int THREADS>
__global__ __launch_bounds__(THREADS) void hist_banked(const uint4* __restrict__ keys, size_t nvec, uint32_t* ghist) {
    extern __shared__ uint32_t sb[];
    for (int j = threadIdx.x; j < 1024 * R; j += THREADS) sb[j] = 0;
    __syncthreads();
    const int copy = threadIdx.x & (R - 1);
    auto count = [&](uint32_t k) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t d = (k >> (8 * p)) & 0xFFu;
            atomicAdd(&sb[(p * 256 + d) * R + copy], 1u);
        }
    };
    const size_t stride = (size_t)gridDim.x * THREADS;
    size_t v = (size_t)blockIdx.x * THREADS + threadIdx.x;
    for (; v + (UNROLL - 1) * stride < nvec; v += UNROLL * stride) {
        uint4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = keys[v + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { count(x[u].x); count(x[u].y); count(x[u].z); count(x[u].w); }
    }
    for (; v < nvec; v += stride) { uint4 x = keys[v]; count(x.x); count(x.y); count(x.z); count(x.w); }
    __syncthreads();
    for (int j = threadIdx.x; j < 1024; j += THREADS) {
        uint32_t c = 0;
        for (int r = 0; r < R; ++r) c += sb[j * R + ((r + j) & (R - 1))];
        if (c) atomicAdd(&ghist[j], c);
    }
}

// V6: load only, xor-reduce (speed of light for this access pattern)
template <int UNROLL>
__global__ __launch_bounds__(256) void load_only(const uint4* __restrict__ keys, size_t nvec, uint32_t* ghist) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (; v + (UNROLL - 1) * stride < nvec; v += UNROLL * stride) {
        uint4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = keys[v + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
    }
    if (acc == 0x12345678u) ghist[0] = acc;
}

template <typename F>
float time_it(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t n = 1ull << 28;
    uint32_t *k, *g; CK(hipMalloc(&k, n * 4)); CK(hipMalloc(&g, 4096));
    fill<<<4096, 256>>>(k, n); CK(hipDeviceSynchronize());
    const uint4* kv = (const uint4*)k; const size_t nvec = n / 4;
    std::vector<uint32_t> ref(1024), got(1024);
    bool have_ref = false;
    auto run = [&](const char* name, auto kern, int grid) {
        CK(hipMemset(g, 0, 4096));
        float ms = time_it([&] { kern<<<grid, 256>>>(kv, nvec, g); });
        CK(hipMemset(g, 0, 4096)); kern<<<grid, 256>>>(kv, nvec, g); CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), g, 4096, hipMemcpyDeviceToHost));
        bool ok = true;
        if (!have_ref) { ref = got; have_ref = true; } else ok = (ref == got);
        printf("%-28s grid %5d  %.3f ms  %.0f GB/s  %s\n", name, grid, ms, n * 4 / ms / 1e6, ok ? "ok" : "MISMATCH");
        fflush(stdout);
    };
    for (int grid : {4096}) {
        run("plain u1", hist_v<1, 1, false>, grid);
        run("plain u2", hist_v<2, 1, false>, grid);
        run("plain u4", hist_v<4, 1, false>, grid);
        run("check u1", hist_v<1, 1, true>, grid);
        run("check u4", hist_v<4, 1, true>, grid);
        run("wave-private u2", hist_v<2, 4, false>, grid);
        run("wave-private u4 check", hist_v<4, 4, true>, grid);
        run("2-private u4", hist_v<4, 2, false>, grid);
    }
    auto run_banked = [&](const char* name, auto kern, int grid, int threads, int copies) {
        const size_t lds = 4096 * (size_t)copies;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipMemset(g, 0, 4096));
        float ms = time_it([&] { kern<<<grid, threads, lds>>>(kv, nvec, g); });
        CK(hipMemset(g, 0, 4096)); kern<<<grid, threads, lds>>>(kv, nvec, g); CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), g, 4096, hipMemcpyDeviceToHost));
        printf("%-28s grid %5d  %.3f ms  %.0f GB/s  %s\n", name, grid, ms, n * 4 / ms / 1e6, ref == got ? "ok" : "MISMATCH");
        fflush(stdout);
    };
    for (int grid : {256, 512}) {
        run_banked("banked R32 1024thr u2", hist_banked<2, 32, 1024>, grid, 1024, 32);
        run_banked("banked R32 1024thr u4", hist_banked<4, 32, 1024>, grid, 1024, 32);
    }
    for (int grid : {512, 1024}) {
        run_banked("banked R16 512thr u2", hist_banked<2, 16, 512>, grid, 512, 16);
        run_banked("banked R16 512thr u4", hist_banked<4, 16, 512>, grid, 512, 16);
        run_banked("banked R16 1024thr u4", hist_banked<4, 16, 1024>, grid / 2, 1024, 16);
    }
    for (int grid : {1024, 2048}) {
        run_banked("banked R8 256thr u4", hist_banked<4, 8, 256>, grid, 256, 8);
        run_banked("banked R4 256thr u4", hist_banked<4, 4, 256>, grid, 256, 4);
    }
    {   // the same kernel over three different inputs in turn: nothing of the previous sweep is cached
        uint32_t* k3[3] = {k, nullptr, nullptr};
        for (int i = 1; i < 3; ++i) { CK(hipMalloc(&k3[i], n * 4)); fill<<<4096, 256>>>(k3[i], n); }
        CK(hipDeviceSynchronize());
        auto kern = hist_banked<4, 16, 1024>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        int turn = 0;
        float ms = time_it([&] { kern<<<512, 1024, 65536>>>((const uint4*)k3[turn++ % 3], nvec, g); }, 9);
        printf("banked R16 1024thr u4 grid 512, rotating inputs  %.3f ms\n", ms);
        ms = time_it([&] { kern<<<512, 1024, 65536>>>((const uint4*)k3[0], nvec, g); }, 9);
        printf("banked R16 1024thr u4 grid 512, same input       %.3f ms\n", ms);
    }
    for (int grid : {2048, 4096}) {
        float ms = time_it([&] { load_only<4><<<grid, 256>>>(kv, nvec, g); });
        printf("load only u4 grid %d  %.3f ms %.0f GB/s\n", grid, ms, n * 4 / ms / 1e6);
    }
    return 0;
}
