// CPU baseline for the sort: TEST/BENCH INFRASTRUCTURE ONLY (never shipped,
// never on the product path).
//
// Packs (key << 32) | index into one u64 and sorts with std::sort -- equal to
// a stable sort by key because the indices ascend -- single-threaded, or with
// libstdc++'s parallel mode (__gnu_parallel::sort, OpenMP) on `threads` cores.
// It is the "CPU std::sort timed on the host cores" leg that the benchmark
// reports beside the GPU number; it is a baseline, not a target.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <parallel/algorithm>
#include <omp.h>
#include <vector>

extern "C" {

// Sorts n (key, idx) pairs in place.  threads <= 1: std::sort; otherwise
// __gnu_parallel::sort on `threads` OpenMP threads.  Returns the seconds
// spent in pack + sort + unpack; *threads_used reports the thread count.
double cpu_sort_pairs_u32(uint32_t* keys, uint32_t* idx, uint64_t n, int threads,
                          int* threads_used)
{
    std::vector<uint64_t> packed(n);
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t i = 0; i < n; ++i) packed[i] = (uint64_t(keys[i]) << 32) | idx[i];
    if (threads <= 1) {
        std::sort(packed.begin(), packed.end());
        if (threads_used) *threads_used = 1;
    } else {
        omp_set_num_threads(threads);
        __gnu_parallel::sort(packed.begin(), packed.end());
        if (threads_used) *threads_used = threads;
    }
    for (uint64_t i = 0; i < n; ++i) {
        keys[i] = uint32_t(packed[i] >> 32);
        idx[i] = uint32_t(packed[i]);
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

int cpu_hardware_threads(void) { return omp_get_num_procs(); }

}  // extern "C"
