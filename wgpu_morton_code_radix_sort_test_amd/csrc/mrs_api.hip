// C ABI of libmrs.so: contexts, plans and the launch schedule of the sort.
//
// mrs_sort_pairs is the host side of the reference's pass loop
// (src/main.rs:594-665): for every digit it enqueues the histogram, prefix-sum
// and scatter stages on one stream with ping-pong buffers
// (src/main.rs:515-576), with no host synchronisation in between.
#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "mrs_hybrid.hpp"
#include "mrs_internal.h"
#include "mrs_radix_impl.hpp"

namespace mrs {

static thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

void clear_error() { g_last_error.clear(); }

// ---- range markers for an external profiler (MRS_ATTR_ROCTX_RANGES) -------------------------
// The reference names its compute passes for NSight (src/main.rs:583-662); roctx ranges are what
// rocprofv3 shows in their place.  The roctx library is bound at run time, only when asked for.
static std::atomic<bool> g_ranges_on{false};
static int (*g_range_push)(const char*) = nullptr;
static int (*g_range_pop)() = nullptr;

static int bind_roctx()
{
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (g_range_push && g_range_pop) return MRS_OK;
    // rocprofv3 listens to the first; the second serves roctracer-based tools
    const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    for (const char* name : names) {
        void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        auto push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        auto pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push && pop) {
            g_range_push = push;
            g_range_pop = pop;
            return MRS_OK;
        }
    }
    return fail(MRS_ERR_UNSUPPORTED, "no roctx library (librocprofiler-sdk-roctx.so, libroctx64.so) could be loaded");
}

ProfilerRange::ProfilerRange(const char* name) : pushed_(false)
{
    if (g_ranges_on.load(std::memory_order_relaxed) && g_range_push) {
        (void)g_range_push(name);
        pushed_ = true;
    }
}
ProfilerRange::~ProfilerRange()
{
    if (pushed_ && g_range_pop) (void)g_range_pop();
}

// Zeroes device memory on a stream with a kernel of the library, not with the runtime's
// asynchronous memset: the memset node that a stream capture makes of the latter was seen to
// leave the memory untouched when the captured graph is replayed by the HIP runtime a PyTorch
// process carries (the sort's ticket counters then keep their old values and every workgroup of
// the passes returns at once), while a kernel node replays like the kernels around it.
__global__ __launch_bounds__(256) void clear_kernel(uint32_t* __restrict__ p, size_t words)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * 256u;
    const size_t tid = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
        uint4* v = reinterpret_cast<uint4*>(p);
        const size_t vecs = words / 4;
        for (size_t i = tid; i < vecs; i += stride) v[i] = make_uint4(0, 0, 0, 0);
        for (size_t i = vecs * 4 + tid; i < words; i += stride) p[i] = 0;
    } else {
        for (size_t i = tid; i < words; i += stride) p[i] = 0;
    }
}

int clear_async(void* p, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return MRS_OK;
    const ProfilerRange range("Clear");
    if ((reinterpret_cast<uintptr_t>(p) & 3u) || (bytes & 3u))
        return fail(MRS_ERR_INVALID_ARGUMENT, "clear: address and size must be multiples of 4");
    const size_t words = bytes / 4;
    size_t blocks = (words / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(clear_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, static_cast<uint32_t*>(p), words);
    MRS_CHECK_LAUNCH("clear_kernel");
    return MRS_OK;
}

// per-key-type entry points (mrs_radix_u32.hip / mrs_radix_u64.hip)
int launch_pass_u32(int mode, int bits, int threads, int kpt, int payload_bytes, const PassArgs& a, hipStream_t s);
int launch_pass_u64(int mode, int bits, int threads, int kpt, int payload_bytes, const PassArgs& a, hipStream_t s);
int launch_block_histogram_u32(int bits, uint32_t n, const void* keys, uint32_t tile_items, uint32_t num_tiles,
                               uint32_t shift, uint32_t mask, uint32_t* hist, hipStream_t s);
int launch_block_histogram_u64(int bits, uint32_t n, const void* keys, uint32_t tile_items, uint32_t num_tiles,
                               uint32_t shift, uint32_t mask, uint32_t* hist, hipStream_t s);
int launch_global_histogram_u32(int bits, int grid, int grid_banked, const GlobalHistArgs& a, hipStream_t s);
int launch_global_histogram_u64(int bits, int grid, int grid_banked, const GlobalHistArgs& a, hipStream_t s);
int probe_lds_atomic_order(bool* in_order);
int probe_die_ids(bool* all_dies);
// mrs_morton.hip
int launch_encode30_with_histogram(int num_cus, const float* xyz_dev, uint64_t n, const float* mn, const float* mx,
                                   uint32_t* keys_dev, uint32_t* idx_dev, uint32_t* ghist, uint32_t begin_bit,
                                   uint32_t end_bit, uint32_t bits, uint32_t num_passes, hipStream_t s);
// mrs_gather.hip
int launch_gather(const void* in, const uint32_t* index, void* out, uint32_t n, uint32_t record_bytes, int num_cus,
                  hipStream_t s);
// mrs_small_sort.hip
uint32_t small_sort_capacity();
int launch_small_sort(int key_words, int bits, const SmallSortArgs& a, hipStream_t s);

bool config_supported(unsigned long long n, int key_words, int payload_bytes, int digit_bits, int tile_items, int algo,
                      KernelConfig* cfg)
{
    (void)algo;
    if (key_words != 1 && key_words != 2) return false;
    if (payload_bytes != 0 && payload_bytes != 4 && payload_bytes != 16) return false;
    if (digit_bits != 4 && digit_bits != 8) return false;
    const bool small_pairs = key_words == 1 && payload_bytes <= 4;  // the shapes of 16384 elements and more
    KernelConfig c;
    switch (tile_items) {
        case 0:
            // Larger tiles scatter longer runs per digit and walk fewer tile status words, as
            // long as the input still yields several tiles per compute unit.
            if (payload_bytes == 16) c = n >= (1ull << 20) ? KernelConfig{1024, 8} : (n >= (1ull << 18) ? KernelConfig{512, 8} : KernelConfig{256, 4});
            else if (key_words == 2) c = n >= (1ull << 20) ? KernelConfig{1024, 16} : KernelConfig{512, 8};
            // measured on whole sorts, back to back: 2^23 pairs 0.145 ms with 32768-key tiles
            // (one per compute unit) against 0.184 with 16384; 2^18 0.060 ms with 4096 against 0.066
            else if (n >= (1ull << 23)) c = {1024, 32};
            else if (n >= (1ull << 20)) c = {1024, 16};
            else if (n >= (1ull << 19)) c = {512, 16};
            else c = {512, 8};
            break;
        case 256: c = {256, 1}; break;
        case 1024: c = {256, 4}; break;
        case 4096: c = {512, 8}; break;
        case 8192: c = payload_bytes == 16 ? KernelConfig{1024, 8} : KernelConfig{512, 16}; break;
        case 16384: c = {1024, 16}; break;
        case 32768: c = {1024, 32}; break;
        default: return false;
    }
    // tuning aid: MRS_TILE_SHAPE=threads,kpt overrides the default shape (read once per process)
    static const KernelConfig forced = [] {
        KernelConfig f{0, 0};
        if (const char* e = std::getenv("MRS_TILE_SHAPE")) {
            int t = 0, k = 0;
            if (std::sscanf(e, "%d,%d", &t, &k) == 2) f = {t, k};
        }
        return f;
    }();
    if (forced.threads != 0 && tile_items == 0) c = forced;
    // what the kernel table holds (mrs_radix_dispatch.hpp): three shapes for every element type,
    // 1024 x 8 for 16-byte payloads, 1024 x 16 for payloads of at most 4 bytes, and the large
    // shapes for 32-bit keys with such payloads.  Anything else (the override above can name any
    // pair) is refused here, at plan creation, not at the first launch.
    auto is = [&](int t, int k) { return c.threads == t && c.kpt == k; };
    bool known = is(256, 1) || is(256, 4) || is(512, 8);
    if (payload_bytes == 16) known = known || is(1024, 8);
    if (payload_bytes <= 4) known = known || is(1024, 16);
    if (small_pairs) known = known || is(512, 16) || is(512, 32) || is(1024, 32);
    if (!known) return false;
    if (cfg) *cfg = c;
    return true;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

constexpr size_t kControlBytes = (sizeof(OnesweepControl) + 255) / 256 * 256;

// ---- scratch of mrs_sort_pairs_noalloc -------------------------------------------------
// The temporary memory of a onesweep plan (error flag, control block, tile status words) for
// inputs of at most kNoallocMaxN (u32 key, u32 payload) pairs, as one array of the code object:
// every device the library runs on holds its own copy from the moment the code object is loaded
// until the process ends.  2048 tiles is the most any n <= 2^26 yields (32768-key tiles from
// 2^23 elements on, at most 512 smaller tiles below); 4 passes of 256 four-byte status words each.
constexpr uint64_t kNoallocMaxN = 1ull << 26;
constexpr size_t kNoallocStatusBytes = 2048ull * 256 * 4 * 4;
constexpr size_t kNoallocScratchBytes = 256 + kControlBytes + kNoallocStatusBytes;
static __device__ __align__(256) unsigned char scratch[kNoallocScratchBytes];

// ---- which sorts may use die-aware tickets --------------------------------------------------
// Two chained-scan passes that both deal their tiles per die can block each other when they run
// at the same time (each may hold the compute units of a die the other needs a slot on), so at
// most one such pass may be in flight per device.  Every sort call that uses die-aware tickets
// leaves an event behind its last launch; a call on another stream that finds the event of the
// previous call still pending runs with a single ticket list, whose progress needs nothing but
// one free slot anywhere.  A single-list pass and a die-aware pass can share the device.
constexpr int kMaxDevices = 64;
struct DeviceActivity {
    std::mutex m;
    hipEvent_t event = nullptr;  // created by mrs_context_create with the device current
    hipStream_t stream = nullptr;
    // Calls of one lineage (the plans of one sharded-sort handle) are ordered among themselves by
    // events even where they use two streams, and count as one stream here.
    const void* lineage = nullptr;
    bool recorded = false;
    // A call that took die-aware tickets and is still being enqueued by its host thread: its event
    // does not exist yet, so a call that begins on another thread meanwhile goes by these.
    int open = 0;
    hipStream_t open_stream = nullptr;
    const void* open_lineage = nullptr;
};
static DeviceActivity g_activity[kMaxDevices];

static bool stream_is_capturing(hipStream_t s)
{
    if (!s) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

static int check_device(const mrs_context* ctx, const char* who)
{
    int cur = -1;
    MRS_HIP_TRY(hipGetDevice(&cur));
    if (cur != ctx->device)
        return fail(MRS_ERR_INVALID_ARGUMENT, "%s: context is for device %d but the current device is %d", who,
                    ctx->device, cur);
    return MRS_OK;
}

static uint32_t pass_shift(const mrs_plan* p, uint32_t pass) { return p->desc.begin_bit + pass * p->radix_bits; }
static uint32_t pass_mask(const mrs_plan* p, uint32_t pass)
{
    const uint32_t shift = pass_shift(p, pass);
    const uint32_t width = p->desc.end_bit - shift < p->radix_bits ? p->desc.end_bit - shift : p->radix_bits;
    return (1u << width) - 1u;
}

// Consecutive tiles are handed to workgroups of the same die in blocks of this many, so that the
// output lines two neighbouring tiles share meet in one L2 (8 % of the pass time at 2^28 pairs).
// Blocks must stay well below the tiles a die has in flight (32 with one workgroup per compute
// unit), or the dies wait for each other.  0 = a single ticket list.
static uint32_t xcd_block_tiles(const mrs_context* ctx)
{
    const int env = ctx->tuning.xcd_block;
    if (!ctx->die_aware_tickets) return 0;
    return env >= 0 ? static_cast<uint32_t>(env > 64 ? 64 : env) : 8u;  // larger blocks only make the dies wait
}

// Ticket block size for the chained-scan passes of one call on stream `s`: the context's, unless
// a die-aware call on another stream of this device may still be running or is being enqueued by
// another thread right now (DeviceActivity).  `*counted` is set when the call was entered in
// DeviceActivity::open and end_ticketed_call has to take it out again.
// Under stream capture nothing can be queried: the captured work keeps the context's setting,
// and replays that overlap on one device are the caller's to avoid.
static uint32_t begin_ticketed_call(const mrs_context* ctx, hipStream_t s, const void* lineage, bool* counted)
{
    *counted = false;
    uint32_t block = xcd_block_tiles(ctx);
    if (block == 0 || stream_is_capturing(s)) return block;
    DeviceActivity& a = g_activity[ctx->device];
    std::lock_guard<std::mutex> lock(a.m);
    bool other = false;
    if (a.open > 0 && !(a.open_stream == s || (lineage && lineage == a.open_lineage))) other = true;
    if (!other && a.recorded && a.event && !(a.stream == s || (lineage && lineage == a.lineage))) {
        if (hipEventQuery(a.event) != hipSuccess) other = true;
        (void)hipGetLastError();  // "not ready" is an answer, not an error to report later
    }
    if (other) return 0;
    ++a.open;
    a.open_stream = s;
    a.open_lineage = lineage;
    *counted = true;
    return block;
}

// Leaves the mark the next call looks for.  Only calls that did use die-aware tickets need one.
static void end_ticketed_call(const mrs_context* ctx, hipStream_t s, uint32_t block, const void* lineage, bool counted)
{
    if (!counted) return;
    DeviceActivity& a = g_activity[ctx->device];
    std::lock_guard<std::mutex> lock(a.m);
    if (a.open > 0) --a.open;
    if (block == 0 || !a.event) return;
    if (hipEventRecord(a.event, s) == hipSuccess) {
        a.stream = s;
        a.lineage = lineage;
        a.recorded = true;
    } else {
        (void)hipGetLastError();
    }
}

// One call's worth of the two functions above: whatever way the call is left, the entry in
// DeviceActivity::open goes with it and the event is recorded behind what was enqueued.
class TicketedCall {
public:
    TicketedCall(const mrs_context* ctx, hipStream_t s, const void* lineage, bool chained_scan = true)
        : ctx_(ctx), s_(s), lineage_(lineage)
    {
        if (chained_scan) block_ = begin_ticketed_call(ctx, s, lineage, &counted_);
    }
    ~TicketedCall() { end_ticketed_call(ctx_, s_, block_, lineage_, counted_); }
    TicketedCall(const TicketedCall&) = delete;
    TicketedCall& operator=(const TicketedCall&) = delete;
    uint32_t block() const { return block_; }

private:
    const mrs_context* ctx_;
    hipStream_t s_;
    const void* lineage_;
    uint32_t block_ = 0;
    bool counted_ = false;
};

// A stream that is about to be destroyed must not stay behind as "where the last die-aware call
// ran": the next call would query an event whose stream is gone.  Waits for the stream first.
void forget_stream_activity(int device, hipStream_t s)
{
    if (device < 0 || device >= kMaxDevices || !s) return;
    (void)hipStreamSynchronize(s);
    DeviceActivity& a = g_activity[device];
    std::lock_guard<std::mutex> lock(a.m);
    if (a.stream == s) {
        a.stream = nullptr;
        a.lineage = nullptr;
        a.recorded = false;
    }
}

static int launch_pass(const mrs_plan* p, int mode, const PassArgs& args, hipStream_t s)
{
    const ProfilerRange range("Indexing");  // src/main.rs:658
    PassArgs a = args;
    a.atomic_rank = p->ctx->lds_atomics_in_lane_order;
    if (p->desc.key_words == 1)
        return launch_pass_u32(mode, p->radix_bits, p->threads, p->kpt, p->desc.payload_bytes, a, s);
    return launch_pass_u64(mode, p->radix_bits, p->threads, p->kpt, p->desc.payload_bytes, a, s);
}

static int launch_block_hist(const mrs_plan* p, uint32_t n, const void* keys, uint32_t num_tiles, uint32_t pass,
                             uint32_t* hist, hipStream_t s)
{
    const ProfilerRange range("Histogram calculation");  // src/main.rs:597
    if (p->desc.key_words == 1)
        return launch_block_histogram_u32(p->radix_bits, n, keys, p->tile_items, num_tiles, pass_shift(p, pass),
                                          pass_mask(p, pass), hist, s);
    return launch_block_histogram_u64(p->radix_bits, n, keys, p->tile_items, num_tiles, pass_shift(p, pass),
                                      pass_mask(p, pass), hist, s);
}

static int launch_global_hist(const mrs_plan* p, const GlobalHistArgs& a, hipStream_t s)
{
    const ProfilerRange range("Histogram calculation");  // src/main.rs:597
    const uint64_t vec = p->desc.key_words == 1 ? 4 : 2;
    // Every workgroup ends with up to passes x radix atomic adds on the same global words, so
    // each gets at least 8 vectors per thread to amortise them.
    const uint64_t per_cu = p->ctx->tuning.hist_blocks_per_cu;
    uint64_t blocks = (a.n / vec + 256 * 8 - 1) / (256 * 8);
    const uint64_t cap = static_cast<uint64_t>(p->ctx->num_cus) * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    // The sorts use the kernel with bank-spread counters: 1024 threads and 64 KiB of LDS per
    // workgroup, one per compute unit (two measured 2-3 % slower), at least 4 vectors per thread
    // (mrs::Tuning: the plain kernel instead, more workgroups).
    const bool plain = p->ctx->tuning.hist_plain;
    const uint64_t banked_per_cu = p->ctx->tuning.hist_banked_per_cu;
    uint64_t banked = 0;
    if (!plain) {
        banked = (a.n / vec + 1024 * 4 - 1) / (1024 * 4);
        const uint64_t banked_cap = static_cast<uint64_t>(p->ctx->num_cus) * banked_per_cu;
        if (banked > banked_cap) banked = banked_cap;
        if (banked == 0) banked = 1;
    }
    if (p->desc.key_words == 1)
        return launch_global_histogram_u32(p->radix_bits, static_cast<int>(blocks), static_cast<int>(banked), a, s);
    return launch_global_histogram_u64(p->radix_bits, static_cast<int>(blocks), static_cast<int>(banked), a, s);
}

// `n` != 0: the scan also fills the plan's pass_plan words (which passes need not run).
static int launch_ghist_scan(const mrs_plan* p, uint32_t num_passes, uint32_t n, hipStream_t s, uint32_t keep_passes = 0)
{
    const ProfilerRange range("Small Prefix Sum");  // src/main.rs:625
    const unsigned threads = p->radix < 64 ? 64 : p->radix;
    uint32_t* pass_plan = n ? p->control->pass_plan : nullptr;
    if (p->radix_bits == 4)
        hipLaunchKernelGGL(global_histogram_scan_kernel<4>, dim3(num_passes), dim3(threads), 0, s, p->control->ghist,
                           pass_plan, n, num_passes, keep_passes);
    else
        hipLaunchKernelGGL(global_histogram_scan_kernel<8>, dim3(num_passes), dim3(threads), 0, s, p->control->ghist,
                           pass_plan, n, num_passes, keep_passes);
    MRS_CHECK_LAUNCH("global_histogram_scan_kernel");
    return MRS_OK;
}

static int validate_call(const mrs_plan* plan, uint64_t n, const char* who)
{
    if (!plan) return fail(MRS_ERR_INVALID_ARGUMENT, "%s: null plan", who);
    if (n > plan->desc.n) return fail(MRS_ERR_INVALID_ARGUMENT, "%s: n = %llu exceeds the plan capacity %llu", who,
                                      (unsigned long long)n, (unsigned long long)plan->desc.n);
    return check_device(plan->ctx, who);
}

// Records a timing mark after the launch called `label` (no-op unless timing is on).
static int mark(mrs_plan* p, const char* label, hipStream_t s)
{
    if (!p->timing) return MRS_OK;
    const int cap = static_cast<int>(sizeof(p->marks) / sizeof(p->marks[0]));
    if (p->num_marks >= cap) return MRS_OK;
    MRS_HIP_TRY(hipEventRecord(p->marks[p->num_marks], s));
    p->mark_labels[p->num_marks] = label;
    p->num_marks++;
    return MRS_OK;
}

static bool misaligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; }

// ---- bucket histogram (multi-GPU split) ---------------------------------------------
// One LDS atomic per key into `bins` <= 4096 counters; 16-byte loads, four in flight per thread,
// when the key array is 16-byte aligned (`vectors` = n / keys per vector, else 0).  Few, large
// workgroups: each ends with up to `bins` 64-bit global atomics on the same words.
template <typename KeyT>
__global__ __launch_bounds__(1024) void bucket_histogram_kernel(const KeyT* __restrict__ keys, unsigned long long n,
                                                                 unsigned long long vectors, uint32_t shift,
                                                                 uint32_t bins, unsigned long long* __restrict__ counts)
{
    constexpr int THREADS = 1024, UNROLL = 4;
    constexpr int VEC = 16 / sizeof(KeyT);
    using VecT = typename std::conditional<sizeof(KeyT) == 4, uint4, ulonglong2>::type;
    extern __shared__ uint32_t s_bins[];
    for (uint32_t j = threadIdx.x; j < bins; j += THREADS) s_bins[j] = 0;
    __syncthreads();
    const uint32_t mask = bins - 1;
    auto count = [&](KeyT k) { atomicAdd(&s_bins[static_cast<uint32_t>(k >> shift) & mask], 1u); };
    auto count_vec = [&](const VecT& x) {
        count(x.x);
        count(x.y);
        if constexpr (sizeof(KeyT) == 4) {
            count(x.z);
            count(x.w);
        }
    };
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * THREADS;
    const unsigned long long first = static_cast<unsigned long long>(blockIdx.x) * THREADS + threadIdx.x;
    const VecT* __restrict__ vkeys = reinterpret_cast<const VecT*>(keys);
    unsigned long long v = first;
    for (; v + (UNROLL - 1) * stride < vectors; v += UNROLL * stride) {
        VecT x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = vkeys[v + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) count_vec(x[u]);
    }
    for (; v < vectors; v += stride) count_vec(vkeys[v]);
    // what the vectors do not cover: everything if the array is misaligned, else at most VEC - 1 keys
    for (unsigned long long i = vectors * VEC + first; i < n; i += stride) count(keys[i]);
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < bins; j += THREADS) {
        const uint32_t c = s_bins[j];
        if (c) atomicAdd(&counts[j], static_cast<unsigned long long>(c));
    }
}

}  // namespace mrs

using namespace mrs;

#ifdef MRS_STAMPS
// diagnostic builds only: device buffer that pass 1 of the next sorts writes its phase clocks to
static unsigned long long* g_debug_stamps = nullptr;
extern "C" void mrs_debug_set_stamps(unsigned long long* dev) { g_debug_stamps = dev; }
#endif

extern "C" const char* mrs_last_error_string(void) { return g_last_error.c_str(); }

extern "C" int mrs_version(void) { return (MRS_VERSION_MAJOR << 16) | MRS_VERSION_MINOR; }

extern "C" int mrs_context_create(int device_id, mrs_context** out)
{
    clear_error();
    if (!out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_context_create: null output pointer");
    *out = nullptr;
    int count = 0;
    MRS_HIP_TRY(hipGetDeviceCount(&count));
    if (device_id < 0 || device_id >= count)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_context_create: device %d out of range (%d devices)", device_id, count);
    hipDeviceProp_t prop;
    MRS_HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MRS_ERR_UNSUPPORTED, "mrs_context_create: device %d is %s; this library is built for gfx950 only",
                    device_id, prop.gcnArchName);
    if (device_id >= kMaxDevices)
        return fail(MRS_ERR_UNSUPPORTED, "mrs_context_create: at most %d devices are supported", kMaxDevices);
    bool in_order = false, all_dies = false;
    void* scratch_ptr = nullptr;
    {
        int prev = -1;
        MRS_HIP_TRY(hipGetDevice(&prev));
        MRS_HIP_TRY(hipSetDevice(device_id));
        int rc = MRS_OK;
        rc = probe_lds_atomic_order(&in_order);
        if (rc == MRS_OK && prop.multiProcessorCount == 256) rc = probe_die_ids(&all_dies);
        // This device's copy of the static scratch array.  `scratch` itself is a host-side stand-in
        // for the symbol: only the address obtained here, with the device current, may be handed
        // to kernels, memsets and copies.
        if (rc == MRS_OK) {
            hipError_t e = hipGetSymbolAddress(&scratch_ptr, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !scratch_ptr)
                rc = fail(MRS_ERR_HIP, "mrs_context_create: hipGetSymbolAddress(scratch) failed on device %d: %s",
                          device_id, hipGetErrorString(e));
        }
        if (rc == MRS_OK) {
            DeviceActivity& a = g_activity[device_id];
            std::lock_guard<std::mutex> lock(a.m);
            if (!a.event) {
                hipError_t e = hipEventCreateWithFlags(&a.event, hipEventDisableTiming);
                if (e != hipSuccess) {
                    a.event = nullptr;
                    rc = fail(MRS_ERR_HIP, "mrs_context_create: hipEventCreate failed: %s", hipGetErrorString(e));
                }
            }
        }
        MRS_HIP_TRY(hipSetDevice(prev));
        if (rc) return rc;
    }
    mrs_context* c = new mrs_context;
    c->device = device_id;
    c->num_cus = prop.multiProcessorCount;
    c->lds_atomics_in_lane_order = in_order;
    c->lds_atomic_probe_passed = in_order;
    c->workgroups_on_all_dies = all_dies;
    c->die_aware_tickets = all_dies;
    // tuning aids of profiling scripts: read here, never inside a sort call (MRS_TILE_SHAPE: at the
    // first plan creation)
    if (const char* e = std::getenv("MRS_XCD_BLOCK")) c->tuning.xcd_block = std::atoi(e);
    if (const char* e = std::getenv("MRS_HIST_BLOCKS_PER_CU")) c->tuning.hist_blocks_per_cu = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("MRS_HIST_BANKED_PER_CU")) c->tuning.hist_banked_per_cu = std::max(1, std::atoi(e));
    c->tuning.hist_plain = std::getenv("MRS_HIST_PLAIN") != nullptr;
    c->live_plans = 0;
    c->static_scratch = scratch_ptr;
    *out = c;
    return MRS_OK;
}

extern "C" int mrs_context_get_attribute(const mrs_context* ctx, int attribute, int64_t* value)
{
    clear_error();
    if (!ctx || !value) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_context_get_attribute: null argument");
    switch (attribute) {
        case MRS_ATTR_COMPUTE_UNITS: *value = ctx->num_cus; return MRS_OK;
        case MRS_ATTR_RANK_MODE: *value = ctx->lds_atomics_in_lane_order ? 1 : 0; return MRS_OK;
        case MRS_ATTR_DIE_AWARE_TICKETS: *value = ctx->die_aware_tickets ? 1 : 0; return MRS_OK;
        case MRS_ATTR_NOALLOC_TIMEOUT: {
            // the error flag is the first word of the device's static scratch array
            if (int rc = check_device(ctx, "mrs_context_get_attribute")) return rc;
            MRS_HIP_TRY(hipDeviceSynchronize());
            uint32_t flag = 0;
            MRS_HIP_TRY(hipMemcpy(&flag, ctx->static_scratch, sizeof flag, hipMemcpyDeviceToHost));
            if (flag) MRS_HIP_TRY(hipMemset(ctx->static_scratch, 0, sizeof flag));
            *value = flag ? 1 : 0;
            return MRS_OK;
        }
        case MRS_ATTR_ROCTX_RANGES: *value = g_ranges_on.load() ? 1 : 0; return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_MIN_N: *value = static_cast<int64_t>(ctx->aids.hybrid_min_n); return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_LEADING_BITS: *value = ctx->aids.hybrid_leading_bits; return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_FORCE_FALLBACK: *value = ctx->aids.hybrid_force_fallback ? 1 : 0; return MRS_OK;
        case MRS_ATTR_TEST_NO_SMALL_SORT: *value = ctx->aids.no_small_sort ? 1 : 0; return MRS_OK;
        case MRS_ATTR_TEST_NO_PASS_SKIP: *value = ctx->aids.no_pass_skip ? 1 : 0; return MRS_OK;
        case MRS_ATTR_TEST_WIDE_STATUS: *value = ctx->aids.wide_status ? 1 : 0; return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_FORCE_ADAPTIVE: *value = ctx->aids.hybrid_force_adaptive ? 1 : 0; return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_NO_ADAPTIVE: *value = ctx->aids.hybrid_no_adaptive ? 1 : 0; return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_CAPACITY: *value = ctx->aids.hybrid_capacity; return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_BIG_LOCAL: *value = ctx->aids.hybrid_big_local ? 1 : 0; return MRS_OK;
        default: return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_context_get_attribute: unknown attribute %d", attribute);
    }
}

extern "C" int mrs_context_set_attribute(mrs_context* ctx, int attribute, int64_t value)
{
    clear_error();
    if (!ctx) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_context_set_attribute: null context");
    switch (attribute) {
        case MRS_ATTR_DIE_AWARE_TICKETS:
            if (value != 0 && !ctx->workgroups_on_all_dies)
                return fail(MRS_ERR_UNSUPPORTED, "mrs_context_set_attribute: workgroups were not seen on all 8 dies of this device");
            ctx->die_aware_tickets = value != 0;
            return MRS_OK;
        case MRS_ATTR_RANK_MODE:
            if (value != 0 && !ctx->lds_atomic_probe_passed)
                return fail(MRS_ERR_UNSUPPORTED, "mrs_context_set_attribute: this device failed the lane-order check of returning LDS atomics");
            ctx->lds_atomics_in_lane_order = value != 0;
            return MRS_OK;
        case MRS_ATTR_ROCTX_RANGES:
            if (value != 0)
                if (int rc = bind_roctx()) return rc;
            g_ranges_on.store(value != 0);
            return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_MIN_N:
            if (value < 0) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_context_set_attribute: negative size");
            ctx->aids.hybrid_min_n = static_cast<uint64_t>(value);
            return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_LEADING_BITS:
            if (value != 0 && (value < 14 || value > 16))
                return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_context_set_attribute: leading bits must be 0 (default) or 14..16");
            ctx->aids.hybrid_leading_bits = static_cast<uint32_t>(value);
            return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_FORCE_FALLBACK: ctx->aids.hybrid_force_fallback = value != 0; return MRS_OK;
        case MRS_ATTR_TEST_NO_SMALL_SORT: ctx->aids.no_small_sort = value != 0; return MRS_OK;
        case MRS_ATTR_TEST_NO_PASS_SKIP: ctx->aids.no_pass_skip = value != 0; return MRS_OK;
        case MRS_ATTR_TEST_WIDE_STATUS: ctx->aids.wide_status = value != 0; return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_FORCE_ADAPTIVE: ctx->aids.hybrid_force_adaptive = value != 0; return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_NO_ADAPTIVE: ctx->aids.hybrid_no_adaptive = value != 0; return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_BIG_LOCAL: ctx->aids.hybrid_big_local = value != 0; return MRS_OK;
        case MRS_ATTR_TEST_HYBRID_CAPACITY:
            if (value < 0 || value > kHybLocalCapCompact)
                return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_context_set_attribute: capacity must be 0 (default) or 1..%u", kHybLocalCapCompact);
            ctx->aids.hybrid_capacity = static_cast<uint32_t>(value);
            return MRS_OK;
        default: return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_context_set_attribute: attribute %d cannot be set", attribute);
    }
}

extern "C" int mrs_context_destroy(mrs_context* ctx)
{
    clear_error();
    if (ctx && ctx->live_plans != 0)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_context_destroy: %d plan(s) of this context are still alive; destroy them first",
                    ctx->live_plans);
    delete ctx;
    return MRS_OK;
}

// Validates a plan description and fills every field of `p` that does not depend on memory:
// shape, pass count, tile count.  `p` must be zero-initialised.
static int fill_plan(mrs_plan* p, mrs_context* ctx, const mrs_plan_desc* d, const char* who)
{
    if (d->n == 0 || d->n >= 0xFFFFFFFFull) return fail(MRS_ERR_INVALID_ARGUMENT, "%s: n must satisfy 0 < n < 2^32 - 1", who);
    if (d->algo != MRS_ALGO_THREE_KERNEL && d->algo != MRS_ALGO_ONESWEEP && d->algo != MRS_ALGO_HYBRID)
        return fail(MRS_ERR_INVALID_ARGUMENT, "%s: unknown algo %d", who, d->algo);
    mrs_plan_desc resolved = *d;
    bool hybrid = false;
    if (d->algo == MRS_ALGO_HYBRID) {
        // The schedule of mrs_hybrid.hpp exists for one element shape; it is an MRS_ALGO_ONESWEEP plan
        // with more temporary memory and a device-side choice between two schedules.
        const int width = d->end_bit - d->begin_bit;
        if (d->key_words != 1 || d->payload_bytes != 4 || d->digit_bits != 8 || width < 25 || width > 32 ||
            d->n > (1ull << 30) || (d->tile_items != 0 && d->tile_items != 4096 && d->tile_items != 32768))
            return fail(MRS_ERR_UNSUPPORTED,
                        "%s: MRS_ALGO_HYBRID sorts (u32 key, 4-byte payload) pairs on 25 to 32 key bits with 8-bit digits, "
                        "n <= 2^30, tile_items 0, 4096 or 32768",
                        who);
        resolved.algo = MRS_ALGO_ONESWEEP;
        // ranking by returning LDS atomics is all the gated kernels are built for; a device that
        // failed the lane-order check gets the plain chained-scan plan, and so does a plan too small
        // ever to be offered the schedule (below 3 * 2^24 pairs; MRS_ATTR_TEST_HYBRID_MIN_N lifts that)
        hybrid = ctx->lds_atomics_in_lane_order && !ctx->aids.wide_status &&
                 (d->n >= (3ull << 24) || ctx->aids.hybrid_min_n != 0);
        // the gated kernels exist for two tile shapes
        if (hybrid && resolved.tile_items == 0) resolved.tile_items = d->n >= (1ull << 20) ? 32768 : 4096;
        d = &resolved;
    }
    KernelConfig cfg;
    if (!config_supported(d->n, d->key_words, d->payload_bytes, d->digit_bits, d->tile_items, d->algo, &cfg))
        return fail(MRS_ERR_UNSUPPORTED, "%s: unsupported shape (key_words %d, payload_bytes %d, digit_bits %d, tile_items %d)",
                    who, d->key_words, d->payload_bytes, d->digit_bits, d->tile_items);
    const int key_bits = 32 * d->key_words;
    if (d->begin_bit < 0 || d->end_bit > key_bits || d->begin_bit >= d->end_bit)
        return fail(MRS_ERR_INVALID_ARGUMENT, "%s: bit range [%d, %d) invalid for a %d-bit key", who, d->begin_bit,
                    d->end_bit, key_bits);
    p->ctx = ctx;
    p->aids = ctx->aids;
    p->desc = *d;
    p->hybrid = hybrid;
    p->radix_bits = d->digit_bits;
    p->radix = 1u << d->digit_bits;
    p->num_passes = (d->end_bit - d->begin_bit + d->digit_bits - 1) / d->digit_bits;
    p->threads = cfg.threads;
    p->kpt = cfg.kpt;
    p->tile_items = cfg.threads * cfg.kpt;
    p->desc.tile_items = p->tile_items;
    p->max_tiles = (d->n + p->tile_items - 1) / p->tile_items;
    p->histogram_len = static_cast<uint64_t>(p->radix) * p->max_tiles;
    if (p->num_passes > static_cast<uint32_t>(kMaxPasses)) return fail(MRS_ERR_UNSUPPORTED, "%s: too many passes", who);
    if (d->algo == MRS_ALGO_ONESWEEP) {
        // 32-bit status words carry counts below 2^30; the kernels instantiated for them also
        // address keys and payload with 32-bit byte offsets, so no array may exceed 4 GiB
        const unsigned long long elem_bytes =
            std::max<unsigned long long>(4ull * d->key_words, static_cast<unsigned long long>(d->payload_bytes));
        p->status_word_bytes = (d->n < (1ull << 30) && d->n * elem_bytes <= (1ull << 32)) ? 4 : 8;
        // test aid: the 64-bit instantiation on inputs small enough to check element by element
        if (p->aids.wide_status) p->status_word_bytes = 8;
    }
    return MRS_OK;
}

// Tiles the passes of the three-move schedule can have on an input of `num_tiles` tiles: pass B has
// up to 256 more than the array (every bucket ends in a tile of its own); the two passes over the
// oversized sub-buckets run only when those hold at most a quarter of the pairs, in at most 256
// sub-buckets, each of which ends in a tile of its own.
static uint64_t hybrid_tiles_b(uint64_t num_tiles) { return num_tiles + kHybBuckets; }
static uint64_t hybrid_tiles_big(uint64_t num_tiles) { return num_tiles / 4 + kHybMaxBig + 1; }

// Status-word bytes one call on `num_tiles` tiles uses (and clears).  The two schedules of a
// hybrid plan never run in the same call, so they share the space: pass A takes the first pass
// region, pass B what follows it, the passes over oversized sub-buckets what follows that.  Pass A
// of the adaptive schedule has a region behind all of that: the ordinary passes may still run
// after it, and they need theirs untouched.
static size_t hybrid_adaptive_status_offset(const mrs_plan* p, uint64_t num_tiles)
{
    const size_t per_tile = static_cast<size_t>(p->radix) * p->status_word_bytes;
    const size_t passes = num_tiles * per_tile * p->num_passes;
    const size_t msd = (num_tiles + hybrid_tiles_b(num_tiles) + 2 * hybrid_tiles_big(num_tiles)) * per_tile;
    return msd > passes ? msd : passes;
}
static size_t status_bytes(const mrs_plan* p, uint64_t num_tiles)
{
    const size_t per_tile = static_cast<size_t>(p->radix) * p->status_word_bytes;
    size_t bytes = num_tiles * per_tile * p->num_passes;
    if (p->hybrid) bytes = hybrid_adaptive_status_offset(p, num_tiles) + num_tiles * per_tile;
    return align_up(bytes, 256);
}

constexpr size_t kHybridClearedBytes = (sizeof(HybridCleared) + 255) / 256 * 256;

// Device bytes a onesweep plan needs: [error flag, 256 B][control block][hybrid plans: header,
// spill counts, digit counts of oversized sub-buckets][status words][hybrid plans: tables, packed
// histograms].
static size_t onesweep_temp_bytes(const mrs_plan* p)
{
    size_t bytes = 256 + kControlBytes + status_bytes(p, p->max_tiles);
    if (p->hybrid)
        bytes += kHybridClearedBytes + hybrid_tables_bytes() +
                 hybrid_packed_bytes(hybrid_histogram_grid(p->ctx->num_cus, p->desc.n));
    return bytes;
}

// Points the plan at `temp`, a zeroed block of onesweep_temp_bytes(p) bytes.  Everything after
// the error flag is cleared by one memset at the start of every sort call.
static void bind_onesweep_memory(mrs_plan* p, void* temp)
{
    p->temp = temp;
    p->error_flag = static_cast<uint32_t*>(temp);
    p->zero_begin = static_cast<char*>(temp) + 256;
    p->control = reinterpret_cast<OnesweepControl*>(p->zero_begin);
    p->status = p->zero_begin + kControlBytes;
    if (p->hybrid) {
        p->hyb_cleared = reinterpret_cast<HybridCleared*>(p->status);
        p->status += kHybridClearedBytes;
        char* after = p->status + status_bytes(p, p->max_tiles);
        p->hyb_tables = reinterpret_cast<HybridTables*>(after);
        p->hyb_packed = reinterpret_cast<uint32_t*>(after + hybrid_tables_bytes());
    }
}

extern "C" int mrs_plan_create(mrs_context* ctx, const mrs_plan_desc* d, mrs_plan** out)
{
    clear_error();
    if (!ctx || !d || !out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_plan_create: null argument");
    *out = nullptr;
    if (int rc = check_device(ctx, "mrs_plan_create")) return rc;
    mrs_plan* p = new mrs_plan();
    if (int rc = fill_plan(p, ctx, d, "mrs_plan_create")) {
        delete p;
        return rc;
    }
    size_t bytes = 0;
    if (d->algo == MRS_ALGO_THREE_KERNEL) {
        const size_t hist_bytes = align_up(p->histogram_len * sizeof(uint32_t), 256);
        p->scan_partials_len = scan_partials_len(p->histogram_len);
        const size_t part_bytes = align_up(p->scan_partials_len * sizeof(uint32_t), 256);
        bytes = hist_bytes + part_bytes;
        hipError_t e = hipMalloc(&p->temp, bytes);
        if (e != hipSuccess) {
            delete p;
            return fail(MRS_ERR_OUT_OF_MEMORY, "mrs_plan_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        }
        p->hist = static_cast<uint32_t*>(p->temp);
        p->scan_partials = reinterpret_cast<uint32_t*>(static_cast<char*>(p->temp) + hist_bytes);
    } else {
        bytes = onesweep_temp_bytes(p);
        void* temp = nullptr;
        hipError_t e = hipMalloc(&temp, bytes);
        if (e != hipSuccess) {
            delete p;
            return fail(MRS_ERR_OUT_OF_MEMORY, "mrs_plan_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        }
        bind_onesweep_memory(p, temp);
        // The first sort may run on a stream that does not wait for the null stream: the fill must
        // have finished, not merely been queued, before the plan is handed out.
        e = hipMemset(p->temp, 0, bytes);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) {
            (void)hipFree(p->temp);
            delete p;
            return fail(MRS_ERR_HIP, "mrs_plan_create: hipMemset failed: %s", hipGetErrorString(e));
        }
    }
    p->temp_bytes = bytes;
    p->owns_temp = true;
    ctx->live_plans++;
    *out = p;
    return MRS_OK;
}

extern "C" int mrs_plan_destroy(mrs_plan* plan)
{
    clear_error();
    if (!plan) return MRS_OK;
    if (plan->timing) {
        const int cap = static_cast<int>(sizeof(plan->marks) / sizeof(plan->marks[0]));
        for (int i = 0; i < cap; ++i) MRS_TEARDOWN(hipEventDestroy(plan->marks[i]));
    }
    if (plan->temp && plan->owns_temp) MRS_TEARDOWN(hipFree(plan->temp));
    if (plan->ctx) plan->ctx->live_plans--;
    delete plan;
    return MRS_OK;
}

extern "C" int mrs_plan_get_info(const mrs_plan* plan, mrs_plan_info* info)
{
    clear_error();
    if (!plan || !info) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_plan_get_info: null argument");
    info->num_passes = plan->num_passes;
    info->radix = plan->radix;
    info->tile_items = plan->tile_items;
    info->threads = plan->threads;
    info->max_tiles = plan->max_tiles;
    info->histogram_len = plan->histogram_len;
    info->temp_bytes = plan->temp_bytes;
    return MRS_OK;
}

extern "C" int mrs_plan_check(mrs_plan* plan, void* stream)
{
    clear_error();
    if (!plan) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_plan_check: null plan");
    if (int rc = check_device(plan->ctx, "mrs_plan_check")) return rc;
    MRS_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    if (!plan->error_flag) return MRS_OK;
    uint32_t flag = 0;
    MRS_HIP_TRY(hipMemcpy(&flag, plan->error_flag, sizeof flag, hipMemcpyDeviceToHost));
    if (flag) {
        MRS_HIP_TRY(hipMemset(plan->error_flag, 0, sizeof flag));
        return fail(MRS_ERR_DEVICE_TIMEOUT, "a chained-scan workgroup timed out waiting for its predecessor tile; "
                                            "the output of that sort call is invalid");
    }
    return MRS_OK;
}

extern "C" int mrs_plan_get_passes_left_out(mrs_plan* plan, void* stream, uint32_t* left_out_mask)
{
    clear_error();
    if (!plan || !left_out_mask) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_plan_get_passes_left_out: null argument");
    if (int rc = check_device(plan->ctx, "mrs_plan_get_passes_left_out")) return rc;
    *left_out_mask = 0;
    MRS_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    if (!plan->pass_plan_live) return MRS_OK;
    uint32_t words[kMaxPasses];
    MRS_HIP_TRY(hipMemcpy(words, plan->control->pass_plan, sizeof words, hipMemcpyDeviceToHost));
    for (uint32_t p = 0; p < plan->num_passes; ++p)
        if (words[p] & 1u) *left_out_mask |= 1u << p;
    return MRS_OK;
}

extern "C" int mrs_plan_get_schedule(mrs_plan* plan, void* stream, int32_t* schedule, uint32_t* largest_sub_bucket)
{
    clear_error();
    if (!plan || !schedule) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_plan_get_schedule: null argument");
    if (int rc = check_device(plan->ctx, "mrs_plan_get_schedule")) return rc;
    *schedule = MRS_SCHEDULE_PASSES;
    if (largest_sub_bucket) *largest_sub_bucket = 0;
    MRS_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    if (!plan->hyb_last) return MRS_OK;
    HybridHeader h;
    MRS_HIP_TRY(hipMemcpy(&h, &plan->hyb_cleared->hdr, sizeof h, hipMemcpyDeviceToHost));
    const uint32_t mode = hybrid_mode(h);
    *schedule = mode == kHybRun ? MRS_SCHEDULE_MSD : (mode == kHybAdapt ? MRS_SCHEDULE_MSD_ADAPTIVE : MRS_SCHEDULE_MSD_REFUSED);
    if (largest_sub_bucket) *largest_sub_bucket = h.max_sub;
    return MRS_OK;
}

extern "C" int mrs_plan_get_oversized(mrs_plan* plan, void* stream, uint32_t* sub_buckets, uint64_t* pairs,
                                      uint32_t* capacity)
{
    clear_error();
    if (!plan) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_plan_get_oversized: null plan");
    if (int rc = check_device(plan->ctx, "mrs_plan_get_oversized")) return rc;
    if (sub_buckets) *sub_buckets = 0;
    if (pairs) *pairs = 0;
    if (capacity) *capacity = 0;
    MRS_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    if (!plan->hyb_last) return MRS_OK;
    HybridHeader h;
    MRS_HIP_TRY(hipMemcpy(&h, &plan->hyb_cleared->hdr, sizeof h, hipMemcpyDeviceToHost));
    const uint32_t mode = hybrid_mode(h);
    const bool ran = mode == kHybRun || mode == kHybAdapt;
    if (sub_buckets) *sub_buckets = ran ? h.num_big : 0;
    if (pairs) *pairs = ran ? h.big_pairs : 0;
    if (capacity) *capacity = plan->hyb_cap;
    return MRS_OK;
}

extern "C" int mrs_plan_get_adaptive(mrs_plan* plan, void* stream, uint32_t* buckets, uint32_t* groups)
{
    clear_error();
    if (!plan) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_plan_get_adaptive: null plan");
    if (int rc = check_device(plan->ctx, "mrs_plan_get_adaptive")) return rc;
    if (buckets) *buckets = 0;
    if (groups) *groups = 0;
    MRS_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    if (!plan->hyb_last) return MRS_OK;
    HybridHeader h;
    MRS_HIP_TRY(hipMemcpy(&h, &plan->hyb_cleared->hdr, sizeof h, hipMemcpyDeviceToHost));
    if (hybrid_mode(h) != kHybAdapt) return MRS_OK;
    if (buckets) *buckets = h.adapt_buckets;
    if (groups) *groups = h.adapt_groups;
    return MRS_OK;
}

static int check_pair_buffers(const mrs_plan* plan, const void* ka, const void* va, const void* kb, const void* vb,
                              const char* who)
{
    const size_t key_align = 4 * plan->desc.key_words;
    if (!ka || !kb) return fail(MRS_ERR_INVALID_ARGUMENT, "%s: null key buffer", who);
    if (misaligned(ka, key_align) || misaligned(kb, key_align))
        return fail(MRS_ERR_INVALID_ARGUMENT, "%s: key buffers must be %zu-byte aligned", who, key_align);
    if (plan->desc.payload_bytes) {
        if (!va || !vb) return fail(MRS_ERR_INVALID_ARGUMENT, "%s: null payload buffer", who);
        if (misaligned(va, plan->desc.payload_bytes) || misaligned(vb, plan->desc.payload_bytes))
            return fail(MRS_ERR_INVALID_ARGUMENT, "%s: payload buffers must be %d-byte aligned", who,
                        plan->desc.payload_bytes);
    }
    if (ka == kb || (va && va == vb)) return fail(MRS_ERR_INVALID_ARGUMENT, "%s: the A and B buffers must differ", who);
    return MRS_OK;
}

// Inputs below this size take the plan's ordinary passes: the local sort visits at least 2^14
// sub-buckets whatever n is, 2^16 when more than 30 key bits are sorted, and each visit has a
// latency of its own (measured break-even: 2^25.5 and 2^27 pairs).  aids.hybrid_min_n: test aid, to
// run the schedule on inputs a checker can follow.
static uint64_t hybrid_min_n(uint32_t width, const TestAids& aids)
{
    if (aids.hybrid_min_n != 0) return aids.hybrid_min_n;
    return width <= 30 ? 1ull << 26 : (width == 31 ? 3ull << 25 : 1ull << 27);
}

// What a workgroup of the local sort holds when the sub-bucket's own bits start at `shift`: with
// 16 bits or fewer below them a pair is staged in 6 bytes (if the device finds no differing bits
// above the sorted range either), else in 8.
static uint32_t hybrid_local_cap(uint32_t shift) { return shift <= 16u ? kHybLocalCapCompact : kHybLocalCapFull; }

// Key bits passes A and B take together: the coarsest split whose sub-buckets hold at most four
// fifths of a workgroup's capacity on evenly spread keys (8192 of 10240 pairs, 6144 of 7680: every
// visit of a sub-bucket costs the local sort about 4 microseconds whatever its size, so fewer,
// fuller sub-buckets are faster -- 2^28 pairs: 0.95 ms with 2^15 sub-buckets, 1.19 ms with 2^16 --
// and the fifth left free takes uneven keys), at least 14, and at least so many that 16 bits or
// fewer are left for the local sort.  A takes the larger half.
static uint32_t hybrid_msd_bits(uint32_t end, uint32_t width, uint32_t n, const TestAids& aids)
{
    uint32_t bits = 14;
    while (bits < 16 && (static_cast<uint64_t>(n) >> bits) > hybrid_local_cap(end - bits) * 4u / 5u) ++bits;
    // test aid: the finer splits on inputs a checker can follow
    if (aids.hybrid_leading_bits >= 14 && aids.hybrid_leading_bits <= 16) bits = aids.hybrid_leading_bits;
    if (bits + 16 < width) bits = width - 16;
    return bits;
}

// How the two scatters and the local sort share the key: passes A and B take `msd_bits` bits from
// `shift` up (of the key itself, or, windowed, of (key >> shift) - base), the local sort bits
// [begin_bit, shift).
struct HybridSplit {
    uint32_t msd_bits, shift, base;
    bool windowed;
    bool big;  // the local sort's workgroups of 1024 threads (20480 pairs): 2^29.1 to 2^30 evenly spread pairs
};

// The split for keys known to lie in [lo, hi]: with 14, 15 or 16 bits for passes A and B, the
// smallest shift that leaves at most that many sub-buckets over the window; the first of the three
// whose sub-buckets would hold about 5000 pairs or fewer if the keys were spread evenly (a window
// that is a little wider than a power of two needs the next shift, and with it twice the
// sub-bucket size) and whose local sort gets at most 16 bits; failing that, the finest split if
// evenly spread keys would still fit a workgroup with a sixth to spare -- a workgroup of 512
// threads, or else one of 1024 (twice the pairs; 6-byte staging only: a quarter of 2^31 pairs, whose
// window is as likely as not a little wider than a quarter of the key space).  `force_big`: test aid,
// the large workgroups or nothing.  False when there is none (a window too wide for this n, or so
// narrow that its keys cannot spread) or the window is empty.
static bool hybrid_window_split(uint32_t begin, uint32_t n, uint64_t lo, uint64_t hi, bool force_big, HybridSplit* out)
{
    if (hi < lo || hi > 0xFFFFFFFFull) return false;
    for (uint32_t attempt = force_big ? 3u : 0u; attempt < 4u; ++attempt) {
        const bool big = attempt == 3u;
        const uint32_t bits = big ? 16u : 14u + attempt;
        uint32_t shift = begin + 2;  // the local sort takes two digits of at least one bit
        while (shift < 32 && ((hi >> shift) - (lo >> shift)) >= (1ull << bits)) ++shift;
        const uint64_t subs = (hi >> shift) - (lo >> shift) + 1;
        const uint64_t limit = big ? kHybLocalCapBig * 5ull / 6
                                   : (bits < 16 ? hybrid_local_cap(shift) * 4ull / 5 : hybrid_local_cap(shift) * 5ull / 6);
        if (shift - begin <= 16 && (!big || shift <= 16u) && n / subs <= limit) {
            *out = HybridSplit{bits, shift, static_cast<uint32_t>(lo >> shift), true, big};
            return true;
        }
    }
    return false;
}

// Whether a sort of n pairs on key bits [begin, end) is offered the three-move schedule at all, and
// with which split.  `window`: null, or two key values every key is known to lie between (the
// sharded sort's local sorts: keys of one slice of the key space).
static bool hybrid_choose_split(uint32_t begin, uint32_t end, uint64_t n, const uint64_t* window, const TestAids& aids,
                                HybridSplit* split)
{
    const uint32_t n32 = static_cast<uint32_t>(n);
    if (window) {
        const uint64_t min_n = aids.hybrid_min_n != 0 ? aids.hybrid_min_n : 3ull << 24;  // slices are never exact powers of two
        return n >= min_n && hybrid_window_split(begin, n32, window[0], window[1], aids.hybrid_big_local != 0, split);
    }
    // Evenly spread keys must leave sub-buckets a workgroup can hold with a sixth to spare (the few
    // that do not fit are sorted by two extra passes, but not most of them): up to about 2^28.7
    // pairs with 8-byte staging, 2^29.1 with 6-byte staging and two workgroups of 512 threads per
    // compute unit, 2^30 with 6-byte staging and one workgroup of 1024 (which holds twice as much but
    // has no neighbour to fill its waits: 1.2 ms per 2^28 pairs instead of 0.98).  Beyond that the
    // device would refuse, so it is not asked.  aids.hybrid_big_local: test aid, the large workgroups
    // at any size.
    const uint32_t msd_bits = hybrid_msd_bits(end, end - begin, n32, aids);
    *split = HybridSplit{msd_bits, end - msd_bits, 0u, false, false};
    if (n < hybrid_min_n(end - begin, aids)) return false;
    const uint64_t per_sub = n >> msd_bits;
    if (per_sub <= hybrid_local_cap(split->shift) * 5ull / 6 && !(aids.hybrid_big_local && split->shift <= 16u)) return true;
    split->big = split->shift <= 16u && per_sub <= kHybLocalCapBig * 5ull / 6;
    return split->big;
}

// The schedule of an MRS_ALGO_HYBRID plan (mrs_hybrid.hpp): both pass sequences are enqueued and the
// device runs one of them.
static int sort_pairs_hybrid(mrs_plan* plan, uint32_t n32, void* keys_a, void* vals_a, void* keys_b, void* vals_b,
                             hipStream_t s, bool indexed, const HybridSplit& split)
{
    const uint32_t num_tiles = (n32 + plan->tile_items - 1) / plan->tile_items;
    const TicketedCall ticketed(plan->ctx, s, plan->lineage);
    const uint32_t xcd_block = ticketed.block();
    const size_t per_tile = static_cast<size_t>(plan->radix) * plan->status_word_bytes;
    const size_t region = num_tiles * per_tile;
    const size_t zero = static_cast<size_t>(plan->status - plan->zero_begin) + status_bytes(plan, num_tiles);
    if (int rc = clear_async(plan->zero_begin, zero, s)) return rc;
    if (int rc = mark(plan, "memset", s)) return rc;
    const uint32_t begin_bit = plan->desc.begin_bit, end_bit = plan->desc.end_bit;
    const int hist_grid = hybrid_histogram_grid(plan->ctx->num_cus, n32);
    const uint32_t msd_bits = split.msd_bits;
#ifdef MRS_SWAP_AB  // A/B builds: pass A takes the smaller half
    const uint32_t bits_a = msd_bits / 2, bits_b = msd_bits - bits_a;
#else
    const uint32_t bits_a = (msd_bits + 1) / 2, bits_b = msd_bits - bits_a;  // 7 or 8 each
#endif
    // a pair of a sub-bucket can be staged as 16 key bits + payload if nothing below the
    // sub-bucket's own bits is lost that way (the device still has a say: bits above end_bit)
#ifdef MRS_NO_COMPACT  // A/B builds: 8-byte staging everywhere
    const bool compact_ok = false;
#else
    const bool compact_ok = split.shift <= 16u;
#endif
    plan->hyb_compact_ok = compact_ok;
    uint32_t cap = split.big ? kHybLocalCapBig : (compact_ok ? kHybLocalCapCompact : kHybLocalCapFull);
    if (plan->aids.hybrid_capacity != 0 && plan->aids.hybrid_capacity < cap) cap = plan->aids.hybrid_capacity;  // test aid
    plan->hyb_cap = cap;
    // Passes loop over tiles with one workgroup per compute unit where that costs nothing (same time
    // per pass within 1 %): a launch with nothing to do is then 256 workgroups instead of 8448 (0.005
    // against 0.10 ms), and one launch can serve several schedules.  Not inside a sharded sort, whose
    // exchange kernels on the other stream would wait for the compute units until a pass ends: there
    // passes A and B get a workgroup per tile and launches of their own.
    const bool loop = !split.windowed && plan->lineage == nullptr;
    // The adaptive schedule (buckets and groups cut to the keys' density) is offered where a group,
    // which lies inside one cell of the 16 bits above begin_bit, can be staged in 6 bytes per pair and
    // the 4096 cells of the leading 12 bits are made of such cells: 28 to 32 sorted bits from bit 0.
    AdaptArgs adapt{};
    adapt.offered = loop && compact_ok && !split.big && begin_bit == 0 && end_bit >= 28 && !plan->aids.hybrid_no_adaptive;
    adapt.force = adapt.offered && plan->aids.hybrid_force_adaptive;
    adapt.cell_shift = end_bit - kAdaptCellBits;
    adapt.low16_shift = begin_bit + 16u;
    // a bucket is cut into at most 256 groups of at most `cap` pairs, which greedy cuts fill to three
    // quarters and more: two thirds of 256 full groups is what a bucket may hold
    adapt.bucket_max = cap * 170u;
    adapt.msd_bits = msd_bits;
    HybridCleared* hc = plan->hyb_cleared;
    const HybridHeader* hdr = &hc->hdr;
    // the same read of the keys counts the two lowest digits of the plan's ordinary passes
    if (int rc = hybrid_launch_histogram(static_cast<const uint32_t*>(keys_a), n32, split.shift, msd_bits, split.windowed,
                                         split.base, begin_bit, end_bit, hist_grid, plan->hyb_packed, hc, plan->control->ghist, s))
        return rc;
    if (int rc = mark(plan, "msd_histogram", s)) return rc;
    // test aid: the device takes the plan's ordinary passes whatever the keys are
    const uint32_t force = plan->aids.hybrid_force_fallback ? kHybFallback : 0u;
    // ... and a decision against the schedule leaves the digit prefix sums of all four ordinary passes
    // behind (the upper two digits are sums over the exact table): no second read of the keys
    if (int rc = hybrid_launch_scan(plan->hyb_packed, hist_grid, plan->hyb_tables, hc, n32, plan->tile_items, force, bits_b,
                                    compact_ok, cap, adapt, plan->control->ghist, begin_bit, end_bit, split.shift,
                                    split.windowed ? split.base : 0u, s))
        return rc;
    if (int rc = mark(plan, "msd_scan", s)) return rc;

    const uint32_t cus = static_cast<uint32_t>(plan->ctx->num_cus);
    const uint32_t persistent_grid = num_tiles < cus ? num_tiles : cus;
    // the plan's ordinary passes, for inputs the device found unfit
    PassArgs ordinary[4];
    if (plan->num_passes != 4) return fail(MRS_ERR_UNSUPPORTED, "an MRS_ALGO_HYBRID plan has four ordinary passes");
    for (uint32_t pass = 0; pass < 4; ++pass) {
        const bool even = (pass % 2u) == 0u;
        PassArgs& f = ordinary[pass];
        f = PassArgs{};
        f.keys_in = even ? keys_a : keys_b;
        f.vals_in = even ? vals_a : vals_b;
        f.keys_out = even ? keys_b : keys_a;
        f.vals_out = even ? vals_b : vals_a;
        f.n = n32;
        f.num_tiles = num_tiles;
        f.shift = pass_shift(plan, pass);
        f.mask = pass_mask(plan, pass);
        f.digit_base = plan->control->ghist + static_cast<size_t>(pass) * plan->radix;
        f.status = plan->status + static_cast<size_t>(pass) * region;
        f.tile_counter = plan->control->tile_counter[pass];
        f.xcd_block = xcd_block;
        f.error_flag = plan->error_flag;
        f.index_payload = indexed && pass == 0;
        f.wide_status = plan->status_word_bytes == 8;  // plans of 2^30 pairs: a digit of theirs can count 2^30 keys
    }
    PassArgs a{};
    a.n = n32;
    a.num_tiles = num_tiles;
    a.wide_status = 0;
    a.xcd_block = xcd_block;
    a.error_flag = plan->error_flag;
    // pass A: a -> b by the top digit
    a.keys_in = keys_a;
    a.vals_in = vals_a;
    a.keys_out = keys_b;
    a.vals_out = vals_b;
    a.shift = split.windowed ? split.shift : split.shift + bits_b;
    a.mask = (1u << bits_a) - 1u;
    a.window_base = split.base;
    a.window_low = bits_b;
    a.digit_base = plan->hyb_tables->bucket_start;
    a.status = plan->status;
    a.tile_counter = plan->control->tile_counter[kMaxPasses - 2];
    a.index_payload = indexed;
    // The same launch holds pass A of the adaptive schedule: by the byte table over the 4096 cells of
    // the leading 12 bits.  The ordinary passes start one launch later, because that schedule can
    // still hand over to them when its buckets have been counted.
    PassArgs aa = a;
    aa.shift = adapt.cell_shift;
    aa.mask = kAdaptCells - 1u;
    aa.digit_map = plan->hyb_tables->map_a;
    aa.window_base = 0;
    aa.window_low = 0;
    aa.status = plan->status + hybrid_adaptive_status_offset(plan, num_tiles);
    if (int rc = hybrid_launch_pass(0, &a, nullptr, nullptr, loop, split.windowed, plan->threads, plan->kpt, hdr,
                                    loop ? persistent_grid : num_tiles, s, adapt.offered ? &aa : nullptr, plan->hyb_tables))
        return rc;
    if (int rc = mark(plan, "msd_pass_a", s)) return rc;
    if (adapt.offered) {
        if (int rc = hybrid_launch_adapt_count(static_cast<const uint32_t*>(keys_b), plan->hyb_tables, hc, n32, plan->tile_items, cap,
                                               adapt, s))
            return rc;
        if (int rc = mark(plan, "msd_adaptive_count", s)) return rc;
    }
    // pass B: b -> a by the next digit, bucket by bucket
    a.keys_in = keys_b;
    a.vals_in = vals_b;
    a.keys_out = keys_a;
    a.vals_out = vals_a;
    a.shift = split.shift;
    a.mask = (1u << bits_b) - 1u;
    a.window_low = 0;
    a.digit_base = nullptr;
    a.status = plan->status + region;
    a.tile_counter = plan->control->tile_counter[kMaxPasses - 1];
    a.index_payload = 0;
    a.xcd_block = xcd_block * 2;  // chains end with their bucket: larger blocks cost less here (0.834 -> 0.822 ms at 2^28)
    const SegmentTable buckets{plan->hyb_tables->first_tile, plan->hyb_tables->bucket_start, plan->hyb_tables->bucket_start + 1,
                               plan->hyb_tables->sub_start, a.mask + 1u, &hc->hdr.num_tiles_b};
    // ... and pass B of the adaptive schedule: over the same tables, by the groups of each bucket
    aa = a;
    aa.shift = adapt.cell_shift;
    aa.mask = kAdaptCells - 1u;
    if (int rc = hybrid_launch_pass(1, &a, &buckets, loop ? &ordinary[0] : nullptr, loop, split.windowed, plan->threads, plan->kpt, hdr,
                                    loop ? persistent_grid : static_cast<uint32_t>(hybrid_tiles_b(num_tiles)), s,
                                    adapt.offered ? &aa : nullptr, plan->hyb_tables))
        return rc;
    if (int rc = mark(plan, "msd_pass_b", s)) return rc;
    if (!loop) {
        if (int rc = hybrid_launch_pass(3, nullptr, nullptr, &ordinary[0], true, false, plan->threads, plan->kpt, hdr, persistent_grid, s))
            return rc;
        if (int rc = mark(plan, "fallback_pass", s)) return rc;
    }
    // the remaining bits: two digits, least significant first
    const uint32_t low_bits = split.shift - begin_bit;  // 9..16, or 2..16 under a window
    const uint32_t bits1 = (low_bits + 1) / 2, bits2 = low_bits - bits1;
    // ... of the sub-buckets that fit a workgroup: inside LDS, in place
    if (int rc = hybrid_launch_local_sort(static_cast<uint32_t*>(keys_a), static_cast<uint32_t*>(vals_a), hdr, plan->hyb_tables,
                                          begin_bit, bits1, bits2, 1u << msd_bits, split.shift, compact_ok,
                                          adapt.offered ? adapt.low16_shift : 0u, hybrid_local_sort_grid(plan->ctx->num_cus, split.big), s,
                                          split.big, /* stream_out: 8 n bytes against a last-level cache of 256 MiB */ n32 > (3u << 26)))
        return rc;
    if (int rc = mark(plan, "msd_local_sort", s)) return rc;
    // ... of the oversized ones (the device's list; usually empty, and then these launches are a few
    // hundred workgroups that return at once): their digit counts, then a -> b -> a inside each
    // sub-bucket's own range, one chain per sub-bucket.  Plain bit-field digits also under a window.
    // The same two launches are the last two ordinary passes of the other schedule.
    // Where the adaptive schedule is offered these passes take the 16 bits above begin_bit in either
    // schedule (a fixed sub-bucket's own bits end at or below them, and what lies above is constant
    // inside it).
    const uint32_t big1 = adapt.offered ? 8u : bits1, big2 = adapt.offered ? 8u : bits2;
    // (the launch is also ordinary pass 1 of the other schedule)
    if (int rc = hybrid_launch_big_histogram(static_cast<const uint32_t*>(keys_a), plan->hyb_tables, hc, plan->tile_items,
                                             begin_bit, big1, big2, &ordinary[1], plan->threads, plan->kpt,
                                             static_cast<int>(persistent_grid), s))
        return rc;
    if (int rc = mark(plan, "msd_oversized_histogram", s)) return rc;
    const size_t big_region = hybrid_tiles_big(num_tiles) * per_tile;
    for (uint32_t hh = 0; hh < 2; ++hh) {
        a.keys_in = hh == 0 ? keys_a : keys_b;
        a.vals_in = hh == 0 ? vals_a : vals_b;
        a.keys_out = hh == 0 ? keys_b : keys_a;
        a.vals_out = hh == 0 ? vals_b : vals_a;
        a.shift = hh == 0 ? begin_bit : begin_bit + big1;
        a.mask = (1u << (hh == 0 ? big1 : big2)) - 1u;
        a.window_base = 0;
        a.window_low = 0;
        a.digit_base = nullptr;
        a.status = plan->status + region + hybrid_tiles_b(num_tiles) * per_tile + hh * big_region;
        a.tile_counter = plan->control->tile_counter[kMaxPasses - 4 + hh];
        a.index_payload = 0;
        a.xcd_block = xcd_block;
        const SegmentTable oversized{plan->hyb_tables->big_first_tile, plan->hyb_tables->big_start, plan->hyb_tables->big_end,
                                     &hc->big_hist[0][hh][0], 512u, &hc->hdr.num_tiles_big};
        if (int rc = hybrid_launch_pass(2, &a, &oversized, &ordinary[2 + hh], true, false, plan->threads, plan->kpt, hdr, persistent_grid, s))
            return rc;
        if (int rc = mark(plan, "msd_oversized_pass", s)) return rc;
    }
    return MRS_OK;
}

// The launch schedule shared by mrs_sort_pairs, mrs_sort_pairs_primed, mrs_sort_pairs_indexed and
// mrs_sort_pairs_noalloc.
// `primed`: the plan's control block already holds the digit histograms of exactly these keys.
// `indexed`: the payload of input element i is i; vals_a is ping-pong space only and is never read
// before it is written.
// `window`: null, or two key values every key is known to lie between (inclusive).
static int sort_pairs_impl(mrs_plan* plan, uint64_t n, void* keys_a, void* vals_a, void* keys_b, void* vals_b,
                           hipStream_t s, int* result_in_b, bool primed, const char* who, bool indexed = false,
                           const uint64_t* window = nullptr)
{
    if (indexed && plan->desc.payload_bytes != 4)
        return fail(MRS_ERR_INVALID_ARGUMENT, "%s: the plan must sort 4-byte payloads (source indices)", who);
    // result side by pass parity, src/main.rs:819-823
    if (result_in_b) *result_in_b = (plan->num_passes % 2u) == 1u;
    plan->primed_keys = nullptr;  // whatever happens below, the histograms of an encoder call are spent
    if (n == 0) return MRS_OK;
    if (int rc = check_pair_buffers(plan, keys_a, vals_a, keys_b, vals_b, who)) return rc;
    const uint32_t n32 = static_cast<uint32_t>(n);
    const uint32_t num_tiles = static_cast<uint32_t>((n + plan->tile_items - 1) / plan->tile_items);

    plan->pass_plan_live = false;
    plan->hyb_last = false;
    plan->num_marks = 0;
    if (int rc = mark(plan, "start", s)) return rc;
    // Small inputs: the whole sort in one launch of one workgroup (aids.no_small_sort: A/B aid).
    if (plan->desc.algo == MRS_ALGO_ONESWEEP && n32 <= small_sort_capacity() && !plan->aids.no_small_sort) {
        const bool in_b = (plan->num_passes % 2u) == 1u;
        SmallSortArgs a{};
        a.keys_in = keys_a;
        a.vals_in = vals_a;
        a.keys_out = in_b ? keys_b : keys_a;
        // made-up indices read nothing, so they can be written where the result belongs
        a.vals_out = indexed && !in_b ? vals_a : vals_b;
        a.vals_final = in_b || indexed ? nullptr : vals_a;
        a.index_payload = indexed;
        a.n = n32;
        a.begin_bit = plan->desc.begin_bit;
        a.end_bit = plan->desc.end_bit;
        a.num_passes = plan->num_passes;
        a.payload_bytes = plan->desc.payload_bytes;
        {
            const ProfilerRange range("Indexing");
            if (int rc = launch_small_sort(plan->desc.key_words, plan->radix_bits, a, s)) return rc;
        }
        return mark(plan, "small_sort", s);
    }
    // (a primed call too: the three-move schedule takes its own 16-bit histogram, whatever the encoder
    // left for the four passes, and is still the faster of the two: 3.9 against 4.5 ms with the encoder
    // at 2^28 points)
    if (plan->hybrid) {
        HybridSplit split{};
        plan->hyb_last = hybrid_choose_split(plan->desc.begin_bit, plan->desc.end_bit, n, window, plan->aids, &split);
        if (plan->hyb_last) return sort_pairs_hybrid(plan, n32, keys_a, vals_a, keys_b, vals_b, s, indexed, split);
    }
    // 64-bit keys: passes whose digit is the same for every key are left out on the device
    // (see global_histogram_scan_kernel; aids.no_pass_skip: A/B aid)
    const bool skip_passes = plan->desc.algo == MRS_ALGO_ONESWEEP && plan->desc.key_words == 2 && !plan->aids.no_pass_skip;
    plan->pass_plan_live = skip_passes;
    const TicketedCall ticketed(plan->ctx, s, plan->lineage, plan->desc.algo == MRS_ALGO_ONESWEEP);
    const uint32_t xcd_block = ticketed.block();
    if (plan->desc.algo == MRS_ALGO_ONESWEEP) {
        const size_t region = static_cast<size_t>(num_tiles) * plan->radix * plan->status_word_bytes;
        const size_t zero = static_cast<size_t>(plan->status - plan->zero_begin) + align_up(region * plan->num_passes, 256);
        // a primed plan keeps its histograms: only the ticket counters and the status words are cleared
        const size_t keep = primed ? offsetof(OnesweepControl, tile_counter) : 0;
        if (int rc = clear_async(plan->zero_begin + keep, zero - keep, s)) return rc;
        if (int rc = mark(plan, "memset", s)) return rc;
        GlobalHistArgs h{};
        h.keys = keys_a;
        h.n = n32;
        h.begin_bit = plan->desc.begin_bit;
        h.end_bit = plan->desc.end_bit;
        h.num_passes = plan->num_passes;
        h.digit_map = nullptr;
        h.map_mask = 0;
        h.ghist = plan->control->ghist;
        if (!primed) {
            if (int rc = launch_global_hist(plan, h, s)) return rc;
            if (int rc = mark(plan, "global_histogram", s)) return rc;
        }
        if (int rc = launch_ghist_scan(plan, plan->num_passes, skip_passes ? n32 : 0, s, indexed ? 2u : 0u)) return rc;
        if (int rc = mark(plan, "histogram_scan", s)) return rc;
    }

    for (uint32_t pass = 0; pass < plan->num_passes; ++pass) {
        const bool even = (pass % 2u) == 0u;  // ping-pong, src/main.rs:519-576
        PassArgs a{};
        a.keys_in = even ? keys_a : keys_b;
        a.vals_in = even ? vals_a : vals_b;
        a.keys_out = even ? keys_b : keys_a;
        a.vals_out = even ? vals_b : vals_a;
        a.n = n32;
        a.num_tiles = num_tiles;
        a.shift = pass_shift(plan, pass);
        a.mask = pass_mask(plan, pass);
        a.digit_map = nullptr;
        a.index_payload = indexed && pass == 0;
        if (plan->desc.algo == MRS_ALGO_THREE_KERNEL) {
            // "Histogram calculation" / prefix sums / "Indexing", src/main.rs:595-664
            if (int rc = launch_block_hist(plan, n32, a.keys_in, num_tiles, pass, plan->hist, s)) return rc;
            if (int rc = mark(plan, "block_histogram", s)) return rc;
            if (int rc = launch_exclusive_scan(plan->hist, static_cast<uint64_t>(plan->radix) * num_tiles,
                                               plan->scan_partials, s))
                return rc;
            if (int rc = mark(plan, "scan", s)) return rc;
            a.scanned = plan->hist;
            if (int rc = launch_pass(plan, kModeScatter, a, s)) return rc;
            if (int rc = mark(plan, "scatter", s)) return rc;
        } else {
            a.digit_base = plan->control->ghist + static_cast<size_t>(pass) * plan->radix;
            a.status = plan->status + static_cast<size_t>(pass) * num_tiles * plan->radix * plan->status_word_bytes;
            a.wide_status = plan->status_word_bytes == 8;
            a.tile_counter = plan->control->tile_counter[pass];
            a.xcd_block = xcd_block;
            a.error_flag = plan->error_flag;
            a.pass_plan = skip_passes ? plan->control->pass_plan + pass : nullptr;
#ifdef MRS_STAMPS
            static const uint32_t stamp_pass = std::getenv("MRS_STAMP_PASS") ? std::atoi(std::getenv("MRS_STAMP_PASS")) : 1;
            a.stamps = pass == stamp_pass ? g_debug_stamps : nullptr;
#endif
            if (int rc = launch_pass(plan, kModeOnesweep, a, s)) return rc;
            if (int rc = mark(plan, "onesweep_pass", s)) return rc;
        }
    }
    return MRS_OK;
}

extern "C" int mrs_hybrid_split_for(uint64_t n, int32_t begin_bit, int32_t end_bit, const uint64_t* window,
                                    mrs_hybrid_split* out)
{
    clear_error();
    if (!out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_hybrid_split_for: null output");
    std::memset(out, 0, sizeof *out);
    const int width = end_bit - begin_bit;
    if (begin_bit < 0 || end_bit > 32 || width < 25 || n == 0 || n > (1ull << 30))
        return fail(MRS_ERR_UNSUPPORTED, "mrs_hybrid_split_for: MRS_ALGO_HYBRID sorts 25 to 32 bits of a 32-bit key, 0 < n <= 2^30");
    HybridSplit split{};
    out->offered = hybrid_choose_split(begin_bit, end_bit, n, window, TestAids{}, &split) ? 1 : 0;
    if (!out->offered && window) return MRS_OK;  // no split exists for this window
    out->leading_bits = split.msd_bits;
    out->shift = split.shift;
    out->base = split.base;
    out->windowed = split.windowed ? 1 : 0;
    out->local_bits = split.shift - begin_bit;
    out->local_capacity = split.big ? kHybLocalCapBig : hybrid_local_cap(split.shift);
    return MRS_OK;
}

extern "C" int mrs_sort_pairs(mrs_plan* plan, uint64_t n, void* keys_a, void* vals_a, void* keys_b, void* vals_b,
                              void* stream, int* result_in_b)
{
    clear_error();
    if (int rc = validate_call(plan, n, "mrs_sort_pairs")) return rc;
    return sort_pairs_impl(plan, n, keys_a, vals_a, keys_b, vals_b, static_cast<hipStream_t>(stream), result_in_b, false,
                           "mrs_sort_pairs");
}

extern "C" int mrs_sort_pairs_primed(mrs_plan* plan, uint64_t n, void* keys_a, void* vals_a, void* keys_b, void* vals_b,
                                     void* stream, int* result_in_b)
{
    clear_error();
    if (int rc = validate_call(plan, n, "mrs_sort_pairs_primed")) return rc;
    if (plan->primed_keys == nullptr || plan->primed_keys != keys_a || plan->primed_n != n) {
        plan->primed_keys = nullptr;
        return fail(MRS_ERR_INVALID_ARGUMENT,
                    "mrs_sort_pairs_primed: the plan does not hold the histograms of these keys: the call must follow "
                    "mrs_morton_encode30_for_sort on the same plan, with its keys_dev and n, and no other call on the "
                    "plan in between");
    }
    return sort_pairs_impl(plan, n, keys_a, vals_a, keys_b, vals_b, static_cast<hipStream_t>(stream), result_in_b, true,
                           "mrs_sort_pairs_primed");
}

extern "C" int mrs_sort_pairs_indexed(mrs_plan* plan, uint64_t n, void* keys_a, uint32_t* vals_a, void* keys_b,
                                      uint32_t* vals_b, void* stream, int* result_in_b)
{
    clear_error();
    if (int rc = validate_call(plan, n, "mrs_sort_pairs_indexed")) return rc;
    return sort_pairs_impl(plan, n, keys_a, vals_a, keys_b, vals_b, static_cast<hipStream_t>(stream), result_in_b, false,
                           "mrs_sort_pairs_indexed", true);
}

// No counterpart among the plan-based calls allocates less: this one owns no plan at all.  The
// temporary memory is the library's static scratch array on the context's device, bound to a
// plan that lives on this call's stack.
extern "C" int mrs_sort_pairs_noalloc(mrs_context* ctx, uint64_t n, uint32_t* keys_a, uint32_t* vals_a,
                                      uint32_t* keys_b, uint32_t* vals_b, int32_t begin_bit, int32_t end_bit,
                                      void* stream, int* result_in_b)
{
    clear_error();
    if (!ctx) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_sort_pairs_noalloc: null context");
    if (int rc = check_device(ctx, "mrs_sort_pairs_noalloc")) return rc;
    if (n > kNoallocMaxN)
        return fail(MRS_ERR_INVALID_ARGUMENT,
                    "mrs_sort_pairs_noalloc: n = %llu exceeds the %llu pairs the static scratch array is sized for; "
                    "use a plan and mrs_sort_pairs",
                    (unsigned long long)n, (unsigned long long)kNoallocMaxN);
    mrs_plan_desc d{};
    d.n = n ? n : 1;
    d.key_words = 1;
    d.payload_bytes = 4;
    d.digit_bits = 8;
    d.begin_bit = begin_bit;
    d.end_bit = end_bit;
    d.algo = MRS_ALGO_ONESWEEP;
    d.tile_items = 0;
    mrs_plan plan{};
    if (int rc = fill_plan(&plan, ctx, &d, "mrs_sort_pairs_noalloc")) return rc;
    if (plan.status_word_bytes != 4 || onesweep_temp_bytes(&plan) > kNoallocScratchBytes)
        return fail(MRS_ERR_UNSUPPORTED, "mrs_sort_pairs_noalloc: this tile shape needs %zu bytes of scratch, the array holds %zu",
                    onesweep_temp_bytes(&plan), kNoallocScratchBytes);
    if (!ctx->static_scratch) return fail(MRS_ERR_HIP, "mrs_sort_pairs_noalloc: the context has no scratch address");
    bind_onesweep_memory(&plan, ctx->static_scratch);
    plan.temp_bytes = kNoallocScratchBytes;
    plan.owns_temp = false;
    return sort_pairs_impl(&plan, n, keys_a, vals_a, keys_b, vals_b, static_cast<hipStream_t>(stream), result_in_b, false,
                           "mrs_sort_pairs_noalloc");
}

extern "C" int mrs_plan_set_timing(mrs_plan* plan, int enable)
{
    clear_error();
    if (!plan) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_plan_set_timing: null plan");
    if (int rc = check_device(plan->ctx, "mrs_plan_set_timing")) return rc;
    const int cap = static_cast<int>(sizeof(plan->marks) / sizeof(plan->marks[0]));
    if (enable && !plan->timing) {
        for (int i = 0; i < cap; ++i) MRS_HIP_TRY(hipEventCreate(&plan->marks[i]));
        plan->timing = true;
        plan->num_marks = 0;
    } else if (!enable && plan->timing) {
        for (int i = 0; i < cap; ++i) (void)hipEventDestroy(plan->marks[i]);
        plan->timing = false;
        plan->num_marks = 0;
    }
    return MRS_OK;
}

extern "C" int mrs_plan_get_timings(mrs_plan* plan, int cap, float* ms_out, char* labels_out, int* count_out)
{
    clear_error();
    if (!plan || !ms_out || !labels_out || !count_out)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_plan_get_timings: null argument");
    *count_out = 0;
    if (!plan->timing) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_plan_get_timings: timing is not enabled");
    if (plan->num_marks < 2) return MRS_OK;
    MRS_HIP_TRY(hipEventSynchronize(plan->marks[plan->num_marks - 1]));
    int count = 0;
    for (int i = 1; i < plan->num_marks && count < cap; ++i) {
        float ms = 0.f;
        MRS_HIP_TRY(hipEventElapsedTime(&ms, plan->marks[i - 1], plan->marks[i]));
        ms_out[count] = ms;
        std::snprintf(labels_out + 32 * count, 32, "%s", plan->mark_labels[i]);
        ++count;
    }
    *count_out = count;
    return MRS_OK;
}

extern "C" int mrs_morton_encode30_for_sort(mrs_plan* plan, const float* xyz_dev, uint64_t n, const float* mn,
                                            const float* mx, uint32_t* keys_dev, uint32_t* idx_dev, void* stream)
{
    clear_error();
    if (int rc = validate_call(plan, n, "mrs_morton_encode30_for_sort")) return rc;
    if (plan->desc.key_words != 1 || plan->desc.algo != MRS_ALGO_ONESWEEP)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_morton_encode30_for_sort: the plan must be an onesweep plan for 32-bit keys");
    if (n == 0) return MRS_OK;
    if (!xyz_dev || !mn || !mx || !keys_dev)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_morton_encode30_for_sort: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    plan->primed_keys = nullptr;
    if (plan->hybrid) {
        // A sort of this size on this plan takes the three-move schedule, whose one read of the keys
        // counts what BOTH of its pass sequences need (the exact table does not fit beside the
        // encoder's tiles in LDS, so the encoder cannot leave it): digit histograms counted here
        // would be thrown away, four LDS atomics per key for nothing.  The plain encoder runs, and
        // mrs_sort_pairs_primed is then mrs_sort_pairs.
        HybridSplit split{};
        if (hybrid_choose_split(plan->desc.begin_bit, plan->desc.end_bit, n, nullptr, plan->aids, &split)) {
            if (int rc = mrs_morton_encode30(plan->ctx, xyz_dev, n, mn, mx, keys_dev, idx_dev, stream)) return rc;
            plan->primed_keys = keys_dev;
            plan->primed_n = n;
            return MRS_OK;
        }
    }
    if (int rc = clear_async(plan->control->ghist, sizeof(plan->control->ghist), s)) return rc;
    {
        const ProfilerRange range("Morton Coding");  // src/main.rs:583
        if (int rc = launch_encode30_with_histogram(plan->ctx->num_cus, xyz_dev, n, mn, mx, keys_dev, idx_dev,
                                                    plan->control->ghist, plan->desc.begin_bit, plan->desc.end_bit,
                                                    plan->radix_bits, plan->num_passes, s))
            return rc;
    }
    plan->primed_keys = keys_dev;
    plan->primed_n = n;
    return MRS_OK;
}

extern "C" int mrs_sort_gather(mrs_plan* plan, uint64_t n, void* keys_a, void* keys_b, uint32_t* index_a,
                               uint32_t* index_b, const void* records_in, void* records_out, uint32_t record_bytes,
                               void* stream, int* result_in_b)
{
    clear_error();
    if (int rc = validate_call(plan, n, "mrs_sort_gather")) return rc;
    if (plan->desc.payload_bytes != 4)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_sort_gather: the plan must sort 4-byte payloads (source indices)");
    if (record_bytes == 0 || record_bytes % 4 != 0)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_sort_gather: record_bytes must be a positive multiple of 4");
    int in_b = (plan->num_passes % 2u) == 1u;
    if (result_in_b) *result_in_b = in_b;
    if (n == 0) return MRS_OK;
    if (!index_a || !index_b || !records_in || !records_out)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_sort_gather: null buffer");
    if (records_in == records_out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_sort_gather: records_out aliases records_in");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the first pass makes the source indices up: no iota is written, none is read
    if (int rc = sort_pairs_impl(plan, n, keys_a, index_a, keys_b, index_b, s, &in_b, false, "mrs_sort_gather", true))
        return rc;
    {
        const ProfilerRange range("Gather");
        if (int rc = launch_gather(records_in, in_b ? index_b : index_a, records_out, static_cast<uint32_t>(n), record_bytes,
                                   plan->ctx->num_cus, s))
            return rc;
    }
    return mark(plan, "gather", s);
}

extern "C" int mrs_histogram(mrs_plan* plan, uint64_t n, const void* keys_dev, uint32_t pass, uint32_t* hist_dev,
                             void* stream)
{
    clear_error();
    if (int rc = validate_call(plan, n, "mrs_histogram")) return rc;
    if (pass >= plan->num_passes) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_histogram: pass %u out of range", pass);
    if (n == 0) return MRS_OK;
    if (!keys_dev || !hist_dev) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_histogram: null buffer");
    const uint32_t num_tiles = static_cast<uint32_t>((n + plan->tile_items - 1) / plan->tile_items);
    return launch_block_hist(plan, static_cast<uint32_t>(n), keys_dev, num_tiles, pass, hist_dev,
                             static_cast<hipStream_t>(stream));
}

extern "C" int mrs_scan(mrs_plan* plan, uint64_t n, uint32_t* hist_dev, void* stream)
{
    clear_error();
    if (int rc = validate_call(plan, n, "mrs_scan")) return rc;
    if (plan->desc.algo != MRS_ALGO_THREE_KERNEL)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_scan: the plan must be MRS_ALGO_THREE_KERNEL");
    if (n == 0) return MRS_OK;
    if (!hist_dev || misaligned(hist_dev, 16))
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_scan: the histogram must be a 16-byte aligned device pointer");
    const uint64_t num_tiles = (n + plan->tile_items - 1) / plan->tile_items;
    return launch_exclusive_scan(hist_dev, plan->radix * num_tiles, plan->scan_partials,
                                 static_cast<hipStream_t>(stream));
}

extern "C" int mrs_scatter(mrs_plan* plan, uint64_t n, const void* keys_in, const void* vals_in, void* keys_out,
                           void* vals_out, const uint32_t* scanned_hist_dev, uint32_t pass, void* stream)
{
    clear_error();
    if (int rc = validate_call(plan, n, "mrs_scatter")) return rc;
    if (pass >= plan->num_passes) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_scatter: pass %u out of range", pass);
    if (n == 0) return MRS_OK;
    if (int rc = check_pair_buffers(plan, keys_in, vals_in, keys_out, vals_out, "mrs_scatter")) return rc;
    if (!scanned_hist_dev) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_scatter: null histogram");
    PassArgs a{};
    a.keys_in = keys_in;
    a.vals_in = vals_in;
    a.keys_out = keys_out;
    a.vals_out = vals_out;
    a.n = static_cast<uint32_t>(n);
    a.num_tiles = static_cast<uint32_t>((n + plan->tile_items - 1) / plan->tile_items);
    a.shift = pass_shift(plan, pass);
    a.mask = pass_mask(plan, pass);
    a.scanned = scanned_hist_dev;
    return launch_pass(plan, kModeScatter, a, static_cast<hipStream_t>(stream));
}

extern "C" int mrs_bucket_histogram(mrs_context* ctx, const void* keys_dev, uint64_t n, int32_t key_words,
                                    uint32_t shift, uint32_t bits, uint64_t* counts_dev, void* stream)
{
    clear_error();
    if (!ctx || !counts_dev) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_bucket_histogram: null argument");
    if (int rc = check_device(ctx, "mrs_bucket_histogram")) return rc;
    if (key_words != 1 && key_words != 2) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_bucket_histogram: key_words");
    if (bits == 0 || bits > 12 || shift + bits > 32u * key_words)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_bucket_histogram: need 1 <= bits <= 12 inside the key");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t bins = 1u << bits;
    if (int rc = clear_async(counts_dev, bins * sizeof(uint64_t), s)) return rc;
    if (n == 0) return MRS_OK;
    if (!keys_dev) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_bucket_histogram: null keys");
    const uint64_t vec = key_words == 1 ? 4 : 2;
    const uint64_t vectors = misaligned(keys_dev, 16) ? 0 : n / vec;
    uint64_t blocks = ((vectors ? vectors : n) + 1024 * 4 - 1) / (1024 * 4);
    const uint64_t cap = static_cast<uint64_t>(ctx->num_cus) * 2;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    unsigned long long* counts = reinterpret_cast<unsigned long long*>(counts_dev);
    if (key_words == 1)
        hipLaunchKernelGGL(bucket_histogram_kernel<uint32_t>, dim3(blocks), dim3(1024), bins * 4, s,
                           static_cast<const uint32_t*>(keys_dev), static_cast<unsigned long long>(n),
                           static_cast<unsigned long long>(vectors), shift, bins, counts);
    else
        hipLaunchKernelGGL(bucket_histogram_kernel<unsigned long long>, dim3(blocks), dim3(1024), bins * 4, s,
                           static_cast<const unsigned long long*>(keys_dev), static_cast<unsigned long long>(n),
                           static_cast<unsigned long long>(vectors), shift, bins, counts);
    MRS_CHECK_LAUNCH("bucket_histogram_kernel");
    return MRS_OK;
}

namespace mrs {

// The partition pass behind mrs_partition and the sharded sort.  `class_starts_dev` (u32[256],
// may be null): the exclusive scan of the digit histogram if the caller already knows it, which
// saves the histogram and scan launches; digits beyond the table (`splitters_dev`) need it.
int partition_internal(mrs_plan* plan, uint64_t n, const void* keys_in, const void* vals_in, void* keys_out,
                       void* vals_out, uint32_t shift, uint32_t bits, const uint8_t* map_dev, const void* splitters_dev,
                       uint32_t num_splitters, uint32_t exact_mask, const uint32_t* class_starts_dev, hipStream_t s)
{
    // the control block is about to be overwritten: histograms an encoder left are gone, and
    // the record of which passes the last sort left out no longer describes it
    plan->primed_keys = nullptr;
    plan->pass_plan_live = false;
    plan->hyb_last = false;
    const uint32_t n32 = static_cast<uint32_t>(n);
    const uint32_t num_tiles = static_cast<uint32_t>((n + plan->tile_items - 1) / plan->tile_items);
    const size_t zero = static_cast<size_t>(plan->status - plan->zero_begin) +
                        align_up(static_cast<size_t>(num_tiles) * plan->radix * plan->status_word_bytes, 256);
    if (int rc = clear_async(plan->zero_begin, zero, s)) return rc;
    if (class_starts_dev) {
        MRS_HIP_TRY(hipMemcpyAsync(plan->control->ghist, class_starts_dev, plan->radix * sizeof(uint32_t),
                                   hipMemcpyDeviceToDevice, s));
    } else {
        if (num_splitters) return fail(MRS_ERR_INVALID_ARGUMENT, "partition: splitters need precomputed class offsets");
        (void)splitters_dev;
        (void)exact_mask;
        GlobalHistArgs h{};
        h.keys = keys_in;
        h.n = n32;
        h.begin_bit = shift;
        h.end_bit = shift + bits;
        h.num_passes = 1;
        h.digit_map = map_dev;
        h.map_mask = (1u << bits) - 1u;
        h.ghist = plan->control->ghist;
        if (int rc = launch_global_hist(plan, h, s)) return rc;
    }
    return MRS_OK;
}

// second half of partition_internal: the pass itself (split so that mrs_partition can copy the
// part sizes out between the histogram and its scan)
int partition_pass(mrs_plan* plan, uint64_t n, const void* keys_in, const void* vals_in, void* keys_out, void* vals_out,
                   uint32_t shift, uint32_t bits, const uint8_t* map_dev, const void* splitters_dev, uint32_t num_splitters,
                   uint32_t exact_mask, bool scan_histogram, hipStream_t s)
{
    if (scan_histogram)
        if (int rc = launch_ghist_scan(plan, 1, 0, s)) return rc;
    PassArgs a{};
    a.keys_in = keys_in;
    a.vals_in = vals_in;
    a.keys_out = keys_out;
    a.vals_out = vals_out;
    a.n = static_cast<uint32_t>(n);
    a.num_tiles = static_cast<uint32_t>((n + plan->tile_items - 1) / plan->tile_items);
    a.shift = shift;
    a.mask = (1u << bits) - 1u;
    a.digit_map = map_dev;
    a.splitters = splitters_dev;
    a.num_splitters = num_splitters;
    a.exact_mask = exact_mask;
    a.digit_base = plan->control->ghist;
    a.status = plan->status;
    a.wide_status = plan->status_word_bytes == 8;
    a.tile_counter = plan->control->tile_counter[0];
    const TicketedCall ticketed(plan->ctx, s, plan->lineage);
    a.xcd_block = ticketed.block();
    a.error_flag = plan->error_flag;
    if (int rc = launch_pass(plan, kModeOnesweep, a, s)) return rc;
    return MRS_OK;
}

int sort_pairs_internal(mrs_plan* plan, uint64_t n, void* keys_a, void* vals_a, void* keys_b, void* vals_b, hipStream_t s,
                        int* result_in_b, const uint64_t* window)
{
    return sort_pairs_impl(plan, n, keys_a, vals_a, keys_b, vals_b, s, result_in_b, false, "mrs_dist_sort_pairs", false, window);
}

int record_mark(mrs_plan* plan, const char* label, hipStream_t s) { return mark(plan, label, s); }

}  // namespace mrs

extern "C" int mrs_partition(mrs_plan* plan, uint64_t n, const void* keys_in, const void* vals_in, void* keys_out,
                             void* vals_out, uint32_t shift, uint32_t bits, const uint8_t* map_dev, uint32_t num_parts,
                             uint32_t* part_counts_dev, void* stream)
{
    clear_error();
    if (int rc = validate_call(plan, n, "mrs_partition")) return rc;
    if (plan->desc.algo != MRS_ALGO_ONESWEEP)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_partition: the plan must be MRS_ALGO_ONESWEEP");
    if (num_parts == 0 || num_parts > 16 || num_parts > plan->radix)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_partition: num_parts must be in [1, 16]");
    if (bits == 0 || (1u << bits) > static_cast<uint32_t>(kMaxMapEntries) || shift + bits > 32u * plan->desc.key_words)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_partition: need 1 <= bits <= 12 inside the key");
    if (!map_dev) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_partition: null map");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {
        if (part_counts_dev)
            if (int rc = clear_async(part_counts_dev, 16 * sizeof(uint32_t), s)) return rc;
        return MRS_OK;
    }
    if (int rc = check_pair_buffers(plan, keys_in, vals_in, keys_out, vals_out, "mrs_partition")) return rc;
    if (int rc = partition_internal(plan, n, keys_in, vals_in, keys_out, vals_out, shift, bits, map_dev, nullptr, 0, 0,
                                    nullptr, s))
        return rc;
    if (part_counts_dev) {
        const size_t have = (plan->radix < 16 ? plan->radix : 16) * sizeof(uint32_t);
        if (int rc = clear_async(part_counts_dev, 16 * sizeof(uint32_t), s)) return rc;
        MRS_HIP_TRY(hipMemcpyAsync(part_counts_dev, plan->control->ghist, have, hipMemcpyDeviceToDevice, s));
    }
    return partition_pass(plan, n, keys_in, vals_in, keys_out, vals_out, shift, bits, map_dev, nullptr, 0, 0, true, s);
}
