// The sort sharded over the GPUs of one node (mrs_comm_*, mrs_dist_*, include/mrs.h).
//
// The reference records its passes for one device (src/main.rs:376, 594-665); this is the
// scale-out of that path: one rank per GPU, sharded by top-digit bucket.  Per call and rank:
// histogram of the top key digit -> RCCL all-reduce of the global digit histogram plus
// all-gather of the per-rank rows -> Splitter (host, mrs_split.hpp) -> one stable partition pass
// by destination -> grouped ncclSend/ncclRecv (all-to-all-v over xGMI) received in source-rank
// order -> local sort.  Because the partition and the local sort are stable and slices arrive
// in rank order, the result equals a stable sort of the concatenated input.
//
// RCCL is bound at run time (dlopen) so that the library loads on hosts and in processes that
// never shard; a second transport, a group of ranks inside one process (one host thread per
// rank, device copies behind a host barrier), runs the same orchestration on one GPU or several.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: every function is resolved with dlsym

#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "mrs_internal.h"
#include "mrs_split.hpp"

namespace mrs {

// ---- RCCL, bound at run time ----------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
};

static int load_rccl(const RcclApi** out)
{
    static std::mutex m;
    static RcclApi api;
    static bool tried = false;
    static std::string load_error = "no candidate name could be opened";  // dlerror() is only valid right after the failing dlopen
    std::lock_guard<std::mutex> lock(m);
    if (!tried) {
        tried = true;
        // A process that already holds an RCCL (PyTorch ships its own) keeps using that one: the
        // first name is the library's SONAME, which the loader resolves to the copy in memory.
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* name : names) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            if (const char* e = dlerror()) load_error = e;
        }
        if (api.handle) {
#define MRS_BIND(field, symbol) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, symbol))
            MRS_BIND(GetUniqueId, "ncclGetUniqueId");
            MRS_BIND(CommInitRank, "ncclCommInitRank");
            MRS_BIND(CommDestroy, "ncclCommDestroy");
            MRS_BIND(AllReduce, "ncclAllReduce");
            MRS_BIND(AllGather, "ncclAllGather");
            MRS_BIND(Send, "ncclSend");
            MRS_BIND(Recv, "ncclRecv");
            MRS_BIND(GroupStart, "ncclGroupStart");
            MRS_BIND(GroupEnd, "ncclGroupEnd");
            MRS_BIND(GetErrorString, "ncclGetErrorString");
            MRS_BIND(CommGetAsyncError, "ncclCommGetAsyncError");
#undef MRS_BIND
        }
    }
    if (!api.handle) return fail(MRS_ERR_UNSUPPORTED, "librccl.so could not be loaded: %s", load_error.c_str());
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.AllGather || !api.Send ||
        !api.Recv || !api.GroupStart || !api.GroupEnd || !api.GetErrorString)
        return fail(MRS_ERR_UNSUPPORTED, "librccl.so lacks a symbol this library binds");
    *out = &api;
    return MRS_OK;
}

#define MRS_NCCL_TRY(api, expr)                                                                         \
    do {                                                                                                \
        ncclResult_t _r = (expr);                                                                       \
        if (_r != ncclSuccess)                                                                          \
            return ::mrs::fail(MRS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(_r),  \
                               __FILE__, __LINE__);                                                     \
    } while (0)

// The two arrays of an exchange (keys, payload); element counts and offsets are shared.
struct ExchangeArrays {
    const void* send[2];
    void* recv[2];
    size_t elem_bytes[2];
    int count;  // 1 or 2
};

}  // namespace mrs

// What the orchestration needs from a transport.  All buffers are device memory; everything is
// enqueued on `s` (the host may block on other ranks' host threads, never on the device).
struct mrs_comm {
    int rank = 0, world = 1;
    virtual ~mrs_comm() {}
    // reduced[i] = sum over ranks of send[i] (may be null); gathered[r * count + i] = rank r's send[i]
    virtual int allreduce_allgather_u64(const uint64_t* send, uint64_t* reduced, uint64_t* gathered, size_t count,
                                        hipStream_t s) = 0;
    // all-to-all-v: send_cnt[d] elements from offset send_off[d] go to rank d, which stores the
    // slice of source r at recv_off[r] (recv_cnt[r] elements)
    virtual int alltoallv(const mrs::ExchangeArrays& x, const uint64_t* send_off, const uint64_t* send_cnt,
                          const uint64_t* recv_off, const uint64_t* recv_cnt, hipStream_t s) = 0;
    // an error the transport noticed asynchronously since the last call
    virtual int check_async() { return MRS_OK; }
    // this rank gives up on the call it is in (it returns an error to its caller): whatever the
    // transport can do so that its peers do not wait for it
    virtual void abandon() {}
};

namespace mrs {

// ---- transport 1: RCCL ------------------------------------------------------------------------
struct RcclComm final : mrs_comm {
    const RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    bool owned = false;
    ~RcclComm() override
    {
        if (owned && comm) (void)api->CommDestroy(comm);
    }
    int allreduce_allgather_u64(const uint64_t* send, uint64_t* reduced, uint64_t* gathered, size_t count,
                                hipStream_t s) override
    {
        MRS_NCCL_TRY(api, api->GroupStart());
        if (reduced) MRS_NCCL_TRY(api, api->AllReduce(send, reduced, count, ncclUint64, ncclSum, comm, s));
        MRS_NCCL_TRY(api, api->AllGather(send, gathered, count, ncclUint64, comm, s));
        MRS_NCCL_TRY(api, api->GroupEnd());
        return MRS_OK;
    }
    int alltoallv(const ExchangeArrays& x, const uint64_t* send_off, const uint64_t* send_cnt, const uint64_t* recv_off,
                  const uint64_t* recv_cnt, hipStream_t s) override
    {
        // RCCL has no all-to-all-v: one send and one receive per peer and array, fused in a group
        // so that they progress together; every peer pair has its own xGMI link.
        // A transfer goes out in operations of at most kMaxOpBytes: a send of 2 GiB and more (seen with
        // a rank's send to itself, RCCL 2.26) delivered part of the data only, without any error.  The
        // sender and the receiver of a transfer know the same count, so both cut it alike, and
        // operations between one pair of ranks match in the order they were posted.
        constexpr size_t kMaxOpBytes = size_t(1) << 30;
        MRS_NCCL_TRY(api, api->GroupStart());
        for (int a = 0; a < x.count; ++a) {
            const size_t eb = x.elem_bytes[a];
            const uint64_t step = kMaxOpBytes / eb;  // elements per operation
            for (int p = 0; p < world; ++p) {
                for (uint64_t at = 0; at < send_cnt[p]; at += step) {
                    const uint64_t cnt = send_cnt[p] - at < step ? send_cnt[p] - at : step;
                    MRS_NCCL_TRY(api, api->Send(static_cast<const char*>(x.send[a]) + (send_off[p] + at) * eb, cnt * eb,
                                                ncclUint8, p, comm, s));
                }
                for (uint64_t at = 0; at < recv_cnt[p]; at += step) {
                    const uint64_t cnt = recv_cnt[p] - at < step ? recv_cnt[p] - at : step;
                    MRS_NCCL_TRY(api, api->Recv(static_cast<char*>(x.recv[a]) + (recv_off[p] + at) * eb, cnt * eb,
                                                ncclUint8, p, comm, s));
                }
            }
        }
        MRS_NCCL_TRY(api, api->GroupEnd());
        return MRS_OK;
    }
    int check_async() override
    {
        if (!api->CommGetAsyncError) return MRS_OK;
        ncclResult_t state = ncclSuccess;
        MRS_NCCL_TRY(api, api->CommGetAsyncError(comm, &state));
        if (state != ncclSuccess && state != ncclInProgress)
            return fail(MRS_ERR_HIP, "RCCL reported an asynchronous error: %s", api->GetErrorString(state));
        return MRS_OK;
    }
};

// ---- transport 2: ranks inside one process ------------------------------------------------------
// One host thread per rank.  A collective is: every rank posts its buffers and an event that
// marks them ready on its stream, the host threads meet, every rank copies what is meant for
// it (on its own stream, behind the owners' events), marks itself done, the threads meet again
// and every rank makes its stream wait for the others' done marks before it may reuse a buffer.
struct LocalGroup {
    int world;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    int handles_alive;
    struct Post {
        const void* send[2] = {nullptr, nullptr};
        const uint64_t* send_off = nullptr;
        const uint64_t* send_cnt = nullptr;
        hipEvent_t ready = nullptr, done = nullptr;
    };
    std::vector<Post> post;
    // Set, for good, by a rank that returns an error from a collective or gives up before one:
    // the other ranks must not wait for it.  A dead group never blocks anybody again.
    bool failed = false;
    explicit LocalGroup(int w) : world(w), handles_alive(w), post(w) {}
    // false: a rank of the group has failed (the caller returns MRS_ERR_PEER_FAILED)
    bool barrier()
    {
        std::unique_lock<std::mutex> lock(m);
        if (failed) return false;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
            return true;
        }
        cv.wait(lock, [&] { return generation != gen || failed; });
        return !failed;
    }
    void fail_group()
    {
        std::lock_guard<std::mutex> lock(m);
        failed = true;
        cv.notify_all();
    }
};

static int peer_failed()
{
    return fail(MRS_ERR_PEER_FAILED, "local transport: another rank of the group failed; the group's collectives no longer run");
}

__global__ void sum_rows_kernel(const unsigned long long* __restrict__ rows, unsigned long long* __restrict__ out,
                                uint32_t count, uint32_t num_rows)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    unsigned long long sum = 0;
    for (uint32_t r = 0; r < num_rows; ++r) sum += rows[static_cast<size_t>(r) * count + i];
    out[i] = sum;
}

struct LocalComm final : mrs_comm {
    std::shared_ptr<LocalGroup> group;
    ~LocalComm() override
    {
        LocalGroup::Post& me = group->post[rank];
        if (me.ready) (void)hipEventDestroy(me.ready);
        if (me.done) (void)hipEventDestroy(me.done);
    }
    int prepare(hipStream_t s)
    {
        LocalGroup::Post& me = group->post[rank];
        if (!me.ready) MRS_HIP_TRY(hipEventCreateWithFlags(&me.ready, hipEventDisableTiming));
        if (!me.done) MRS_HIP_TRY(hipEventCreateWithFlags(&me.done, hipEventDisableTiming));
        MRS_HIP_TRY(hipEventRecord(me.ready, s));
        return MRS_OK;
    }
    int finish(hipStream_t s)
    {
        MRS_HIP_TRY(hipEventRecord(group->post[rank].done, s));
        if (!group->barrier()) return peer_failed();
        for (int p = 0; p < world; ++p)
            if (p != rank) MRS_HIP_TRY(hipStreamWaitEvent(s, group->post[p].done, 0));
        // the posts may be overwritten only after every rank has read them
        if (!group->barrier()) return peer_failed();
        return MRS_OK;
    }
    void abandon() override { group->fail_group(); }
    // A rank that fails inside a collective would leave the others at the next host barrier for
    // ever: it marks the group first, and they return MRS_ERR_PEER_FAILED.
    int allreduce_allgather_u64(const uint64_t* send, uint64_t* reduced, uint64_t* gathered, size_t count,
                                hipStream_t s) override
    {
        const int rc = allreduce_allgather_body(send, reduced, gathered, count, s);
        if (rc != MRS_OK) group->fail_group();
        return rc;
    }
    int alltoallv(const ExchangeArrays& x, const uint64_t* send_off, const uint64_t* send_cnt, const uint64_t* recv_off,
                  const uint64_t* recv_cnt, hipStream_t s) override
    {
        const int rc = alltoallv_body(x, send_off, send_cnt, recv_off, recv_cnt, s);
        if (rc != MRS_OK) group->fail_group();
        return rc;
    }
    int allreduce_allgather_body(const uint64_t* send, uint64_t* reduced, uint64_t* gathered, size_t count, hipStream_t s)
    {
        if (int rc = prepare(s)) return rc;
        group->post[rank].send[0] = send;
        if (!group->barrier()) return peer_failed();
        for (int p = 0; p < world; ++p) {
            if (p != rank) MRS_HIP_TRY(hipStreamWaitEvent(s, group->post[p].ready, 0));
            MRS_HIP_TRY(hipMemcpyAsync(gathered + static_cast<size_t>(p) * count, group->post[p].send[0],
                                       count * sizeof(uint64_t), hipMemcpyDefault, s));
        }
        if (reduced) {
            hipLaunchKernelGGL(sum_rows_kernel, dim3((count + 255) / 256), dim3(256), 0, s,
                               reinterpret_cast<const unsigned long long*>(gathered),
                               reinterpret_cast<unsigned long long*>(reduced), static_cast<uint32_t>(count),
                               static_cast<uint32_t>(world));
            MRS_CHECK_LAUNCH("sum_rows_kernel");
        }
        return finish(s);
    }
    int alltoallv_body(const ExchangeArrays& x, const uint64_t* send_off, const uint64_t* send_cnt, const uint64_t* recv_off,
                       const uint64_t* recv_cnt, hipStream_t s)
    {
        if (int rc = prepare(s)) return rc;
        LocalGroup::Post& me = group->post[rank];
        me.send[0] = x.send[0];
        me.send[1] = x.count > 1 ? x.send[1] : nullptr;
        me.send_off = send_off;
        me.send_cnt = send_cnt;
        if (!group->barrier()) return peer_failed();
        for (int p = 0; p < world; ++p) {
            const LocalGroup::Post& src = group->post[p];
            if (src.send_cnt[rank] != recv_cnt[p])
                return fail(MRS_ERR_INVALID_ARGUMENT, "local exchange: rank %d sends %llu elements to rank %d, which expects %llu",
                            p, (unsigned long long)src.send_cnt[rank], rank, (unsigned long long)recv_cnt[p]);
            if (recv_cnt[p] == 0) continue;
            if (p != rank) MRS_HIP_TRY(hipStreamWaitEvent(s, src.ready, 0));
            for (int a = 0; a < x.count; ++a) {
                const size_t eb = x.elem_bytes[a];
                MRS_HIP_TRY(hipMemcpyAsync(static_cast<char*>(x.recv[a]) + recv_off[p] * eb,
                                           static_cast<const char*>(src.send[a]) + src.send_off[rank] * eb,
                                           recv_cnt[p] * eb, hipMemcpyDefault, s));
            }
        }
        return finish(s);
    }
};

// ---- digit histogram of the keys under given prefixes (splitter levels after the first) -------
struct PrefixHistArgs {
    const void* keys;
    unsigned long long n;
    uint32_t shift, bits, num_prefixes;
    unsigned long long prefix[kSplitMaxPrefixes];
    unsigned long long* counts;  // [num_prefixes][1 << bits], zeroed
};

// counts[j][d] = number of keys with (key >> (shift + bits)) == prefix[j] and digit d.  Only skewed
// inputs get here, and those are the inputs whose keys pile up on one counter: a wave whose
// matching lanes all hold the same digit adds their number from one lane.
template <typename KeyT>
__global__ __launch_bounds__(1024) void prefix_histogram_kernel(const PrefixHistArgs a)
{
    __shared__ uint32_t s_bins[kSplitMaxPrefixes * 256];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t j = tid; j < a.num_prefixes * 256u; j += 1024u) s_bins[j] = 0;
    __syncthreads();
    const KeyT* __restrict__ keys = static_cast<const KeyT*>(a.keys);
    const uint32_t above = a.shift + a.bits;  // below the key width: the first level never comes here
    const uint32_t mask = (1u << a.bits) - 1u;
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * 1024ull;
    for (unsigned long long i0 = static_cast<unsigned long long>(blockIdx.x) * 1024ull + (tid - lane); i0 < a.n; i0 += stride) {
        const unsigned long long i = i0 + lane;
        const bool valid = i < a.n;
        const KeyT k = valid ? keys[i] : static_cast<KeyT>(0);
        const unsigned long long p = static_cast<unsigned long long>(k) >> above;
        const uint32_t d = static_cast<uint32_t>(k >> a.shift) & mask;
        for (uint32_t j = 0; j < a.num_prefixes; ++j) {
            const bool hit = valid && p == a.prefix[j];
            const unsigned long long m = __ballot(hit);
            if (m == 0) continue;
            const int first = __ffsll(static_cast<long long>(m)) - 1;
            const uint32_t d0 = __shfl(d, first);
            if (__ballot(hit && d != d0) == 0) {
                if (static_cast<int>(lane) == first) atomicAdd(&s_bins[j * 256u + d0], static_cast<uint32_t>(__popcll(m)));
            } else if (hit) {
                atomicAdd(&s_bins[j * 256u + d], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t idx = tid; idx < a.num_prefixes * 256u; idx += 1024u) {
        const uint32_t c = s_bins[idx];
        const uint32_t j = idx >> 8, d = idx & 255u;
        if (c && d <= mask) atomicAdd(&a.counts[(static_cast<size_t>(j) << a.bits) + d], static_cast<unsigned long long>(c));
    }
}

static int launch_prefix_histogram(const mrs_context* ctx, int key_words, const PrefixHistArgs& a, hipStream_t s)
{
    unsigned long long blocks = (a.n + 4095) / 4096;
    const unsigned long long cap = static_cast<unsigned long long>(ctx->num_cus) * 2;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    if (key_words == 1)
        hipLaunchKernelGGL(prefix_histogram_kernel<uint32_t>, dim3(blocks), dim3(1024), 0, s, a);
    else
        hipLaunchKernelGGL(prefix_histogram_kernel<unsigned long long>, dim3(blocks), dim3(1024), 0, s, a);
    MRS_CHECK_LAUNCH("prefix_histogram_kernel");
    return MRS_OK;
}

constexpr size_t kHistWords = 4096;  // u64 counters of one rank's histogram row (12 key bits, or 15 prefixes x 256)
static_assert(kSplitMaxPrefixes * 256 <= kHistWords && (1 << kSplitTopBits) <= kHistWords, "histogram rows share one size");

// what the host hands to the partition pass, in one upload
struct PartitionTables {
    uint8_t map[kMaxMapEntries];
    unsigned long long splitters[kMaxSplitters + 1];
    uint32_t class_starts[256];
};

}  // namespace mrs

struct mrs_dist {
    mrs_context* ctx = nullptr;
    mrs_comm* comm = nullptr;
    mrs_dist_desc desc{};
    uint64_t tolerance = 0, recv_capacity = 0;
    mrs_plan* part_plan = nullptr;
    // Every rank receives its share in `chunks` pieces of consecutive key ranges, exchanged one
    // after the other: the local sort of a piece (on `side`) runs beside the exchange of the next.
    int chunks = 1;
    uint64_t chunk_capacity = 0;
    mrs_plan* sort_plan[2] = {nullptr, nullptr};
    hipStream_t side = nullptr;
    hipEvent_t arrived[2] = {nullptr, nullptr};
    hipEvent_t sorted_all = nullptr;
    size_t key_bytes = 4, val_bytes = 0;
    // device memory owned by the handle
    char* small = nullptr;  // [hist_local][hist_global][hist_all: world rows][PartitionTables]
    uint64_t *hist_local = nullptr, *hist_global = nullptr, *hist_all = nullptr;
    mrs::PartitionTables* tables_dev = nullptr;
    void *part_k = nullptr, *part_v = nullptr, *recv_k = nullptr, *recv_v = nullptr, *tmp_k = nullptr, *tmp_v = nullptr;
    size_t device_bytes = 0;
    // pinned host mirrors
    uint64_t* h_hist = nullptr;  // [hist_global][hist_all]
    mrs::PartitionTables* h_tables = nullptr;
    // last call
    bool timing = false;
    hipEvent_t marks[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool marks_valid = false;
    uint32_t last_levels = 0, last_exact_cuts = 0;
    uint64_t last_n_out = 0;
    uint64_t last_piece_pairs = 0;  // pairs of the last piece (the one whose plan records timings)
};

using namespace mrs;

extern "C" int mrs_comm_get_unique_id(void* id_out)
{
    clear_error();
    if (!id_out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_comm_get_unique_id: null output");
    const RcclApi* api = nullptr;
    if (int rc = load_rccl(&api)) return rc;
    static_assert(sizeof(ncclUniqueId) == MRS_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    MRS_NCCL_TRY(api, api->GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return MRS_OK;
}

extern "C" int mrs_comm_create_rccl(const void* unique_id, int32_t rank, int32_t world, mrs_comm** out)
{
    clear_error();
    if (!unique_id || !out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_comm_create_rccl: null argument");
    *out = nullptr;
    if (world < 1 || world > kSplitMaxRanks || rank < 0 || rank >= world)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_comm_create_rccl: need 0 <= rank < world <= %d", kSplitMaxRanks);
    const RcclApi* api = nullptr;
    if (int rc = load_rccl(&api)) return rc;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm = nullptr;
    MRS_NCCL_TRY(api, api->CommInitRank(&comm, world, id, rank));
    RcclComm* c = new RcclComm;
    c->api = api;
    c->comm = comm;
    c->owned = true;
    c->rank = rank;
    c->world = world;
    *out = c;
    return MRS_OK;
}

extern "C" int mrs_comm_wrap_rccl(void* nccl_comm, int32_t rank, int32_t world, mrs_comm** out)
{
    clear_error();
    if (!nccl_comm || !out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_comm_wrap_rccl: null argument");
    *out = nullptr;
    if (world < 1 || world > kSplitMaxRanks || rank < 0 || rank >= world)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_comm_wrap_rccl: need 0 <= rank < world <= %d", kSplitMaxRanks);
    const RcclApi* api = nullptr;
    if (int rc = load_rccl(&api)) return rc;
    RcclComm* c = new RcclComm;
    c->api = api;
    c->comm = static_cast<ncclComm_t>(nccl_comm);
    c->owned = false;
    c->rank = rank;
    c->world = world;
    *out = c;
    return MRS_OK;
}

extern "C" int mrs_comm_create_local(int32_t world, mrs_comm** out)
{
    clear_error();
    if (!out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_comm_create_local: null output");
    if (world < 1 || world > kSplitMaxRanks)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_comm_create_local: world must be in [1, %d]", kSplitMaxRanks);
    auto group = std::make_shared<LocalGroup>(world);
    for (int r = 0; r < world; ++r) {
        LocalComm* c = new LocalComm;
        c->group = group;
        c->rank = r;
        c->world = world;
        out[r] = c;
    }
    return MRS_OK;
}

extern "C" int mrs_comm_destroy(mrs_comm* comm)
{
    clear_error();
    delete comm;
    return MRS_OK;
}

// Handles of several in-process ranks are destroyed by their host threads at the same time.  A
// process was once seen to abort with two threads inside this function and none anywhere else in
// the library.  Nothing of the library's own can dangle here: the per-device activity record
// (mrs_api.hip) owns its event for the life of the process and keeps the stream of the last
// die-aware call only as a value to compare, never to hand to the runtime; so what died were
// runtime calls of two threads running side by side -- hipStreamDestroy of one handle's side
// stream against the hipFree / hipHostFree of another (both walk the device's stream list), under
// the HIP runtime a PyTorch process carries.  Until a run of the -DMRS_DESTROY_TRACE build names
// the pair, the teardown of a handle is made by one thread at a time; creation and the sort calls
// are not serialised.
#ifndef MRS_DESTROY_TRACE
static std::mutex g_destroy_mutex;
#endif

static void free_dist(mrs_dist* d)
{
    if (!d) return;
#ifndef MRS_DESTROY_TRACE
    std::lock_guard<std::mutex> lock(g_destroy_mutex);
#endif
    // not for the abort above: a later stream may be given the same handle value, and must not
    // pass for the stream the last die-aware call ran on
    if (d->side && d->ctx) forget_stream_activity(d->ctx->device, d->side);
    if (d->part_plan) MRS_TEARDOWN(mrs_plan_destroy(d->part_plan));
    for (mrs_plan* p : d->sort_plan)
        if (p) MRS_TEARDOWN(mrs_plan_destroy(p));
    if (d->side) MRS_TEARDOWN(hipStreamDestroy(d->side));
    for (hipEvent_t e : d->arrived)
        if (e) MRS_TEARDOWN(hipEventDestroy(e));
    if (d->sorted_all) MRS_TEARDOWN(hipEventDestroy(d->sorted_all));
    void* dev[] = {d->small, d->part_k, d->part_v, d->recv_k, d->recv_v, d->tmp_k, d->tmp_v};
    for (void* p : dev)
        if (p) MRS_TEARDOWN(hipFree(p));
    if (d->h_hist) MRS_TEARDOWN(hipHostFree(d->h_hist));
    if (d->h_tables) MRS_TEARDOWN(hipHostFree(d->h_tables));
    for (hipEvent_t e : d->marks)
        if (e) MRS_TEARDOWN(hipEventDestroy(e));
    delete d;
}

extern "C" int mrs_dist_create(mrs_context* ctx, mrs_comm* comm, const mrs_dist_desc* desc, mrs_dist** out)
{
    clear_error();
    if (!ctx || !comm || !desc || !out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_create: null argument");
    *out = nullptr;
    int cur = -1;
    MRS_HIP_TRY(hipGetDevice(&cur));
    if (cur != ctx->device)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_create: context is for device %d but the current device is %d",
                    ctx->device, cur);
    if (desc->key_words != 1 && desc->key_words != 2) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_create: key_words must be 1 or 2");
    if (desc->payload_bytes != 0 && desc->payload_bytes != 4)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_create: payload_bytes must be 0 or 4");
    if (desc->end_bit < 1 || desc->end_bit > 32 * desc->key_words)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_create: end_bit must be in [1, %d]", 32 * desc->key_words);
    if (desc->n_local_max == 0 || desc->n_local_max >= (1ull << 31))
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_create: n_local_max must satisfy 0 < n_local_max < 2^31");
    mrs_dist* d = new mrs_dist;
    d->ctx = ctx;
    d->comm = comm;
    d->desc = *desc;
    d->tolerance = desc->tolerance ? desc->tolerance : desc->n_local_max / 256;
    d->recv_capacity = desc->n_local_max + 2 * d->tolerance + 2;
    if (desc->exchange_chunks < 0 || desc->exchange_chunks > 2) {
        delete d;
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_create: exchange_chunks must be 0 (chosen here), 1 or 2");
    }
    // two pieces wherever that leaves the partition within its 31 key classes and there is a peer to overlap with
    d->chunks = desc->exchange_chunks ? desc->exchange_chunks : ((comm->world >= 2 && comm->world <= kSplitMaxRanks / 2) ? 2 : 1);
    if (d->chunks * comm->world > kSplitMaxRanks) {
        delete d;
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_create: exchange_chunks x world must not exceed %d", kSplitMaxRanks);
    }
    d->chunk_capacity = (desc->n_local_max + d->chunks - 1) / d->chunks + 2 * d->tolerance + 2;
    d->key_bytes = 4u * desc->key_words;
    d->val_bytes = desc->payload_bytes;
    if (d->recv_capacity >= 0xFFFFFFFFull) {
        free_dist(d);
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_create: n_local_max + 2 * tolerance must stay below 2^32");
    }
    auto bail = [&](int rc) {
        free_dist(d);
        return rc;
    };
    // the partition runs the chained-scan pass with 8-bit digits (up to 31 key classes)
    mrs_plan_desc pd{};
    pd.n = desc->n_local_max;
    pd.key_words = desc->key_words;
    pd.payload_bytes = desc->payload_bytes;
    pd.digit_bits = 8;
    pd.begin_bit = 0;
    pd.end_bit = desc->end_bit;
    pd.algo = MRS_ALGO_ONESWEEP;
    pd.tile_items = 0;
    if (int rc = mrs_plan_create(ctx, &pd, &d->part_plan)) return bail(rc);
    pd.n = d->chunks == 1 ? d->recv_capacity : d->chunk_capacity;
    // (u32 key, u32 payload) pieces of 2^26 pairs and more take the three-move schedule, split over
    // the piece's own window of the key space (the cut keys on either side of it)
    if (desc->key_words == 1 && desc->payload_bytes == 4 && desc->end_bit >= 25 && desc->end_bit <= 32 && pd.n < (1ull << 30))
        pd.algo = MRS_ALGO_HYBRID;
    for (int c = 0; c < d->chunks; ++c)
        if (int rc = mrs_plan_create(ctx, &pd, &d->sort_plan[c])) return bail(rc);
    d->part_plan->lineage = d;
    for (int c = 0; c < d->chunks; ++c) d->sort_plan[c]->lineage = d;
    if (d->chunks > 1) {
        // Default priority on purpose: a low-priority stream whose chained-scan pass holds compute
        // units while it waits for tiles that the dispatcher keeps behind another stream's
        // workgroups stalled two ranks sharing one GPU for minutes.
        if (hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&d->arrived[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&d->arrived[1], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&d->sorted_all, hipEventDisableTiming) != hipSuccess)
            return bail(fail(MRS_ERR_HIP, "mrs_dist_create: stream or event creation failed"));
    }
    const size_t rows = static_cast<size_t>(comm->world) + 2;
    const size_t small_bytes = rows * kHistWords * sizeof(uint64_t) + sizeof(PartitionTables);
    auto dev_alloc = [&](void** p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes ? bytes : 16);
        if (e != hipSuccess) return fail(MRS_ERR_OUT_OF_MEMORY, "mrs_dist_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        d->device_bytes += bytes;
        return MRS_OK;
    };
    if (int rc = dev_alloc(reinterpret_cast<void**>(&d->small), small_bytes)) return bail(rc);
    d->hist_local = reinterpret_cast<uint64_t*>(d->small);
    d->hist_global = d->hist_local + kHistWords;
    d->hist_all = d->hist_global + kHistWords;
    d->tables_dev = reinterpret_cast<PartitionTables*>(d->hist_all + static_cast<size_t>(comm->world) * kHistWords);
    if (int rc = dev_alloc(&d->part_k, desc->n_local_max * d->key_bytes)) return bail(rc);
    if (int rc = dev_alloc(&d->recv_k, d->recv_capacity * d->key_bytes)) return bail(rc);
    if (int rc = dev_alloc(&d->tmp_k, d->recv_capacity * d->key_bytes)) return bail(rc);
    if (d->val_bytes) {
        if (int rc = dev_alloc(&d->part_v, desc->n_local_max * d->val_bytes)) return bail(rc);
        if (int rc = dev_alloc(&d->recv_v, d->recv_capacity * d->val_bytes)) return bail(rc);
        if (int rc = dev_alloc(&d->tmp_v, d->recv_capacity * d->val_bytes)) return bail(rc);
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&d->h_hist), (static_cast<size_t>(comm->world) + 1) * kHistWords * sizeof(uint64_t),
                      hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&d->h_tables), sizeof(PartitionTables), hipHostMallocDefault) != hipSuccess)
        return bail(fail(MRS_ERR_OUT_OF_MEMORY, "mrs_dist_create: hipHostMalloc failed"));
    mrs_plan_info pi{};
    (void)mrs_plan_get_info(d->part_plan, &pi);
    d->device_bytes += pi.temp_bytes;
    for (int c = 0; c < d->chunks; ++c) {
        (void)mrs_plan_get_info(d->sort_plan[c], &pi);
        d->device_bytes += pi.temp_bytes;
    }
    *out = d;
    return MRS_OK;
}

extern "C" int mrs_dist_destroy(mrs_dist* dist)
{
    clear_error();
    free_dist(dist);
    return MRS_OK;
}

extern "C" int mrs_dist_set_timing(mrs_dist* dist, int enable)
{
    clear_error();
    if (!dist) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_set_timing: null handle");
    if (enable && !dist->marks[0])
        for (hipEvent_t& e : dist->marks) MRS_HIP_TRY(hipEventCreate(&e));
    dist->timing = enable != 0;
    dist->marks_valid = false;
    return mrs_plan_set_timing(dist->sort_plan[dist->chunks - 1], enable);
}

extern "C" int mrs_dist_get_info(mrs_dist* dist, mrs_dist_info* info)
{
    clear_error();
    if (!dist || !info) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_get_info: null argument");
    std::memset(info, 0, sizeof *info);
    info->recv_capacity = dist->recv_capacity;
    info->device_bytes = dist->device_bytes;
    info->last_levels = dist->last_levels;
    info->last_exact_cuts = dist->last_exact_cuts;
    info->last_n_out = dist->last_n_out;
    info->last_sort_piece_pairs = dist->last_piece_pairs;
    {
        // what the device chose is in device memory: wait for everything this rank enqueued
        MRS_HIP_TRY(hipDeviceSynchronize());
        for (int c = 0; c < dist->chunks && c < 2; ++c) {
            int32_t schedule = 0;
            if (int rc = mrs_plan_get_schedule(dist->sort_plan[c], nullptr, &schedule, nullptr)) return rc;
            info->last_sort_schedule[c] = static_cast<uint32_t>(schedule);
        }
    }
    if (dist->timing && dist->marks_valid) {
        MRS_HIP_TRY(hipEventSynchronize(dist->marks[4]));
        for (int i = 0; i < 4; ++i) MRS_HIP_TRY(hipEventElapsedTime(&info->last_ms[i], dist->marks[i], dist->marks[i + 1]));
        float ms[kMaxPasses * 4 + 8];
        char labels[(kMaxPasses * 4 + 8) * 32];
        int count = 0;
        if (int rc = mrs_plan_get_timings(dist->sort_plan[dist->chunks - 1], kMaxPasses * 4 + 8, ms, labels, &count)) return rc;
        float sum = 0.f;
        uint32_t passes = 0;
        // the digit passes that ran: a launch of the schedule the device did not choose (or without
        // oversized sub-buckets to move) returns at once, in a few microseconds
        for (int i = 0; i < count; ++i) {
            const char* lab = labels + 32 * i;
            const bool pass_label = std::strcmp(lab, "onesweep_pass") == 0 || std::strcmp(lab, "msd_pass_a") == 0 ||
                                    std::strcmp(lab, "msd_pass_b") == 0 || std::strcmp(lab, "fallback_pass") == 0 ||
                                    std::strcmp(lab, "msd_oversized_pass") == 0;
            if (pass_label && ms[i] > 0.03f) {
                sum += ms[i];
                ++passes;
            }
        }
        info->last_sort_passes = passes;
        info->last_sort_pass_ms = passes ? sum / passes : 0.f;
    }
    return MRS_OK;
}

extern "C" int mrs_dist_check(mrs_dist* dist, void* stream)
{
    clear_error();
    if (!dist) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_check: null handle");
    if (int rc = mrs_plan_check(dist->part_plan, stream)) return rc;
    for (int c = 0; c < dist->chunks; ++c)
        if (int rc = mrs_plan_check(dist->sort_plan[c], stream)) return rc;
    return dist->comm->check_async();
}

static int dist_sort_pairs_impl(mrs_dist* dist, uint64_t n_local, const void* keys_in, const void* vals_in,
                                void** keys_out, void** vals_out, uint64_t* n_out, void* stream);

extern "C" int mrs_dist_sort_pairs(mrs_dist* dist, uint64_t n_local, const void* keys_in, const void* vals_in,
                                   void** keys_out, void** vals_out, uint64_t* n_out, void* stream)
{
    clear_error();
    if (!dist || !keys_out || !n_out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_sort_pairs: null argument");
    const int rc = dist_sort_pairs_impl(dist, n_local, keys_in, vals_in, keys_out, vals_out, n_out, stream);
    // a rank that leaves the call with an error tells its transport, so that peers that can be
    // told (ranks of one process) do not wait for it in a collective
    if (rc != MRS_OK) dist->comm->abandon();
    return rc;
}

static int dist_sort_pairs_impl(mrs_dist* dist, uint64_t n_local, const void* keys_in, const void* vals_in,
                                void** keys_out, void** vals_out, uint64_t* n_out, void* stream)
{
    mrs_dist* d = dist;
    int cur = -1;
    MRS_HIP_TRY(hipGetDevice(&cur));
    if (cur != d->ctx->device)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_sort_pairs: context is for device %d but the current device is %d",
                    d->ctx->device, cur);
    // Argument errors must not leave the other ranks waiting in a collective: everything that can
    // be wrong on one rank alone is checked here, before the first one.
    if (n_local > d->desc.n_local_max)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_sort_pairs: n_local = %llu exceeds n_local_max = %llu",
                    (unsigned long long)n_local, (unsigned long long)d->desc.n_local_max);
    if (n_local && (!keys_in || (d->val_bytes && !vals_in)))
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_sort_pairs: null input buffer");
    if (d->val_bytes && !vals_out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_sort_pairs: null vals_out");
    hipStream_t s = static_cast<hipStream_t>(stream);
    mrs_comm* comm = d->comm;
    const int P = comm->world, me = comm->rank;
    const int kw = d->desc.key_words;
    d->marks_valid = false;
    if (d->timing) MRS_HIP_TRY(hipEventRecord(d->marks[0], s));

    // ---- 1-3. histograms, collectives, cuts -------------------------------------------------
    const int K = d->chunks;  // destination slices per rank
    Splitter sp(P, P * K, d->desc.end_bit, d->tolerance);
    uint32_t shift = 0, bits = 0;
    std::vector<uint64_t> prefixes;
    uint64_t* h_global = d->h_hist;
    uint64_t* h_all = d->h_hist + kHistWords;
    while (sp.next(&shift, &bits, &prefixes)) {
        const size_t count = prefixes.size() << bits;
        const bool first = sp.levels_fed() == 0;
        if (first) {
            if (int rc = mrs_bucket_histogram(d->ctx, keys_in, n_local, kw, shift, bits, d->hist_local, s)) return rc;
        } else {
            if (int rc = clear_async(d->hist_local, count * sizeof(uint64_t), s)) return rc;
            PrefixHistArgs a{};
            a.keys = keys_in;
            a.n = n_local;
            a.shift = shift;
            a.bits = bits;
            a.num_prefixes = static_cast<uint32_t>(prefixes.size());
            for (size_t j = 0; j < prefixes.size(); ++j) a.prefix[j] = prefixes[j];
            a.counts = reinterpret_cast<unsigned long long*>(d->hist_local);
            if (n_local)
                if (int rc = launch_prefix_histogram(d->ctx, kw, a, s)) return rc;
        }
        // the global digit histogram (all-reduce) and every rank's row (all-gather), one group
        if (int rc = comm->allreduce_allgather_u64(d->hist_local, first ? d->hist_global : nullptr, d->hist_all, count, s))
            return rc;
        if (first) MRS_HIP_TRY(hipMemcpyAsync(h_global, d->hist_global, count * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        MRS_HIP_TRY(hipMemcpyAsync(h_all, d->hist_all, static_cast<size_t>(P) * count * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        MRS_HIP_TRY(hipStreamSynchronize(s));  // the host round trip of this level
        if (first) {
            // the two collectives must tell the same story
            for (size_t i = 0; i < count; ++i) {
                uint64_t sum = 0;
                for (int r = 0; r < P; ++r) sum += h_all[static_cast<size_t>(r) * count + i];
                if (sum != h_global[i])
                    return fail(MRS_ERR_HIP, "mrs_dist_sort_pairs: the all-reduced histogram differs from the sum of the gathered rows at bucket %zu",
                                i);
            }
            if (sp.levels_fed() == 0 && n_local != 0) {
                uint64_t mine = 0;
                for (size_t i = 0; i < count; ++i) mine += h_all[static_cast<size_t>(me) * count + i];
                if (mine != n_local) return fail(MRS_ERR_HIP, "mrs_dist_sort_pairs: the local histogram counts %llu keys of %llu",
                                                 (unsigned long long)mine, (unsigned long long)n_local);
            }
        }
        if (int rc = sp.feed(h_all))
            return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_sort_pairs: inconsistent histograms (code %d): are key bits above end_bit set?", rc);
    }
    d->last_levels = static_cast<uint32_t>(sp.levels_fed());
    uint32_t exact_mask = 0;
    for (int r = 0; r < sp.num_cuts(); ++r)
        if (sp.exact(r)) exact_mask |= 1u << r;
    d->last_exact_cuts = static_cast<uint32_t>(__builtin_popcount(exact_mask));
    // every rank can see every rank's load: all fail together, before any exchange
    for (int r = 0; r < P; ++r) {
        uint64_t total = 0;
        for (int c = 0; c < K; ++c) {
            const uint64_t piece = sp.recv_total(r * K + c);
            total += piece;
            if (K > 1 && piece > d->chunk_capacity)
                return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_sort_pairs: piece %d of rank %d would hold %llu pairs, above the capacity %llu "
                                                      "(n_local_max is smaller than some rank's share of the total)",
                            c, r, (unsigned long long)piece, (unsigned long long)d->chunk_capacity);
        }
        if (total > d->recv_capacity)
            return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_dist_sort_pairs: rank %d would receive %llu pairs, above the capacity %llu "
                                                  "(n_local_max is smaller than some rank's share of the total)",
                        r, (unsigned long long)total, (unsigned long long)d->recv_capacity);
    }
    uint64_t n_recv = 0;
    for (int c = 0; c < K; ++c) n_recv += sp.recv_total(me * K + c);
    if (d->timing) MRS_HIP_TRY(hipEventRecord(d->marks[1], s));

    // ---- 4. stable partition by destination --------------------------------------------------
    const void* send_k = keys_in;
    const void* send_v = vals_in;
    if (P * K > 1 && n_local) {
        PartitionTables* t = d->h_tables;
        sp.bucket_table(t->map);
        for (int r = 0; r < sp.num_cuts(); ++r) {
            if (kw == 1) reinterpret_cast<uint32_t*>(t->splitters)[r] = static_cast<uint32_t>(sp.key(r));
            else t->splitters[r] = sp.key(r);
        }
        sp.class_starts(me, t->class_starts);
        // the slower pass that compares keys with the cut values is only needed if some bucket holds a cut
        bool marked = false;
        for (uint32_t b = 0; b < (1u << sp.top_bits()); ++b) marked = marked || t->map[b] == kMapBySplitters;
        const uint32_t splitters_in_use = marked ? static_cast<uint32_t>(sp.num_cuts()) : 0u;
        MRS_HIP_TRY(hipMemcpyAsync(d->tables_dev, t, sizeof(PartitionTables), hipMemcpyHostToDevice, s));
        if (int rc = partition_internal(d->part_plan, n_local, keys_in, vals_in, d->part_k, d->part_v, sp.top_shift(),
                                        sp.top_bits(), d->tables_dev->map, d->tables_dev->splitters, splitters_in_use,
                                        exact_mask, d->tables_dev->class_starts, s))
            return rc;
        if (int rc = partition_pass(d->part_plan, n_local, keys_in, vals_in, d->part_k, d->part_v, sp.top_shift(),
                                    sp.top_bits(), d->tables_dev->map, d->tables_dev->splitters, splitters_in_use, exact_mask,
                                    false, s))
            return rc;
        send_k = d->part_k;
        send_v = d->part_v;
    }
    if (d->timing) MRS_HIP_TRY(hipEventRecord(d->marks[2], s));

    // ---- 5-6. exchange piece by piece, received in source-rank order; local sort of each piece ----
    // The slice of this rank is the concatenation of its K pieces (consecutive key ranges), so
    // sorting each piece where it arrived leaves the whole slice sorted.  With K = 2 the sort of
    // piece 0 runs on the side stream while piece 1 is still on the links.
    int in_b = 0;
    uint64_t region = 0;  // first element of the piece in the receive buffers
    for (int c = 0; c < K; ++c) {
        uint64_t send_off[kSplitMaxRanks], send_cnt[kSplitMaxRanks], recv_off[kSplitMaxRanks], recv_cnt[kSplitMaxRanks];
        uint64_t ro = region;
        for (int p = 0; p < P; ++p) {
            const int v = p * K + c;  // destination slice: piece c of rank p
            send_cnt[p] = sp.send_count(me, v);
            send_off[p] = v == 0 ? 0 : sp.boundary(v - 1, me);
            recv_cnt[p] = sp.send_count(p, me * K + c);
            recv_off[p] = ro;
            ro += recv_cnt[p];
        }
        const uint64_t piece = ro - region;
        if (c == K - 1) d->last_piece_pairs = piece;
        ExchangeArrays x{};
        x.send[0] = send_k;
        x.recv[0] = d->recv_k;
        x.elem_bytes[0] = d->key_bytes;
        x.count = 1;
        if (d->val_bytes) {
            x.send[1] = send_v;
            x.recv[1] = d->recv_v;
            x.elem_bytes[1] = d->val_bytes;
            x.count = 2;
        }
        if (int rc = comm->alltoallv(x, send_off, send_cnt, recv_off, recv_cnt, s)) return rc;
        hipStream_t sort_stream = s;
        if (K > 1) {
            MRS_HIP_TRY(hipEventRecord(d->arrived[c], s));
            MRS_HIP_TRY(hipStreamWaitEvent(d->side, d->arrived[c], 0));
            sort_stream = d->side;
        }
        if (c == K - 1 && d->timing) MRS_HIP_TRY(hipEventRecord(d->marks[3], s));
        auto at = [&](void* base, size_t elem) { return base ? static_cast<char*>(base) + region * elem : nullptr; };
        // every key of piece c lies between the cuts on either side of destination slice me * K + c
        const int v = me * K + c;
        const uint64_t top_key = d->desc.end_bit >= 64 ? ~0ull : (1ull << d->desc.end_bit) - 1ull;
        const uint64_t window[2] = {v == 0 ? 0ull : sp.key(v - 1), v == P * K - 1 ? top_key : sp.key(v)};
        if (int rc = sort_pairs_internal(d->sort_plan[c], piece, at(d->recv_k, d->key_bytes), at(d->recv_v, d->val_bytes),
                                         at(d->tmp_k, d->key_bytes), at(d->tmp_v, d->val_bytes), sort_stream, &in_b,
                                         P * K > 1 ? window : nullptr))
            return rc;
        region = ro;
    }
    if (K > 1) {
        MRS_HIP_TRY(hipEventRecord(d->sorted_all, d->side));
        MRS_HIP_TRY(hipStreamWaitEvent(s, d->sorted_all, 0));
    }
    if (d->timing) {
        MRS_HIP_TRY(hipEventRecord(d->marks[4], s));
        d->marks_valid = true;
    }
    *keys_out = in_b ? d->tmp_k : d->recv_k;
    if (vals_out) *vals_out = d->val_bytes ? (in_b ? d->tmp_v : d->recv_v) : nullptr;
    *n_out = n_recv;
    d->last_n_out = n_recv;
    return MRS_OK;
}
