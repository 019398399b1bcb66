// SURVEY.md section 8e, the one exchange step of the path behind the C ABI: every rank's local top-k
// (score, GLOBAL id) -> ONE ncclAllGather over RCCL / xGMI -> merge (rl_merge_topk's kernel).  The reference is
// single-process (its "merge" is SQL ORDER BY ... LIMIT over one table, src/raglite/_search.py:75-79,143-149); this
// is what lets a pure-ctypes caller -- the boundary BASELINE.json names -- shard the corpus by chunk over the GPUs of a
// node without going through torch.distributed.
//
// librccl is loaded at the first rl_comm_* call with dlopen (no link-time dependency: libraglite_hip.so keeps loading
// on a box without RCCL, and inside a PyTorch process the loader hands back the librccl.so.1 torch already mapped, so
// both talk to ONE RCCL).  B x k x 8 bytes per rank and step (102 KB at 128 queries x top-100): one all-gather,
// no bucketing -- over xGMI's point-to-point links that is latency, not bandwidth.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace rl {
namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) { api.error = std::string("librccl not found: ") + dlerror(); return; }
#define RL_SYM(field, sym)                                                       \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym));  \
    if (!api.field) { api.error = std::string("librccl lacks ") + sym; return; }
        RL_SYM(GetUniqueId, "ncclGetUniqueId")
        RL_SYM(CommInitRank, "ncclCommInitRank")
        RL_SYM(AllGather, "ncclAllGather")
        RL_SYM(AllReduce, "ncclAllReduce")
        RL_SYM(CommDestroy, "ncclCommDestroy")
        RL_SYM(GetErrorString, "ncclGetErrorString")
#undef RL_SYM
    });
    return api;
}

#define RL_NCCL(expr)                                                                                   \
    do {                                                                                                \
        ncclResult_t _r = (expr);                                                                       \
        if (_r != ncclSuccess) return ::rl::fail(RL_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(_r)); \
    } while (0)

// (score bits, id + offset) pairs, negative ids stay what they are (-1: an empty slot; RL_ID_SHARD_MISSING: the rank that sends it failed
// in its local step and its lists are empty): the record every rank contributes.  Zeroes the communicator's "a shard is missing" word for
// the unpack that follows the all-gather on the same stream.
__global__ __launch_bounds__(256) void pack_topk_kernel(const float* __restrict__ scores, const int32_t* __restrict__ ids, int64_t n,
                                                         int32_t id_offset, int2* __restrict__ out, uint32_t* __restrict__ missing) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *missing = 0u;
    if (i >= n) return;
    const int32_t id = ids[i];
    out[i] = make_int2(__float_as_int(scores[i]), id >= 0 ? id + id_offset : id);
}
__global__ __launch_bounds__(256) void unpack_topk_kernel(const int2* __restrict__ in, int64_t n, float* __restrict__ scores,
                                                           int32_t* __restrict__ ids, uint32_t* __restrict__ missing) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int2 v = in[i];
    scores[i] = __int_as_float(v.x);
    ids[i] = v.y;
    if (v.y == RL_ID_SHARD_MISSING) atomicOr(missing, 1u);
}
// A merge that lacks a shard must never look like an answer: every score NaN, every id -1, on every rank alike, without a host read-back.
__global__ __launch_bounds__(256) void poison_if_missing_kernel(const uint32_t* __restrict__ missing, int64_t n, float* __restrict__ scores,
                                                                 int32_t* __restrict__ ids) {
    if (*missing == 0u) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    scores[i] = __int_as_float(0x7fc00000);
    ids[i] = -1;
}

}  // namespace
}  // namespace rl

struct rl_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    std::mutex mu;
    void* send = nullptr;   // [n x int2]
    void* recv = nullptr;   // [world x n x int2]
    float* g_scores = nullptr;   // rl_allgather_merge_topk: [world x n]
    int32_t* g_ids = nullptr;
    uint32_t* missing = nullptr;  // device word: some rank's records of the last exchange carried RL_ID_SHARD_MISSING
    size_t cap = 0;         // records (n) the buffers hold
    // The buffers above are shared by all calls on this communicator; `mu` serialises only their host side.  Calls are asynchronous,
    // so a call arriving on a DIFFERENT stream than the previous one first waits for that stream (as rl_index does for its scratch).
    hipStream_t last_stream = nullptr;
    bool last_stream_set = false;
};

using namespace rl;

namespace {
int comm_use(rl_comm* c, hipStream_t s) {
    if (c->last_stream_set && c->last_stream != s) RL_HIP(hipStreamSynchronize(c->last_stream));
    c->last_stream = s;
    c->last_stream_set = true;
    return RL_OK;
}
int comm_reserve(rl_comm* c, size_t n) {
    if (!c->missing) RL_HIP(hipMalloc(&c->missing, sizeof(uint32_t)));
    if (n <= c->cap) return RL_OK;
    for (void* p : {c->send, c->recv, (void*)c->g_scores, (void*)c->g_ids}) if (p) (void)hipFree(p);
    c->send = c->recv = nullptr; c->g_scores = nullptr; c->g_ids = nullptr; c->cap = 0;
    RL_HIP(hipMalloc(&c->send, n * sizeof(int2)));
    RL_HIP(hipMalloc(&c->recv, (size_t)c->world * n * sizeof(int2)));
    RL_HIP(hipMalloc(&c->g_scores, (size_t)c->world * n * sizeof(float)));
    RL_HIP(hipMalloc(&c->g_ids, (size_t)c->world * n * sizeof(int32_t)));
    c->cap = n;
    return RL_OK;
}
// pack -> ONE all-gather -> unpack, on stream s; out_* are [world x n]
int allgather_locked(rl_comm* c, const float* local_scores, const int32_t* local_ids, int64_t n, int32_t id_offset, float* out_scores,
                     int32_t* out_ids, hipStream_t s) {
    RL_TRY(comm_reserve(c, (size_t)n));
    const unsigned blocks = (unsigned)((n + 255) / 256), gblocks = (unsigned)((n * c->world + 255) / 256);
    hipLaunchKernelGGL(pack_topk_kernel, dim3(blocks), dim3(256), 0, s, local_scores, local_ids, n, id_offset, static_cast<int2*>(c->send),
                       c->missing);
    RL_HIP(hipGetLastError());
    RL_NCCL(rccl().AllGather(c->send, c->recv, (size_t)n * 2, ncclInt32, c->comm, s));
    hipLaunchKernelGGL(unpack_topk_kernel, dim3(gblocks), dim3(256), 0, s, static_cast<const int2*>(c->recv), n * c->world, out_scores, out_ids,
                       c->missing);
    RL_HIP(hipGetLastError());
    return RL_OK;
}
}  // namespace

extern "C" {

int rl_comm_unique_id(void* out_id) {
    if (!out_id) return fail(RL_ERR_INVALID, "rl_comm_unique_id: null argument");
    RcclApi& api = rccl();
    if (!api.error.empty()) return fail(RL_ERR_UNSUPPORTED, "rl_comm_unique_id: " + api.error);
    ncclUniqueId id;
    RL_NCCL(api.GetUniqueId(&id));
    static_assert(sizeof(id) == RL_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(out_id, &id, sizeof(id));
    return RL_OK;
}

int rl_comm_init(rl_comm** out, int rank, int world, const void* unique_id) {
    if (!out) return fail(RL_ERR_INVALID, "rl_comm_init: null output handle");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !unique_id) return fail(RL_ERR_INVALID, "rl_comm_init: bad rank / world / id");
    RcclApi& api = rccl();
    if (!api.error.empty()) return fail(RL_ERR_UNSUPPORTED, "rl_comm_init: " + api.error);
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    rl_comm* c = new rl_comm();
    c->rank = rank;
    c->world = world;
    const ncclResult_t r = api.CommInitRank(&c->comm, world, id, rank);  // collective: every rank of `world` must call it
    if (r != ncclSuccess) {
        delete c;
        return fail(RL_ERR_HIP, std::string("ncclCommInitRank: ") + api.GetErrorString(r));
    }
    *out = c;
    return RL_OK;
}

int rl_comm_info(const rl_comm* comm, int* rank, int* world) {
    if (!comm) return fail(RL_ERR_INVALID, "rl_comm_info: null communicator");
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return RL_OK;
}

int rl_comm_destroy(rl_comm* comm) {
    if (!comm) return RL_OK;
    if (comm->comm) (void)rccl().CommDestroy(comm->comm);
    for (void* p : {comm->send, comm->recv, (void*)comm->g_scores, (void*)comm->g_ids, (void*)comm->missing}) if (p) (void)hipFree(p);
    delete comm;
    return RL_OK;
}

int rl_allgather_topk(rl_comm* comm, const float* local_scores, const int32_t* local_ids, int32_t n_queries, int32_t k,
                      int32_t id_offset, float* out_scores, int32_t* out_ids, void* stream) {
    if (!comm) return fail(RL_ERR_INVALID, "rl_allgather_topk: null communicator");
    if (n_queries < 0 || k < 1) return fail(RL_ERR_INVALID, "rl_allgather_topk: bad sizes");
    if (n_queries == 0) return RL_OK;
    if (!local_scores || !local_ids || !out_scores || !out_ids) return fail(RL_ERR_INVALID, "rl_allgather_topk: null argument");
    std::lock_guard<std::mutex> lock(comm->mu);
    RL_TRY(comm_use(comm, reinterpret_cast<hipStream_t>(stream)));
    return allgather_locked(comm, local_scores, local_ids, (int64_t)n_queries * k, id_offset, out_scores, out_ids,
                            reinterpret_cast<hipStream_t>(stream));
}

int rl_allgather_merge_topk(rl_comm* comm, const float* local_scores, const int32_t* local_ids, int32_t n_queries, int32_t k_in,
                            int32_t id_offset, int32_t k, float* out_scores, int32_t* out_ids, void* stream) {
    if (!comm) return fail(RL_ERR_INVALID, "rl_allgather_merge_topk: null communicator");
    if (n_queries < 0 || k_in < 1 || k < 1) return fail(RL_ERR_INVALID, "rl_allgather_merge_topk: bad sizes");
    if (k > K_MAX) return fail(RL_ERR_UNSUPPORTED, "rl_allgather_merge_topk: k must be <= 2048");
    if ((int64_t)comm->world * k_in > MERGE_CAP) return fail(RL_ERR_UNSUPPORTED, "rl_allgather_merge_topk: world * k_in must be <= 8192");
    if (n_queries == 0) return RL_OK;
    if (!local_scores || !local_ids || !out_scores || !out_ids) return fail(RL_ERR_INVALID, "rl_allgather_merge_topk: null argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lock(comm->mu);
    RL_TRY(comm_use(comm, s));
    RL_TRY(comm_reserve(comm, (size_t)n_queries * k_in));
    RL_TRY(allgather_locked(comm, local_scores, local_ids, (int64_t)n_queries * k_in, id_offset, comm->g_scores, comm->g_ids, s));
    RL_TRY(launch_merge_topk(comm->g_scores, comm->g_ids, comm->world, n_queries, k_in, k, out_scores, out_ids, s));
    const int64_t n_out = (int64_t)n_queries * k;
    hipLaunchKernelGGL(poison_if_missing_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, comm->missing, n_out, out_scores, out_ids);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int rl_allreduce_sum_u32(rl_comm* comm, uint32_t* buf, int64_t count, void* stream) {
    if (!comm || (!buf && count > 0) || count < 0) return fail(RL_ERR_INVALID, "rl_allreduce_sum_u32: bad arguments");
    if (count == 0 || comm->world == 1) return RL_OK;
    std::lock_guard<std::mutex> lock(comm->mu);
    RL_NCCL(rccl().AllReduce(buf, buf, (size_t)count, ncclUint32, ncclSum, comm->comm, reinterpret_cast<hipStream_t>(stream)));
    return RL_OK;
}

int rl_allgather_u32(rl_comm* comm, const uint32_t* local, int64_t count, uint32_t* out, void* stream) {
    if (!comm || count < 0 || (count > 0 && (!local || !out))) return fail(RL_ERR_INVALID, "rl_allgather_u32: bad arguments");
    if (count == 0) return RL_OK;
    std::lock_guard<std::mutex> lock(comm->mu);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (comm->world == 1) {
        if (out != local) RL_HIP(hipMemcpyAsync(out, local, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        return RL_OK;
    }
    RL_NCCL(rccl().AllGather(local, out, (size_t)count, ncclUint32, comm->comm, s));
    return RL_OK;
}

}  // extern "C"
