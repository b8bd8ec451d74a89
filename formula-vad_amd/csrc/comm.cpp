// comm.cpp -- the one collective of the path: gathering the per-stream Evaluator statistics of a plan whose
// streams were dealt to several GPUs (BASELINE config 4).
//
// The reference runs one pipeline per file on its own thread (src/simulator.zig:221-232) and builds the
// report from the per-instance statistics IN PLAN ORDER (src/simulator/report_generator.zig:48-68,
// src/Evaluator/statistics.zig:116-172: in-order f32 sums).  Here the instances live on different ranks, so
// the 11-float SingleStats of every stream are all-gathered (ncclAllGather of one fixed block per rank,
// <= 48 B per stream: latency only, xGMI bandwidth is irrelevant) and every rank can then run
// fvad_stats_aggregate over the plan-ordered array -- bit-identical to the single-process aggregate.  A bare
// all-reduce of the sums would reorder the f32 additions.
//
// RCCL is opened lazily (dlopen) the first time a communicator is asked for: single-GPU users of the library
// never load it, and a Zig / C host needs no PyTorch for the multi-GPU leg.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "internal.h"

using namespace fvad;

namespace {
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

RcclApi* rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            const char* why = dlerror(); // one call: dlerror() clears the message it returns
            api.err = std::string("dlopen(librccl.so) failed: ") + (why ? why : "?");
            return;
        }
#define SYM(field, sym)                                                             \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym));      \
    if (!api.field) { api.err = std::string("RCCL symbol missing: ") + sym; return; }
        SYM(GetUniqueId, "ncclGetUniqueId")
        SYM(CommInitRank, "ncclCommInitRank")
        SYM(CommDestroy, "ncclCommDestroy")
        SYM(AllGather, "ncclAllGather")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    });
    return &api;
}
constexpr int kRow = 12; // stream id (bit pattern) + the 11 floats of fvad_single_stats
static_assert(sizeof(fvad_single_stats) == 11 * sizeof(float), "SingleStats layout");
static_assert(sizeof(ncclUniqueId) == FVAD_COMM_ID_BYTES, "ncclUniqueId size");
} // namespace

struct fvad_comm {
    fvad_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    float* d_send = nullptr;
    float* d_recv = nullptr;
    size_t cap_rows = 0; // rows per rank the device blocks can hold
    uint32_t* d_hdr = nullptr; // [2 + 2 * world] words: this rank's {n_streams, status}, then every rank's
};

extern "C" {

int fvad_comm_unique_id(uint8_t* id, size_t n)
{
    if (!id || n < FVAD_COMM_ID_BYTES) return FVAD_ERR_INVALID_ARGUMENT;
    RcclApi* a = rccl();
    if (!a->err.empty()) return FVAD_ERR_NO_DEVICE;
    ncclUniqueId u;
    if (a->GetUniqueId(&u) != ncclSuccess) return FVAD_ERR_HIP;
    memcpy(id, &u, sizeof u);
    return FVAD_OK;
}

int fvad_comm_create(fvad_ctx* ctx, const uint8_t* id, size_t n, int world, int rank, fvad_comm** out)
{
    if (!ctx || !id || !out || n < FVAD_COMM_ID_BYTES || world < 1 || rank < 0 || rank >= world) return FVAD_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    RcclApi* a = rccl();
    if (!a->err.empty()) return set_err(ctx, FVAD_ERR_NO_DEVICE, a->err);
    hipSetDevice(ctx->device);
    auto* c = new (std::nothrow) fvad_comm();
    if (!c) return FVAD_ERR_ALLOC_FAILED;
    c->ctx = ctx; c->world = world; c->rank = rank;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    const ncclResult_t r = a->CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return set_err(ctx, FVAD_ERR_HIP, std::string("ncclCommInitRank: ") + a->GetErrorString(r));
    }
    if (hipMalloc((void**)&c->d_hdr, (size_t)(2 + 2 * world) * sizeof(uint32_t)) != hipSuccess) {
        a->CommDestroy(c->comm);
        delete c;
        return set_err(ctx, FVAD_ERR_ALLOC_FAILED, "hipMalloc failed");
    }
    *out = c;
    return FVAD_OK;
}

void fvad_comm_destroy(fvad_comm* c)
{
    if (!c) return;
    hipSetDevice(c->ctx->device);
    hipStreamSynchronize(c->ctx->stream);
    if (c->comm) rccl()->CommDestroy(c->comm);
    if (c->d_send) hipFree(c->d_send);
    if (c->d_recv) hipFree(c->d_recv);
    if (c->d_hdr) hipFree(c->d_hdr);
    delete c;
}

int fvad_comm_world(const fvad_comm* c) { return c ? c->world : 0; }
int fvad_comm_rank(const fvad_comm* c) { return c ? c->rank : -1; }

int fvad_stats_allgather(fvad_comm* c, const uint32_t* local_ids, const fvad_single_stats* local_stats, size_t n_local,
                         size_t n_streams, fvad_single_stats* out)
{
    if (!c || !out || (n_local && (!local_ids || !local_stats)) || n_streams == 0) return FVAD_ERR_INVALID_ARGUMENT;
    fvad_ctx* ctx = c->ctx;
    RcclApi* a = rccl();
    hipSetDevice(ctx->device);
    const size_t per_rank = (n_streams + (size_t)c->world - 1) / (size_t)c->world;
    // Everything that can fail on ONE rank is checked before any rank enters the data collective, and the verdict is
    // shared first: a header all-gather of {n_streams, local status} per rank (its count does not depend on the
    // arguments).  A rank with a bad argument or a failed allocation still takes part in it, so no rank is left
    // waiting in ncclAllGather for one that returned early, and ranks that disagree about n_streams (which sets the
    // data collective's count) find out here.
    int local_rc = FVAD_OK;
    const char* local_msg = "";
    if (n_local > per_rank) { local_rc = FVAD_ERR_INVALID_ARGUMENT; local_msg = "more local streams than ceil(n_streams / world)"; }
    for (size_t j = 0; j < n_local && !local_rc; ++j)
        if (local_ids[j] >= n_streams) { local_rc = FVAD_ERR_OUT_OF_RANGE; local_msg = "stream id >= n_streams"; }
    if (!local_rc && per_rank > c->cap_rows) {
        hipStreamSynchronize(ctx->stream);
        if (c->d_send) hipFree(c->d_send);
        if (c->d_recv) hipFree(c->d_recv);
        c->d_send = c->d_recv = nullptr; c->cap_rows = 0;
        if (hipMalloc((void**)&c->d_send, per_rank * kRow * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&c->d_recv, per_rank * kRow * sizeof(float) * (size_t)c->world) != hipSuccess) {
            (void)hipGetLastError();
            local_rc = FVAD_ERR_ALLOC_FAILED; local_msg = "hipMalloc failed";
        } else c->cap_rows = per_rank;
    }
    {
        const uint32_t hdr[2] = {(uint32_t)n_streams, (uint32_t)(-local_rc)};
        std::vector<uint32_t> all_hdr(2 * (size_t)c->world);
        FVAD_HIP(ctx, hipMemcpyAsync(c->d_hdr, hdr, sizeof hdr, hipMemcpyHostToDevice, ctx->stream));
        const ncclResult_t r = a->AllGather(c->d_hdr, c->d_hdr + 2, 2, ncclUint32, c->comm, ctx->stream);
        if (r != ncclSuccess) return set_err(ctx, FVAD_ERR_HIP, std::string("ncclAllGather (header): ") + a->GetErrorString(r));
        FVAD_HIP(ctx, hipMemcpyAsync(all_hdr.data(), c->d_hdr + 2, all_hdr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        FVAD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (local_rc) return set_err(ctx, local_rc, local_msg);
        for (int rk = 0; rk < c->world; ++rk) {
            if (all_hdr[2 * rk + 1]) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "fvad_stats_allgather failed on rank " + std::to_string(rk) + " (status " + fvad_status_name(-(int)all_hdr[2 * rk + 1]) + ")");
            if (all_hdr[2 * rk] != (uint32_t)n_streams) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "ranks disagree about n_streams (rank " + std::to_string(rk) + ")");
        }
    }
    // this rank's block: [per_rank][12], unused rows carry the id 0xFFFFFFFF
    std::vector<float> block(per_rank * kRow, 0.0f);
    for (size_t j = 0; j < per_rank; ++j) {
        uint32_t id = 0xFFFFFFFFu;
        if (j < n_local) {
            id = local_ids[j];
            memcpy(&block[j * kRow + 1], &local_stats[j], sizeof(fvad_single_stats));
        }
        memcpy(&block[j * kRow], &id, sizeof id);
    }
    FVAD_HIP(ctx, hipMemcpyAsync(c->d_send, block.data(), block.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    const ncclResult_t r = a->AllGather(c->d_send, c->d_recv, per_rank * kRow, ncclFloat, c->comm, ctx->stream);
    if (r != ncclSuccess) return set_err(ctx, FVAD_ERR_HIP, std::string("ncclAllGather: ") + a->GetErrorString(r));
    std::vector<float> all(per_rank * kRow * (size_t)c->world);
    FVAD_HIP(ctx, hipMemcpyAsync(all.data(), c->d_recv, all.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    FVAD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // plan order: row `id` of the result is stream `id`, whichever rank computed it
    std::vector<char> seen(n_streams, 0);
    for (size_t j = 0; j < per_rank * (size_t)c->world; ++j) {
        uint32_t id;
        memcpy(&id, &all[j * kRow], sizeof id);
        if (id == 0xFFFFFFFFu) continue;
        if (id >= n_streams || seen[id]) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "stream id gathered twice or out of range");
        memcpy(&out[id], &all[j * kRow + 1], sizeof(fvad_single_stats));
        seen[id] = 1;
    }
    for (size_t i = 0; i < n_streams; ++i)
        if (!seen[i]) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "a stream's statistics never arrived");
    return FVAD_OK;
}

} // extern "C"
