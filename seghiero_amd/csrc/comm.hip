// RCCL behind the C ABI (last row): what a host WITHOUT PyTorch needs to drive the data-parallel step -- the Python
// package itself uses torch.distributed (backend "nccl" = RCCL), see seghiero_amd/ddp.py.  One communicator per process (one process per
// GPU), created from a 128-byte unique id that rank 0 generates and the launcher hands to every rank; collectives run either in order on
// the caller's stream, or on the communicator's own side stream with event hand-off (the gradient buckets of ddp.GradSync: the side
// stream waits for what the producer stream has queued, the consumer stream later waits for the side stream).
// RCCL is bound lazily (dlopen): libseghiero_hip.so has no link-time dependency on it, and a single-GPU process never loads it.
#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>
#include "common.h"

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r = [] {
        Rccl q;
        // an RCCL that is already mapped (PyTorch's own copy) is reused; otherwise ROCm's
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) { q.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (q.lib) break; }
        if (!q.lib) for (const char* n : names) { q.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (q.lib) break; }
        if (!q.lib) return q;
        q.GetUniqueId = reinterpret_cast<decltype(q.GetUniqueId)>(dlsym(q.lib, "ncclGetUniqueId"));
        q.CommInitRank = reinterpret_cast<decltype(q.CommInitRank)>(dlsym(q.lib, "ncclCommInitRank"));
        q.CommDestroy = reinterpret_cast<decltype(q.CommDestroy)>(dlsym(q.lib, "ncclCommDestroy"));
        q.AllReduce = reinterpret_cast<decltype(q.AllReduce)>(dlsym(q.lib, "ncclAllReduce"));
        q.Broadcast = reinterpret_cast<decltype(q.Broadcast)>(dlsym(q.lib, "ncclBroadcast"));
        q.ok = q.GetUniqueId && q.CommInitRank && q.CommDestroy && q.AllReduce && q.Broadcast;
        return q;
    }();
    return r;
}
struct ShComm {
    ncclComm_t comm;
    hipStream_t side;
    hipEvent_t ev_in, ev_out;
    int world, rank;
    bool pending;          // something was issued on the side stream since the last sh_comm_wait
};
bool dtype_of(int dtype, ncclDataType_t& t) {
    switch (dtype) {
    case 0: t = ncclFloat32; return true;
    case 1: t = ncclFloat64; return true;
    case 2: t = ncclInt64; return true;
    default: return false;
    }
}
bool op_of(int op, ncclRedOp_t& o) {
    switch (op) {
    case 0: o = ncclSum; return true;
    case 1: o = ncclMin; return true;
    case 2: o = ncclMax; return true;
    default: return false;
    }
}
}  // namespace

// id128 <- a fresh 128-byte communicator id (rank 0 calls this; the launcher distributes the bytes to every rank)
extern "C" int sh_comm_unique_id(void* id128) {
    if (!id128) return SH_EINVAL;
    Rccl& r = rccl();
    if (!r.ok) return SH_EUNSUPPORTED;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return SH_ELAUNCH;
    std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return SH_OK;
}
// *comm_out <- communicator of this process's CURRENT HIP device in a job of `world` ranks (collective over the ranks: blocks until all
// of them have called it with the same id); also creates the side stream and the two hand-off events
extern "C" int sh_comm_init(const void* id128, int world, int rank, void** comm_out) {
    if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return SH_EINVAL;
    Rccl& r = rccl();
    if (!r.ok) return SH_EUNSUPPORTED;
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ShComm* c = new ShComm{};
    c->world = world; c->rank = rank; c->pending = false;
    if (r.CommInitRank(&c->comm, world, id, rank) != ncclSuccess) { delete c; return SH_ELAUNCH; }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) {
        r.CommDestroy(c->comm); delete c; return SH_ELAUNCH;
    }
    *comm_out = c;
    return SH_OK;
}
extern "C" int sh_comm_destroy(void* comm) {
    if (!comm) return SH_EINVAL;
    ShComm* c = static_cast<ShComm*>(comm);
    (void)hipStreamSynchronize(c->side);
    rccl().CommDestroy(c->comm);
    (void)hipEventDestroy(c->ev_in); (void)hipEventDestroy(c->ev_out); (void)hipStreamDestroy(c->side);
    delete c;
    return SH_OK;
}
// in-place all-reduce of `count` elements (dtype 0 f32, 1 f64, 2 i64; op 0 sum, 1 min, 2 max), in order on `stream`
extern "C" int sh_comm_all_reduce(void* comm, void* buf, int64_t count, int dtype, int op, void* stream) {
    ncclDataType_t t; ncclRedOp_t o;
    if (!comm || !buf || count <= 0 || !dtype_of(dtype, t) || !op_of(op, o)) return SH_EINVAL;
    ShComm* c = static_cast<ShComm*>(comm);
    return rccl().AllReduce(buf, buf, (size_t)count, t, o, c->comm, (hipStream_t)stream) == ncclSuccess ? SH_OK : SH_ELAUNCH;
}
// the same on the communicator's side stream: it first waits for everything `producer_stream` has queued so far (the kernels that wrote
// buf), the producer stream itself does not wait -- backward continues while the bucket is reduced.  sh_comm_wait orders a consumer.
extern "C" int sh_comm_all_reduce_async(void* comm, void* buf, int64_t count, int dtype, int op, void* producer_stream) {
    ncclDataType_t t; ncclRedOp_t o;
    if (!comm || !buf || count <= 0 || !dtype_of(dtype, t) || !op_of(op, o)) return SH_EINVAL;
    ShComm* c = static_cast<ShComm*>(comm);
    if (hipEventRecord(c->ev_in, (hipStream_t)producer_stream) != hipSuccess || hipStreamWaitEvent(c->side, c->ev_in, 0) != hipSuccess) return SH_ELAUNCH;
    c->pending = true;
    return rccl().AllReduce(buf, buf, (size_t)count, t, o, c->comm, c->side) == ncclSuccess ? SH_OK : SH_ELAUNCH;
}
// `consumer_stream` waits for every collective issued on the side stream so far (before the optimizer reads the reduced gradients)
extern "C" int sh_comm_wait(void* comm, void* consumer_stream) {
    if (!comm) return SH_EINVAL;
    ShComm* c = static_cast<ShComm*>(comm);
    if (!c->pending) return SH_OK;
    if (hipEventRecord(c->ev_out, c->side) != hipSuccess || hipStreamWaitEvent((hipStream_t)consumer_stream, c->ev_out, 0) != hipSuccess) return SH_ELAUNCH;
    c->pending = false;
    return SH_OK;
}
// buf (bytes) of rank `root` to every rank, in order on `stream` (weights, BatchNorm running statistics and `step` at start-up)
extern "C" int sh_comm_broadcast(void* comm, void* buf, int64_t bytes, int root, void* stream) {
    if (!comm || !buf || bytes <= 0) return SH_EINVAL;
    ShComm* c = static_cast<ShComm*>(comm);
    if (root < 0 || root >= c->world) return SH_EINVAL;
    return rccl().Broadcast(buf, buf, (size_t)bytes, ncclUint8, root, c->comm, (hipStream_t)stream) == ncclSuccess ? SH_OK : SH_ELAUNCH;
}
