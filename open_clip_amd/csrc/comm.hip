// Collectives of the hot path behind the C ABI (SURVEY.md 8b "Exports needed": ocn_comm_*): the feature all-gather of
// gather_features (reference src/open_clip/loss.py:29-54), its reduce-scatter backward (:23-26) and a gradient all-reduce, as direct
// RCCL calls on the caller's stream -- no process-group object, no Python in between.  RCCL is bound at run time with dlopen /
// dlsym (the library torch has already loaded: its bundled librccl.so), so libopenclip_hip.so carries no link-time dependency on it.
// One process per GPU; a communicator is created from a 128-byte unique id that rank 0 makes and the caller distributes (the
// reference broadcasts such things with distributed.broadcast_object, open_clip_train/distributed.py:186-193).
#include "ocn_common.h"

#include <dlfcn.h>
#include <string.h>

namespace {

typedef struct { char internal[128]; } UniqueId;  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                               // ncclComm_t
enum { kSum = 0, kAvg = 4, kUint8 = 1, kFloat32 = 7, kBfloat16 = 9 };   // ncclSum, ncclAvg, ncclUint8, ncclFloat32, ncclBfloat16 (rccl.h)

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*CommCount)(Comm, int*) = nullptr;
    int (*CommUserRank)(Comm, int*) = nullptr;
    int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.handle ? &r : nullptr;
    tried = true;
    for (const char* name : {"librccl.so", "librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);  // the copy torch.distributed has loaded, if any
        if (!r.handle) r.handle = dlopen(name, RTLD_NOW);
        if (r.handle) break;
    }
    if (!r.handle) return nullptr;
    *(void**)&r.GetUniqueId = dlsym(r.handle, "ncclGetUniqueId");
    *(void**)&r.CommInitRank = dlsym(r.handle, "ncclCommInitRank");
    *(void**)&r.CommDestroy = dlsym(r.handle, "ncclCommDestroy");
    *(void**)&r.AllGather = dlsym(r.handle, "ncclAllGather");
    *(void**)&r.ReduceScatter = dlsym(r.handle, "ncclReduceScatter");
    *(void**)&r.AllReduce = dlsym(r.handle, "ncclAllReduce");
    *(void**)&r.Broadcast = dlsym(r.handle, "ncclBroadcast");
    *(void**)&r.GetErrorString = dlsym(r.handle, "ncclGetErrorString");
    *(void**)&r.CommCount = dlsym(r.handle, "ncclCommCount");
    *(void**)&r.CommUserRank = dlsym(r.handle, "ncclCommUserRank");
    *(void**)&r.Send = dlsym(r.handle, "ncclSend");
    *(void**)&r.Recv = dlsym(r.handle, "ncclRecv");
    *(void**)&r.GroupStart = dlsym(r.handle, "ncclGroupStart");
    *(void**)&r.GroupEnd = dlsym(r.handle, "ncclGroupEnd");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.ReduceScatter || !r.AllReduce) r.handle = nullptr;
    return r.handle ? &r : nullptr;
}

int dtype_of(int d) { return d == 1 ? kBfloat16 : (d == 2 ? kUint8 : kFloat32); }  // 0 = fp32, 1 = bf16, 2 = raw bytes (broadcast only)

#define OCN_RCCL(call, what)                                                                          \
    do {                                                                                              \
        const int rc__ = (call);                                                                      \
        if (rc__ != 0) {                                                                              \
            ocn_set_error("%s: RCCL error %d (%s)", what, rc__, R->GetErrorString ? R->GetErrorString(rc__) : "?"); \
            return OCN_ERR_LAUNCH;                                                                    \
        }                                                                                             \
    } while (0)

}  // namespace

extern "C" int ocn_comm_unique_id(void* id_out_128) {
    OCN_CHECK_ARG(id_out_128, "ocn_comm_unique_id: null output");
    Rccl* R = rccl();
    OCN_CHECK_ARG(R, "ocn_comm_unique_id: librccl.so could not be loaded");
    UniqueId id;
    OCN_RCCL(R->GetUniqueId(&id), "ocn_comm_unique_id");
    memcpy(id_out_128, &id, sizeof(id));
    return OCN_OK;
}

extern "C" int ocn_comm_init(const void* id_128, int rank, int world, void** comm_out) {
    OCN_CHECK_ARG(id_128 && comm_out && world > 0 && rank >= 0 && rank < world, "ocn_comm_init: bad arguments (rank %d of %d)", rank, world);
    Rccl* R = rccl();
    OCN_CHECK_ARG(R, "ocn_comm_init: librccl.so could not be loaded");
    UniqueId id;
    memcpy(&id, id_128, sizeof(id));
    Comm c = nullptr;
    OCN_RCCL(R->CommInitRank(&c, world, id, rank), "ocn_comm_init");
    *comm_out = c;
    return OCN_OK;
}

extern "C" int ocn_comm_destroy(void* comm) {
    Rccl* R = rccl();
    OCN_CHECK_ARG(R && comm, "ocn_comm_destroy: bad arguments");
    // ncclCommDestroy does not order itself behind the streams its collectives were enqueued on: drain the device first, here, so that no
    // caller can destroy a communicator under a running collective
    if (hipDeviceSynchronize() != hipSuccess) {
        ocn_set_error("ocn_comm_destroy: hipDeviceSynchronize failed");
        return OCN_ERR_LAUNCH;
    }
    OCN_RCCL(R->CommDestroy((Comm)comm), "ocn_comm_destroy");
    return OCN_OK;
}

// recv [world * count] = concatenation over ranks of send [count]   (loss.py:43-46: the packed [B, 2E] feature all-gather)
extern "C" int ocn_comm_allgather(void* comm, const void* send, void* recv, int64_t count_per_rank, int dtype, ocn_stream_t stream) {
    Rccl* R = rccl();
    OCN_CHECK_ARG(R && comm && send && recv && count_per_rank > 0, "ocn_comm_allgather: bad arguments");
    OCN_RCCL(R->AllGather(send, recv, (size_t)count_per_rank, dtype_of(dtype), (Comm)comm, (hipStream_t)stream), "ocn_comm_allgather");
    return OCN_OK;
}

// recv [count] = this rank's slice of the element-wise sum over ranks of send [world * count]   (loss.py:23-26: backward of the gather)
extern "C" int ocn_comm_reduce_scatter_sum(void* comm, const void* send, void* recv, int64_t count_per_rank, int dtype, ocn_stream_t stream) {
    Rccl* R = rccl();
    OCN_CHECK_ARG(R && comm && send && recv && count_per_rank > 0, "ocn_comm_reduce_scatter_sum: bad arguments");
    OCN_RCCL(R->ReduceScatter(send, recv, (size_t)count_per_rank, dtype_of(dtype), kSum, (Comm)comm, (hipStream_t)stream), "ocn_comm_reduce_scatter_sum");
    return OCN_OK;
}

// buf [count] = element-wise sum over ranks, in place   (the scalar sums of the row-sharded loss; gradient buckets)
extern "C" int ocn_comm_allreduce_sum(void* comm, void* buf, int64_t count, int dtype, ocn_stream_t stream) {
    Rccl* R = rccl();
    OCN_CHECK_ARG(R && comm && buf && count > 0, "ocn_comm_allreduce_sum: bad arguments");
    OCN_RCCL(R->AllReduce(buf, buf, (size_t)count, dtype_of(dtype), kSum, (Comm)comm, (hipStream_t)stream), "ocn_comm_allreduce_sum");
    return OCN_OK;
}

// buf [count] on every rank = buf of rank `root`, in place (ncclBroadcast): the start-of-training parameter broadcast (what DDP does at
// construction, base_task.py:227) -- exact for every dtype, one collective per tensor
extern "C" int ocn_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, ocn_stream_t stream) {
    Rccl* R = rccl();
    OCN_CHECK_ARG(R, "ocn_comm_broadcast: librccl.so could not be loaded");
    OCN_CHECK_ARG(R->Broadcast, "ocn_comm_broadcast: the loaded librccl.so exports no ncclBroadcast");
    OCN_CHECK_ARG(comm && buf && count > 0 && root >= 0 && dtype >= 0 && dtype <= 2, "ocn_comm_broadcast: bad arguments");
    OCN_RCCL(R->Broadcast(buf, buf, (size_t)count, dtype_of(dtype), root, (Comm)comm, (hipStream_t)stream), "ocn_comm_broadcast");
    return OCN_OK;
}

// buf [count] = element-wise MEAN over ranks, in place (ncclAvg): the gradient all-reduce of the data-parallel step -- what DDP's reducer
// computes (base_task.py:219-232), issued per residual block on the block's own gradient arena (open_clip_amd/grad_sync.py)
extern "C" int ocn_comm_allreduce_avg(void* comm, void* buf, int64_t count, int dtype, ocn_stream_t stream) {
    Rccl* R = rccl();
    OCN_CHECK_ARG(R && comm && buf && count > 0, "ocn_comm_allreduce_avg: bad arguments");
    OCN_RCCL(R->AllReduce(buf, buf, (size_t)count, dtype_of(dtype), kAvg, (Comm)comm, (hipStream_t)stream), "ocn_comm_allreduce_avg");
    return OCN_OK;
}

// What the COMMUNICATOR says about itself (ncclCommCount / ncclCommUserRank): the number of ranks RCCL connected and this process's rank among them.
// bench.py puts the count on its line next to torch.distributed's world size: a line that claims N GPUs shows how many ranks the transport saw.
extern "C" int ocn_comm_count(void* comm, int* count_out, int* rank_out) {
    Rccl* R = rccl();
    OCN_CHECK_ARG(R && comm && count_out, "ocn_comm_count: bad arguments");
    OCN_CHECK_ARG(R->CommCount && R->CommUserRank, "ocn_comm_count: the loaded librccl.so exports no ncclCommCount / ncclCommUserRank");
    OCN_RCCL(R->CommCount((Comm)comm, count_out), "ocn_comm_count");
    if (rank_out) OCN_RCCL(R->CommUserRank((Comm)comm, rank_out), "ocn_comm_count");
    return OCN_OK;
}

// One neighbour exchange (reference src/open_clip/loss.py:226-243 `neighbour_exchange`: batch_isend_irecv of one isend + one irecv): send [count] to
// `to_rank` and receive [count] from `from_rank` as ONE grouped RCCL operation on the caller's stream (to_rank == from_rank == own rank is a copy).
extern "C" int ocn_comm_sendrecv(void* comm, const void* send, int to_rank, void* recv, int from_rank, int64_t count, int dtype, ocn_stream_t stream) {
    Rccl* R = rccl();
    OCN_CHECK_ARG(R && comm && send && recv && count > 0 && to_rank >= 0 && from_rank >= 0 && dtype >= 0 && dtype <= 2, "ocn_comm_sendrecv: bad arguments");
    OCN_CHECK_ARG(R->Send && R->Recv && R->GroupStart && R->GroupEnd, "ocn_comm_sendrecv: the loaded librccl.so exports no ncclSend / ncclRecv / ncclGroup*");
    OCN_RCCL(R->GroupStart(), "ocn_comm_sendrecv");
    const int rs = R->Send(send, (size_t)count, dtype_of(dtype), to_rank, (Comm)comm, (hipStream_t)stream);
    const int rr = R->Recv(recv, (size_t)count, dtype_of(dtype), from_rank, (Comm)comm, (hipStream_t)stream);
    const int re = R->GroupEnd();
    OCN_RCCL(rs, "ocn_comm_sendrecv (send)");
    OCN_RCCL(rr, "ocn_comm_sendrecv (recv)");
    OCN_RCCL(re, "ocn_comm_sendrecv");
    return OCN_OK;
}
