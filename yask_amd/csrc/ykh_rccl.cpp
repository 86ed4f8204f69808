// ykh_rccl.cpp -- built-in halo transport: RCCL point-to-point over xGMI.
//
// Replaces the reference's MPI_Isend/MPI_Irecv/MPI_Wait halo traffic (src/kernel/lib/halo.cpp:223-225,
// 331-332, 370-374) and its MPI_Allreduce scalars (src/kernel/lib/utils.cpp:74-86). All sends and
// receives of one exchange are issued inside a single ncclGroupStart/End on the communication
// stream, so they are stream-ordered after the pack kernels and before the unpack kernels and can
// run concurrently with the interior stencil kernel on the compute stream.  xGMI is point-to-point
// (one link per neighbour pair), so one large message per neighbour is the efficient shape.
//
// librccl is resolved at run time (dlopen) so that the stencil library itself has no link-time RCCL
// dependency and shares the copy PyTorch has already loaded when used from Python.
#include <dlfcn.h>

#include <cstring>

#include "../../include/yask_hip_c_api.h"
#include "ykh_handles.hpp"
#include "ykh_runtime.hpp"

namespace {

struct ncclUniqueId_t { char internal[128]; };
typedef void* ncclComm_t;
enum { NCCL_INT8 = 0, NCCL_INT64 = 4 };
enum { NCCL_SUM = 0, NCCL_MAX = 2, NCCL_MIN = 3 };

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(ncclUniqueId_t*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId_t, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    if (r.h) return r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (auto n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (r.h) break; }
    if (!r.h)
        for (auto n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
    if (!r.h) YKH_THROW(std::string("cannot load librccl: ") + dlerror());
#define SYM(f, n) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.h, n)); if (!r.f) YKH_THROW(std::string("librccl lacks symbol ") + n)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
    SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return r;
}

struct RcclState {
    ncclComm_t comm = nullptr;
    long long* dscalar = nullptr;
    hipStream_t stream = nullptr;
};

#define NCCL_OK(call)                                                                              \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        if (rc_ != 0) { fprintf(stderr, "RCCL error in %s: %s\n", #call, rccl().GetErrorString(rc_)); return 1; } \
    } while (0)

int rccl_start(void* user, int n, const ykh::HaloMsg* m, void* stream) {
    RcclState* st = static_cast<RcclState*>(user);
    Rccl& r = rccl();
    NCCL_OK(r.GroupStart());
    // a failing send / recv must not leave the group open (the next exchange would nest inside it): remember the first error, stop
    // queueing, and ALWAYS reach GroupEnd
    int err = 0;
    const char* where = "";
    for (int i = 0; i < n && !err; i++) {
        if (m[i].recv_bytes && (err = r.Recv(m[i].recv_buf, m[i].recv_bytes, NCCL_INT8, m[i].peer, st->comm, (hipStream_t)stream))) where = "ncclRecv";
        if (!err && m[i].send_bytes && (err = r.Send(m[i].send_buf, m[i].send_bytes, NCCL_INT8, m[i].peer, st->comm, (hipStream_t)stream))) where = "ncclSend";
    }
    int end = r.GroupEnd();
    if (err) { fprintf(stderr, "RCCL error in %s: %s\n", where, r.GetErrorString(err)); return 1; }
    if (end) { fprintf(stderr, "RCCL error in ncclGroupEnd: %s\n", r.GetErrorString(end)); return 1; }
    return 0;
}
int rccl_wait(void*, int, const ykh::HaloMsg*, void*) { return 0; }   // stream-ordered already

int rccl_allreduce(void* user, int op, long long* val) {
    RcclState* st = static_cast<RcclState*>(user);
    Rccl& r = rccl();
    if (hipMemcpyAsync(st->dscalar, val, sizeof(long long), hipMemcpyHostToDevice, st->stream) != hipSuccess) return 1;
    int nop = op == 0 ? NCCL_SUM : (op == 1 ? NCCL_MIN : NCCL_MAX);
    NCCL_OK(r.AllReduce(st->dscalar, st->dscalar, 1, NCCL_INT64, nop, st->comm, st->stream));
    if (hipMemcpyAsync(val, st->dscalar, sizeof(long long), hipMemcpyDeviceToHost, st->stream) != hipSuccess) return 1;
    if (hipStreamSynchronize(st->stream) != hipSuccess) return 1;
    return 0;
}

}  // namespace

extern "C" {

int yk_rccl_get_unique_id(void* out128) {
    try {
        ncclUniqueId_t id;
        int rc = rccl().GetUniqueId(&id);
        if (rc != 0) return 1;
        std::memcpy(out128, &id, sizeof(id));
        return 0;
    } catch (...) { return 1; }
}

int yk_env_init_rccl(yk_env_h e, const void* id128, int rank, int nranks) {
    try {
        if (!e) return 1;
        e->env->set_ranks(rank, nranks);
        ncclUniqueId_t id;
        std::memcpy(&id, id128, sizeof(id));
        auto* st = new RcclState;
        auto fail = [&](const char* what, const char* why) {
            fprintf(stderr, "yk_env_init_rccl: %s failed%s%s\n", what, why ? ": " : "", why ? why : "");
            if (st->dscalar) (void)hipFree(st->dscalar);
            if (st->stream) (void)hipStreamDestroy(st->stream);
            delete st;
            return 1;
        };
        if (hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate", nullptr);
        if (hipMalloc(&st->dscalar, sizeof(long long)) != hipSuccess) return fail("hipMalloc", nullptr);
        int rc = rccl().CommInitRank(&st->comm, nranks, id, rank);
        if (rc != 0) return fail("ncclCommInitRank", rccl().GetErrorString(rc));
        e->env->drop_transport();               // an earlier built-in transport
        e->env->exch_start = rccl_start;
        e->env->exch_wait = rccl_wait;
        e->env->allreduce = rccl_allreduce;
        e->env->user = st;
        e->env->user_free = [](void* p) {
            RcclState* s = static_cast<RcclState*>(p);
            // (the communicator itself is left to process exit: envs are usually released while the host's own RCCL
            //  user -- torch.distributed -- is shutting down, and ncclCommDestroy is a collective that can block there)
            if (s->dscalar) (void)hipFree(s->dscalar);
            if (s->stream) (void)hipStreamDestroy(s->stream);
            delete s;
        };
        return 0;
    } catch (...) { return 1; }
}

}  // extern "C"
