// Data-parallel gradient exchange over RCCL: vag_comm_{unique_id, init, allreduce, destroy}.
// One communicator per process (one process per GPU); the all-reduce is an in-place fp32 sum of a slice of the flat gradient
// buffer, enqueued on the caller's stream (RCCL work on a stream can be captured into a HIP graph with the step's kernels).
// librccl is resolved at first use with dlopen: a single-GPU user of libvagnmt.so never loads it, and inside a torch process
// the copy torch already mapped is the one that is found (same soname), never a second RCCL.
#include "kernels.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <new>

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    bool ok = false;
};

RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {          // a copy that is already mapped (torch's) wins
            a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (a.handle) break;
        }
        for (int i = 0; !a.handle && i < 3; ++i) a.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!a.handle) return a;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.handle, "ncclAllReduce"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
        a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(a.handle, "ncclCommCount"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.CommDestroy && a.CommCount;
        return a;
    }();
    return api;
}

// ncclResult_t values are small positive integers like hipError_t; keep them apart from those in the return value
constexpr int VAG_RCCL_BASE = 10000;
inline int rc_of(ncclResult_t r) { return r == ncclSuccess ? VAG_OK : VAG_RCCL_BASE + (int)r; }

}  // namespace

struct vag_comm_s {
    ncclComm_t comm;
    int nranks, rank;
};

extern "C" {

int vag_comm_unique_id(void* id) {
    VAG_CHECK_ARG(id != nullptr);
    static_assert(sizeof(ncclUniqueId) == VAG_COMM_ID_BYTES, "vag_nmt.h: VAG_COMM_ID_BYTES");
    if (!rccl().ok) return VAG_ENOSYS;
    ncclUniqueId u;
    const int rc = rc_of(rccl().GetUniqueId(&u));
    if (rc == VAG_OK) std::memcpy(id, &u, sizeof(u));
    return rc;
}

int vag_comm_init(vag_comm_t* comm, int nranks, int rank, const void* id) {
    VAG_CHECK_ARG(comm && id && nranks >= 1 && rank >= 0 && rank < nranks);
    if (!rccl().ok) return VAG_ENOSYS;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    const int rc = rc_of(rccl().CommInitRank(&c, nranks, u, rank));       // collective over the ranks; uses the current device
    if (rc != VAG_OK) return rc;
    vag_comm_s* h = new (std::nothrow) vag_comm_s{c, nranks, rank};
    if (!h) { rccl().CommDestroy(c); return VAG_EINVAL; }
    *comm = h;
    return VAG_OK;
}

int vag_comm_allreduce(vag_comm_t comm, float* buf, int64_t n, vag_stream_t stream) {
    VAG_CHECK_ARG(comm && comm->comm && n >= 0 && (buf || n == 0));
    if (n == 0) return VAG_OK;
    return rc_of(rccl().AllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, comm->comm, reinterpret_cast<hipStream_t>(stream)));
}

int vag_comm_size(vag_comm_t comm) {
    VAG_CHECK_ARG(comm && comm->comm);
    return comm->nranks;
}

int vag_comm_destroy(vag_comm_t comm) {
    if (!comm) return VAG_OK;
    int rc = VAG_OK;
    if (comm->comm) rc = rc_of(rccl().CommDestroy(comm->comm));
    delete comm;
    return rc;
}

}  // extern "C"
