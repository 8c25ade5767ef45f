// comm.hip -- the gradient exchange of data-parallel training behind C entry points (SURVEY 8 row a9 / boundary list).
//
// Replaces what `accelerate` / DistributedDataParallel do for the reference's training script (train_vit_decorr.py:68-78:
// all-reduce of every .grad across ranks): one in-place all-reduce (sum or average) of a contiguous gradient range -- the flat
// buffer parallel.FlatGradSink owns, or a chunk of it -- over RCCL on the caller's stream.  The Python package keeps using
// torch.distributed (whose "nccl" backend IS RCCL on ROCm) for this; these entry points serve hosts that bind libvitk directly.
// librccl is opened at first use (dlopen) so that libvitk.so has no link-time dependency on it.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <string>

// The handful of RCCL (= NCCL API) declarations used here, stated locally so that libvitk builds on hosts without the RCCL
// headers (the library is dlopen'ed, never linked).  Values are those of rccl.h (NCCL ABI: stable across 2.x).
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0, ncclAvg = 4 } ncclRedOp_t;
typedef enum { ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat = 7, ncclBfloat16 = 9 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
const char* ncclGetErrorString(ncclResult_t result);
}

namespace {

struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) init_rank = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclCommDestroy) destroy = nullptr;
    decltype(&ncclGetErrorString) err = nullptr;
    bool ok = false;
    std::string why;        // dlerror() text captured ONCE where dlopen failed (a second dlerror() call returns NULL)
};

const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
            const char* e = dlerror();
            r.why = e ? e : "dlopen failed";
        }
        if (!r.h) return;
        r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.h, "ncclGetUniqueId");
        r.init_rank = (decltype(r.init_rank))dlsym(r.h, "ncclCommInitRank");
        r.all_reduce = (decltype(r.all_reduce))dlsym(r.h, "ncclAllReduce");
        r.destroy = (decltype(r.destroy))dlsym(r.h, "ncclCommDestroy");
        r.err = (decltype(r.err))dlsym(r.h, "ncclGetErrorString");
        r.ok = r.get_unique_id && r.init_rank && r.all_reduce && r.destroy && r.err;
        if (!r.ok) r.why = "symbols missing";
    });
    return r;
}

struct Comm { ncclComm_t c; int rank, world; };

}  // namespace

#define VITK_RCCL_OR_FAIL(R) do { if (!(R).ok) VITK_FAIL(VITK_E_UNAVAILABLE, "vitk_comm: librccl.so.1 could not be opened (%s)", (R).why.c_str()); } while (0)
#define VITK_RCCL_CALL(R, expr, what) do { const ncclResult_t rc__ = (expr); if (rc__ != ncclSuccess) VITK_FAIL(VITK_E_COMM, "%s: %s", what, (R).err(rc__)); } while (0)

extern "C" int vitk_comm_unique_id(void* out128) {
    if (!out128) VITK_FAIL(VITK_E_ARG, "comm_unique_id: null pointer");
    const Rccl& r = rccl();
    VITK_RCCL_OR_FAIL(r);
    static_assert(sizeof(ncclUniqueId) == VITK_COMM_ID_BYTES, "ncclUniqueId size");
    VITK_RCCL_CALL(r, r.get_unique_id((ncclUniqueId*)out128), "ncclGetUniqueId");
    return 0;
}

extern "C" int vitk_comm_init(const void* id128, int rank, int world, vitk_comm_t* out) {
    if (!id128 || !out) VITK_FAIL(VITK_E_ARG, "comm_init: null pointer");
    if (world < 1 || rank < 0 || rank >= world) VITK_FAIL(VITK_E_ARG, "comm_init: rank %d of %d", rank, world);
    const Rccl& r = rccl();
    VITK_RCCL_OR_FAIL(r);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm* c = new Comm{nullptr, rank, world};
    const ncclResult_t rc = r.init_rank(&c->c, world, id, rank);
    if (rc != ncclSuccess) { delete c; VITK_FAIL(VITK_E_COMM, "ncclCommInitRank: %s", r.err(rc)); }
    *out = c;
    return 0;
}

extern "C" int vitk_comm_allreduce(vitk_comm_t comm, void* buf, int64_t n, int dt, int average, void* stream) {
    if (!comm || !buf) VITK_FAIL(VITK_E_ARG, "comm_allreduce: null pointer");
    if (n <= 0) return 0;
    const Rccl& r = rccl();
    VITK_RCCL_OR_FAIL(r);
    ncclDataType_t t;
    if (dt == VITK_F32) t = ncclFloat32;
    else if (dt == VITK_BF16) t = (vitk_half_type() == VITK_F16) ? ncclFloat16 : ncclBfloat16;      // the library's 16-bit type
    else VITK_FAIL(VITK_E_DTYPE, "comm_allreduce: bad dtype tag %d", dt);
    Comm* c = (Comm*)comm;
    VITK_RCCL_CALL(r, r.all_reduce(buf, buf, (size_t)n, t, average ? ncclAvg : ncclSum, c->c, (hipStream_t)stream), "ncclAllReduce");
    return 0;
}

extern "C" int vitk_comm_destroy(vitk_comm_t comm) {
    if (!comm) return 0;
    const Rccl& r = rccl();
    VITK_RCCL_OR_FAIL(r);
    Comm* c = (Comm*)comm;
    const ncclResult_t rc = r.destroy(c->c);
    delete c;
    if (rc != ncclSuccess) VITK_FAIL(VITK_E_COMM, "ncclCommDestroy: %s", r.err(rc));
    return 0;
}
