// Data-parallel learner replicas: the gradient exchange of SURVEY.md §8(b)/(e) behind the C-ABI (include/sf_hip.h).
//
// One RCCL communicator per rank (created once, on the rank's current device), one in-place SUM all-reduce of the flat
// fp32 gradient (or of a tail/head slice of it — the two-bucket overlap of algo/learning/learner.py) per SGD step,
// enqueued on the stream the caller passes; nothing here synchronises the device.  The reference has a single learner
// per policy (algo/utils/shared_buffers.py:26-32), so there is no reference call site to cite: this is the entry point
// a non-torch host binds instead of torch.distributed.
//
// librccl is resolved at first use with dlopen (SONAME librccl.so.1): a process that already loaded RCCL (PyTorch
// bundles one) keeps using that copy, and libsf_hip.so has no load-time dependency on it (single-GPU hosts never touch
// it).  Host code only; built with hipcc like the rest for one toolchain.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "sf_common.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

template <class F>
bool bind(F &fn, const char *name) {
    fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
    return fn != nullptr;
}

int rccl_load() {
    if (g_rccl.handle) return SF_OK;
    static const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // a copy the process already holds (PyTorch's) wins
        if (h) break;
    }
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        snprintf(sf_err_buf, sizeof(sf_err_buf), "sf_dp: cannot load librccl.so.1: %s", dlerror());
        return SF_ERR_LAUNCH;
    }
    g_rccl.handle = h;
    const bool ok = bind(g_rccl.GetUniqueId, "ncclGetUniqueId") && bind(g_rccl.CommInitRank, "ncclCommInitRank") &&
                    bind(g_rccl.CommDestroy, "ncclCommDestroy") && bind(g_rccl.CommCount, "ncclCommCount") &&
                    bind(g_rccl.CommUserRank, "ncclCommUserRank") && bind(g_rccl.AllReduce, "ncclAllReduce") &&
                    bind(g_rccl.Broadcast, "ncclBroadcast") && bind(g_rccl.GetErrorString, "ncclGetErrorString");
    if (!ok) {
        g_rccl.handle = nullptr;
        snprintf(sf_err_buf, sizeof(sf_err_buf), "sf_dp: librccl lacks an expected symbol: %s", dlerror());
        return SF_ERR_LAUNCH;
    }
    return SF_OK;
}

int rccl_status(ncclResult_t r, const char *what) {
    if (r == ncclSuccess) return SF_OK;
    snprintf(sf_err_buf, sizeof(sf_err_buf), "%s: RCCL error %d (%s)", what, (int)r,
             g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return SF_ERR_LAUNCH;
}

}  // namespace

static_assert(SF_DP_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "sf_hip.h mirrors the RCCL id size");

extern "C" int sf_dp_unique_id(void *out_id) {
    SF_REQUIRE(out_id, "sf_dp_unique_id: NULL output");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    rc = rccl_status(g_rccl.GetUniqueId(&id), "sf_dp_unique_id");
    if (rc) return rc;
    memcpy(out_id, id.internal, NCCL_UNIQUE_ID_BYTES);
    return SF_OK;
}

extern "C" int sf_dp_comm_create(const void *id_bytes, int nranks, int rank, void **comm_out) {
    SF_REQUIRE(id_bytes && comm_out && nranks >= 1 && rank >= 0 && rank < nranks,
               "sf_dp_comm_create: bad args (nranks=%d rank=%d)", nranks, rank);
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    rc = rccl_status(g_rccl.CommInitRank(&comm, nranks, id, rank), "sf_dp_comm_create");  // collective over the ranks
    if (rc) return rc;
    *comm_out = comm;
    return SF_OK;
}

extern "C" int sf_dp_comm_destroy(void *comm) {
    SF_REQUIRE(comm, "sf_dp_comm_destroy: NULL communicator");
    int rc = rccl_load();
    if (rc) return rc;
    return rccl_status(g_rccl.CommDestroy((ncclComm_t)comm), "sf_dp_comm_destroy");
}

extern "C" int sf_dp_comm_info(void *comm, int *nranks, int *rank) {
    SF_REQUIRE(comm && nranks && rank, "sf_dp_comm_info: bad args");
    int rc = rccl_load();
    if (rc) return rc;
    rc = rccl_status(g_rccl.CommCount((ncclComm_t)comm, nranks), "sf_dp_comm_info");
    if (rc) return rc;
    return rccl_status(g_rccl.CommUserRank((ncclComm_t)comm, rank), "sf_dp_comm_info");
}

extern "C" int sf_allreduce_grads(void *comm, float *grads, int64_t n, void *stream) {
    SF_REQUIRE(comm && grads && n > 0, "sf_allreduce_grads: bad args");
    int rc = rccl_load();
    if (rc) return rc;
    return rccl_status(g_rccl.AllReduce(grads, grads, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm,
                                        (hipStream_t)stream), "sf_allreduce_grads");
}

extern "C" int sf_dp_allreduce_f64(void *comm, double *buf, int64_t n, int op, void *stream) {
    SF_REQUIRE(comm && buf && n > 0 && (op == 0 || op == 1), "sf_dp_allreduce_f64: bad args (op: 0 = sum, 1 = max)");
    int rc = rccl_load();
    if (rc) return rc;
    return rccl_status(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat64, op == 0 ? ncclSum : ncclMax, (ncclComm_t)comm,
                                        (hipStream_t)stream), "sf_dp_allreduce_f64");
}

extern "C" int sf_dp_broadcast(void *comm, void *buf, int64_t nbytes, int root, void *stream) {
    SF_REQUIRE(comm && buf && nbytes > 0 && root >= 0, "sf_dp_broadcast: bad args");
    int rc = rccl_load();
    if (rc) return rc;
    return rccl_status(g_rccl.Broadcast(buf, buf, (size_t)nbytes, ncclUint8, root, (ncclComm_t)comm, (hipStream_t)stream),
                       "sf_dp_broadcast");
}
