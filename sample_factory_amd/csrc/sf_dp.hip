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

// ============================================================================================== one-shot small-bucket exchange
// "one-shot / direct for <= 8 MB ... never ring".  The conv bucket of the Nature-CNN gradient is 0.31 MB and
// the scalar buckets are 24 ... 400 bytes: a ring over 8 ranks costs them 14 latency-bound hops on the critical path (the
// conv bucket completes LAST in the backward pass, nothing is left to hide it behind).  Here every rank owns a MAILBOX in
// its own HBM (uncached allocation, exported with hipIpcGetMemHandle and mapped by every peer over xGMI): one kernel per
// rank and call
//   1. copies the bucket into its mailbox slot, fences at system scope and raises one flag per block;
//   2. waits for the same flag of every peer (rank order 0 .. W-1, own copy included) and adds the peers' slots IN RANK
//      ORDER — every rank computes the same sum in the same order: the results are bit-identical across the ranks;
//   3. writes the sum back in place.
// One hop, W-1 concurrent peer reads per rank.  Blocks are independent (block b owns elements [b*chunk, (b+1)*chunk) and
// flag b), so no grid-wide barrier exists.  Two slots alternate by the call's sequence number: slot s is rewritten at call
// q+2 only after this rank has seen every peer's flag of call q+1, which the peer raised after it had finished reading call
// q (stream order) — no extra acknowledgement round.  Every spin is bounded (wall clock): a peer that never arrives sets the
// context's error word instead of hanging the queue, and the next call returns SF_ERR_LAUNCH.
// No reference line: the reference has one learner per policy (algo/utils/shared_buffers.py:26-32).
namespace {

constexpr int OS_MAX_RANKS = 16, OS_MAX_BLOCKS = 32, OS_THREADS = 256;

struct OneShot {
    int nranks = 0, rank = 0;
    int64_t cap = 0;            // bytes per data slot
    char *mine = nullptr;       // own mailbox: [2][OS_MAX_BLOCKS] u32 flags (256 B), then 2 slots of `cap` bytes
    char *peer[OS_MAX_RANKS] = {};
    bool opened[OS_MAX_RANKS] = {};
    uint32_t *err = nullptr;    // device word: != 0 after a timed-out wait
    uint32_t seq = 0;
    bool uncached = false;
};
constexpr int64_t OS_HDR = 256;

struct OneShotArgs {
    char *box[OS_MAX_RANKS];
    int nranks, rank;
    int64_t cap;
    uint32_t seq;
    uint32_t *err;
};

__device__ __forceinline__ bool os_wait_flag(const uint32_t *flag, uint32_t seq, uint32_t *err) {
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 500000000ull) {  // 5 s: a peer never launched its half of the exchange
            atomicExch(err, 1u);
            return false;
        }
    }
    return true;
}

// T = float (SUM) or double (op 0 = SUM, 1 = MAX); n elements, in place
template <typename T>
__global__ __launch_bounds__(OS_THREADS) void k_oneshot_allreduce(OneShotArgs a, T *__restrict__ buf, int64_t n, int op) {
    __shared__ int ok_s;
    const int b = blockIdx.x, nb = gridDim.x;
    const int64_t chunk = ((n + nb - 1) / nb + 3) & ~(int64_t)3;
    const int64_t lo = (int64_t)b * chunk, hi = lo + chunk < n ? lo + chunk : n;
    const uint32_t slot = a.seq & 1u;
    char *own = a.box[a.rank];
    T *mine = reinterpret_cast<T *>(own + OS_HDR + (int64_t)slot * a.cap);
    for (int64_t i = lo + threadIdx.x; i < hi; i += OS_THREADS) __builtin_nontemporal_store(buf[i], &mine[i]);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(reinterpret_cast<uint32_t *>(own) + slot * OS_MAX_BLOCKS + b, a.seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    for (int r = 0; r < a.nranks; ++r) {
        if (threadIdx.x == 0)
            ok_s = os_wait_flag(reinterpret_cast<const uint32_t *>(a.box[r]) + slot * OS_MAX_BLOCKS + b, a.seq, a.err) ? 1 : 0;
        __syncthreads();
        if (!ok_s) return;
        __threadfence_system();  // acquire for the whole block: the peer's slot bytes are read after its flag
        const T *src = reinterpret_cast<const T *>(a.box[r] + OS_HDR + (int64_t)slot * a.cap);
        for (int64_t i = lo + threadIdx.x; i < hi; i += OS_THREADS) {
            const T v = __builtin_nontemporal_load(&src[i]);
            if (r == 0) buf[i] = v;
            else buf[i] = op == 1 ? (buf[i] > v ? buf[i] : v) : buf[i] + v;
        }
        __syncthreads();
    }
}

int os_check(OneShot *c, const char *what) {
    uint32_t e = 0;
    if (hipMemcpy(&e, c->err, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return SF_OK;  // (stream busy: checked later)
    if (e) {
        snprintf(sf_err_buf, sizeof(sf_err_buf), "%s: an earlier one-shot exchange timed out waiting for a peer (5 s)", what);
        return SF_ERR_LAUNCH;
    }
    return SF_OK;
}

}  // namespace

static_assert(SF_DP_IPC_HANDLE_BYTES == sizeof(hipIpcMemHandle_t), "sf_hip.h mirrors the HIP IPC handle size");

extern "C" int sf_dp_oneshot_create(int nranks, int rank, int64_t max_bytes, void **ctx_out, void *handle_out) {
    SF_REQUIRE(ctx_out && handle_out && nranks >= 1 && nranks <= OS_MAX_RANKS && rank >= 0 && rank < nranks && max_bytes > 0,
               "sf_dp_oneshot_create: bad args (nranks=%d rank=%d max_bytes=%lld; at most %d ranks)", nranks, rank,
               (long long)max_bytes, OS_MAX_RANKS);
    OneShot *c = new OneShot();
    c->nranks = nranks;
    c->rank = rank;
    c->cap = (max_bytes + 255) & ~(int64_t)255;
    const size_t total = (size_t)(OS_HDR + 2 * c->cap);
    void *p = nullptr;
    // fine-grained (uncached) memory: peer reads over xGMI and the flags are coherent without cache maintenance; a plain
    // allocation + the system-scope fences of the kernel is the fallback
    if (hipExtMallocWithFlags(&p, total, hipDeviceMallocUncached) == hipSuccess) c->uncached = true;
    else {
        (void)hipGetLastError();
        int rc = sf_hip_status(hipMalloc(&p, total), "sf_dp_oneshot_create: mailbox");
        if (rc) { delete c; return rc; }
    }
    c->mine = (char *)p;
    int rc = sf_hip_status(hipMemset(p, 0, total), "sf_dp_oneshot_create: memset");
    if (!rc) rc = sf_hip_status(hipMalloc((void **)&c->err, sizeof(uint32_t)), "sf_dp_oneshot_create: error word");
    if (!rc) rc = sf_hip_status(hipMemset(c->err, 0, sizeof(uint32_t)), "sf_dp_oneshot_create: memset");
    hipIpcMemHandle_t h;
    if (!rc) rc = sf_hip_status(hipIpcGetMemHandle(&h, p), "sf_dp_oneshot_create: hipIpcGetMemHandle (HSA_ENABLE_IPC_MODE_LEGACY=0?)");
    if (rc) { (void)hipFree(p); if (c->err) (void)hipFree(c->err); delete c; return rc; }
    memcpy(handle_out, &h, sizeof(h));
    c->peer[rank] = c->mine;
    *ctx_out = c;
    return SF_OK;
}

extern "C" int sf_dp_oneshot_connect(void *ctx, const void *all_handles) {
    OneShot *c = (OneShot *)ctx;
    SF_REQUIRE(c && all_handles, "sf_dp_oneshot_connect: bad args");
    for (int r = 0; r < c->nranks; ++r) {
        if (r == c->rank || c->opened[r]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)all_handles + (size_t)r * sizeof(h), sizeof(h));
        void *p = nullptr;
        int rc = sf_hip_status(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "sf_dp_oneshot_connect: hipIpcOpenMemHandle");
        if (rc) return rc;
        c->peer[r] = (char *)p;
        c->opened[r] = true;
    }
    return SF_OK;
}

static int oneshot_launch(OneShot *c, void *buf, int64_t n, int elem, int op, void *stream) {
    SF_REQUIRE(c && buf && n > 0 && n * elem <= c->cap, "sf_dp_oneshot_allreduce: %lld bytes exceed the mailbox slot (%lld)",
               (long long)(n * elem), (long long)(c ? c->cap : 0));
    for (int r = 0; r < c->nranks; ++r) SF_REQUIRE(c->peer[r], "sf_dp_oneshot_allreduce: rank %d is not connected", r);
    OneShotArgs a;
    for (int r = 0; r < OS_MAX_RANKS; ++r) a.box[r] = c->peer[r];
    a.nranks = c->nranks;
    a.rank = c->rank;
    a.cap = c->cap;
    a.seq = ++c->seq;
    a.err = c->err;
    int nb = (int)((n * elem + 16383) / 16384);  // 16 KiB per block: the 0.31 MB conv bucket runs on 20 blocks
    nb = nb < 1 ? 1 : (nb > OS_MAX_BLOCKS ? OS_MAX_BLOCKS : nb);
    if (elem == 4) k_oneshot_allreduce<float><<<dim3(nb), dim3(OS_THREADS), 0, (hipStream_t)stream>>>(a, (float *)buf, n, op);
    else k_oneshot_allreduce<double><<<dim3(nb), dim3(OS_THREADS), 0, (hipStream_t)stream>>>(a, (double *)buf, n, op);
    return sf_launch_status("sf_dp_oneshot_allreduce");
}

extern "C" int sf_dp_oneshot_allreduce_f32(void *ctx, float *buf, int64_t n, void *stream) {
    return oneshot_launch((OneShot *)ctx, buf, n, 4, 0, stream);
}

extern "C" int sf_dp_oneshot_allreduce_f64(void *ctx, double *buf, int64_t n, int op, void *stream) {
    SF_REQUIRE(op == 0 || op == 1, "sf_dp_oneshot_allreduce_f64: op 0 = sum, 1 = max");
    return oneshot_launch((OneShot *)ctx, buf, n, 8, op, stream);
}

/* 0: no exchange of this context has timed out so far (reads one device word: synchronises with the copy, not the stream) */
extern "C" int sf_dp_oneshot_status(void *ctx) {
    SF_REQUIRE(ctx, "sf_dp_oneshot_status: NULL context");
    return os_check((OneShot *)ctx, "sf_dp_oneshot_status");
}

extern "C" int sf_dp_oneshot_destroy(void *ctx) {
    OneShot *c = (OneShot *)ctx;
    SF_REQUIRE(c, "sf_dp_oneshot_destroy: NULL context");
    for (int r = 0; r < c->nranks; ++r)
        if (c->opened[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    (void)hipFree(c->mine);
    (void)hipFree(c->err);
    delete c;
    return SF_OK;
}
