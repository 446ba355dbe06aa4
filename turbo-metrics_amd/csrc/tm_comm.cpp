// tm_comm.cpp -- libturbometrics_rccl.so: the one collective of the frame-pair path (include/turbo_metrics_comm.h) over RCCL.
// Host code only (HIP runtime + RCCL calls); built with hipcc for its include paths and linked against librccl.
// 16 KB of scores per 2 048 frames: the reduce is bound by latency, not by the per-link bandwidth of xGMI -- one call, no bucketing.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/turbo_metrics_comm.h"

static_assert(TM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

namespace {
thread_local char g_err[320] = "";
int fail_hip(hipError_t e, const char *what) { snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e)); return 3; }
int fail_nccl(ncclResult_t r, const char *what) { snprintf(g_err, sizeof g_err, "%s: %s", what, ncclGetErrorString(r)); return 4; }
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail_hip(e_, #call); } while (0)
#define NCCLCHK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail_nccl(r_, #call); } while (0)
} // namespace

struct tm_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int device = 0, rank = 0, n_ranks = 0;
    double *d_send = nullptr, *d_recv = nullptr;
    size_t cap = 0; // doubles each buffer holds
};

extern "C" {

const char *tm_comm_last_error(void) { return g_err; }

int tm_comm_get_unique_id(void *id128)
{
    if (!id128) { snprintf(g_err, sizeof g_err, "tm_comm_get_unique_id: null pointer"); return 1; }
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int tm_comm_init(tm_comm **out, int n_ranks, int rank, const void *id128)
{
    if (!out || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) { snprintf(g_err, sizeof g_err, "tm_comm_init: invalid argument"); return 1; }
    *out = nullptr;
    tm_comm *c = new (std::nothrow) tm_comm();
    if (!c) { snprintf(g_err, sizeof g_err, "tm_comm_init: out of memory"); return 2; }
    c->rank = rank; c->n_ranks = n_ranks;
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    // (no stream of its own while the communicator idles: the runtime maps streams onto a handful of hardware queues, and one more stream in
    // the process put the engine's second upload stream on a shared queue -- the CLI's uploads ran at half their rate with the communicator
    // merely existing, profiles/r06l_ranks_time.log.  The reduce makes its stream and gives it back.)
    hipError_t he = hipGetDevice(&c->device);
    if (he != hipSuccess) { const int rc = fail_hip(he, "hipGetDevice"); tm_comm_destroy(c); return rc; }
    const ncclResult_t r = ncclCommInitRank(&c->comm, n_ranks, id, rank);
    if (r != ncclSuccess) { const int rc = fail_nccl(r, "ncclCommInitRank"); c->comm = nullptr; tm_comm_destroy(c); return rc; }
    *out = c;
    return 0;
}

int tm_comm_reduce_sum_f64(tm_comm *c, double *v, size_t n, int root)
{
    if (!c || !v || root < 0 || root >= c->n_ranks) { snprintf(g_err, sizeof g_err, "tm_comm_reduce_sum_f64: invalid argument"); return 1; }
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    if (n > c->cap) {
        if (c->d_send) (void)hipFree(c->d_send);
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_send = c->d_recv = nullptr; c->cap = 0;
        HIPCHK(hipMalloc((void **)&c->d_send, n * sizeof(double)));
        HIPCHK(hipMalloc((void **)&c->d_recv, n * sizeof(double)));
        c->cap = n;
    }
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    int rc = 0;
    do {
        hipError_t he = hipMemcpyAsync(c->d_send, v, n * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (he != hipSuccess) { rc = fail_hip(he, "hipMemcpyAsync"); break; }
        const ncclResult_t nr = ncclReduce(c->d_send, c->d_recv, n, ncclDouble, ncclSum, root, c->comm, c->stream);
        if (nr != ncclSuccess) { rc = fail_nccl(nr, "ncclReduce"); break; }
        if (c->rank == root && (he = hipMemcpyAsync(v, c->d_recv, n * sizeof(double), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) { rc = fail_hip(he, "hipMemcpyAsync"); break; }
        if ((he = hipStreamSynchronize(c->stream)) != hipSuccess) rc = fail_hip(he, "hipStreamSynchronize");
    } while (0);
    (void)hipStreamDestroy(c->stream);
    c->stream = nullptr;
    return rc;
}

void tm_comm_destroy(tm_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

} // extern "C"
