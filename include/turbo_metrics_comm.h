/*
 * turbo_metrics_comm.h -- C ABI of libturbometrics_rccl.so: the ONE collective of the frame-pair path.
 *
 * Frame pairs shard embarrassingly across the GPUs of a node (one process per GPU); the only exchange is a single
 * reduce(sum, f64) of the zero-padded per-frame score vector to rank 0, over RCCL / xGMI.  The reference has no counterpart -- it is
 * single-GPU, device 0 hard-coded (crates/turbo-metrics/src/lib.rs:442) --; a multi-GPU `turbo-metrics` would add exactly this after its
 * compute_all (lib.rs:362-433).  Kept apart from libturbometrics_hip.so so that the engine library does not depend on RCCL (573 MB):
 * the CLI loads this one at run time, and only for `--ranks N` (host/ranks.cpp).  A Rust binder declares the same five functions in an
 * `extern "C"` block.
 *
 * Plain C types.  Every function returns 0 on success and a non-zero code otherwise; tm_comm_last_error() gives the text (the
 * RCCL / HIP error string and the call that failed).  A communicator belongs to the device that was current when it was created.
 */
#ifndef TURBO_METRICS_COMM_H
#define TURBO_METRICS_COMM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tm_comm tm_comm;

enum { TM_COMM_ID_BYTES = 128 }; /* == NCCL_UNIQUE_ID_BYTES */

/* rank 0: a fresh id for one communicator (ncclGetUniqueId); the caller carries the 128 bytes to the other ranks (the CLI: over the
 * launcher's pipes) */
int tm_comm_get_unique_id(void *id128);
/* every rank, after binding to its device (tm_init): ncclCommInitRank + a stream and staging buffers of its own */
int tm_comm_init(tm_comm **out, int n_ranks, int rank, const void *id128);
/* ONE ncclReduce(sum, ncclDouble) of `n` doubles to `root`: host vector in, on the root the sum over the ranks out (the other ranks'
 * vectors are left as they were).  Blocks until the result is in `v`. */
int tm_comm_reduce_sum_f64(tm_comm *c, double *v, size_t n, int root);
void tm_comm_destroy(tm_comm *c);
const char *tm_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
