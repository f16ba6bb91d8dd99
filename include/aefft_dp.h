/* aefft_dp.h -- the data-parallel training step over RCCL, as a C ABI (libaefft_dp.so = libaefft.so + librccl).
 *
 * SURVEY.md section 8e (the reference is single-GPU: no counterpart; the loop body being distributed is
 * source/fft_backproplib.cu:1446-1465).  One process per GPU; frames shard, weights replicate.  Per step and rank:
 *
 *     aefft_net_step_grad  ->  ncclAllReduce(SUM) of aefft_net_grad_buffer on aefft_stream(ctx)  ->  aefft_net_step_apply(1/world)
 *
 * all three enqueued by ONE host call, stream-ordered, no host synchronisation and no Python between the two halves
 * (examples/rccl_step.cpp is the same sequence written out against rccl.h; autoencoder-fft_amd/dp.py the same over torch.distributed).
 * libaefft.so itself does not link RCCL: a host that brings its own communicator uses aefft_stream + aefft_net_grad_buffer directly.
 *
 * Conventions as include/aefft.h: plain pointers and sizes, "_d" = device pointer, every function returns AEFFT_OK (0) or an error code
 * (RCCL failures: AEFFT_EHIP, text through aefft_dp_last_error).
 */
#ifndef AEFFT_DP_H
#define AEFFT_DP_H

#include "aefft.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct aefft_dp aefft_dp;

enum { AEFFT_DP_ID_BYTES = 128 };      /* sizeof(ncclUniqueId) */

/* rank 0: a fresh ncclUniqueId into id_h[AEFFT_DP_ID_BYTES]; the host hands the bytes to the other ranks (a file, a socket, MPI,
 * torch.distributed.broadcast -- bench.py does the latter) before every rank calls aefft_dp_create with them. */
int aefft_dp_unique_id(void* id_h);
/* Collective over the `world` ranks: the communicator for `net` (whose context must own its device).  world = 1 is allowed (the
 * all-reduce then runs through the same library path on one rank). */
int aefft_dp_create(aefft_net* net, aefft_ctx* ctx, int rank, int world, const void* id_h, aefft_dp** out);
void aefft_dp_destroy(aefft_dp* dp);
const char* aefft_dp_last_error(const aefft_dp* dp);

/* One data-parallel step (arguments as aefft_net_step_grad / aefft_net_step_apply; grad_scale is 1/world).  mse_d nullable. */
int aefft_dp_step(aefft_dp* dp, const float* frames_d, float* recon_d, float del0, int maxdiff, int sym, float* mse_d);
/* nsteps steps on the same buffers in one host call (a resident-batch training loop; bench.py's timed region). */
int aefft_dp_run(aefft_dp* dp, const float* frames_d, float* recon_d, float del0, int maxdiff, int sym, int nsteps);
/* nsteps steps with events on the library's stream around the gradient half, the collective and the update half: mean milliseconds
 * of each into phase_ms[3], and the mean HOST time per step (microseconds spent enqueueing one step, no synchronisation inside) into
 * *host_us.  Blocks until done. */
int aefft_dp_profile(aefft_dp* dp, const float* frames_d, float* recon_d, float del0, int maxdiff, int sym, int nsteps,
                     double* phase_ms, double* host_us);
/* global-batch post-update MSE per pair of the LAST step: one small all-reduce of the packed buffer's tail, times 1/world, to
 * mse_h[L] (host).  Blocks. */
int aefft_dp_flush_mse(aefft_dp* dp, float* mse_h);
/* bytes of one gradient all-reduce (the packed buffer: [dck | dfk | db | dp] per pair | one MSE float per pair) */
size_t aefft_dp_allreduce_bytes(const aefft_dp* dp);
/* 1 when every rank holds bit-identical weights (a checksum of every pair's c, b, f, p all-reduced as min and max), 0 when not,
 * negative on error.  Blocks. */
int aefft_dp_replicas_agree(aefft_dp* dp);

#ifdef __cplusplus
}
#endif
#endif /* AEFFT_DP_H */
