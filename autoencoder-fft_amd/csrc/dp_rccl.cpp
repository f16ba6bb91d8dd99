// libaefft_dp.so: the data-parallel training step over RCCL behind a C ABI (include/aefft_dp.h).  Host orchestration only:
//     aefft_net_step_grad -> ncclAllReduce(SUM) of the packed buffer on the library's stream -> aefft_net_step_apply(1/world)
// SURVEY.md section 8e; the loop body being distributed is fft_backproplib.cu:1446-1465.  No torch, no Python in the loop.
#include "../../include/aefft_dp.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static_assert(sizeof(ncclUniqueId) == AEFFT_DP_ID_BYTES, "ncclUniqueId size");

struct aefft_dp {
    aefft_net* net = nullptr;
    aefft_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t st = nullptr;
    int rank = 0, world = 1, L = 0;
    float* gbuf = nullptr; size_t gn = 0;       // packed buffer: gradients | one MSE float per pair
    float inv_world = 1.f;
    double* chk_d = nullptr;                    // replicas_agree scratch
    float* tail_d = nullptr;                    // flush_mse scratch
    std::string err;
};

static int dp_fail(aefft_dp* dp, int code, const std::string& what) { if (dp) dp->err = what; return code; }
#define DP_NCCL(dp, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return dp_fail(dp, AEFFT_EHIP, std::string(#call) + ": " + ncclGetErrorString(r_)); } while (0)
#define DP_HIP(dp, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return dp_fail(dp, AEFFT_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define DP_AE(dp, call) do { int a_ = (call); if (a_ != AEFFT_OK) return dp_fail(dp, a_, std::string(#call) + ": " + aefft_last_error((dp)->ctx)); } while (0)

extern "C" int aefft_dp_unique_id(void* id_h)
{
    if (!id_h) return AEFFT_EINVAL;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return AEFFT_EHIP;
    memcpy(id_h, &id, sizeof id);
    return AEFFT_OK;
}

extern "C" const char* aefft_dp_last_error(const aefft_dp* dp) { return dp ? dp->err.c_str() : "null handle"; }

extern "C" void aefft_dp_destroy(aefft_dp* dp)
{
    if (!dp) return;
    if (dp->st) (void)hipStreamSynchronize(dp->st);
    if (dp->comm) (void)ncclCommDestroy(dp->comm);
    if (dp->chk_d) (void)hipFree(dp->chk_d);
    if (dp->tail_d) (void)hipFree(dp->tail_d);
    delete dp;
}

extern "C" int aefft_dp_create(aefft_net* net, aefft_ctx* ctx, int rank, int world, const void* id_h, aefft_dp** out)
{
    if (!out) return AEFFT_EINVAL;
    *out = nullptr;
    if (!net || !ctx || !id_h || world < 1 || rank < 0 || rank >= world) return AEFFT_EINVAL;
    aefft_dp* dp = new aefft_dp();
    dp->net = net; dp->ctx = ctx; dp->rank = rank; dp->world = world; dp->inv_world = 1.0f / (float)world;
    dp->st = (hipStream_t)aefft_stream(ctx);
    dp->L = aefft_net_npairs(net);
    int rc = aefft_net_grad_buffer(net, &dp->gbuf, &dp->gn);
    if (rc != AEFFT_OK || dp->L <= 0) { delete dp; return rc != AEFFT_OK ? rc : AEFFT_EINVAL; }
    ncclUniqueId id;
    memcpy(&id, id_h, sizeof id);
    ncclResult_t r = ncclCommInitRank(&dp->comm, world, id, rank);
    if (r != ncclSuccess) { fprintf(stderr, "aefft_dp_create: ncclCommInitRank: %s\n", ncclGetErrorString(r)); delete dp; return AEFFT_EHIP; }
    if (hipMalloc(&dp->chk_d, sizeof(double) * 8 * (size_t)dp->L) != hipSuccess || hipMalloc(&dp->tail_d, sizeof(float) * (size_t)dp->L) != hipSuccess) {
        aefft_dp_destroy(dp);
        return AEFFT_ENOMEM;
    }
    *out = dp;
    return AEFFT_OK;
}

extern "C" size_t aefft_dp_allreduce_bytes(const aefft_dp* dp) { return dp ? dp->gn * sizeof(float) : 0; }

extern "C" int aefft_dp_step(aefft_dp* dp, const float* frames_d, float* recon_d, float del0, int maxdiff, int sym, float* mse_d)
{
    if (!dp) return AEFFT_EINVAL;
    DP_AE(dp, aefft_net_step_grad(dp->net, frames_d, recon_d));
    DP_NCCL(dp, ncclAllReduce(dp->gbuf, dp->gbuf, dp->gn, ncclFloat, ncclSum, dp->comm, dp->st));      // on the library's stream: no host sync
    DP_AE(dp, aefft_net_step_apply(dp->net, del0, maxdiff, sym, dp->inv_world, mse_d));
    return AEFFT_OK;
}

extern "C" int aefft_dp_run(aefft_dp* dp, const float* frames_d, float* recon_d, float del0, int maxdiff, int sym, int nsteps)
{
    if (!dp || nsteps < 0) return AEFFT_EINVAL;
    for (int i = 0; i < nsteps; ++i) {
        const int rc = aefft_dp_step(dp, frames_d, recon_d, del0, maxdiff, sym, nullptr);
        if (rc != AEFFT_OK) return rc;
    }
    return AEFFT_OK;
}

extern "C" int aefft_dp_profile(aefft_dp* dp, const float* frames_d, float* recon_d, float del0, int maxdiff, int sym, int nsteps,
                                double* phase_ms, double* host_us)
{
    if (!dp || nsteps <= 0 || !phase_ms) return AEFFT_EINVAL;
    std::vector<hipEvent_t> ev(4 * (size_t)nsteps);
    for (auto& e : ev) DP_HIP(dp, hipEventCreate(&e));
    DP_AE(dp, aefft_sync(dp->ctx));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nsteps; ++i) {
        hipEvent_t* e = &ev[4 * (size_t)i];
        DP_HIP(dp, hipEventRecord(e[0], dp->st));
        DP_AE(dp, aefft_net_step_grad(dp->net, frames_d, recon_d));
        DP_HIP(dp, hipEventRecord(e[1], dp->st));
        DP_NCCL(dp, ncclAllReduce(dp->gbuf, dp->gbuf, dp->gn, ncclFloat, ncclSum, dp->comm, dp->st));
        DP_HIP(dp, hipEventRecord(e[2], dp->st));
        DP_AE(dp, aefft_net_step_apply(dp->net, del0, maxdiff, sym, dp->inv_world, nullptr));
        DP_HIP(dp, hipEventRecord(e[3], dp->st));
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (host_us) *host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / nsteps;
    DP_AE(dp, aefft_sync(dp->ctx));
    phase_ms[0] = phase_ms[1] = phase_ms[2] = 0.0;
    for (int i = 0; i < nsteps; ++i)
        for (int k = 0; k < 3; ++k) {
            float ms = 0.f;
            DP_HIP(dp, hipEventElapsedTime(&ms, ev[4 * (size_t)i + k], ev[4 * (size_t)i + k + 1]));
            phase_ms[k] += ms / nsteps;
        }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return AEFFT_OK;
}

extern "C" int aefft_dp_flush_mse(aefft_dp* dp, float* mse_h)
{
    if (!dp || !mse_h) return AEFFT_EINVAL;
    DP_AE(dp, aefft_net_last_mse(dp->net, dp->tail_d));            // (a step_apply without an MSE output leaves the sums to the next step: formed now)
    DP_NCCL(dp, ncclAllReduce(dp->tail_d, dp->tail_d, (size_t)dp->L, ncclFloat, ncclSum, dp->comm, dp->st));
    DP_HIP(dp, hipMemcpyAsync(mse_h, dp->tail_d, sizeof(float) * (size_t)dp->L, hipMemcpyDeviceToHost, dp->st));
    DP_AE(dp, aefft_sync(dp->ctx));
    for (int l = 0; l < dp->L; ++l) mse_h[l] *= dp->inv_world;
    return AEFFT_OK;
}

extern "C" int aefft_dp_replicas_agree(aefft_dp* dp)
{
    if (!dp) return -AEFFT_EINVAL;
    const int L = dp->L;
    std::vector<double> sums(4 * (size_t)L), lo(4 * (size_t)L), hi(4 * (size_t)L);
    for (int l = 0; l < L; ++l) {
        int dD, dM, Nk, Nl;
        if (aefft_net_pair_shape(dp->net, l, &dD, &dM, &Nk, &Nl) != AEFFT_OK) return -AEFFT_EINVAL;
        const size_t nk = (size_t)dM * dD * Nk * Nl;
        std::vector<float> c(nk), f(nk), b((size_t)dM), p((size_t)dD);
        if (aefft_net_get_pair(dp->net, l, c.data(), b.data(), f.data(), p.data()) != AEFFT_OK) return -AEFFT_EHIP;
        // position-weighted sums: a permutation of equal values does not pass
        auto chk = [](const std::vector<float>& v) { double s = 0; for (size_t i = 0; i < v.size(); ++i) s += (double)v[i] * (double)(1 + i % 251); return s; };
        sums[4 * l] = chk(c); sums[4 * l + 1] = chk(b); sums[4 * l + 2] = chk(f); sums[4 * l + 3] = chk(p);
    }
    double* d = dp->chk_d;
    const size_t n = 4 * (size_t)L;
    if (hipMemcpy(d, sums.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d + n, sums.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess) return -AEFFT_EHIP;
    if (ncclAllReduce(d, d, n, ncclDouble, ncclMin, dp->comm, dp->st) != ncclSuccess) return -AEFFT_EHIP;
    if (ncclAllReduce(d + n, d + n, n, ncclDouble, ncclMax, dp->comm, dp->st) != ncclSuccess) return -AEFFT_EHIP;
    if (hipStreamSynchronize(dp->st) != hipSuccess) return -AEFFT_EHIP;
    if (hipMemcpy(lo.data(), d, n * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hi.data(), d + n, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return -AEFFT_EHIP;
    for (size_t i = 0; i < n; ++i) if (lo[i] != hi[i]) return 0;
    return 1;
}
