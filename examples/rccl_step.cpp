// Data-parallel training steps from plain C++ over the flat C ABI and RCCL: one process per GPU,
//     aefft_net_step_grad  ->  ncclAllReduce(SUM) of aefft_net_grad_buffer ON aefft_stream(ctx)  ->  aefft_net_step_apply(1/world)
// No torch, no Python.  Ranks: environment RANK / WORLD_SIZE (default 1 rank); the ncclUniqueId travels through a file named by
// AEFFT_NCCL_ID_FILE when WORLD_SIZE > 1 (rank 0 writes it).  Exit code 0 and "rccl_step ok ..." on success.
//
// What it checks at any world size: the collective and the two halves stay ordered on the library's stream without a host
// synchronisation; the packed buffer's MSE tail (one float per pair, previous step) comes back as the sum over ranks; at
// world size 1 the weights after 3 steps equal those of a second net trained without the collective.
#include "../include/aefft.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include <vector>

#define CK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s failed: %d (%s)\n", #x, r_, ctx ? aefft_last_error(ctx) : ""); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define NK(x) do { ncclResult_t n_ = (x); if (n_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(n_)); return 1; } } while (0)

static float lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.0f; }

int main()
{
    aefft_ctx* ctx = nullptr;
    const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0, world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
    int ndev = 0;
    HK(hipGetDeviceCount(&ndev));
    const int dev = rank % (ndev > 0 ? ndev : 1);
    HK(hipSetDevice(dev));
    CK(aefft_ctx_create(&ctx, dev, nullptr, 1));
    hipStream_t st = (hipStream_t)aefft_stream(ctx);

    ncclUniqueId id;
    const char* idf = getenv("AEFFT_NCCL_ID_FILE");
    if (rank == 0) {
        NK(ncclGetUniqueId(&id));
        if (world > 1) {
            // written under a temporary name and renamed into place: a reader never sees a partial file.  The launcher removes a stale
            // file of an earlier run before it starts the ranks (tests/test_gpu_round4.py does).
            if (!idf) { fprintf(stderr, "AEFFT_NCCL_ID_FILE not set\n"); return 1; }
            const std::string tmp = std::string(idf) + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(&id, sizeof id, 1, f) != 1 || fclose(f) != 0 || rename(tmp.c_str(), idf) != 0) { fprintf(stderr, "cannot write the nccl id to %s\n", idf); return 1; }
        }
    } else {
        if (!idf) { fprintf(stderr, "AEFFT_NCCL_ID_FILE not set\n"); return 1; }
        bool got = false;
        for (int t = 0; t < 600 && !got; ++t) {                       // up to 60 s
            FILE* f = fopen(idf, "rb");
            if (f) { got = fread(&id, sizeof id, 1, f) == 1; fclose(f); }
            if (!got) usleep(100 * 1000);
        }
        if (!got) { fprintf(stderr, "cannot read the nccl id from %s\n", idf); return 1; }
    }
    ncclComm_t comm;
    NK(ncclCommInitRank(&comm, world, id, rank));

    const int D = 3, N = 64, L = 2, B = 4;
    int maps[L] = {4, 6}, k[L] = {5, 5}, s[L] = {2, 2};
    aefft_net_desc d = {D, N, N, L, maps, k, k, s, B};
    aefft_net *net = nullptr, *ref = nullptr;
    CK(aefft_net_create(ctx, &d, &net));
    CK(aefft_net_create(ctx, &d, &ref));
    unsigned seed = 12345;                                            // identical weights on every rank
    int dD = D;
    for (int l = 0; l < L; ++l) {
        const int nk = maps[l] * dD * 25;
        std::vector<float> c(nk), f(nk), b(maps[l]), p(dD);
        for (auto& v : c) v = 2 * lcg(seed) - 1;
        for (auto& v : f) v = 2 * lcg(seed) - 1;
        for (auto& v : b) v = 2 * lcg(seed) - 1;
        for (auto& v : p) v = 2 * lcg(seed) - 1;
        CK(aefft_net_set_pair(net, l, c.data(), b.data(), f.data(), p.data()));
        CK(aefft_net_set_pair(ref, l, c.data(), b.data(), f.data(), p.data()));
        dD = maps[l];
    }
    const size_t nf = (size_t)B * D * N * N;
    std::vector<float> fh(nf);
    unsigned fs = 777u + 31u * (unsigned)rank;                        // each rank its own shard of the global batch
    for (auto& v : fh) v = std::floor(256 * lcg(fs));
    float *frames = nullptr, *recon = nullptr, *mse = nullptr;
    HK(hipMalloc(&frames, nf * 4)); HK(hipMalloc(&recon, nf * 4)); HK(hipMalloc(&mse, L * 4));
    HK(hipMemcpy(frames, fh.data(), nf * 4, hipMemcpyHostToDevice));

    float* gbuf = nullptr; size_t gn = 0;
    CK(aefft_net_grad_buffer(net, &gbuf, &gn));
    std::vector<float> tail_before(L), tail_after(L), m_prev(L);
    for (int it = 0; it < 3; ++it) {
        CK(aefft_net_step_grad(net, frames, recon));
        if (it == 2) HK(hipMemcpyAsync(tail_before.data(), gbuf + gn - L, L * 4, hipMemcpyDeviceToHost, st));     // this rank's MSEs of step 1
        NK(ncclAllReduce(gbuf, gbuf, gn, ncclFloat, ncclSum, comm, st));                                         // on the library's stream
        if (it == 2) HK(hipMemcpyAsync(tail_after.data(), gbuf + gn - L, L * 4, hipMemcpyDeviceToHost, st));
        CK(aefft_net_step_apply(net, 0.2f, 0, 0, 1.0f / (float)world, mse));
        if (it == 1) HK(hipMemcpyAsync(m_prev.data(), mse, L * 4, hipMemcpyDeviceToHost, st));
    }
    std::vector<float> saved(L);
    HK(hipMemcpyAsync(saved.data(), gbuf + gn, L * 4, hipMemcpyDeviceToHost, st));      // behind the buffer: the reduced tail times grad_scale, saved by the last step_apply
    CK(aefft_sync(ctx));
    int bad = 0;
    for (int l = 0; l < L; ++l)
        if (std::fabs(saved[l] - tail_after[l] / (float)world) > 1e-6f * std::fabs(saved[l])) { fprintf(stderr, "saved global MSE != reduced tail / world (pair %d: %g vs %g)\n", l, saved[l], tail_after[l] / (float)world); ++bad; }
    for (int l = 0; l < L; ++l) {
        if (std::fabs(tail_before[l] - m_prev[l]) > 1e-6f * std::fabs(m_prev[l])) { fprintf(stderr, "tail != previous step's MSE (pair %d: %g vs %g)\n", l, tail_before[l], m_prev[l]); ++bad; }
        if (world == 1 && tail_after[l] != tail_before[l]) { fprintf(stderr, "all-reduce at world 1 changed the tail\n"); ++bad; }
        if (!(tail_after[l] > 0.f) || !std::isfinite(tail_after[l])) { fprintf(stderr, "reduced MSE tail not finite\n"); ++bad; }
    }
    if (world == 1) {
        for (int it = 0; it < 3; ++it) { CK(aefft_net_step_grad(ref, frames, nullptr)); CK(aefft_net_step_apply(ref, 0.2f, 0, 0, 1.0f, nullptr)); }
        dD = D;
        for (int l = 0; l < L; ++l) {
            const int nk = maps[l] * dD * 25;
            std::vector<float> c1(nk), c2(nk), f1(nk), f2(nk), b1(maps[l]), b2(maps[l]), p1(dD), p2(dD);
            CK(aefft_net_get_pair(net, l, c1.data(), b1.data(), f1.data(), p1.data()));
            CK(aefft_net_get_pair(ref, l, c2.data(), b2.data(), f2.data(), p2.data()));
            if (memcmp(c1.data(), c2.data(), nk * 4) || memcmp(f1.data(), f2.data(), nk * 4) || memcmp(b1.data(), b2.data(), maps[l] * 4) || memcmp(p1.data(), p2.data(), dD * 4)) {
                fprintf(stderr, "pair %d: weights differ from the run without the collective\n", l); ++bad;
            }
            dD = maps[l];
        }
    }
    printf("%s rank %d/%d: packed buffer %zu floats, global MSE of step 1: %g %g\n", bad ? "rccl_step FAILED" : "rccl_step ok", rank, world, gn,
           tail_after[0] / world, tail_after[1] / world);
    (void)ncclCommDestroy(comm);
    aefft_net_destroy(net); aefft_net_destroy(ref);
    (void)hipFree(frames); (void)hipFree(recon); (void)hipFree(mse);
    aefft_ctx_destroy(ctx);
    return bad ? 1 : 0;
}
