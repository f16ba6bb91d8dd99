// Frequency-space ("momentum space") kernels of the training path for gfx950.
//
//   contract_kernel : per-bin complex channel contraction.  One primitive serves
//                     conv_k              (fft_backproplib.cu:162-189)   O[b][m] = sum_d (X[b][d]/dM) * C[m][d]  (+bias at DC)
//                     gradient_k_io       (fft_backproplib.cu:395-475)   split into its four sums:
//                        G [b][m] = sum_d1 conj(F[d1][m]) * E[b][d1]          (:412-424, sumcRR..sumcII)
//                        H'[b][m] = sum_d1 C[m][d1] * X[b][d1] + b[m]*Nx*Ny   (:426-429,448-450, sumfR/I + b0)
//                        dc[m][d] = sum_b  G[b][m] * conj(X[b][d]) / Norm     (:438-445)
//                        df[d][m] = sum_b  E[b][d] * conj(H'[b][m]) / Norm    (:452-461)
//                     The reference recomputes G and H' inside every (m,d,bin) thread (dD-fold
//                     redundant reads); here they are computed once per (b,m,bin).
//   resize_kernel   : spectral pooling index remap (fft_backproplib.cu:87-157).
//   diff_mse_kernel : E = O - T fused with calc_mse (fft_backproplib.cu:480-498).
//   bias_grad_kernel: db, dp from the DC bins (fft_backproplib.cu:463-473).
//
// Bins are the fastest dimension everywhere, so a wave reads 64 x 16 B = 1 KiB contiguous per
// load instruction; each thread owns two bins (one float4) and a TR x TC register tile of outputs.
#include "../../include/aefft.h"
#include "internal.h"
#include "update_device.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "device_util.h"

namespace aefft {

__device__ __forceinline__ void cfma(float2& acc, float2 a, float2 b)
{
    acc.x += a.x * b.x - a.y * b.y;
    acc.y += a.x * b.y + a.y * b.x;
}

// Thread layout: threadIdx.x = 64 consecutive bin groups (VEC bins each: one float2 or float4 per
// load, so a wave reads 512 B / 1 KiB contiguous), threadIdx.y = up to 4 row tiles that share
// the B operand through L1.  Each thread owns a TR x TC register tile of outputs.
template <int VEC, int TR, int TC>
__device__ __forceinline__ void contract_body(const Contract& q, int bx, int by, int zblk)
{
    using V = typename std::conditional<VEC == 2, float4, float2>::type;
    const long grp = (long)bx * 64 + threadIdx.x;       // index of the bin group
    if (grp * VEC >= q.P) return;
    const int r0 = (by * blockDim.y + threadIdx.y) * TR, c0 = zblk * TC;
    if (r0 >= q.R || c0 >= q.C) return;
    const V* Ap = reinterpret_cast<const V*>(q.A);
    const V* A2p = reinterpret_cast<const V*>(q.A2);
    const V* Bp = reinterpret_cast<const V*>(q.B);
    long bgrp = grp;                    // bin group inside a B plane
    bool live = true;                   // false: destination bin has no source under the zero-pad remap
    if (VEC == 1 && q.upNx) {
        const int Nyr = q.upNy / 2 + 1, Nyrs = q.upNys / 2 + 1;
        const int i = (int)(grp / Nyr), j = (int)(grp - (long)i * Nyr);
        int si = -1, sj = -1;
        if (i < q.upNxs / 2) si = i;
        else if (i > q.upNx - q.upNxs / 2) si = i - q.upNx + q.upNxs;
        else if (i == q.upNx / 2) si = q.upNxs / 2;
        if (j < Nyrs - 1) sj = j;
        else if (j == Nyr - 1) sj = Nyrs - 1;
        live = (si >= 0 && sj >= 0);
        bgrp = (long)si * Nyrs + sj;
    }
    long aoff[TR], boff[TC];
    int rr[TR];
#pragma unroll
    for (int i = 0; i < TR; ++i) { rr[i] = (r0 + i < q.R) ? r0 + i : q.R - 1; aoff[i] = (rr[i] * q.a_r) / VEC + grp; }
#pragma unroll
    for (int j = 0; j < TC; ++j) { const int cc = (c0 + j < q.C) ? c0 + j : q.C - 1; boff[j] = (cc * q.b_c) / VEC + bgrp; }
    const long a_kv = q.a_k / VEC, b_kv = q.b_k / VEC;
    const float sa = q.conjA ? -1.f : 1.f, sb = q.conjB ? -1.f : 1.f;
    // x/dM as x*(1/dM): exact for power-of-two dM, <= 1 ulp otherwise (the reference itself is built with
    // --use_fast_math, Makefile:18, i.e. its division is already approximate)
    const float bmul = q.preDivB != 0.f ? 1.0f / q.preDivB : 1.0f;
    const float omul = q.postDiv != 0.f ? 1.0f / q.postDiv : 1.0f;

    float2 acc[VEC][TR][TC];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
            for (int j = 0; j < TC; ++j) acc[v][i][j] = make_float2(0.f, 0.f);

    const bool dc_thread = (grp == 0);
    if (live) {
        // deep unroll: the loop is latency bound (operands come from L2 / Infinity Cache), so keep
        // UNR iterations of loads in flight per wave
        constexpr int UNR = (TR * TC * VEC <= 8) ? 8 : ((TR * TC * VEC <= 16) ? 4 : 2);
#pragma unroll UNR
        for (int k = 0; k < q.K; ++k) {
            V a[TR], b[TC];
#pragma unroll
            for (int i = 0; i < TR; ++i) a[i] = Ap[aoff[i] + k * a_kv];
            if (A2p) {
#pragma unroll
                for (int i = 0; i < TR; ++i) {
                    const V a2 = A2p[aoff[i] + k * a_kv];
                    float* af = reinterpret_cast<float*>(&a[i]);
                    const float* a2f = reinterpret_cast<const float*>(&a2);
#pragma unroll
                    for (int e = 0; e < 2 * VEC; ++e) af[e] -= a2f[e];
                }
            }
#pragma unroll
            for (int j = 0; j < TC; ++j) b[j] = Bp[boff[j] + k * b_kv];
#pragma unroll
            for (int j = 0; j < TC; ++j) {
                float* bf = reinterpret_cast<float*>(&b[j]);
#pragma unroll
                for (int v = 0; v < VEC; ++v) { bf[2 * v] *= bmul; bf[2 * v + 1] *= bmul * sb; }
            }
#pragma unroll
            for (int i = 0; i < TR; ++i) {
                float* af = reinterpret_cast<float*>(&a[i]);
#pragma unroll
                for (int v = 0; v < VEC; ++v) af[2 * v + 1] *= sa;
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v)
#pragma unroll
                for (int i = 0; i < TR; ++i)
#pragma unroll
                    for (int j = 0; j < TC; ++j) {
                        const float* af = reinterpret_cast<const float*>(&a[i]);
                        const float* bf = reinterpret_cast<const float*>(&b[j]);
                        cfma(acc[v][i][j], make_float2(af[2 * v], af[2 * v + 1]), make_float2(bf[2 * v], bf[2 * v + 1]));
                    }
            if (k == 0 && q.bias && q.biasAfterFirst && dc_thread) {
#pragma unroll
                for (int i = 0; i < TR; ++i)
#pragma unroll
                    for (int j = 0; j < TC; ++j) if (q.biasColP1 == 0 || c0 + j == q.biasColP1 - 1) acc[0][i][j].x += q.bias[rr[i]] * q.biasScale;
            }
        }
    }
    V* Op = reinterpret_cast<V*>(q.Out);
    long cdst[VEC];                      // fused down-sampling: destination bin of each of this thread's bins (or -1)
#pragma unroll
    for (int v = 0; v < VEC; ++v) cdst[v] = (q.Out2 && grp * VEC + v < q.P) ? crop_dest(grp * VEC + v, q.dnNx, q.dnNy, q.dnNxs, q.dnNys) : -1;
    const long Ps = (long)q.dnNxs * (q.dnNys / 2 + 1);
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            if (r0 + i >= q.R || c0 + j >= q.C) continue;
            V o;
            float* of = reinterpret_cast<float*>(&o);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float2 val = acc[v][i][j];
                if (v == 0 && q.bias && !q.biasAfterFirst && dc_thread && (q.biasColP1 == 0 || c0 + j == q.biasColP1 - 1)) val.x += q.bias[rr[i]] * q.biasScale;
                of[2 * v] = val.x * omul; of[2 * v + 1] = val.y * omul;
            }
            Op[((r0 + i) * q.o_r + (c0 + j) * q.o_c) / VEC + grp] = o;
            if (q.Out2) {
                const long plane = ((r0 + i) * q.o_r + (c0 + j) * q.o_c) / q.P;
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (cdst[v] >= 0) q.Out2[plane * Ps + cdst[v]] = make_float2(of[2 * v], of[2 * v + 1]);
            }
        }
}

// XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs by linear id, and every
// XCD has its own 4 MiB L2.  Bins are independent, so the launch is 1-D and decoded such that all
// workgroups whose bin tile is congruent to x mod 8 land on XCD x: each L2 then caches 1/8 of every
// operand instead of all of it (8x less fabric traffic; speed only -- any placement is correct).
template <int VEC, int TR, int TC>
__global__ __launch_bounds__(256) void contract_kernel(const Contract2 qq, int gx, int gy, int gz, int z0)
{
    const BlockId b = xcd_decode(gx, gy, gz);
    if (!b.ok) return;
    // z = [0, z0) -> problem 0, [z0, ...) -> problem 1
    if (b.bz < z0) contract_body<VEC, TR, TC>(qq.q[0], b.bx, b.by, b.bz);
    else contract_body<VEC, TR, TC>(qq.q[1], b.bx, b.by, b.bz - z0);
}

template <int VEC, int TR, int TC> static hipError_t contract_tile(const Contract2& qq, hipStream_t st)
{
    long gx = 0; int gy = 0, by = 1, z[2] = {0, 0};
    for (int p = 0; p < qq.n; ++p) {
        const Contract& q = qq.q[p];
        const long groups = (q.P + VEC - 1) / VEC;
        const int rtiles = (q.R + TR - 1) / TR;
        gx = std::max(gx, (groups + 63) / 64);
        by = std::max(by, rtiles < 4 ? rtiles : 4);
        z[p] = (q.C + TC - 1) / TC;
    }
    for (int p = 0; p < qq.n; ++p) gy = std::max(gy, ((qq.q[p].R + TR - 1) / TR + by - 1) / by);
    const int gz = z[0] + z[1];
    contract_kernel<VEC, TR, TC><<<dim3(xcd_grid(gx, gy, gz)), dim3(64, by), 0, st>>>(qq, (int)gx, gy, gz, z[0]);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Lean variant of the register-tile kernel for the shapes the training loop actually launches.
// PMC showed the generic kernel issue-bound on bookkeeping (5 VALU instructions per useful FMA:
// 64-bit address arithmetic per load, sign and scale multiplies per operand element).  Here
//   * operands are fetched with buffer loads: a per-lane 32-bit byte offset computed once, the K
//     stride advanced in a scalar register (no vector address arithmetic in the loop),
//   * conjugation is folded into the FMA signs (template flags),
//   * the operand scaling x/dM is applied once to the sum (sum_d (x/dM) c == (sum_d x c)/dM up to
//     float32 rounding; power-of-two dM: bit-identical unless an intermediate is subnormal).
// ------------------------------------------------------------------------------------------
// four scalar FMAs per complex MAC (conjugation folded into the signs); the file is built with
// -fno-slp-vectorize because hipcc otherwise packs these into v_pk_mul/v_pk_add/v_pk_fma triples
// (5 packed instructions per complex MAC instead of 4 plain FMAs)
template <bool CA, bool CB> __device__ __forceinline__ void cfma_s(float2& acc, float ax, float ay, float bx, float by)
{
    if (CA && !CB) { acc.x = fmaf(ax, bx, acc.x); acc.x = fmaf(ay, by, acc.x); acc.y = fmaf(ax, by, acc.y); acc.y = fmaf(-ay, bx, acc.y); }        // conj(a) * b
    else if (CB && !CA) { acc.x = fmaf(ax, bx, acc.x); acc.x = fmaf(ay, by, acc.x); acc.y = fmaf(ay, bx, acc.y); acc.y = fmaf(-ax, by, acc.y); }   // a * conj(b)
    else if (CA && CB) { acc.x = fmaf(ax, bx, acc.x); acc.x = fmaf(-ay, by, acc.x); acc.y = fmaf(-ax, by, acc.y); acc.y = fmaf(-ay, bx, acc.y); }  // conj(a*b)
    else { acc.x = fmaf(ax, bx, acc.x); acc.x = fmaf(-ay, by, acc.x); acc.y = fmaf(ax, by, acc.y); acc.y = fmaf(ay, bx, acc.y); }
}

template <int VEC, int TR, int TC, bool CA, bool CB, bool DIFF, int KS, bool MSE = false>
__device__ __forceinline__ void contract_fast_body(const Contract& q, int bx, int by, int zblk, float* red)
{
    using L = BufLoad<VEC>;
    using V = typename L::T;
    const long grp = (long)bx * 64 + threadIdx.x;
    const int ks = KS > 1 ? threadIdx.y : 0;
    const int r0 = (KS > 1 ? by : by * blockDim.y + threadIdx.y) * TR, c0 = zblk * TC;
    const bool inside = grp * VEC < q.P && r0 < q.R && c0 < q.C;
    if (KS == 1 && !inside && !MSE) return;             // (the MSE epilogue ends in a workgroup reduction)
    long bgrp = grp;
    bool live = inside;
    if (VEC == 1 && q.upNx) {
        const int Nyr = q.upNy / 2 + 1, Nyrs = q.upNys / 2 + 1;
        const int i = (int)(grp / Nyr), j = (int)(grp - (long)i * Nyr);
        int si = -1, sj = -1;
        if (i < q.upNxs / 2) si = i;
        else if (i > q.upNx - q.upNxs / 2) si = i - q.upNx + q.upNxs;
        else if (i == q.upNx / 2) si = q.upNxs / 2;
        if (j < Nyrs - 1) sj = j;
        else if (j == Nyr - 1) sj = Nyrs - 1;
        live = inside && (si >= 0 && sj >= 0);
        bgrp = live ? (long)si * Nyrs + sj : 0;
    }
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)q.A, 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DIFF ? q.A2 : q.A), 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)q.B, 0, 0xFFFFFFFFu, 0x00020000);
    unsigned aoff[TR], boff[TC];
#pragma unroll
    for (int i = 0; i < TR; ++i) { const int rr = (r0 + i < q.R) ? r0 + i : q.R - 1; aoff[i] = (unsigned)((rr * q.a_r + grp * VEC) * 8); }
#pragma unroll
    for (int j = 0; j < TC; ++j) { const int cc = (c0 + j < q.C) ? c0 + j : q.C - 1; boff[j] = (unsigned)((cc * q.b_c + bgrp * VEC) * 8); }
    const unsigned a_ks = (unsigned)(q.a_k * 8), b_ks = (unsigned)(q.b_k * 8);

    float2 acc[VEC][TR][TC];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
            for (int j = 0; j < TC; ++j) acc[v][i][j] = make_float2(0.f, 0.f);

    if (live) {
        constexpr int UNR = ((TR + TC) * VEC <= 8) ? 8 : (((TR + TC) * VEC <= 12) ? 4 : 2);   // <= ~96 operand VGPRs in flight
        const int kq = (q.K + KS - 1) / KS;
        const int kbeg = ks * kq, kend = (kbeg + kq < q.K) ? kbeg + kq : q.K;
        // Explicit software pipelining: hipcc keeps "load k; wait; FMA k" order inside an unrolled loop (and sinks
        // hoisted loads back past any branch), which exposes one full memory round trip per k (measured:
        // 0.4 us per k).  A branch-free group of UNR iterations puts UNR*(TR+TC) loads in flight before the first wait.
        auto group = [&](int k0, auto NU) {
            constexpr int U = decltype(NU)::value;            // straight-line: U*(TR+TC) loads, then the FMAs -- no branch in between
            V a[U][TR], b[U][TC], a2[DIFF ? U : 1][DIFF ? TR : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned sa = (k0 + u) * a_ks, sb = (k0 + u) * b_ks;
#pragma unroll
                for (int i = 0; i < TR; ++i) a[u][i] = L::ld(ra, aoff[i], sa);
#pragma unroll
                for (int j = 0; j < TC; ++j) b[u][j] = L::ld(rb, boff[j], sb);
                if (DIFF) {
#pragma unroll
                    for (int i = 0; i < TR; ++i) a2[u][i] = L::ld(ra2, aoff[i], sa);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (DIFF) {
#pragma unroll
                    for (int i = 0; i < TR; ++i) {
                        float* af = reinterpret_cast<float*>(&a[u][i]);
                        const float* a2f = reinterpret_cast<const float*>(&a2[u][i]);
#pragma unroll
                        for (int e = 0; e < 2 * VEC; ++e) af[e] -= a2f[e];
                    }
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v)
#pragma unroll
                    for (int i = 0; i < TR; ++i)
#pragma unroll
                        for (int j = 0; j < TC; ++j) {
                            const float* af = reinterpret_cast<const float*>(&a[u][i]);
                            const float* bf = reinterpret_cast<const float*>(&b[u][j]);
                            cfma_s<CA, CB>(acc[v][i][j], af[2 * v], af[2 * v + 1], bf[2 * v], bf[2 * v + 1]);
                        }
            }
        };
        int k0 = kbeg;
        for (; k0 + UNR <= kend; k0 += UNR) group(k0, std::integral_constant<int, UNR>{});
        for (; k0 < kend; ++k0) group(k0, std::integral_constant<int, 1>{});
    }
    if (KS > 1) {
        // waves 1..KS-1 park their partial tiles in LDS ([slice][value][lane]: conflict-free), wave 0 sums them in slice order
        constexpr int NV = VEC * TR * TC * 2;
        if (ks > 0) {
#pragma unroll
            for (int v = 0; v < VEC; ++v)
#pragma unroll
                for (int i = 0; i < TR; ++i)
#pragma unroll
                    for (int j = 0; j < TC; ++j) {
                        const int e = ((v * TR + i) * TC + j) * 2;
                        red[((ks - 1) * NV + e) * 64 + threadIdx.x] = acc[v][i][j].x;
                        red[((ks - 1) * NV + e + 1) * 64 + threadIdx.x] = acc[v][i][j].y;
                    }
        }
        __syncthreads();
        if (ks > 0 || (!inside && !MSE)) return;
#pragma unroll
        for (int s2 = 0; s2 < KS - 1; ++s2)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
#pragma unroll
                for (int i = 0; i < TR; ++i)
#pragma unroll
                    for (int j = 0; j < TC; ++j) {
                        const int e = ((v * TR + i) * TC + j) * 2;
                        acc[v][i][j].x += red[(s2 * NV + e) * 64 + threadIdx.x];
                        acc[v][i][j].y += red[(s2 * NV + e + 1) * 64 + threadIdx.x];
                    }
    }
    if (MSE) {
        // tile = pair-local reconstruction O_b[d'] (rows d' = r, cols b = c); compare with X_b[d'] = B[k = r][c]
        float part = 0.f;
        if (inside) {
            float2 beta[TR];
#pragma unroll
            for (int i = 0; i < TR; ++i) beta[i] = make_float2(0.f, 0.f);
            if (bx == 0) {                                    // this wave holds the DC bin (lane 0): wave-parallel bias term
#pragma unroll
                for (int i = 0; i < TR; ++i) {
                    const int rr = (r0 + i < q.R) ? r0 + i : q.R - 1;
                    float sx = 0.f, sy = 0.f;
                    for (int m = threadIdx.x; m < q.mse.dM; m += 64) {
                        const float2 f = q.mse.F[((long)rr * q.mse.dM + m) * q.P];
                        const float bb = q.mse.b[m];
                        sx = fmaf(f.x, bb, sx); sy = fmaf(f.y, bb, sy);
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) { sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); }
                    beta[i] = make_float2((sx / (float)q.R + q.mse.p[rr]) * q.mse.norm, sy / (float)q.R * q.mse.norm);
                }
            }
            float w[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const unsigned j = (unsigned)((grp * VEC + v) % q.mse.Nyr);
                w[v] = (j > 0 && j < (unsigned)q.mse.Nyr - 1) ? 2.f : 1.f;        // Hermitian half-plane: interior columns count twice
            }
#pragma unroll
            for (int i = 0; i < TR; ++i)
#pragma unroll
                for (int j = 0; j < TC; ++j) {
                    if (r0 + i >= q.R || c0 + j >= q.C) continue;
                    const V t = L::ld(rb, boff[j], (unsigned)(r0 + i) * b_ks);
                    const float* tf = reinterpret_cast<const float*>(&t);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        float ox = acc[v][i][j].x, oy = acc[v][i][j].y;
                        if (v == 0 && grp == 0) { ox += beta[i].x; oy += beta[i].y; }
                        const float ex = tf[2 * v] - ox, ey = tf[2 * v + 1] - oy;
                        part = fmaf(w[v], ex * ex + ey * ey, part);
                    }
                }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        const float sc = q.mse.scale / q.mse.nfull;
        // one atomic per workgroup, spread over MSE_SLOTS addresses (64 B apart): device-scope float atomics on ONE
        // address retire at ~22 ns each on MI355X (measured), which made a 2112-workgroup launch take 35 us
        float* slot = q.mse.acc + (blockIdx.x % MSE_SLOTS) * MSE_SLOT_STRIDE;
        if (KS > 1) { if (threadIdx.x == 0 && part != 0.f) atomicAdd(slot, part * sc); return; }
        if (threadIdx.x == 0) red[threadIdx.y] = part;
        __syncthreads();
        if (threadIdx.x == 0 && threadIdx.y == 0) {
            float s = 0.f;
            for (int y = 0; y < (int)blockDim.y; ++y) s += red[y];
            if (s != 0.f) atomicAdd(slot, s * sc);
        }
        return;
    }
    const float bmul = q.preDivB != 0.f ? 1.0f / q.preDivB : 1.0f;
    const float omul = q.postDiv != 0.f ? 1.0f / q.postDiv : 1.0f;
    V* Op = reinterpret_cast<V*>(q.Out);
    long cdst[VEC];                      // fused down-sampling: destination bin of each of this thread's bins (or -1), computed once
#pragma unroll
    for (int v = 0; v < VEC; ++v) cdst[v] = (q.Out2 && grp * VEC + v < q.P) ? crop_dest(grp * VEC + v, q.dnNx, q.dnNy, q.dnNxs, q.dnNys) : -1;
    const long Ps = (long)q.dnNxs * (q.dnNys / 2 + 1);
    const long orp = q.Out2 ? q.o_r / q.P : 0, ocp = q.Out2 ? q.o_c / q.P : 0;
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            if (r0 + i >= q.R || c0 + j >= q.C) continue;
            V o;
            float* of = reinterpret_cast<float*>(&o);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float2 val = make_float2(acc[v][i][j].x * bmul, acc[v][i][j].y * bmul);
                if (v == 0 && q.bias && grp == 0 && (q.biasColP1 == 0 || c0 + j == q.biasColP1 - 1)) val.x += q.bias[r0 + i] * q.biasScale;
                of[2 * v] = val.x * omul; of[2 * v + 1] = val.y * omul;
            }
            Op[((r0 + i) * q.o_r + (c0 + j) * q.o_c) / VEC + grp] = o;
            if (q.Out2) {
                // Out2 keeps Out's [c][r] plane order with the small plane size: plane index = ((c)*o_c + (r)*o_r) / P
                const long plane = (long)(r0 + i) * orp + (long)(c0 + j) * ocp;
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (cdst[v] >= 0) q.Out2[plane * Ps + cdst[v]] = make_float2(of[2 * v], of[2 * v + 1]);
            }
        }
}

// FL: 0 = plain (conv_k), 1 = a*conj(b) with fused subtraction (the S contraction),
//     2 = dual launch: problem 0 conj(a)*b (dc), problem 1 a*conj(b) (df), 3 = plain with the MSE epilogue (no store)
template <int VEC, int TR, int TC, int FL, int KS>
__global__ __launch_bounds__(256) void contract_fast_kernel(const Contract2 qq, int gx, int gy, int gz, int z0)
{
    __shared__ float red[KS > 1 ? (KS - 1) * VEC * TR * TC * 2 * 64 : 4];
    const BlockId b = xcd_decode(gx, gy, gz);
    if (!b.ok) return;                                   // uniform per workgroup
    if (FL == 3) contract_fast_body<VEC, TR, TC, false, false, false, KS, true>(qq.q[0], b.bx, b.by, b.bz, red);
    else if (FL == 0) contract_fast_body<VEC, TR, TC, false, false, false, KS>(qq.q[0], b.bx, b.by, b.bz, red);
    else if (FL == 1) contract_fast_body<VEC, TR, TC, false, true, true, KS>(qq.q[0], b.bx, b.by, b.bz, red);
    else {
        if (b.bz < z0) contract_fast_body<VEC, TR, TC, true, false, false, KS>(qq.q[0], b.bx, b.by, b.bz, red);
        else contract_fast_body<VEC, TR, TC, false, true, false, KS>(qq.q[1], b.bx, b.by, b.bz - z0, red);
    }
}

template <int VEC, int TR, int TC, int FL, int KS> static hipError_t contract_fast_tile(const Contract2& qq, hipStream_t st)
{
    long gx = 0; int gy = 0, by = 1, z[2] = {0, 0};
    for (int p = 0; p < qq.n; ++p) {
        const Contract& q = qq.q[p];
        const long groups = (q.P + VEC - 1) / VEC;
        const int rtiles = (q.R + TR - 1) / TR;
        gx = std::max(gx, (groups + 63) / 64);
        by = std::max(by, rtiles < 4 ? rtiles : 4);
        z[p] = (q.C + TC - 1) / TC;
    }
    if (KS > 1) by = KS;
    for (int p = 0; p < qq.n; ++p) { const int rtiles = (qq.q[p].R + TR - 1) / TR; gy = std::max(gy, KS > 1 ? rtiles : (rtiles + by - 1) / by); }
    const int gz = z[0] + z[1];
    contract_fast_kernel<VEC, TR, TC, FL, KS><<<dim3(xcd_grid(gx, gy, gz)), dim3(64, by), 0, st>>>(qq, (int)gx, gy, gz, z[0]);
    return hipGetLastError();
}

static bool contract_even(const Contract& q);
// ---- grouped launch: several problems, one grid.  Each problem owns a contiguous, 8-aligned range of linear
// workgroup ids (so the XCD-aware decode keeps working inside its range).
template <int VEC, int TR, int TC, int FL, int KS>
__global__ __launch_bounds__(256) void contract_group_kernel(const ContractN g)
{
    __shared__ float red[KS > 1 ? (KS - 1) * VEC * TR * TC * 2 * 64 : 1];
    const int lin = blockIdx.x;
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && lin >= g.start[i]) p = i;
    const BlockId b = xcd_decode_lin(lin - g.start[p], g.gx[p], g.gy[p], g.gz[p]);
    if (!b.ok) return;                                   // uniform per workgroup
    if (FL == 0) contract_fast_body<VEC, TR, TC, false, false, false, KS>(g.q[p], b.bx, b.by, b.bz, red);
    else if (FL == 1) contract_fast_body<VEC, TR, TC, false, true, true, KS>(g.q[p], b.bx, b.by, b.bz, red);
    else {
        if (p < g.nA) contract_fast_body<VEC, TR, TC, true, false, false, KS>(g.q[p], b.bx, b.by, b.bz, red);
        else contract_fast_body<VEC, TR, TC, false, true, false, KS>(g.q[p], b.bx, b.by, b.bz, red);
    }
}

template <int VEC, int TR, int TC, int FL, int KS> static hipError_t contract_group_tile(ContractN& g, hipStream_t st)
{
    const int by = 4;
    int total = 0;
    for (int p = 0; p < g.n; ++p) {
        const Contract& q = g.q[p];
        const long groups = (q.P + VEC - 1) / VEC;
        const int rtiles = (q.R + TR - 1) / TR;
        g.gx[p] = (int)((groups + 63) / 64);
        g.gy[p] = KS > 1 ? rtiles : (rtiles + by - 1) / by;
        g.gz[p] = (q.C + TC - 1) / TC;
        g.start[p] = total;
        total += (int)((xcd_grid(g.gx[p], g.gy[p], g.gz[p]) + 7) / 8 * 8);
    }
    g.start[g.n] = total;
    contract_group_kernel<VEC, TR, TC, FL, KS><<<dim3(total), dim3(64, by), 0, st>>>(g);
    return hipGetLastError();
}

static bool use_mfma() { return !flag(AEFFT_F_NOMFMA); }

hipError_t launch_contract_group(ContractN& g, int cls, hipStream_t st)
{
    if (g.n < 1 || g.n > 8 || cls < 0 || cls > 2) return hipErrorInvalidValue;
    if (use_mfma()) { const hipError_t e = launch_contract_mfma(g, st); if (e != hipErrorInvalidValue) return e; }
    bool even = true;
    int Cmin = 1 << 30, Rmin = 1 << 30, Kmax = 0;
    for (int p = 0; p < g.n; ++p) {
        const Contract& q = g.q[p];
        if (q.R <= 0 || q.C <= 0 || q.K <= 0 || q.P <= 0) return hipErrorInvalidValue;
        const double a = ((double)(q.R - 1) * q.a_r + (double)(q.K - 1) * q.a_k + q.P) * 8.0;
        const double b = ((double)(q.C - 1) * q.b_c + (double)(q.K - 1) * q.b_k + q.P) * 8.0;
        if (a >= 4.0e9 || b >= 4.0e9 || q.upNx || q.Out2) return hipErrorInvalidValue;     // callers fall back to per-problem launches
        even = even && contract_even(q);
        Cmin = std::min(Cmin, q.C); Rmin = std::min(Rmin, q.R); Kmax = std::max(Kmax, q.K);
    }
    if (Rmin < 2 || Cmin < 2) return hipErrorInvalidValue;
    const int vec = even ? 2 : 1, tc = Cmin >= 4 ? 4 : 2;
    const bool ks = Kmax >= 16;
#define AEFFT_CG(V, C_, F) if (vec == V && tc == C_ && cls == F) return ks ? contract_group_tile<V, 2, C_, F, 4>(g, st) : contract_group_tile<V, 2, C_, F, 1>(g, st);
    AEFFT_CG(2, 4, 0) AEFFT_CG(2, 4, 1) AEFFT_CG(2, 4, 2) AEFFT_CG(2, 2, 0) AEFFT_CG(2, 2, 1) AEFFT_CG(2, 2, 2)
    AEFFT_CG(1, 4, 0) AEFFT_CG(1, 4, 1) AEFFT_CG(1, 4, 2) AEFFT_CG(1, 2, 0) AEFFT_CG(1, 2, 1) AEFFT_CG(1, 2, 2)
#undef AEFFT_CG
    return hipErrorInvalidValue;
}

// which lean instantiation (if any) serves this launch: -1 = none (generic kernel)
static int contract_fast_class(const Contract2& qq)
{
    auto fits32 = [](const Contract& q) {
        const double a = ((double)(q.R - 1) * q.a_r + (double)(q.K - 1) * q.a_k + q.P) * 8.0;
        const double b = ((double)(q.C - 1) * q.b_c + (double)(q.K - 1) * q.b_k + q.P) * 8.0;
        return a < 4.0e9 && b < 4.0e9;
    };
    for (int p = 0; p < qq.n; ++p) if (!fits32(qq.q[p]) || (!qq.q[p].biasAfterFirst && qq.q[p].bias)) return -1;
    if (qq.q[0].Out2 && (qq.n != 1 || qq.q[0].conjA || qq.q[0].conjB || qq.q[0].A2)) return -1;
    const Contract& a = qq.q[0];
    if (a.mse.acc) return (qq.n == 1 && !a.conjA && !a.conjB && !a.A2 && !a.upNx && !a.Out2 && a.R == a.K) ? 3 : -1;
    if (qq.n == 1) {
        if (!a.conjA && !a.conjB && !a.A2) return 0;
        if (!a.conjA && a.conjB && a.A2 && !a.upNx) return 1;
        return -1;
    }
    const Contract& b = qq.q[1];
    if (a.conjA && !a.conjB && !a.A2 && !a.upNx && !b.conjA && b.conjB && !b.A2 && !b.upNx) return 2;
    return -1;
}

static bool contract_even(const Contract& q)
{
    return !((q.P & 1) || (q.a_r & 1) || (q.a_k & 1) || (q.b_k & 1) || (q.b_c & 1) || (q.o_r & 1) || (q.o_c & 1)) && !q.upNx;
}

hipError_t launch_contract2(const Contract2& qq, hipStream_t st)
{
    if (qq.n < 1 || qq.n > 2) return hipErrorInvalidValue;
    bool even = true;
    int Rmin = 1 << 30, Cmin = 1 << 30;
    for (int p = 0; p < qq.n; ++p) {
        const Contract& q = qq.q[p];
        if (q.R <= 0 || q.C <= 0 || q.K <= 0 || q.P <= 0) return hipErrorInvalidValue;
        even = even && contract_even(q);          // the float4 path needs 16-byte aligned plane offsets and no remap
        Rmin = std::min(Rmin, q.R); Cmin = std::min(Cmin, q.C);
    }
    if (use_mfma()) {                                          // matrix-core kernel first; it declines (InvalidValue) what it does not serve
        ContractN g{};
        for (int p = 0; p < qq.n; ++p) g.q[p] = qq.q[p];
        g.n = qq.n;
        const hipError_t e = launch_contract_mfma(g, st);
        if (e != hipErrorInvalidValue) return e;
    }
    for (int p = 0; p < qq.n; ++p) if (qq.q[p].gdNx) return hipErrorInvalidValue;     // gather form: matrix-core kernel only
    // start from the largest register tile the shapes allow and shrink until the launch has enough waves
    int tr = Rmin >= 4 ? 4 : (Rmin >= 2 ? 2 : 1), tc = Cmin >= 4 ? 4 : (Cmin >= 2 ? 2 : 1), vec = even ? 2 : 1;
    auto waves = [&](int v, int r, int c) {
        long w = 0;
        for (int p = 0; p < qq.n; ++p) { const Contract& q = qq.q[p]; w += ((q.P + 64L * v - 1) / (64L * v)) * ((q.R + r - 1) / r) * ((q.C + c - 1) / c); }
        return w;
    };
    // Measured on MI355X (tools_mb.py sweep): these kernels are bound by the number of wave-level load
    // instructions the texture path must process (~16 clk each, whatever their width) and by waves in flight.
    // 16-byte loads (VEC = 2) halve the instruction count; TR = 4 only pays when the launch still has >= 4096 waves.
    if (waves(vec, tr, tc) < 4096 && tr == 4) tr = 2;
    if (waves(vec, tr, tc) < 1024 && vec == 2) vec = 1;
    const bool nofast = flag(AEFFT_F_NOFAST);
    const int fc = nofast ? -1 : contract_fast_class(qq);
    if (fc >= 0 && tr >= 2 && tc >= 2) {
        // split-K when even the shrunk tiles leave the chip short of waves and the K chain is long
        int Kmin = 1 << 30;
        for (int p = 0; p < qq.n; ++p) Kmin = std::min(Kmin, qq.q[p].K);
        const bool nosplit = flag(AEFFT_F_NOSPLITK);
        const bool splitk = !nosplit && Kmin >= 16 && waves(vec, tr, tc) < 4096 && vec * tr * tc <= 16;
#define AEFFT_CF(V, R_, C_) if (vec == V && tr == R_ && tc == C_) { \
        if (V * R_ * C_ <= 16 && splitk) { \
            if (fc == 0) return contract_fast_tile<V, R_, C_, 0, (V * R_ * C_ <= 16 ? 4 : 1)>(qq, st); \
            if (fc == 1) return contract_fast_tile<V, R_, C_, 1, (V * R_ * C_ <= 16 ? 4 : 1)>(qq, st); \
            if (fc == 3) return contract_fast_tile<V, R_, C_, 3, (V * R_ * C_ <= 16 ? 4 : 1)>(qq, st); \
            return contract_fast_tile<V, R_, C_, 2, (V * R_ * C_ <= 16 ? 4 : 1)>(qq, st); } \
        if (fc == 0) return contract_fast_tile<V, R_, C_, 0, 1>(qq, st); \
        if (fc == 1) return contract_fast_tile<V, R_, C_, 1, 1>(qq, st); \
        if (fc == 3) return contract_fast_tile<V, R_, C_, 3, 1>(qq, st); \
        return contract_fast_tile<V, R_, C_, 2, 1>(qq, st); }
        AEFFT_CF(2, 4, 4) AEFFT_CF(2, 4, 2) AEFFT_CF(2, 2, 4) AEFFT_CF(2, 2, 2)
        AEFFT_CF(1, 4, 4) AEFFT_CF(1, 4, 2) AEFFT_CF(1, 2, 4) AEFFT_CF(1, 2, 2)
#undef AEFFT_CF
    }
    if (qq.q[0].mse.acc) return hipErrorInvalidValue;        // the MSE epilogue exists only in the lean kernel: callers use conv, conv, diff_mse instead
#define AEFFT_CT(V, R_, C_) if (vec == V && tr == R_ && tc == C_) return contract_tile<V, R_, C_>(qq, st);
    AEFFT_CT(2, 4, 4) AEFFT_CT(2, 4, 2) AEFFT_CT(2, 2, 4) AEFFT_CT(2, 2, 2) AEFFT_CT(2, 4, 1) AEFFT_CT(2, 1, 4) AEFFT_CT(2, 2, 1) AEFFT_CT(2, 1, 2) AEFFT_CT(2, 1, 1)
    AEFFT_CT(1, 4, 4) AEFFT_CT(1, 4, 2) AEFFT_CT(1, 2, 4) AEFFT_CT(1, 2, 2) AEFFT_CT(1, 4, 1) AEFFT_CT(1, 1, 4) AEFFT_CT(1, 2, 1) AEFFT_CT(1, 1, 2) AEFFT_CT(1, 1, 1)
#undef AEFFT_CT
    return hipErrorInvalidValue;
}

hipError_t launch_contract(const Contract& q, hipStream_t st)
{
    Contract2 qq{};
    qq.q[0] = q; qq.n = 1;
    return launch_contract2(qq, st);
}

// ------------------------------------------------------------------------------------------
// spectral pooling (fft_backproplib.cu:87-157).  Destination fully written (zeros where the
// reference relies on its cudaMemset, :990).
// ------------------------------------------------------------------------------------------
template <typename I>   // 32-bit indexing whenever the tensors allow it: 64-bit div/mod costs hundreds of cycles per element
__global__ __launch_bounds__(256) void resize_kernel(const float2* __restrict__ in, float2* __restrict__ out, I planes,
                                                     int Nx, int Ny, int Nxs, int Nys)
{
    const int Nyr = Ny / 2 + 1, Nyrs = Nys / 2 + 1;
    const I psz = (I)Nxs * Nyrs;
    const I total = planes * psz;
    for (I idx = (I)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (I)gridDim.x * 256) {
        const I d = idx / psz;
        const int rem = (int)(idx - d * psz);
        const int i = rem / Nyrs, j = rem - i * Nyrs;
        int si = -1, sj = -1;
        if (Nxs <= Nx) {
            si = (i < Nxs / 2) ? i : (i == Nxs / 2 ? Nx / 2 : i + Nx - Nxs);
            sj = (j < Nyrs - 1) ? j : Nyr - 1;
        } else {
            if (i < Nx / 2) si = i;
            else if (i > Nxs - Nx / 2) si = i - Nxs + Nx;
            else if (i == Nxs / 2) si = Nx / 2;
            if (j < Nyr - 1) sj = j;
            else if (j == Nyrs - 1) sj = Nyr - 1;
        }
        float2 v = make_float2(0.f, 0.f);
        if (si >= 0 && sj >= 0) v = in[(d * Nx + si) * (I)Nyr + sj];
        out[idx] = v;
    }
}

hipError_t launch_resize(const float2* in, float2* out, long planes, int Nx, int Ny, int Nxs, int Nys, hipStream_t st)
{
    const long total = planes * Nxs * (long)(Nys / 2 + 1);
    const long total_in = planes * Nx * (long)(Ny / 2 + 1);
    if (total <= 0) return hipSuccess;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (total < (1L << 31) && total_in < (1L << 31))
        resize_kernel<unsigned><<<dim3((unsigned)blocks), 256, 0, st>>>(in, out, (unsigned)planes, Nx, Ny, Nxs, Nys);
    else
        resize_kernel<long><<<dim3((unsigned)blocks), 256, 0, st>>>(in, out, planes, Nx, Ny, Nxs, Nys);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// spectrum display (fft_backproplib.cu:27-63): `magnitude` followed by `shift_magnitude`, fused.  Output pixel (i, j) takes
// the magnitude at (i', j') = the quadrant-swapped position when shift != 0 (:33-36), else (i, j); the magnitude itself is
// sqrt(|X| / Ntot) of X[i'][j'] for j' < Nyr and of the mirrored element X[Nx-1-i'][2*Nyr-1-j'] otherwise (:52-53, literally).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void magnitude_kernel(const float2* __restrict__ X, float* __restrict__ mag, unsigned planes, int ch, int Nx, int Ny, int shift)
{
    const unsigned Nyr = Ny / 2 + 1, psz = (unsigned)Nx * Ny;
    const unsigned total = planes * psz;
    const float inv = 1.0f / ((float)ch * (float)Nx * (float)Ny);
    for (unsigned idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const unsigned d = idx / psz, rem = idx - d * psz;
        int i = (int)(rem / (unsigned)Ny), j = (int)(rem - (unsigned)i * Ny);
        if (shift) { i = i < Nx / 2 ? i + Nx / 2 : i - Nx / 2; j = j < Ny / 2 ? j + Ny / 2 : j - Ny / 2; }
        const float2 v = j < (int)Nyr ? X[((size_t)d * Nx + i) * Nyr + j] : X[((size_t)d * Nx + (Nx - 1 - i)) * Nyr + (2 * Nyr - 1 - j)];
        mag[idx] = sqrtf(sqrtf(v.x * v.x + v.y * v.y) * inv);
    }
}
hipError_t launch_magnitude(const float2* X, float* mag, long planes, int ch, int Nx, int Ny, int shift, hipStream_t st)
{
    const long total = planes * Nx * Ny;
    if (total <= 0) return hipSuccess;
    if (total >= (1L << 32) || ch < 1) return hipErrorInvalidValue;
    magnitude_kernel<<<dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), 256, 0, st>>>(X, mag, (unsigned)planes, ch, Nx, Ny, shift);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// E = O - T, and the spectral MSE (fft_backproplib.cu:480-498 + 1188-1190):
//   mse = sum_bins |T-O|^2 / n_bin / (2*dM*Nx*Ny),  n_bin = dD*Nx*Ny, halved for columns 0<j<Nyr-1.
// For a batch the mean over frames is accumulated: *mse_acc += partial * scale.
// ------------------------------------------------------------------------------------------
// Two bins per thread (P is even, so a float4 never straddles a plane); at most 256 blocks so the
// single-address atomics stay cheap.  es[d] += E_b[d](0,0) collects the DC bins of the error for
// the bias gradients (fft_backproplib.cu:432,471).
template <typename I>
__global__ __launch_bounds__(256) void diff_mse_kernel(const float4* __restrict__ T, const float4* __restrict__ O,
                                                       float4* __restrict__ E, float* __restrict__ mse_acc,
                                                       float* __restrict__ es, I npairs, I P, int ch, int Nyr,
                                                       float nfull, float scale)
{
    float part = 0.f;
    const I hp = P / 2;                                  // float4 pairs per plane
    // four float4 pairs per thread and trip: 8 loads in flight before the first use
    const I stride = (I)gridDim.x * 256;
    for (I idx0 = (I)blockIdx.x * 256 + threadIdx.x; idx0 < npairs; idx0 += 4 * stride) {
        float4 tv[4], ov[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const I idx = idx0 + u * stride;
            const I ci = idx < npairs ? idx : npairs - 1;
            tv[u] = T[ci]; ov[u] = O[ci];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const I idx = idx0 + u * stride;
            if (idx >= npairs) break;
            const float4 t = tv[u], o = ov[u];
            const float4 e = make_float4(o.x - t.x, o.y - t.y, o.z - t.z, o.w - t.w);
            if (E) E[idx] = e;
            const I plane = idx / hp;
            const unsigned bin = (unsigned)(idx - plane * hp) * 2u;
            if (es && bin == 0) {
                const int d = (int)(plane % (I)ch);
                atomicAdd(&es[2 * d], e.x); atomicAdd(&es[2 * d + 1], e.y);
            }
            const unsigned j0 = bin % (unsigned)Nyr, j1 = (j0 + 1 == (unsigned)Nyr) ? 0u : j0 + 1;
            const float n0 = (j0 > 0 && j0 < (unsigned)Nyr - 1) ? nfull / 2 : nfull, n1 = (j1 > 0 && j1 < (unsigned)Nyr - 1) ? nfull / 2 : nfull;
            part += (e.x * e.x + e.y * e.y) / n0 + (e.z * e.z + e.w * e.w) / n1;
        }
    }
    if (!mse_acc) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    __shared__ float wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(mse_acc, (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * scale);
}

hipError_t launch_diff_mse(const float2* T, const float2* O, float2* E, float* mse_acc, float* es, int B, int ch, int Nx, int Ny,
                           float scale, hipStream_t st)
{
    const int Nyr = Ny / 2 + 1;
    const long P = (long)Nx * Nyr;
    const long npairs = (long)B * ch * P / 2;
    if (npairs <= 0) return hipSuccess;
    if (P & 1) return hipErrorInvalidValue;
    long blocks = (npairs + 1023) / 1024;                // >= 4 float4 per thread, at most 512 single-address atomics
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    const float4 *T4 = reinterpret_cast<const float4*>(T), *O4 = reinterpret_cast<const float4*>(O);
    float4* E4 = reinterpret_cast<float4*>(E);
    if (npairs < (1L << 30))
        diff_mse_kernel<unsigned><<<dim3((unsigned)blocks), 256, 0, st>>>(T4, O4, E4, mse_acc, es, (unsigned)npairs, (unsigned)P, ch, Nyr, (float)ch * Nx * Ny, scale);
    else
        diff_mse_kernel<long><<<dim3((unsigned)blocks), 256, 0, st>>>(T4, O4, E4, mse_acc, es, npairs, P, ch, Nyr, (float)ch * Nx * Ny, scale);
    return hipGetLastError();
}

// DC-bin terms of gradient_k_io.  es[d] = sum_b (O_b[d](0,0) - T_b[d](0,0)) is rebuilt per workgroup in LDS, then
//   db[m]       = Re( sum_d1 es[d1] * conj(F[d1][m](0,0)) ) * norm / (Norm*B)      (fft_backproplib.cu:432,465)
//   dp[d]       = Re es[d] * norm / (Norm*B)                                       (:471)
//   df[d][m](0,0) += es[d] * b[m]*norm / (Norm*B)    -- the b0 term of :448-455 (H' bias at DC)
// sums (and clears) the slot accumulators of the MSE epilogue: out[l] += sum, copy[l] = out[l].  Optionally also forms the
// DC-bin bias of the collapsed pair operator O = G X + beta (conv_k o conv_k, fft.cu:183-184 twice):
//   beta[d'] = p[d'] + sum_m F[d'][m](0,0) b[m] / dD          (times Nx*Ny where it is applied)
__global__ __launch_bounds__(MSE_SLOTS) void mse_finish_kernel(float* __restrict__ slots, float* __restrict__ out, float* __restrict__ copy, float* __restrict__ copy2, int L,
                                                               const BetaArgs ba, float prev_scale)
{
    __shared__ float ws[MSE_SLOTS / 64];
    __shared__ float bacc[256];
    if (blockIdx.x == 1) {                               // second workgroup (only launched when beta is wanted): runs beside the sums
        for (int d = threadIdx.x; d < ba.dD; d += MSE_SLOTS) bacc[d] = 0.f;
        __syncthreads();
        for (int i = threadIdx.x; i < ba.dD * ba.dM; i += MSE_SLOTS) {
            const int d = i / ba.dM, m = i - d * ba.dM;
            atomicAdd(&bacc[d], ba.F[(long)i * ba.P].x * ba.b[m]);
        }
        __syncthreads();
        for (int d = threadIdx.x; d < ba.dD; d += MSE_SLOTS) ba.beta[d] = ba.p[d] + bacc[d] / (float)ba.dD;
        return;
    }
    mse_finish_body(slots, out, copy, copy2, L, prev_scale, ws);
}

hipError_t launch_mse_finish(float* slots, float* out, float* copy, int L, hipStream_t st, const BetaArgs* ba, float* copy2, float prev_scale)
{
    BetaArgs a{};
    if (ba && ba->dD <= 256) a = *ba;
    mse_finish_kernel<<<a.beta ? 2 : 1, MSE_SLOTS, 0, st>>>(slots, out, copy, copy2, L, a, prev_scale);
    return hipGetLastError();
}


__global__ __launch_bounds__(256) void bias_grad_kernel(const float2* __restrict__ O, const float2* __restrict__ T,
                                                        const float2* __restrict__ F, const float* __restrict__ b,
                                                        float2* __restrict__ df, float* __restrict__ db, float* __restrict__ dp,
                                                        int B, int dM, int dD, long P, float norm, float Norm, int fix_blocks)
{
    extern __shared__ float2 es[];
    bias_grad_body(O, T, F, b, df, db, dp, B, dM, dD, P, norm, Norm, fix_blocks, blockIdx.x, es, P, nullptr);
}

__global__ __launch_bounds__(256) void bias_grad_group_kernel(const BiasGradGroup g)
{
    extern __shared__ float2 es[];
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
    const BiasGradArgs& a = g.a[p];
    bias_grad_body(a.O, a.T, a.F, a.b, a.df, a.db, a.dp, a.B, a.dM, a.dD, a.P, a.norm, a.Norm, g.fix[p], blockIdx.x - g.start[p], es, a.PO, a.es_out, a.es_in);
}

hipError_t launch_bias_grad_group(BiasGradGroup& g, hipStream_t st)
{
    if (g.n < 1 || g.n > 8) return hipErrorInvalidValue;
    int total = 0; size_t lds = 0;
    for (int i = 0; i < g.n; ++i) {
        const BiasGradArgs& a = g.a[i];
        g.fix[i] = (a.dM * a.dD + 255) / 256;
        g.start[i] = total; total += g.fix[i] + (a.dM + 3) / 4;
        lds = std::max(lds, bias_grad_lds(a.B, a.dD));
    }
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    g.start[g.n] = total;
    bias_grad_group_kernel<<<dim3(total), 256, lds, st>>>(g);
    return hipGetLastError();
}

hipError_t launch_bias_grad(const float2* O, const float2* T, const float2* F, const float* b, float2* df, float* db, float* dp,
                            int B, int dM, int dD, long P, float norm, float Norm, hipStream_t st)
{
    const int fix_blocks = (dM * dD + 255) / 256;
    if (bias_grad_lds(B, dD) > 64 * 1024) return hipErrorInvalidValue;
    bias_grad_kernel<<<dim3(fix_blocks + (dM + 3) / 4), 256, bias_grad_lds(B, dD), st>>>(O, T, F, b, df, db, dp, B, dM, dD, P, norm, Norm, fix_blocks);
    return hipGetLastError();
}

}  // namespace aefft
