// Operator form of the FFT-mode training step (gfx950): the batch-dependent and batch-contracted pieces.
//
// The FFT-mode network is linear -- the activation is the identity (backproplib.cu:38-51) and conv_k / pool_fft are linear
// maps (fft_backproplib.cu:162-189, 87-157) -- so every spectrum a training step looks at is an affine function of the frame's
// input spectrum x_b on pair 0's grid:
//       X_l,b[s] = A_l[s] [x_b[u]; 1],     O_l,b[t] = O^_l[t] [x_b[u']; 1]            (u, u' = the grid-0 bins s, t map to)
// A_l [dD_l x OPC] and O^_l come out of the ordinary forward run on OPC basis frames (unit inputs + the zero input, whose
// response is the bias terms).  With M^[u] = sum_b [x_b;1][x_b;1]^H the per-frame sums of gradient_k_io and mse_fft
// (fft_backproplib.cu:395-498) become small per-bin matrix products in which the batch never appears again:
//       S_l[s]   = sum_b (O_l,b - X_l,b) X_l,b^H            = (O^_l[t] [s on the support] - A_l[s]) M^[u] A_l[s]^H
//       es_l     = sum_b (O_l,b - X_l,b)(0,0)               = ((O^_l - A_l) M^[.][OPC-1])(0,0)
//       mse_l    = mean_b sum_s w |X_l,b - O'_l,b|^2 / ...  = sum_s w tr(R M^ R^H) / ...,   R = A_l - F'(C' A_l / dM + b^) / dD - p^
// Same sums, batch contracted first (float32 rounding only); parity: tests/test_gpu_fft_path.py::test_step_*.
#include "../../include/aefft.h"
#include "internal.h"
#include <hip/hip_ext.h>
#include "device_util.h"
#include "opform_device.h"
#include <algorithm>
#include <type_traits>

namespace aefft {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
// (complex multiply-accumulates as two v_pk_fma_f32 with op_sel / neg modifiers -- no swizzle instructions, bit-identical -- were measured in round 4:
// the tail launch 28 -> 34 us, the step 0.160 -> 0.166 ms; these kernels wait on memory round trips, not on VALU issue slots: DESIGN.md section 6)
__device__ __forceinline__ void cfma2(float2& acc, float2 a, float2 b) { acc.x = fmaf(a.x, b.x, acc.x); acc.x = fmaf(-a.y, b.y, acc.x); acc.y = fmaf(a.x, b.y, acc.y); acc.y = fmaf(a.y, b.x, acc.y); }
__device__ __forceinline__ void cfmac(float2& acc, float2 a, float2 b) { acc.x = fmaf(a.x, b.x, acc.x); acc.x = fmaf(a.y, b.y, acc.x); acc.y = fmaf(a.y, b.x, acc.y); acc.y = fmaf(-a.x, b.y, acc.y); }   // += a * conj(b)
// element `e` of a base with a 32-bit BYTE offset: with a uniform base the load takes the scalar-base + 32-bit lane offset form (callers
// guarantee e * 8 < 2^32)
__device__ __forceinline__ float2 ld8(const float2* base, unsigned e) { return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base) + e * 8u); }
__device__ __forceinline__ void st8(float2* base, unsigned e, float2 v) { *reinterpret_cast<float2*>(reinterpret_cast<char*>(base) + e * 8u) = v; }

// r M^ r^H for one row r of an operator, with M^ in its stored CENTRED form (msgrad_kernel): r~_3 = r_3 + sum_j r_j xbar_j, then
// sum_{j,k<3} r_j M[j][k] conj(r_k) + B |r~_3|^2.  Mget(e): entry e = j*OPC + k of M^ at the row's bin.
template <typename MG> __device__ __forceinline__ float quad_centred(const float2 (&r)[OPC], MG Mget)
{
    float2 r3 = r[OPC - 1];
#pragma unroll
    for (int j = 0; j < OPC - 1; ++j) cfma2(r3, r[j], Mget(j * OPC + OPC - 1));
    float part = Mget(OPC * OPC - 1).x * (r3.x * r3.x + r3.y * r3.y);
#pragma unroll
    for (int k = 0; k < OPC - 1; ++k) {
        float2 v = make_float2(0.f, 0.f);
#pragma unroll
        for (int k2 = 0; k2 < OPC - 1; ++k2) cfmac(v, Mget(k * OPC + k2), r[k2]);      // M[k][k2] conj(r_k2)
        part += r[k].x * v.x - r[k].y * v.y;
    }
    return part;
}

// ------------------------------------------------------------------------------------------
// basis frames: A_0[j][d][u] = 1 if j == d < D0 else 0 (column OPC-1 = the zero input)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void basis_fill_kernel(float2* __restrict__ A0, int D0, long P0)
{
    const long total = (long)OPC * D0 * P0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pl = i / P0;
        const int j = (int)(pl / D0), d = (int)(pl - (long)j * D0);
        A0[i] = make_float2((j == d && j < OPC - 1) ? 1.f : 0.f, 0.f);
    }
}
hipError_t launch_basis_fill(float2* A0, int D0, long P0, hipStream_t st)
{
    if (D0 < 1 || D0 > OPC - 1) return hipErrorInvalidValue;
    basis_fill_kernel<<<dim3(1024), 256, 0, st>>>(A0, D0, P0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Batch moments and S_l, es_l of every pair in ONE launch (msgrad_kernel).
//
// CENTRED moments.  With xbar[u] = mean_b x_b[u] and dl_b = x_b - xbar the affine form X_l,b = A [x_b; 1] reads
// A_x dl_b + (A_x xbar + A_1): the operators keep their linear columns and the affine column becomes the response to the MEAN
// frame, A~_1 = A_1 + sum_j A_j xbar_j.  In that basis the batch moment matrix is block diagonal, [sum_b dl dl^H, 0; 0, B], and
//       S   = sum_b (O_b - X_b) X_b^H = E_x (sum dl dl^H) A_x^H + B E~_1 A~_1^H,          E = O^ - A
//       mse = sum_b |R [x_b; 1]|^2    = R_x (sum dl dl^H) R_x^H + B |R~_1|^2
// are sums of terms that are each as small as the per-frame quantities themselves.  The uncentred form M^ = sum [x;1][x;1]^H
// cancels catastrophically wherever the frames share a mean that R (a trained net) or E maps to almost zero -- the DC bin of
// any video, and every bin of the smooth component the synthetic frames share.
// M^ as stored ([OPC*OPC][P0], consumed by the MSE kernels): (j,k<3) sum_b dl_j conj(dl_k); (j,3) xbar_j; (3,j) conj(xbar_j); (3,3) B.
// The sums use frame 0 as a provisional mean (one pass: sum (x-K), sum (x-K)(x-K)^H, then the exact shift to the true mean).
//
// Workgroup = BT consecutive bins of one pair's grid.  Phase A: threads (bin, frame slice) accumulate the moments of the tile's
// bins from the input spectra (every pair's workgroups form the moments of their own bins: no workgroup waits for another; the
// workgroups of pair 0, whose grid is the moments' grid, also store them).  Phase B: threads (bin, row a): U[a][.] = E~ M,
// S[a][b] = sum_k U[a][k] conj(A~[b][k]) for every b.  All global loads of a workgroup are issued up front (one round trip).
// ------------------------------------------------------------------------------------------
#ifndef AEFFT_X_MSGRAD_W
#define AEFFT_X_MSGRAD_W 1
#endif
// FC: frames per batch of loads.  8 -- every frame of a pair-0 thread at cfg3 in ONE round trip, 118 registers, 4 workgroups per CU: the launch is one
// round of 976 workgroups there.  4 -- 96 registers, 5 per CU: for launches of several rounds (cfg5: 3 940 workgroups), which are bound by their slots.
template <int FC>
__global__ __launch_bounds__(256, AEFFT_X_MSGRAD_W) void msgrad_kernel(const SgradGroup g)
{
    AEFFT_WGTIME(0);
    extern __shared__ float2 sh[];
    const int tid = threadIdx.x;
    int p = g.n - 1;                                                // pair n-1 owns the first workgroups (most dependent steps per workgroup), pair 0 the last
#pragma unroll
    for (int i = 6; i >= 0; --i) if (i < g.n - 1 && (int)blockIdx.x >= g.start[i]) p = i;
    const OpPair q = g.q[p];                                        // (by value: one bulk scalar load, not one per field use)
    const int BT = g.bt[p], RT = 256 / BT;                          // bins per workgroup; row threads == frame slices
    const int dD = q.dD, D0 = g.D0, B = g.B;
    const long s0 = (long)(blockIdx.x - g.start[p]) * BT;
    const int bl = tid % BT, ry = tid / BT;
    const long s = s0 + bl;
    const bool ok = s < q.P;
    const long sc = ok ? s : q.P - 1;
    const unsigned u = (unsigned)map_up(sc, q.Nx, q.Ny, g.Nx0, g.Ny0);
    const unsigned P0 = (unsigned)g.P0;
    const int nA = OPC * dD * BT;
    float2* red = sh;                                               // [RT][9][BT]     per-slice partial moments; dead after the slice sums, then:
    float2* As = sh;                                                // [OPC][dD][BT]   A  (affine column centred in place)
    float2* Os = As + nA;                                           // [OPC][dD][BT]   O^ on the support, 0 elsewhere
    float2* mom = sh + max(9 * 256, 2 * nA);                        // [9][BT]         slice sums: s_0..2, m_00 m_01 m_02 m_11 m_12 m_22
    float2* Kl = mom + 9 * BT;                                      // [3][BT]         provisional mean (frame 0)
    float2* Ms = Kl + 3 * BT;                                       // [16][BT]        M^ of the tile's bins (layout above)
    // ---- every global load of the workgroup: the A and O^ tiles, then this thread's frames ----
    float2 va[4], vo[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int idx = min(w * 256 + tid, nA - 1);
        const int kd = idx / BT, b2 = idx - kd * BT;
        const long s2 = min(s0 + b2, q.P - 1);
        const int t2 = crop_dest32(s2, q.Nx, q.Ny, q.NxO, q.NyO);
        va[w] = q.A[(long)kd * q.P + s2];
        vo[w] = q.O[(long)kd * q.PO + (t2 >= 0 ? t2 : 0)];
        if (t2 < 0) vo[w] = make_float2(0.f, 0.f);
    }
    float2 K[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) K[d] = d < D0 ? ld8(g.Xf + (size_t)d * P0, u) : make_float2(0.f, 0.f);
    float2 sx[3], mm[6];
#pragma unroll
    for (int d = 0; d < 3; ++d) sx[d] = make_float2(0.f, 0.f);
#pragma unroll
    for (int e = 0; e < 6; ++e) mm[e] = make_float2(0.f, 0.f);
    for (int b0 = ry; b0 < B; b0 += RT * FC) {
        float2 x[FC][3];
#pragma unroll
        for (int f = 0; f < FC; ++f)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int bb = min(b0 + f * RT, B - 1);
                x[f][d] = d < D0 ? ld8(g.Xf + ((size_t)bb * D0 + d) * P0, u) : make_float2(0.f, 0.f);
            }
#pragma unroll
        for (int f = 0; f < FC; ++f) {
            if (b0 + f * RT >= B) break;
            const float2 y0 = make_float2(x[f][0].x - K[0].x, x[f][0].y - K[0].y), y1 = make_float2(x[f][1].x - K[1].x, x[f][1].y - K[1].y),
                         y2 = make_float2(x[f][2].x - K[2].x, x[f][2].y - K[2].y);
            sx[0].x += y0.x; sx[0].y += y0.y; sx[1].x += y1.x; sx[1].y += y1.y; sx[2].x += y2.x; sx[2].y += y2.y;
            cfmac(mm[0], y0, y0); cfmac(mm[1], y0, y1); cfmac(mm[2], y0, y2); cfmac(mm[3], y1, y1); cfmac(mm[4], y1, y2); cfmac(mm[5], y2, y2);
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) red[(ry * 9 + d) * BT + bl] = sx[d];
#pragma unroll
    for (int e = 0; e < 6; ++e) red[(ry * 9 + 3 + e) * BT + bl] = mm[e];
    if (ry == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) Kl[d * BT + bl] = K[d];
    }
    __syncthreads();
    AEFFT_WGSTAMP(0, 0);
#pragma unroll 3
    for (int i = tid; i < 9 * BT; i += 256) {                       // slices -> one sum, in slice order (deterministic)
        const int e = i / BT, b2 = i - e * BT;
        float2 a = red[e * BT + b2];
#pragma unroll 4
        for (int r2 = 1; r2 < RT; ++r2) { const float2 v = red[(r2 * 9 + e) * BT + b2]; a.x += v.x; a.y += v.y; }
        mom[i] = a;
    }
    __syncthreads();
    AEFFT_WGSTAMP(0, 1);
    // the tiles take the place of the per-slice partial sums (held in registers since the first round trip)
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int idx = w * 256 + tid; if (idx < nA) { As[idx] = va[w]; Os[idx] = vo[w]; } }
    for (int i0 = 1024; i0 < nA; i0 += 1024) {                      // tiles larger than 4 elements per thread (dD > 32): further round trips
        float2 wa[4], wo[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int idx = min(i0 + w * 256 + tid, nA - 1);
            const int kd = idx / BT, b2 = idx - kd * BT;
            const long s2 = min(s0 + b2, q.P - 1);
            const int t2 = crop_dest32(s2, q.Nx, q.Ny, q.NxO, q.NyO);
            wa[w] = q.A[(long)kd * q.P + s2];
            wo[w] = t2 >= 0 ? q.O[(long)kd * q.PO + t2] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int idx = i0 + w * 256 + tid; if (idx < nA) { As[idx] = wa[w]; Os[idx] = wo[w]; } }
    }
    const float fB = (float)B, iB = 1.0f / (float)B;
    const bool store_m = g.Mout != nullptr && p == 0;                               // (uniform) pair 0: its grid is the moments' grid
#pragma unroll 4
    for (int i = tid; i < 16 * BT; i += 256) {
        const int e = i / BT, b2 = i - e * BT;
        const int j = e >> 2, k = e & 3;
        float2 v;
        if (j == 3 && k == 3) v = make_float2(fB, 0.f);
        else if (j == 3 || k == 3) {
            const int d = j == 3 ? k : j;
            const float2 sd = mom[d * BT + b2], kd = Kl[d * BT + b2];
            v = make_float2(kd.x + sd.x * iB, kd.y + sd.y * iB);                     // the mean
            if (j == 3) v.y = -v.y;
        } else {
            const int lo = j < k ? j : k, hi = j < k ? k : j;
            const int idx = lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);                   // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
            float2 m2 = mom[(3 + idx) * BT + b2];
            const float2 sl = mom[lo * BT + b2], sh2 = mom[hi * BT + b2];
            float2 c = make_float2(0.f, 0.f);
            cfmac(c, sl, sh2);                                                       // s_lo conj(s_hi)
            m2.x -= c.x * iB; m2.y -= c.y * iB;
            if (j == k) m2.y = 0.f;
            if (j > k) m2.y = -m2.y;
            v = m2;
        }
        Ms[i] = v;
        if (store_m && s0 + b2 < q.P) g.Mout[(size_t)e * P0 + (size_t)(s0 + b2)] = v;
    }
    __syncthreads();
    AEFFT_WGSTAMP(0, 2);
    // ---- centre the affine columns: A~_1 = A_1 + sum_j A_j xbar_j (the same for O^) ----
    for (int a = ry; a < dD; a += RT) {
        float2 a3 = As[((OPC - 1) * dD + a) * BT + bl], o3 = Os[((OPC - 1) * dD + a) * BT + bl];
#pragma unroll
        for (int j = 0; j < OPC - 1; ++j) {
            const float2 xb = Ms[(j * OPC + 3) * BT + bl];
            cfma2(a3, As[(j * dD + a) * BT + bl], xb);
            cfma2(o3, Os[(j * dD + a) * BT + bl], xb);
        }
        As[((OPC - 1) * dD + a) * BT + bl] = a3; Os[((OPC - 1) * dD + a) * BT + bl] = o3;
    }
    __syncthreads();
    AEFFT_WGSTAMP(0, 3);
    for (int a = ry; a < dD; a += RT) {
        float2 E[OPC], U[OPC];
#pragma unroll
        for (int j = 0; j < OPC; ++j) { const float2 av = As[(j * dD + a) * BT + bl], ov = Os[(j * dD + a) * BT + bl]; E[j] = make_float2(ov.x - av.x, ov.y - av.y); }
#pragma unroll
        for (int k = 0; k < OPC - 1; ++k) {
            U[k] = make_float2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < OPC - 1; ++j) cfma2(U[k], E[j], Ms[(j * OPC + k) * BT + bl]);
        }
        U[OPC - 1] = make_float2(E[OPC - 1].x * fB, E[OPC - 1].y * fB);
        if (!ok) continue;
        if (s == 0) { q.es[2 * a] = U[OPC - 1].x; q.es[2 * a + 1] = U[OPC - 1].y; }
        float2* dst = q.S + (long)a * dD * q.P + s;
#pragma unroll 4
        for (int b = 0; b < dD; ++b) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < OPC; ++k) cfmac(acc, U[k], As[(k * dD + b) * BT + bl]);
            dst[(long)b * q.P] = acc;
        }
    }
}
// bins per workgroup: 256 threads = bins x row threads.  Cache-sized grids: as many row threads as rows (shortest dependent path per
// workgroup); HBM-sized grids (no pooling): at least 16 bins, so that the plane accesses are whole 128-byte lines, while the tiles fit
static int msgrad_bt(int dD, long P)
{
    int bt = dD <= 4 ? 64 : (dD <= 8 ? 32 : (dD <= 16 ? 16 : 8));
    if (P >= 32768 && bt < 16 && (size_t)2 * OPC * dD * 16 * sizeof(float2) <= 64 * 1024) bt = 16;
    return bt;
}
hipError_t launch_msgrad_group(SgradGroup& g, hipStream_t st)
{
    if (g.n < 1 || g.n > 8 || !g.Xf || g.B < 1 || g.D0 < 1 || g.D0 > OPC - 1 || g.P0 >= (1L << 28)) return hipErrorInvalidValue;
    long total = 0;
    size_t lds = 0;
    for (int i = g.n - 1; i >= 0; --i) {                               // (start[] is DEscending in i: see the kernel's lookup)
        const int bt = msgrad_bt(g.q[i].dD, g.q[i].P);
        g.bt[i] = bt;
        g.start[i] = (int)total;
        total += (g.q[i].P + bt - 1) / bt;
        lds = std::max(lds, sizeof(float2) * (std::max((size_t)2 * OPC * g.q[i].dD * bt, (size_t)9 * 256) + (size_t)(9 + 3 + 16) * bt));
    }
    if (total >= (1L << 31) || lds > 150 * 1024) return hipErrorInvalidValue;
    g.start[g.n] = (int)total;
    const bool rounds = total > 1024;                                // (more workgroups than 4 per CU hold at once)
    if (lds > 64 * 1024) {
        const hipError_t e = rounds ? hipFuncSetAttribute(reinterpret_cast<const void*>(msgrad_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                    : hipFuncSetAttribute(reinterpret_cast<const void*>(msgrad_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (rounds) msgrad_kernel<4><<<dim3((unsigned)total), 256, lds, st>>>(g);
    else msgrad_kernel<8><<<dim3((unsigned)total), 256, lds, st>>>(g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// out[b][r][s] = sum_j A[j][r][s] x^_b[j][u(s)]   (per-frame spectra from an operator: the reconstruction's input, layer exports)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void op_expand_kernel(const float2* __restrict__ A, const float2* __restrict__ Xf, float2* __restrict__ out,
                                                        int B, int D0, int dD, int Nx0, int Ny0, int Nx, int Ny)
{
    const long P = (long)Nx * (Ny / 2 + 1), P0 = (long)Nx0 * (Ny0 / 2 + 1);
    const long s = (long)blockIdx.x * 64 + threadIdx.x;
    const int pl = blockIdx.y * 4 + threadIdx.y;                      // (b, r)
    if (s >= P || pl >= B * dD) return;
    const int b = pl / dD, r = pl - b * dD;
    const long u = map_up(s, Nx, Ny, Nx0, Ny0);
    float2 acc = A[((long)(OPC - 1) * dD + r) * P + s];              // the affine column times 1
#pragma unroll
    for (int j = 0; j < OPC - 1; ++j)
        if (j < D0) cfma2(acc, A[((long)j * dD + r) * P + s], Xf[((long)b * D0 + j) * P0 + u]);
    out[((long)b * dD + r) * P + s] = acc;
}
hipError_t launch_op_expand(const float2* A, const float2* Xf, float2* out, int B, int D0, int dD, int Nx0, int Ny0, int Nx, int Ny, hipStream_t st)
{
    const long P = (long)Nx * (Ny / 2 + 1);
    if (B < 1 || dD < 1 || D0 < 1 || D0 > OPC - 1 || (long)B * dD > 4L * 65535) return hipErrorInvalidValue;
    op_expand_kernel<<<dim3((unsigned)((P + 63) / 64), (unsigned)((B * dD + 3) / 4)), dim3(64, 4), 0, st>>>(A, Xf, out, B, D0, dD, Nx0, Ny0, Nx, Ny);
    return hipGetLastError();
}
hipError_t launch_recon_expand(const float2* O0, const float2* Xf, float2* Of, int B, int D0, int Nx0, int Ny0, int NxO, int NyO, hipStream_t st)
{
    return launch_op_expand(O0, Xf, Of, B, D0, D0, Nx0, Ny0, NxO, NyO, st);
}

constexpr int CH_WL = 6144;       // (CH_VMAX: internal.h)

// out[r][c] = scale * sum_k W[r][k] V[k][c] (+ bias[r] NN on the affine column at the DC bin); W = a row-major matrix of the
// item's packed record, read straight from global memory (contiguous rows, L2-resident), V in LDS.  Thread = one output.
__device__ __forceinline__ void chain_stage_rec(const float2* __restrict__ Ws, const float2* __restrict__ Vin, float2* __restrict__ Vout,
                                                int R, int K, float scale, const float* __restrict__ bias, float NN, bool dc,
                                                float2* __restrict__ out, long outP, long outS)
{
    for (int o = threadIdx.x; o < R * OPC; o += 256) {
        const int r = o / OPC, col = o - r * OPC;
        float2 acc = make_float2(0.f, 0.f);
        const float2* wr = Ws + r * K;
        // the row (K contiguous elements of the record) in groups of 16 loads: all in flight before the first use
        auto grp = [&](int k0, auto NU) {
            constexpr int U = decltype(NU)::value;
            float2 w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = wr[k0 + u];
#pragma unroll
            for (int u = 0; u < U; ++u) cfma2(acc, w[u], Vin[(k0 + u) * OPC + col]);
        };
        int k0 = 0;
        for (; k0 + 16 <= K; k0 += 16) grp(k0, std::integral_constant<int, 16>{});
        if (k0 + 8 <= K) { grp(k0, std::integral_constant<int, 8>{}); k0 += 8; }
        if (k0 + 4 <= K) { grp(k0, std::integral_constant<int, 4>{}); k0 += 4; }
        if (k0 + 2 <= K) { grp(k0, std::integral_constant<int, 2>{}); k0 += 2; }
        if (k0 < K) grp(k0, std::integral_constant<int, 1>{});
        acc.x *= scale; acc.y *= scale;
        if (dc && col == OPC - 1) acc.x += bias[r] * NN;
        Vout[r * OPC + col] = acc;
        if (out) out[((long)col * R + r) * outP + outS] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// post-update MSE of every pair (fft_backproplib.cu:1460-1463 + 480-498) in operator form, one launch.
// Workgroup = BT consecutive bins of one pair x (256 / BT) row threads.  Phase 1: T = C' A / dM + b^ (dM x OPC per bin, rows over
// the row threads, A staged in LDS); phase 2: R = A - F' T / dD - p^ (dD x OPC) and sum_a R[a] M^ R[a]^H; block sum -> one
// atomic per workgroup into MSE_SLOTS accumulators (launch_mse_finish sums them).  The updated spectra are read exactly once.
// ------------------------------------------------------------------------------------------
template <int BT>
__device__ __forceinline__ void opmse_body(const OpMseGroup& g, int p, float2* sh)
{
    constexpr int RT = 256 / BT;
    const OpMsePair q = g.q[p];                                     // (by value: one bulk scalar load)
    const int dD = q.dD, dM = q.dM;
    float2* As = sh;
    float2* Ts = sh + (size_t)OPC * dD * BT;
    const long blk = blockIdx.x - g.start[p];
    const int bl = threadIdx.x % BT, ry = threadIdx.x / BT;
    const long s = blk * BT + bl;
    const bool ok = s < q.P;
    const long sc = ok ? s : q.P - 1;
    // the 4x4 moments of the tile's bins, once per workgroup (every row thread needs them in phase 2: 16 loads each otherwise);
    // issued here, in the same round trip as the A tile
    float2* Ms = Ts + (size_t)dM * OPC * BT + 8 + 256 * OPC;           // [OPC*OPC][BT], behind Rs
    float2 mval = make_float2(0.f, 0.f);
    if ((int)threadIdx.x < OPC * OPC * BT) {
        const int e = threadIdx.x / BT, b2 = threadIdx.x - e * BT;
        const long s2 = min(blk * BT + b2, q.P - 1);
        mval = g.Mhat[(long)e * g.P0 + map_up(s2, q.Nx, q.Ny, g.Nx0, g.Ny0)];
    }
    for (int i0 = 0; i0 < OPC * dD * BT; i0 += 256 * 4) {
        float2 v[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int idx = min(i0 + w * 256 + (int)threadIdx.x, OPC * dD * BT - 1);
            const int kd = idx / BT, b2 = idx - kd * BT;
            const long s2 = min(blk * BT + b2, q.P - 1);
            v[w] = q.A[(long)kd * q.P + s2];
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int idx = i0 + w * 256 + threadIdx.x; if (idx < OPC * dD * BT) As[idx] = v[w]; }
    }
    if ((int)threadIdx.x < OPC * OPC * BT) Ms[threadIdx.x] = mval;
    __syncthreads();
    const float NN = (float)q.Nx * (float)q.Ny;
    const float idM = 1.0f / (float)dM, idD = 1.0f / (float)dD;
    for (int m = ry; m < dM; m += RT) {
        float2 t[OPC];
#pragma unroll
        for (int k = 0; k < OPC; ++k) t[k] = make_float2(0.f, 0.f);
        const float2* Cp = q.C + (long)m * dD * q.P + sc;
        {
            auto grp = [&](int d0, auto NU) {
                constexpr int U = decltype(NU)::value;
                float2 c[U];
#pragma unroll
                for (int w = 0; w < U; ++w) c[w] = Cp[(long)(d0 + w) * q.P];
#pragma unroll
                for (int w = 0; w < U; ++w)
#pragma unroll
                    for (int k = 0; k < OPC; ++k) cfma2(t[k], c[w], As[(k * dD + d0 + w) * BT + bl]);
            };
            int d0 = 0;
            for (; d0 + 16 <= dD; d0 += 16) grp(d0, std::integral_constant<int, 16>{});
            if (d0 + 8 <= dD) { grp(d0, std::integral_constant<int, 8>{}); d0 += 8; }
            if (d0 + 4 <= dD) { grp(d0, std::integral_constant<int, 4>{}); d0 += 4; }
            if (d0 + 2 <= dD) { grp(d0, std::integral_constant<int, 2>{}); d0 += 2; }
            if (d0 < dD) grp(d0, std::integral_constant<int, 1>{});
        }
#pragma unroll
        for (int k = 0; k < OPC; ++k) { t[k].x *= idM; t[k].y *= idM; }
        if (s == 0) t[OPC - 1].x += q.b[m] * NN;
#pragma unroll
        for (int k = 0; k < OPC; ++k) Ts[(m * OPC + k) * BT + bl] = t[k];
    }
    __syncthreads();
    float part = 0.f;
    {
        // rows a over the row threads; when there are more row threads than rows, KS of them share a row and split the sum over m
        int KS = 1;
        while (KS < 8 && dD * KS * 2 <= RT) KS *= 2;
        const int ks = ry % KS, ar = ry / KS;
        float2* Rs = Ts + (size_t)dM * OPC * BT + 8;                      // partial rows of the split: [RT][OPC][BT] (KS > 1 only)
        for (int a0 = 0; a0 < dD; a0 += RT / KS) {
            const int a = a0 + ar;
            const bool act = a < dD;
            const int ac = act ? a : dD - 1;
            float2 r[OPC];
#pragma unroll
            for (int k = 0; k < OPC; ++k) r[k] = make_float2(0.f, 0.f);
            const float2* Fp = q.F + (long)ac * dM * q.P + sc;
            const int mq = (dM + KS - 1) / KS;
            const int mb = ks * mq, me = min(dM, mb + mq);
            {
                auto grp = [&](int m0, auto NU) {
                    constexpr int U = decltype(NU)::value;
                    float2 f[U];
#pragma unroll
                    for (int w = 0; w < U; ++w) f[w] = Fp[(long)(m0 + w) * q.P];
#pragma unroll
                    for (int w = 0; w < U; ++w)
#pragma unroll
                        for (int k = 0; k < OPC; ++k) cfma2(r[k], f[w], Ts[((m0 + w) * OPC + k) * BT + bl]);
                };
                int m0 = mb;
                for (; m0 + 16 <= me; m0 += 16) grp(m0, std::integral_constant<int, 16>{});
                if (m0 + 8 <= me) { grp(m0, std::integral_constant<int, 8>{}); m0 += 8; }
                if (m0 + 4 <= me) { grp(m0, std::integral_constant<int, 4>{}); m0 += 4; }
                if (m0 + 2 <= me) { grp(m0, std::integral_constant<int, 2>{}); m0 += 2; }
                if (m0 < me) grp(m0, std::integral_constant<int, 1>{});
            }
            if (KS > 1) {                                                // uniform per workgroup
#pragma unroll
                for (int k = 0; k < OPC; ++k) Rs[(ry * OPC + k) * BT + bl] = r[k];
                __syncthreads();
                if (ks == 0) {
                    for (int k2 = 1; k2 < KS; ++k2)
#pragma unroll
                        for (int k = 0; k < OPC; ++k) { const float2 v = Rs[((ry + k2) * OPC + k) * BT + bl]; r[k].x += v.x; r[k].y += v.y; }
                }
                __syncthreads();
            }
            if (!act || ks != 0) continue;
#pragma unroll
            for (int k = 0; k < OPC; ++k) {
                const float2 av = As[(k * dD + a) * BT + bl];
                r[k] = make_float2(av.x - r[k].x * idD, av.y - r[k].y * idD);
            }
            if (s == 0) r[OPC - 1].x -= q.p[a] * NN;
            part += quad_centred(r, [&](int e) { return Ms[e * BT + bl]; });
        }
        const int nyr = q.Ny / 2 + 1;
        const int j = (int)((unsigned)sc % (unsigned)nyr);
        part *= !ok ? 0.f : ((j > 0 && j < nyr - 1) ? 2.f : 1.f);          // Hermitian half-plane: interior columns count twice
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    __syncthreads();
    float* red = reinterpret_cast<float*>(Ts + (size_t)dM * OPC * BT);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (red[0] + red[1]) + (red[2] + red[3]);
        if (tot != 0.f) atomicAdd(q.slots + (blockIdx.x % MSE_SLOTS) * MSE_SLOT_STRIDE, tot * q.scale);
    }
}

// The same MSE from the collapsed operator G' = F'.C' / (dM dD) of the UPDATED weights ([dD][dD] planes: the spectrum of the
// (2Nk-1)^2-tap kernel f' (*) c', written by the spectra launch -- gspec_gbody, pruned_kernels.hip):  R = (I - G') A - beta^, with
// beta^ = Nx Ny (p' + F'(0,0) b' / dD) on the affine column at the DC bin (the two bias terms of conv_k o conv_k, fft_backproplib.cu:
// 183-184).  dD*dD instead of 2*dM*dD planes are read, and the difference to the identity is taken on the matrix elements (1 - G'_aa)
// before anything is multiplied by the signal.  Workgroup = BT bins x (256 / BT) row threads; the first 16 elements of a thread's row
// of G' are requested in the same round trip as the A tile and the moments.
__device__ __forceinline__ void opmse_gbody(const OpMseGroup& g, int p, float2* sh)
{
    const OpMsePair q = g.q[p];
    const int BT = g.bt[p], RT = 256 / BT, dD = q.dD;
    const int tid = threadIdx.x;
    float2* As = sh;                                                 // [OPC][dD][BT]
    float2* Ms = As + (size_t)OPC * dD * BT;                         // [OPC*OPC][BT]
    float* red = reinterpret_cast<float*>(Ms + OPC * OPC * BT);      // [4]
    const long s0 = ((long)blockIdx.x - g.start[p]) * BT;
    const int bl = tid % BT, ry = tid / BT;
    const long s = s0 + bl;
    const bool ok = s < q.P;
    const long sc = ok ? s : q.P - 1;
    const int nA = OPC * dD * BT, nM = OPC * OPC * BT;
    float2 va[4], mv[4], gpre[16];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int idx = min(w * 256 + tid, nA - 1);
        const int kd = idx / BT, b2 = idx - kd * BT;
        va[w] = q.A[(long)kd * q.P + min(s0 + b2, q.P - 1)];
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w * 256 >= nM) break;                                    // uniform
        const int idx = min(w * 256 + tid, nM - 1);
        const int e = idx / BT, b2 = idx - e * BT;
        mv[w] = g.Mhat[(long)e * g.P0 + map_up(min(s0 + b2, q.P - 1), q.Nx, q.Ny, g.Nx0, g.Ny0)];
    }
    {
        // (element offsets in 32 bits off the uniform base: one address register per load in flight; launch_opmse_group checks dD*dD*P*8 < 2^32)
        const unsigned P = (unsigned)q.P, e0 = (unsigned)(min(ry, dD - 1) * dD) * P + (unsigned)sc;
#pragma unroll
        for (int w = 0; w < 16; ++w) gpre[w] = ld8(q.G, e0 + (unsigned)min(w, dD - 1) * P);
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int idx = w * 256 + tid; if (idx < nA) As[idx] = va[w]; }
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int idx = w * 256 + tid; if (idx < nM) Ms[idx] = mv[w]; }
    for (int i0 = 1024; i0 < nA; i0 += 1024) {
        float2 wa[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int idx = min(i0 + w * 256 + tid, nA - 1);
            const int kd = idx / BT, b2 = idx - kd * BT;
            wa[w] = q.A[(long)kd * q.P + min(s0 + b2, q.P - 1)];
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int idx = i0 + w * 256 + tid; if (idx < nA) As[idx] = wa[w]; }
    }
    __syncthreads();
    const float NN = (float)q.Nx * (float)q.Ny;
    float part = 0.f;
    // one row a of R = (I - G') A - beta^ and its quadratic form; FIRST: the row whose first 16 elements of G' were requested up front
    auto row = [&](const int a, auto FIRST) {
        float2 r[OPC];
#pragma unroll
        for (int k = 0; k < OPC; ++k) r[k] = make_float2(0.f, 0.f);
        const float2* Gp = q.G + (long)a * dD * q.P + sc;
        auto use = [&](int d, float2 gv) {
            const float2 im = make_float2((d == a ? 1.f : 0.f) - gv.x, -gv.y);          // (I - G')[a][d]
#pragma unroll
            for (int k = 0; k < OPC; ++k) cfma2(r[k], im, As[(k * dD + d) * BT + bl]);
        };
        int d0 = 0;
        if constexpr (decltype(FIRST)::value) {
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                if (w < dD) use(w, gpre[w]);
                if (w & 1) __builtin_amdgcn_sched_barrier(0);       // (keeps the LDS reads of all 16 uses from being hoisted into live registers)
            }
            d0 = 16;
        }
        auto grp = [&](int dd, auto NU) {
            constexpr int U = decltype(NU)::value;
            float2 gv[U];
#pragma unroll
            for (int w = 0; w < U; ++w) gv[w] = Gp[(long)(dd + w) * q.P];
#pragma unroll
            for (int w = 0; w < U; ++w) {
                use(dd + w, gv[w]);
                if (w & 1) __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (; d0 + 8 <= dD; d0 += 8) grp(d0, std::integral_constant<int, 8>{});
        if (d0 + 4 <= dD) { grp(d0, std::integral_constant<int, 4>{}); d0 += 4; }
        if (d0 + 2 <= dD) { grp(d0, std::integral_constant<int, 2>{}); d0 += 2; }
        if (d0 < dD) grp(d0, std::integral_constant<int, 1>{});
        if (s == 0) {
            // beta[a] = p'[a] + sum_m F'[a][m](0,0) b'[m] / dD  (F' at the DC bin is real: the sum of the taps)
            float acc = 0.f;
            for (int m = 0; m < q.dM; ++m) acc = fmaf(q.Fdc[((long)a * q.dM + m) * q.fdc_stride].x, q.b[m], acc);
            r[OPC - 1].x -= (q.p[a] + acc / (float)dD) * NN;
        }
        if (ok) {
            const int nyr = q.Ny / 2 + 1;
            const int j = (int)((unsigned)sc % (unsigned)nyr);
            part += quad_centred(r, [&](int e) { return Ms[e * BT + bl]; }) * ((j > 0 && j < nyr - 1) ? 2.f : 1.f);      // Hermitian half-plane: interior columns count twice
        }
    };
    if (ry < dD) row(ry, std::true_type{});
    for (int a = ry + RT; a < dD; a += RT) row(a, std::false_type{});
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        const float tot = (red[0] + red[1]) + (red[2] + red[3]);
        if (tot != 0.f) atomicAdd(q.slots + (blockIdx.x % MSE_SLOTS) * MSE_SLOT_STRIDE, tot * q.scale);
    }
}

// Pairs with few maps (dM == DM = 8, dD <= 4: the outermost pair, which has the most bins): the generic body spends its time in
// three dependent memory round trips separated by barriers for a few loads each.  Here every global load of the workgroup -- the
// A tile, this thread's row of C', its row of F', the 4x4 moments -- is issued up front (one round trip), then the same two
// small products run out of LDS.  Workgroup = 32 bins x 8 row threads.
// element `e` of a uniform base with a 32-bit BYTE offset: the load takes the scalar-base + 32-bit lane offset form (callers
// guarantee e * 8 < 2^32)

template <int DM>
__device__ __forceinline__ void opmse_small_body(const OpMseGroup& g, int p, float2* sh)
{
    constexpr int BT = 32, RT = 256 / BT, DDMAX = 4;
    static_assert(DM == RT, "one row of C' per row thread");
    const OpMsePair q = g.q[p];
    const int dD = q.dD;
    float2* As = sh;                                                 // [OPC][dD][BT]
    float2* Ts = sh + OPC * DDMAX * BT;                              // [DM][OPC][BT]
    const long blk = blockIdx.x - g.start[p];
    const int bl = threadIdx.x % BT, ry = threadIdx.x / BT;
    const long s = blk * BT + bl;
    const bool ok = s < q.P;
    const long sc = ok ? s : q.P - 1;
    const int nA = OPC * dD * BT;                                    // <= 512: two elements per thread
    // (element offsets in 32 bits -- at most 32 planes of at most 2048 x 1025 bins here --, so the loads take the scalar-base +
    // 32-bit lane offset form and the 28 addresses cost one register each; M^ is Hermitian: upper triangle only)
    const unsigned P = (unsigned)q.P, scu = (unsigned)sc;
    float2 ast[2], c[DDMAX], f[DM], Mu[10];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const unsigned idx = (unsigned)min(w * 256 + (int)threadIdx.x, nA - 1);
        const unsigned kd = idx / BT, b2 = idx - kd * BT;
        ast[w] = ld8(q.A, kd * P + min((unsigned)blk * BT + b2, P - 1));
    }
#pragma unroll
    for (int d = 0; d < DDMAX; ++d) c[d] = ld8(q.C + (size_t)min(d, dD - 1) * P, (unsigned)(ry * dD) * P + scu);      // (uniform row base + one lane offset)
    const int a = min(ry, dD - 1);
#pragma unroll
    for (int m = 0; m < DM; ++m) f[m] = ld8(q.F + (size_t)m * P, (unsigned)(a * DM) * P + scu);
    {
        const unsigned u = (unsigned)map_up(sc, q.Nx, q.Ny, g.Nx0, g.Ny0), P0 = (unsigned)g.P0;
        int e = 0;
#pragma unroll
        for (int i = 0; i < OPC; ++i)
#pragma unroll
            for (int j = i; j < OPC; ++j) Mu[e++] = ld8(g.Mhat + (size_t)(i * OPC + j) * P0, u);
    }
#pragma unroll
    for (int w = 0; w < 2; ++w) { const int idx = w * 256 + threadIdx.x; if (idx < nA) As[idx] = ast[w]; }
    __syncthreads();
    const float NN = (float)q.Nx * (float)q.Ny;
    {
        float2 t[OPC];
#pragma unroll
        for (int k = 0; k < OPC; ++k) {
            t[k] = make_float2(0.f, 0.f);
#pragma unroll
            for (int d = 0; d < DDMAX; ++d) if (d < dD) cfma2(t[k], c[d], As[(k * dD + d) * BT + bl]);
            t[k].x *= 1.0f / (float)DM; t[k].y *= 1.0f / (float)DM;
        }
        if (s == 0) t[OPC - 1].x += q.b[ry] * NN;
#pragma unroll
        for (int k = 0; k < OPC; ++k) Ts[(ry * OPC + k) * BT + bl] = t[k];
    }
    __syncthreads();
    float part = 0.f;
    if (ry < dD) {
        const float idD = 1.0f / (float)dD;
        float2 r[OPC];
#pragma unroll
        for (int k = 0; k < OPC; ++k) r[k] = make_float2(0.f, 0.f);
#pragma unroll
        for (int m = 0; m < DM; ++m) {
#pragma unroll
            for (int k = 0; k < OPC; ++k) cfma2(r[k], f[m], Ts[(m * OPC + k) * BT + bl]);
            if (m & 1) __builtin_amdgcn_sched_barrier(0);           // (keeps the 32 LDS reads from being hoisted into 64 live registers)
        }
#pragma unroll
        for (int k = 0; k < OPC; ++k) {
            const float2 av = As[(k * dD + a) * BT + bl];
            r[k] = make_float2(av.x - r[k].x * idD, av.y - r[k].y * idD);
        }
        if (s == 0) r[OPC - 1].x -= q.p[a] * NN;
        // (upper triangle of the stored M^ in Mu, row-major: (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3); the lower one is its conjugate)
        part = quad_centred(r, [&](int e) {
            const int j = e >> 2, k = e & 3;
            const int lo = j < k ? j : k, hi = j < k ? k : j;
            const int idx = lo == 0 ? hi : (lo == 1 ? 3 + hi : (lo == 2 ? 5 + hi : 9));
            float2 v = Mu[idx];
            if (j > k) v.y = -v.y;
            return v;
        });
        const int nyr = q.Ny / 2 + 1;
        const int j = (int)((unsigned)sc % (unsigned)nyr);
        part *= !ok ? 0.f : ((j > 0 && j < nyr - 1) ? 2.f : 1.f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    float* red = reinterpret_cast<float*>(Ts + (size_t)DM * OPC * BT);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (red[0] + red[1]) + (red[2] + red[3]);
        if (tot != 0.f) atomicAdd(q.slots + (blockIdx.x % MSE_SLOTS) * MSE_SLOT_STRIDE, tot * q.scale);
    }
}

// ---- the per-bin item as a software pipeline over its record ----
// The item's 2L stages are small dependent products whose matrices do NOT depend on the running vector: only the multiply waits for the
// previous stage.  One output per thread and groups of 16 loads (chain_stage_rec) made every stage one to four memory round trips of its
// own -- 24 us per item at cfg3, the long pole of the tail launch.  Here a STEP is up to CH_NE elements per thread of one stage's matrix,
// thread <-> (row r, lane ks of KS <= 16 adjacent lanes: k = k0 + ks + KS u), every element of the record is loaded exactly once, and the loads
// of step i + CH_DEPTH are issued before step i is computed: after the first round trip the stages run out of registers and LDS.  A stage
// whose K range exceeds CH_NE * KS takes several steps (the row sums stay in registers); the KS partial sums of a row meet by lane exchange.
// What keeps the pipeline a pipeline (hipcc's s_waitcnt insertion): every load is unconditional (clamped addresses, masked in the product;
// the bias element rides along whether the item is the DC bin or not), so the number of loads in flight at each wait is static; the running
// vectors are addressed off ONE LDS base (a select between two pointers turns the reads into flat loads, which wait for everything).
// The step list is built by the host (chain_geometry).
#ifndef AEFFT_X_CHAINPIPE
#define AEFFT_X_CHAINPIPE 1
#endif
#ifndef AEFFT_X_CH_DEPTH
#define AEFFT_X_CH_DEPTH 4
#endif
// acc += a * b as two v_pk_fma_f32 (the four FMAs of cfma2 in the same order per component: the same bits).  The steps below are bound by
// their instruction count, not by latency -- where the latter holds (the stage-by-stage bodies) the packed form measured slower, DESIGN.md 6.
#ifndef AEFFT_X_CH_PK
#define AEFFT_X_CH_PK 1
#endif
__device__ __forceinline__ void cfma_pk(float2& acc, float2 a, float2 b)
{
#if AEFFT_X_CH_PK
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f r = {acc.x, acc.y};
    const v2f av = {a.x, a.y}, bv = {b.x, b.y};
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(r) : "v"(av), "v"(bv));                          // + (a.x b.x, a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(r) : "v"(av), "v"(bv));         // + (-a.y b.y, a.y b.x)
    acc = make_float2(r.x, r.y);
#else
    cfma2(acc, a, b);
#endif
}
constexpr int CH_NE = 4, CH_DEPTH = AEFFT_X_CH_DEPTH, CH_NB = CH_DEPTH + 1, CH_KSH_MAX = 4, CH_BLOCK = 12;
// lane i <- lane i + N of the same 16-lane row (0 past the row's end)
template <int N> __device__ __forceinline__ float dpp_row_shl(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}
// desc: R (8 bits) | K (8) | k0 (7) | log2 KS (3) | last chunk of its stage | first chunk | encoder stage | level (3);  off: record offset of
// the stage (24 bits) | bit 31: which of the two running vectors the stage reads
// (SRC: what hosts the step list -- ChainSrc for the chain's items, PackedMseSrc for the innermost pair's post-update MSE)
struct ChainSrc {
    const ChainArgs& g;
    __device__ __forceinline__ unsigned off(int i) const { return g.st_off[i]; }
    __device__ __forceinline__ unsigned desc(int i) const { return g.st_desc[i]; }
    __device__ __forceinline__ const float* bias(unsigned lv, bool enc) const { return enc ? g.lv[lv].b : g.lv[lv].p; }
    __device__ __forceinline__ float NN(unsigned lv) const { return (float)g.lv[lv].Nx * (float)g.lv[lv].Ny; }
    __device__ __forceinline__ float2* out(unsigned lv) const { return g.lv[lv].O; }        // a decoder stage's rows: O_l[c][r] at bin t of the support
    __device__ __forceinline__ unsigned Pc() const { return (unsigned)g.Pc; }
};
template <class SRC>
__device__ __forceinline__ void chain_step_load(const SRC& g, const float2* __restrict__ rec, const int i, float2 (&w)[CH_NE], float& bv)
{
    const unsigned d = g.desc(i), off = g.off(i) & 0xffffffu;
    const unsigned R = d & 255u, K = (d >> 8) & 255u, k0 = (d >> 16) & 127u, ksh = (d >> 23) & 7u;
    const unsigned tid = threadIdx.x;
    const unsigned r = min(tid >> ksh, R - 1u), ks = tid & ((1u << ksh) - 1u);
    const unsigned row = off + r * K, e0 = row + k0 + ks, elast = row + K - 1u;
#pragma unroll
    for (int u = 0; u < CH_NE; ++u) w[u] = ld8(rec, min(e0 + ((unsigned)u << ksh), elast));
    bv = g.bias(d >> 29, (d >> 28) & 1u)[r];
}
// DUAL (the chain items of a tail launch that also owes the innermost pair's post-update MSE): the two innermost stages -- C'_{L-1} and F'_{L-1} of the
// UPDATED weights, marked by bit 30 of their steps' offset word -- are applied to a SECOND set of OPC columns as well, the operator A_{L-1} of the step
// that is ending (regions 2, 3, 4 of Wl: A, C' A / dM + b^, F'(.) / dD + p^).  The matrices are loaded once for both; the workgroups that used to read
// them again for the MSE (one per bin, a quarter of the launch's instructions) are gone.
template <bool DUAL, class SRC>
__device__ __forceinline__ void chain_step_compute(const SRC& g, const int i, const float2 (&w)[CH_NE], const float bv, float2 (&acc)[OPC], float2 (&acc2)[OPC],
                                                   float2* const Wl, const int t)
{
    static_assert(OPC == 4, "two 16-byte LDS accesses per row of V");
    const unsigned d = g.desc(i), par = g.off(i) >> 31;
    const bool dual = DUAL && ((g.off(i) >> 30) & 1u);            // (uniform)
    const unsigned R = d & 255u, K = (d >> 8) & 255u, k0 = (d >> 16) & 127u, ksh = (d >> 23) & 7u;
    const unsigned tid = threadIdx.x;
    const unsigned rr = tid >> ksh, ks = tid & ((1u << ksh) - 1u);
    const bool valid = rr < R;
    const bool enc = (d >> 28) & 1u;
    if ((d >> 27) & 1u) {
#pragma unroll
        for (int c = 0; c < OPC; ++c) { acc[c] = make_float2(0.f, 0.f); if (DUAL) acc2[c] = make_float2(0.f, 0.f); }
    }
    const float2* Vin = Wl + par * (CH_VMAX * OPC);
    const float2* Vin2 = Wl + (enc ? 2 : 3) * (CH_VMAX * OPC);
#pragma unroll
    for (int u = 0; u < CH_NE; ++u) {
        const unsigned k = k0 + ks + ((unsigned)u << ksh);
        const bool ok = valid && k < K;
        const float2 wv = ok ? w[u] : make_float2(0.f, 0.f);
        const unsigned kk = min(k, K - 1u) * OPC;
        {
            const float4* vp = reinterpret_cast<const float4*>(Vin + kk);
            const float4 va = vp[0], vb = vp[1];
            cfma_pk(acc[0], wv, make_float2(va.x, va.y)); cfma_pk(acc[1], wv, make_float2(va.z, va.w));
            cfma_pk(acc[2], wv, make_float2(vb.x, vb.y)); cfma_pk(acc[3], wv, make_float2(vb.z, vb.w));
        }
        if (dual) {
            const float4* vp = reinterpret_cast<const float4*>(Vin2 + kk);
            const float4 va = vp[0], vb = vp[1];
            cfma_pk(acc2[0], wv, make_float2(va.x, va.y)); cfma_pk(acc2[1], wv, make_float2(va.z, va.w));
            cfma_pk(acc2[2], wv, make_float2(vb.x, vb.y)); cfma_pk(acc2[3], wv, make_float2(vb.z, vb.w));
        }
    }
    if (!((d >> 26) & 1u)) return;                                // (uniform) more chunks of this stage follow
    // the KS <= 16 lanes of a row are adjacent and aligned inside a 16-lane DPP row: lane i += lane i + 2^j (row_shl, one v_add_f32_dpp each; lanes
    // shifted in from outside the row read 0) leaves the row's sum in its lane ks == 0 -- the only one that stores.  (__shfl_xor is a
    // ds_bpermute with its own wait per value: 48 of them per stage were 1.3 us of every stage.)
    if (ksh > 0) {
#pragma unroll
        for (int c = 0; c < OPC; ++c) { acc[c].x += dpp_row_shl<1>(acc[c].x); acc[c].y += dpp_row_shl<1>(acc[c].y); }
        if (dual) {
#pragma unroll
            for (int c = 0; c < OPC; ++c) { acc2[c].x += dpp_row_shl<1>(acc2[c].x); acc2[c].y += dpp_row_shl<1>(acc2[c].y); }
        }
    }
    if (ksh > 1) {
#pragma unroll
        for (int c = 0; c < OPC; ++c) { acc[c].x += dpp_row_shl<2>(acc[c].x); acc[c].y += dpp_row_shl<2>(acc[c].y); }
        if (dual) {
#pragma unroll
            for (int c = 0; c < OPC; ++c) { acc2[c].x += dpp_row_shl<2>(acc2[c].x); acc2[c].y += dpp_row_shl<2>(acc2[c].y); }
        }
    }
    if (ksh > 2) {
#pragma unroll
        for (int c = 0; c < OPC; ++c) { acc[c].x += dpp_row_shl<4>(acc[c].x); acc[c].y += dpp_row_shl<4>(acc[c].y); }
        if (dual) {
#pragma unroll
            for (int c = 0; c < OPC; ++c) { acc2[c].x += dpp_row_shl<4>(acc2[c].x); acc2[c].y += dpp_row_shl<4>(acc2[c].y); }
        }
    }
    if (ksh > 3) {
#pragma unroll
        for (int c = 0; c < OPC; ++c) { acc[c].x += dpp_row_shl<8>(acc[c].x); acc[c].y += dpp_row_shl<8>(acc[c].y); }
        if (dual) {
#pragma unroll
            for (int c = 0; c < OPC; ++c) { acc2[c].x += dpp_row_shl<8>(acc2[c].x); acc2[c].y += dpp_row_shl<8>(acc2[c].y); }
        }
    }
    const unsigned lv = d >> 29;
    if (valid && ks == 0) {
        const float scale = __frcp_rn((float)R);
#pragma unroll
        for (int c = 0; c < OPC; ++c) { acc[c].x *= scale; acc[c].y *= scale; }
        if (t == 0) acc[OPC - 1].x += bv * g.NN(lv);
        if (dual) {
#pragma unroll
            for (int c = 0; c < OPC; ++c) { acc2[c].x *= scale; acc2[c].y *= scale; }
            if (t == 0) acc2[OPC - 1].x += bv * g.NN(lv);
            float4* v2 = reinterpret_cast<float4*>(Wl + (enc ? 3 : 4) * (CH_VMAX * OPC) + rr * OPC);
            v2[0] = make_float4(acc2[0].x, acc2[0].y, acc2[1].x, acc2[1].y); v2[1] = make_float4(acc2[2].x, acc2[2].y, acc2[3].x, acc2[3].y);
        }
        const float4 o0 = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y), o1 = make_float4(acc[2].x, acc[2].y, acc[3].x, acc[3].y);
        float4* vo = reinterpret_cast<float4*>(Wl + (par ^ 1u) * (CH_VMAX * OPC) + rr * OPC);
        vo[0] = o0; vo[1] = o1;
        if (!enc) {                                               // (uniform) a decoder stage: its rows may be an output of the item
            float2* O = g.out(lv);
            if (O) {
                const unsigned Pc = g.Pc();
#pragma unroll
                for (int c = 0; c < OPC; ++c) st8(O, ((unsigned)c * R + rr) * Pc + (unsigned)t, acc[c]);
            }
        }
    }
    __syncthreads();                                              // (uniform branch: every thread of the workgroup is here) the stage's output is complete
#if defined(AEFFT_X_WGTIME) && AEFFT_X_WGTIME
    if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(Wl + 5 * CH_VMAX * OPC + 4 + OPC * OPC)[(enc ? lv : 7 - lv) & 7] = wall_clock64();    // (in LDS: a global store here would drain the pipeline it times)
#endif
}
// straight-line code, one copy per step, the exits nested (a template recursion): with a LOOP around the rotation hipcc's wait insertion merges
// the loop's entry and back edge into a vmcnt(0) at the header -- a drained pipeline every CH_NB steps -- and a rolled loop would index the
// register sets dynamically (scratch)
template <int I, bool DUAL, class SRC>
__device__ __forceinline__ void chain_steps(const SRC& g, const float2* __restrict__ rec, const int base, const int nb, float2 (&w)[CH_NB][CH_NE], float (&bv)[CH_NB],
                                            float2 (&acc)[OPC], float2 (&acc2)[OPC], float2* const Wl, const int t)
{
    if constexpr (I < CH_BLOCK) {
        if (I >= nb) return;                                      // (uniform)
        chain_step_load(g, rec, base + min(I + CH_DEPTH, nb - 1), w[(I + CH_DEPTH) % CH_NB], bv[(I + CH_DEPTH) % CH_NB]);
        chain_step_compute<DUAL>(g, base + I, w[I % CH_NB], bv[I % CH_NB], acc, acc2, Wl, t);
        chain_steps<I + 1, DUAL, SRC>(g, rec, base, nb, w, bv, acc, acc2, Wl, t);
    }
}
// every step of a list: blocks of CH_BLOCK straight-line steps (cfg3: one block); the pipeline drains and refills between blocks.  `init`: fills the
// first running vector (buffer 0 of Wl) while the first loads are in flight; the barrier behind it is this function's.
template <bool DUAL, class SRC, class INIT>
__device__ __forceinline__ void chain_run_steps(const SRC& g, const float2* __restrict__ rec, const int n, float2* const Wl, const int t, INIT init)
{
    float2 w[CH_NB][CH_NE];
    float bv[CH_NB];
    float2 acc[OPC], acc2[OPC];
    for (int base = 0; base < n; base += CH_BLOCK) {
        const int nb = min(n - base, CH_BLOCK);
#pragma unroll
        for (int j = 0; j < CH_DEPTH; ++j) chain_step_load(g, rec, base + min(j, nb - 1), w[j], bv[j]);
        if (base == 0) { init(); __syncthreads(); }
        chain_steps<0, DUAL, SRC>(g, rec, base, nb, w, bv, acc, acc2, Wl, t);
    }
}
// host: the steps of one stage (R x K matrix at record offset `off`, reading running vector `par`) appended to a step list; false: the list is full
static bool chain_add_stage(unsigned* st_off, unsigned* st_desc, int* n, int cap, int R, int K, unsigned off, int par, bool enc, int lvl)
{
    int ksh = 0;                                                     // KS = min(largest power of two <= 256 / R, smallest power of two >= K, 16)
    while (ksh < CH_KSH_MAX && (2 << ksh) * R <= 256 && (1 << ksh) < K) ++ksh;
    const int kc = CH_NE << ksh;
    for (int k0 = 0; k0 < K; k0 += kc) {
        if (*n >= cap || off >= (1u << 24) || k0 > 127 || R > 255 || K > 255) return false;
        st_off[*n] = off | ((unsigned)(par & 1) << 31);
        st_desc[*n] = (unsigned)R | ((unsigned)K << 8) | ((unsigned)k0 << 16) | ((unsigned)ksh << 23) | ((k0 + kc >= K ? 1u : 0u) << 26) |
                      ((k0 == 0 ? 1u : 0u) << 27) | ((enc ? 1u : 0u) << 28) | ((unsigned)lvl << 29);
        ++*n;
    }
    return true;
}

// The innermost pair from the bin-major copy of the updated spectra (kspec_packed_body): one workgroup per bin, C' and F' read as
// contiguous rows (the planar layout makes them 8 K scattered 32-byte pieces per bin tile), two chain stages and the quadratic form.
constexpr size_t OPMSE_PACKED_LDS = sizeof(float2) * (3 * CH_VMAX * OPC + 2 + OPC * OPC) + 128;
struct PackedMseSrc {                                   // the two stages C', F' of the innermost pair as a step list (chain_run_steps)
    const OpMseGroup& g; const OpMsePair& q;
    __device__ __forceinline__ unsigned off(int i) const { return g.pst_off[i]; }
    __device__ __forceinline__ unsigned desc(int i) const { return g.pst_desc[i]; }
    __device__ __forceinline__ const float* bias(unsigned, bool enc) const { return enc ? q.b : q.p; }
    __device__ __forceinline__ float NN(unsigned) const { return (float)q.Nx * (float)q.Ny; }
    __device__ __forceinline__ float2* out(unsigned) const { return nullptr; }
    __device__ __forceinline__ unsigned Pc() const { return 0u; }
};
__device__ __forceinline__ void opmse_packed(const OpMseGroup& g, int p, long t, float2* sh)
{
    float2 *Va = sh + 2 * CH_VMAX * OPC;                 // A of the bin (kept); the running vectors are buffers 0 and 1 of sh
    float* redp = reinterpret_cast<float*>(sh + 3 * CH_VMAX * OPC);
    const OpMsePair q = g.q[p];
    const int dD = q.dD, dM = q.dM;
    const float2* rec = g.Wp + t * g.E;
    const float2* Vc;                                    // F'(C' A / dM + b^) / dD + p^ of the bin
    float2* Ms = sh + 3 * CH_VMAX * OPC + 2;             // the bin's 4x4 moments, requested with A (behind the stages they were a round trip of their own)
    const long um = map_up(t, q.Nx, q.Ny, g.Nx0, g.Ny0);
    if (g.pst_n > 0) {
        // (as a software pipeline over the record: the stage-by-stage form below was six dependent round trips, 11 us per workgroup at cfg3)
        const PackedMseSrc src{g, q};
        chain_run_steps<false>(src, rec, g.pst_n, sh, (int)t, [&]() {
            float2 mv = make_float2(0.f, 0.f);
            if (threadIdx.x < OPC * OPC) mv = g.Mhat[(long)threadIdx.x * g.P0 + um];
            for (int i = threadIdx.x; i < dD * OPC; i += 256) { const int a = i / OPC, k = i - a * OPC; const float2 v = q.A[((long)k * dD + a) * q.P + t]; sh[i] = v; Va[i] = v; }
            if (threadIdx.x < OPC * OPC) Ms[threadIdx.x] = mv;
        });
        Vc = sh;                                         // (stage 0 reads buffer 0 and writes 1, stage 1 writes 0; its barrier has been passed)
    } else {
        float2 *Vb = sh, *Vd = sh + CH_VMAX * OPC;
        if (threadIdx.x < OPC * OPC) Ms[threadIdx.x] = g.Mhat[(long)threadIdx.x * g.P0 + um];
        for (int i = threadIdx.x; i < dD * OPC; i += 256) { const int a = i / OPC, k = i - a * OPC; Va[i] = q.A[((long)k * dD + a) * q.P + t]; }
        const float NN = (float)q.Nx * (float)q.Ny;
        __syncthreads();
        chain_stage_rec(rec + g.offC, Va, Vb, dM, dD, 1.0f / (float)dM, q.b, NN, t == 0, nullptr, 0, 0);
        __syncthreads();
        chain_stage_rec(rec + g.offF, Vb, Vd, dD, dM, 1.0f / (float)dD, q.p, NN, t == 0, nullptr, 0, 0);
        __syncthreads();
        Vc = Vd;
    }
    float part = 0.f;
    if ((int)threadIdx.x < dD) {
        const int a = threadIdx.x;
        float2 r[OPC];
#pragma unroll
        for (int k = 0; k < OPC; ++k) { const float2 av = Va[a * OPC + k], fv = Vc[a * OPC + k]; r[k] = make_float2(av.x - fv.x, av.y - fv.y); }
        part = quad_centred(r, [&](int e) { return Ms[e]; });
        const int nyr = q.Ny / 2 + 1;
        const int j = (int)((unsigned)t % (unsigned)nyr);
        part *= (j > 0 && j < nyr - 1) ? 2.f : 1.f;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if ((threadIdx.x & 63) == 0) redp[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (redp[0] + redp[1]) + (redp[2] + redp[3]);
        if (tot != 0.f) atomicAdd(q.slots + (blockIdx.x % MSE_SLOTS) * MSE_SLOT_STRIDE, tot * q.scale);
    }
}

// (the bodies take their workgroup index from blockIdx.x - g.base: the launch may host other work in front, tail_kernel)
// LEAN: every pair of the launch is served by opmse_packed or opmse_gbody (launch_opmse_group checks) -- the bodies that read the planar
// C' | F' are not instantiated: they need 103-121 registers against 62-78 for the rest, and a launch's allocation is its largest body's
template <bool LEAN>
__device__ __forceinline__ void opmse_dispatch(const OpMseGroup& g, float2* sh)
{
    const int blk = (int)blockIdx.x - g.base;
    int p = g.n - 1;                                                 // pair n-1 owns the first workgroups, pair 0 the last
#pragma unroll
    for (int i = 6; i >= 0; --i) if (i < g.n - 1 && blk >= g.start[i] - g.base) p = i;
    if (p == g.n - 1 && g.Wp) { opmse_packed(g, p, (long)blockIdx.x - g.start[p], sh); return; }
    if (LEAN || g.q[p].G) { opmse_gbody(g, p, sh); return; }         // (uniform) the pair's collapsed operator G' is at hand
    if constexpr (!LEAN) {
        const int bt = g.bt[p];                                      // uniform per workgroup
        if (bt == 32) opmse_small_body<8>(g, p, sh);
        else if (bt == 16) opmse_body<16>(g, p, sh);
        else if (bt == 8) opmse_body<8>(g, p, sh);
        else opmse_body<4>(g, p, sh);
    }
}

// workgroup ranges and LDS of the MSE part; `base` = workgroups of the launch in front of it (start[] are launch-wide indices)
static hipError_t opmse_geometry(OpMseGroup& g, int base, long* nblocks, size_t* lds_out, bool fused = false /* the chain's items form the innermost pair's MSE: no workgroups for it here */)
{
    if (g.n < 1 || g.n > 8) return hipErrorInvalidValue;
    long total = base;
    size_t lds = 0;
    g.base = base;
    for (int i = g.n - 1; i >= 0; --i) {                               // innermost (deepest K) pairs first: they are the long poles
        const OpMsePair& q = g.q[i];
        // bins per workgroup: whole 128-byte lines when the pair still yields >= 128 workgroups and its tiles fit 64 KB of LDS
        int bt = 16;
        while (bt > 4 && ((q.P + bt - 1) / bt < 128 || (size_t)OPC * (q.dD + q.dM) * bt * sizeof(float2) > 48 * 1024)) bt >>= 1;
        size_t need = (size_t)OPC * (q.dD + q.dM) * bt * sizeof(float2) + 64 + (size_t)256 * OPC * sizeof(float2) + (size_t)OPC * OPC * bt * sizeof(float2);
        const bool pk = i == g.n - 1 && g.Wp && q.dD <= CH_VMAX && q.dM <= CH_VMAX;
        if (pk) {
            need = OPMSE_PACKED_LDS;
            int n = 0;
            const bool fits = chain_add_stage(g.pst_off, g.pst_desc, &n, OPMSE_PACKED_STEPS, q.dM, q.dD, (unsigned)g.offC, 0, true, 0) &&
                              chain_add_stage(g.pst_off, g.pst_desc, &n, OPMSE_PACKED_STEPS, q.dD, q.dM, (unsigned)g.offF, 1, false, 0);
            g.pst_n = fits && AEFFT_X_CHAINPIPE ? n : 0;
        }
        else if (q.G) {                                                  // opmse_gbody
            if ((double)q.dD * q.dD * q.P * 8.0 >= 4294967296.0) return hipErrorInvalidValue;
            bt = msgrad_bt(q.dD, q.P);
            need = sizeof(float2) * ((size_t)OPC * q.dD * bt + (size_t)OPC * OPC * bt) + 64;
        } else if (q.dM == 8 && q.dD <= 4 && !flag(AEFFT_F_NOFAST)) { bt = 32; need = (size_t)OPC * (4 + 8) * 32 * sizeof(float2) + 64; }     // opmse_small_body (bt == 32 selects it)
        if (i == g.n - 1 && !pk) g.Wp = nullptr;
        if (need > 150 * 1024) return hipErrorInvalidValue;
        g.bt[i] = bt;
        lds = std::max(lds, need);
        g.start[i] = (int)total;
        total += pk ? (fused ? 0 : q.P) : (q.P + bt - 1) / bt;
    }
    if (total >= (1L << 31)) return hipErrorInvalidValue;
    g.start[g.n] = (int)total;                                         // (start[] is DEscending in i: see opmse_dispatch's lookup)
    *nblocks = total - base; *lds_out = lds;
    return hipSuccess;
}

}  // namespace aefft

namespace aefft {

// ------------------------------------------------------------------------------------------
// The network on the basis frames in ONE launch (the per-bin operator chain).
//
// Every bin is independent: A_l[s] = C_{l-1}[m(s)] A_{l-1}[m(s)] / dM + b^ along the pooling maps, then on the coarsest grid
// H^ = C_{L-1} A_{L-1} / dM + b^, O^_{L-1} = F_{L-1} H^ / dD + p^, O^_l = F_l[M_l(t)] O^_{l+1} / dD_l + p^_l  (conv_k,
// fft_backproplib.cu:162-189, on OPC columns; pool_fft's index maps, :87-157).  Work item = one bin of grid l that no bin of
// grid l+1 maps to (all bins of the coarsest grid): it walks its ancestor chain from grid 0 and stores A_j at each ancestor
// (every bin of every grid is stored exactly once); items of the coarsest grid continue through the decoder.
//
// The kernel spectra are planar ([m][d][bin]): all (m, d) of ONE bin is a gather of 8-byte elements from as many cache lines,
// and a CU sustains only about one such request per 15 cycles (measured: 3.2 M requests = 87 us chip-wide).  The coarsest-grid
// items, which need every matrix of every pair (43 KB each at cfg3), therefore read a bin-major copy Wp[t][E] that
// kspec_packed_kernel evaluates straight from the Nk x Nl taps (a pruned DFT like kspec_kernel, pruned_kernels.hip) whenever the
// weights change: one coalesced record per item, one memory round trip, the chain of small dependent products then runs out of
// LDS.  The few middle-grid items (two small matrices each) keep the gather.
// ------------------------------------------------------------------------------------------
template <int NK> __global__ __launch_bounds__(256) void kspec_packed_kernel(const PackArgs g)
{
    extern __shared__ float2 pk_lds[];
    kspec_packed_body<NK>(g, blockIdx.x, blockIdx.y, pk_lds);
}

void pack_blocks(PackArgs& g)
{
    int nb = 0;
    for (int i = 0; i < g.nseg; ++i)
        for (int l0 = 0; l0 < g.seg[i].n && nb < 128; l0 += 256) { g.blk_seg[nb] = (unsigned char)i; g.blk_start[nb] = l0; ++nb; }
    g.nblk = nb;
}

hipError_t launch_kspec_packed(PackArgs& g, hipStream_t st)
{
    if (g.nseg < 1 || g.nseg > 16 || g.L < 1 || g.L > 8 || g.E < 1 || (g.Nk != 3 && g.Nk != 5)) return hipErrorInvalidValue;
    pack_blocks(g);
    const dim3 grid((unsigned)g.nblk, (unsigned)pack_yblocks(g));
    if (g.Nk == 3) kspec_packed_kernel<3><<<grid, 256, kspec_packed_lds(3), st>>>(g);
    else kspec_packed_kernel<5><<<grid, 256, kspec_packed_lds(5), st>>>(g);
    return hipGetLastError();
}

// One launch, two kinds of workgroups, all independent:
//  * blockIdx < Pc: the coarsest-grid bin t.  A_1 .. A_{L-1}, H^ and O^_{L-1} .. O^_0 of that bin, every matrix read from the
//    bin's packed record (contiguous rows); only the running OPC-column vector lives in LDS, so many workgroups share a CU
//    and hide each other's stage-to-stage latency.  The decoder outputs are stored.
//  * the rest: a tile of CH_BT consecutive bins of grid j (1 <= j < L), threads = (bin, row group).  It recomputes the tile's
//    ancestor chain A_1 .. A_j from the PLANAR spectra (small matrices; lanes along the bins: coalesced) and stores A_j.
// No workgroup waits for another one; the only cost of the independence is the re-evaluation of a few small products.
// mse (nullable): the tail launch's MSE description when THIS item also forms the innermost pair's post-update MSE at its bin (ChainArgs::fuse_mse:
// the steps of the two innermost stages are marked dual) -- what opmse_packed does in a workgroup of its own otherwise
template <bool DUAL>
__device__ __forceinline__ void chain_item_run(const ChainArgs& g, const int t, float2* Wl, const OpMseGroup* mse)
{
    const ChainSrc src{g};
    constexpr int VS = CH_VMAX * OPC;
    float2* Ms = Wl + 5 * VS;                             // [OPC*OPC] the bin's moments (DUAL)
    chain_run_steps<DUAL>(src, g.Wp + (long)t * g.E, g.st_n, Wl, t, [&]() {
        float2 mv = make_float2(0.f, 0.f);
        if (DUAL) {
            const OpMsePair& q = mse->q[mse->n - 1];
            if (threadIdx.x < OPC * OPC) mv = mse->Mhat[(long)threadIdx.x * mse->P0 + map_up(t, q.Nx, q.Ny, mse->Nx0, mse->Ny0)];
            const int dD = q.dD;
            for (int i = threadIdx.x; i < dD * OPC; i += 256) { const int a = i / OPC, k = i - a * OPC; Wl[2 * VS + i] = q.A[((long)k * dD + a) * q.P + t]; }
        }
        for (int i = threadIdx.x; i < VS; i += 256) { const int k = i / OPC, c = i - k * OPC; Wl[i] = make_float2((k == c && c < OPC - 1) ? 1.f : 0.f, 0.f); }
        if (DUAL && threadIdx.x < OPC * OPC) Ms[threadIdx.x] = mv;
    });
    if (DUAL) {
        // R = A - (F'(C' A / dM + b^) / dD + p^) at this bin and tr(R M^ R^H): opmse_packed's epilogue (the last stage's barrier has made region 4 visible)
        const OpMsePair& q = mse->q[mse->n - 1];
        const float2 *Va = Wl + 2 * VS, *Vc = Wl + 4 * VS;
        float* redp = reinterpret_cast<float*>(Ms + OPC * OPC);
        float part = 0.f;
        if ((int)threadIdx.x < q.dD) {
            const int a = threadIdx.x;
            float2 r[OPC];
#pragma unroll
            for (int k = 0; k < OPC; ++k) { const float2 av = Va[a * OPC + k], fv = Vc[a * OPC + k]; r[k] = make_float2(av.x - fv.x, av.y - fv.y); }
            part = quad_centred(r, [&](int e) { return Ms[e]; });
            const int nyr = q.Ny / 2 + 1;
            const int j = (int)((unsigned)t % (unsigned)nyr);
            part *= (j > 0 && j < nyr - 1) ? 2.f : 1.f;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if ((threadIdx.x & 63) == 0) redp[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float tot = (redp[0] + redp[1]) + (redp[2] + redp[3]);
            if (tot != 0.f) atomicAdd(q.slots + (blockIdx.x % MSE_SLOTS) * MSE_SLOT_STRIDE, tot * q.scale);
        }
    }
#if defined(AEFFT_X_WGTIME) && AEFFT_X_WGTIME
    if (threadIdx.x == 0 && g_wgtime && blockIdx.x < WGT_MAX)
        for (int k = 0; k < 8; ++k) g_wgtime[(size_t)5 * WGT_MAX * 2 + ((size_t)4 * WGT_MAX + blockIdx.x) * 8 + k] = reinterpret_cast<unsigned long long*>(Wl + 5 * CH_VMAX * OPC + 4 + OPC * OPC)[k];
#endif
}
// (ONE instantiation per kernel: two inlined copies of the straight-line steps in one kernel spill)

#if defined(AEFFT_X_WGTIME) && AEFFT_X_WGTIME
constexpr int CH_ITEM_VECS = 5;      // (the stage stamps live behind the fifth vector)
#else
constexpr int CH_ITEM_VECS = 2;
#endif
constexpr int CH_BT = 8;
// bx: the workgroup's index inside the chain part of the launch; Wl: dynamic LDS (per-bin items: two running vectors; planar tiles:
// two V tiles [rows][OPC][CH_BT])
template <bool FUSE>
__device__ __forceinline__ void chain_body(const ChainArgs& g, const int bx, float2* Wl, const OpMseGroup* mse = nullptr)
{
    const int tid = threadIdx.x;
    const int L = g.L;
    if (AEFFT_X_CHAINPIPE && (long)bx < g.Pc && g.st_n > 0) { chain_item_run<FUSE>(g, bx, Wl, mse); return; }      // (st_n == 0: more steps than the step table holds)
    if ((long)bx < g.Pc) {
        const int t = bx;
        const bool dc = t == 0;
        const float2* rec = g.Wp + (long)t * g.E;
        for (int i = tid; i < CH_VMAX * OPC; i += 256) { const int k = i / OPC, c = i - k * OPC; Wl[i] = make_float2((k == c && c < OPC - 1) ? 1.f : 0.f, 0.f); }
        int off = 0, si = 0;
        auto run = [&](int R, int K, float scale, const float* bias, float NN, float2* out) {
            __syncthreads();                                         // the previous stage's output is complete
            AEFFT_WGSTAMP(4, si & 7);
            // (the two running vectors as offsets off ONE LDS base: a select between two pointers compiles to flat loads)
            chain_stage_rec(rec + off, Wl + (si & 1) * (CH_VMAX * OPC), Wl + ((si + 1) & 1) * (CH_VMAX * OPC), R, K, scale, bias, NN, dc, out, g.Pc, t);
            off += (R * K + 1) & ~1; ++si;
        };
        for (int j = 1; j < L; ++j) { const ChainLevel& w = g.lv[j - 1]; run(w.dM, w.dD, 1.0f / (float)w.dM, w.b, (float)w.Nx * (float)w.Ny, nullptr); }
        { const ChainLevel& w = g.lv[L - 1]; run(w.dM, w.dD, 1.0f / (float)w.dM, w.b, (float)w.Nx * (float)w.Ny, nullptr); }
        for (int j = L - 1; j >= 0; --j) { const ChainLevel& v = g.lv[j]; run(v.dD, v.dM, 1.0f / (float)v.dD, v.p, (float)v.Nx * (float)v.Ny, v.O); }
        return;
    }
    // ---- planar tile of grid j ----
    int j = 1;
#pragma unroll
    for (int i = 2; i < 8; ++i) if (i < L && bx >= g.tile_start[i]) j = i;
    // bins per tile: 256 / (rows of the tile's last stage), 8 .. 32 -- every thread owns a (bin, row) of that stage (8 bins for every level left three
    // quarters of the threads of a grid-1 tile without a row: 1 372 tile workgroups at cfg3, 460 now), and a tile's plane reads are 64 .. 256-byte pieces
    const int bt = g.tile_bt[j], RT = 256 / bt;
    const int bl = tid & (bt - 1), ry = tid / bt;
    const ChainLevel lj = g.lv[j];
    const long s0 = (long)(bx - g.tile_start[j]) * bt;
    const long s = min(s0 + bl, lj.P - 1);
    const bool ok = s0 + bl < lj.P;
    // V tiles in LDS: [row][col][bin], ping-pong halves of Wl
    // (addressed as offsets off ONE LDS base: a select between two pointers compiles to flat loads, which wait for every counter)
    for (int i = tid; i < OPC * OPC * bt; i += 256) { const int c = (i / bt) % OPC, k = i / (bt * OPC); Wl[i] = make_float2((k == c && c < OPC - 1) ? 1.f : 0.f, 0.f); }
    // (a ROLLED loop over the levels: unrolled, the seven copies of the stage body cost 121 registers -- half the occupancy of the
    // launch, which hosts the per-bin items and the MSE as well; the level's descriptor is a uniform, scalar-loaded index)
#pragma unroll 1
    for (int i = 1; i <= j; ++i) {
        const ChainLevel& w = g.lv[i - 1];
        // the tile's bin on grid i and on grid i-1 (pool_fft's index map composes to the same form)
        const long sbi = i == j ? s : map_up(s, lj.Nx, lj.Ny, g.lv[i].Nx, g.lv[i].Ny);
        const long sbm = map_up(s, lj.Nx, lj.Ny, w.Nx, w.Ny);
        const int R = w.dM, K = w.dD;
        const float scale = 1.0f / (float)w.dM, NN = (float)w.Nx * (float)w.Ny;
        const float2* Vin = Wl + ((i - 1) & 1) * g.vt_elems;
        float2* Vout = Wl + (i & 1) * g.vt_elems;
        __syncthreads();
        for (int r = ry; r < R; r += RT) {
            float2 acc[OPC];
#pragma unroll
            for (int c = 0; c < OPC; ++c) acc[c] = make_float2(0.f, 0.f);
            // (the pair's planar spectrum, or its compact copy Cc sampled where grid i lands: then bin sb[i] of planes of lv[i].P bins)
            const long pst = w.Cc ? g.lv[i].P : w.P;
            const float2* Wp = (w.Cc ? w.Cc + sbi : w.C + sbm) + (long)r * K * pst;
            auto grp = [&](int k0, auto NU) {
                constexpr int U = decltype(NU)::value;
                float2 wv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) wv[u] = Wp[(long)(k0 + u) * pst];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int c = 0; c < OPC; ++c) cfma2(acc[c], wv[u], Vin[((k0 + u) * OPC + c) * bt + bl]);
            };
            int k0 = 0;
            for (; k0 + 16 <= K; k0 += 16) grp(k0, std::integral_constant<int, 16>{});
            if (k0 + 8 <= K) { grp(k0, std::integral_constant<int, 8>{}); k0 += 8; }
            if (k0 + 4 <= K) { grp(k0, std::integral_constant<int, 4>{}); k0 += 4; }
            if (k0 + 2 <= K) { grp(k0, std::integral_constant<int, 2>{}); k0 += 2; }
            if (k0 < K) grp(k0, std::integral_constant<int, 1>{});
#pragma unroll
            for (int c = 0; c < OPC; ++c) { acc[c].x *= scale; acc[c].y *= scale; }
            if (sbi == 0) acc[OPC - 1].x += w.b[r] * NN;
#pragma unroll
            for (int c = 0; c < OPC; ++c) {
                Vout[(r * OPC + c) * bt + bl] = acc[c];
                if (i == j && ok) lj.A[((long)c * R + r) * lj.P + s] = acc[c];
            }
        }
    }
}

__global__ __launch_bounds__(256) void chain_kernel(const ChainArgs g)
{
    extern __shared__ float2 Wl[];
    chain_body<false>(g, blockIdx.x, Wl);
}

// fuse (in/out, nullable): the caller would like the per-bin items to form the innermost pair's post-update MSE as well (chain_item_run<true>); granted
// when the items run pipelined and the record offsets of that pair's C and F are the ones the MSE description names (offC, offF): then the steps of
// those two stages carry the dual mark (bit 30 of the offset word)
static hipError_t chain_geometry(ChainArgs& g, long* nblocks, size_t* lds_out, bool* fuse = nullptr, unsigned offC = 0, unsigned offF = 0)
{
    if (g.L < 1 || g.L > 8 || !g.Wp || g.Pc < 1) return hipErrorInvalidValue;
    for (int l = 0; l < g.L; ++l) {
        if (g.lv[l].dD > CH_VMAX || g.lv[l].dM > CH_VMAX) return hipErrorInvalidValue;
        if (l + 1 < g.L && (size_t)2 * g.lv[l].dM * OPC * CH_BT > CH_WL) return hipErrorInvalidValue;     // planar tiles: two V tiles share the LDS buffer (at the narrowest tile)
    }
    {
        // the per-bin item's step list (chain_item_pipelined): stage si = level si's encoder matrix, then the decoders from the innermost level out
        int n = 0; unsigned off = 0;
        bool fits = true, offs_ok = true;
        int dual_lo = 0, dual_hi = 0;                                   // steps [dual_lo, dual_hi): the innermost pair's two stages
        for (int si = 0; si < 2 * g.L && fits; ++si) {
            const bool enc = si < g.L;
            const int l = enc ? si : 2 * g.L - 1 - si;
            const int R = enc ? g.lv[l].dM : g.lv[l].dD, K = enc ? g.lv[l].dD : g.lv[l].dM;
            if (si == g.L - 1) { dual_lo = n; offs_ok = offs_ok && off == offC; }
            if (si == g.L) offs_ok = offs_ok && off == offF;
            fits = chain_add_stage(g.st_off, g.st_desc, &n, CH_MAXSTEPS, R, K, off, si & 1, enc, l);
            if (si == g.L) dual_hi = n;
            off += (unsigned)((R * K + 1) & ~1);
        }
        g.st_n = fits ? n : 0;                                          // (0: the item runs stage by stage, chain_stage_rec)
        if (fuse) {
            *fuse = *fuse && fits && offs_ok && AEFFT_X_CHAINPIPE;
            if (*fuse) for (int i = dual_lo; i < dual_hi; ++i) g.st_off[i] |= 1u << 30;
        }
        if ((double)OPC * CH_VMAX * (double)g.Pc * sizeof(float2) >= 4294967296.0) return hipErrorInvalidValue;      // (32-bit byte offsets of the output stores, st8)
    }
    long total = g.Pc;
    int vt = OPC * OPC * 32;                                            // (the identity start vector of the widest tile)
    for (int j = 1; j < g.L; ++j) {
        int rj = OPC;                                                   // most rows of a stage of a grid-j tile (levels 0 .. j-1)
        for (int l = 0; l < j; ++l) rj = std::max(rj, g.lv[l].dM);
        int bt = 32;
        while (bt > CH_BT && bt * rj > 256) bt >>= 1;
        g.tile_bt[j] = bt;
        vt = std::max(vt, rj * OPC * bt);
        g.tile_start[j] = (int)total; total += (g.lv[j].P + bt - 1) / bt;
    }
    g.tile_start[g.L] = (int)total;
    if (total >= (1L << 31)) return hipErrorInvalidValue;
    g.vt_elems = vt;
    *nblocks = total;
    // per-bin items: two running vectors, and with the fused MSE three more (A, C'A, F'C'A of the ending step), the bin's moments and the wave sums
    *lds_out = std::max(sizeof(float2) * 2 * g.vt_elems, sizeof(float2) * ((fuse && *fuse ? 5 : CH_ITEM_VECS) * (size_t)CH_VMAX * OPC + OPC * OPC + 4) + 256 /* (experiment builds: stage stamps) */);
    return hipSuccess;
}

hipError_t launch_chain(ChainArgs& g, hipStream_t st, hipEvent_t done)
{
    long total; size_t lds;
    const hipError_t eg = chain_geometry(g, &total, &lds);
    if (eg != hipSuccess) return eg;
    // `done`: recorded by this dispatch's own completion signal (no marker packet behind it on the stream: a side stream forks here)
    if (done) hipExtLaunchKernelGGL(chain_kernel, dim3((unsigned)total), dim3(256), lds, st, nullptr, done, 0, g);
    else chain_kernel<<<dim3((unsigned)total), 256, lds, st>>>(g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// The tail of a training step in ONE launch: the NEXT step's operator chain on the updated weights (its per-bin workgroups first:
// eight dependent stages each), the post-update MSE of every pair, and the in-place store of the updated taps (the deferred half
// of a fused update).  The chain reads the record Wp / the compact planes Cc that the spectra launch has just written and writes
// the OTHER set of operator buffers; the MSE reads the operators of the step that is ending.  Nothing in the launch depends on
// anything else in it.
// ------------------------------------------------------------------------------------------
// (LEAN: 80 registers -- six workgroups per CU instead of four; the launch is bound by its resident workgroups: 3 700 of them, 4-20 us each)
#ifndef AEFFT_X_TAIL_W
#define AEFFT_X_TAIL_W 6
#endif
// (with the fused MSE the items carry a second set of sums: 5 waves per SIMD, 96 registers -- at 6 the back end spills 20 bytes; cfg5 82.6-84 vs 82 us)
#ifndef AEFFT_X_TAIL_WF
#define AEFFT_X_TAIL_WF 5
#endif
template <bool LEAN, bool FUSE>
__global__ __launch_bounds__(256, LEAN ? (FUSE ? AEFFT_X_TAIL_WF : AEFFT_X_TAIL_W) : 4) void tail_kernel(const OpMseGroup g, const ChainArgs ch, const UpdateGroup ug, const int nchain, const int nupd_start)
{
    AEFFT_WGTIME(4);
    extern __shared__ float2 sh[];
    if ((int)blockIdx.x < nchain) { chain_body<FUSE>(ch, blockIdx.x, sh, &g); return; }
    if ((int)blockIdx.x >= nupd_start) {
        const int blk = blockIdx.x - nupd_start;
        int p = 0;
#pragma unroll
        for (int i = 1; i < 8; ++i) if (i < ug.n && blk >= ug.start[i]) p = i;
        update_weights_part(ug.a[p], blk - ug.start[p]);
        return;
    }
    opmse_dispatch<LEAN>(g, sh);
}

static UpdateGroup g_tail_ug_none{};
static ChainArgs g_tail_chain_none{};
// chain == null: the MSE (and the tap stores) alone
hipError_t launch_opmse_group(OpMseGroup& g, hipStream_t st, ChainArgs* chain, const UpdateGroup* weights_upd)
{
    long nchain = 0, nmse = 0;
    size_t lds_c = 0, lds_m = 0;
#ifndef AEFFT_X_FUSEMSE
#define AEFFT_X_FUSEMSE 1
#endif
    // the innermost pair's MSE inside the chain's per-bin items: both read that pair's C', F' from the same record at the same bin
    bool fuse = AEFFT_X_FUSEMSE && chain && g.Wp && chain->Wp == g.Wp && chain->E == g.E && chain->Pc == g.q[g.n - 1].P &&
                g.q[g.n - 1].dD <= CH_VMAX && g.q[g.n - 1].dM <= CH_VMAX && g.q[g.n - 1].dD <= 256;
    if (chain) { const hipError_t e = chain_geometry(*chain, &nchain, &lds_c, &fuse, (unsigned)g.offC, (unsigned)g.offF); if (e != hipSuccess) return e; }
    { const hipError_t e = opmse_geometry(g, (int)nchain, &nmse, &lds_m, false); if (e != hipSuccess) return e; }
    // ... where the launch is bound by its resident slots (cfg5: 11 000 workgroups, tail 89 -> 82 us).  Where the items themselves are the launch's
    // long pole (cfg3-P2: 3 634 workgroups) the two wider stages lengthen it: 23.9 -> 26.1 us.
    if (fuse && nchain + nmse < 4 * 1536 && !flag(AEFFT_F_CHAINMSE)) {
        fuse = false;
        for (int i = 0; i < chain->st_n; ++i) chain->st_off[i] &= ~(1u << 30);
        const hipError_t e = chain_geometry(*chain, &nchain, &lds_c); if (e != hipSuccess) return e;      // (the items' LDS without the MSE's vectors)
    }
    if (fuse) {
        const hipError_t e = opmse_geometry(g, (int)nchain, &nmse, &lds_m, true); if (e != hipSuccess) return e;
        if (!g.Wp) return hipErrorInvalidValue;                          // (cannot happen: the same conditions keep the pair packed)
    }
    int nupd = 0;
    UpdateGroup ug = g_tail_ug_none;
    if (weights_upd && weights_upd->n > 0) {
        ug = *weights_upd;
        for (int i = 0; i < ug.n; ++i) { ug.start[i] = nupd; nupd += (ug.a[i].dM * ug.a[i].dD * ug.a[i].Nk * ug.a[i].Nl + 255) / 256; }
        ug.start[ug.n] = nupd;
    }
    const size_t lds = std::max(lds_c, lds_m);
#ifndef AEFFT_X_LEANTAIL
#define AEFFT_X_LEANTAIL 1
#endif
    bool lean = AEFFT_X_LEANTAIL != 0;
    for (int i = 0; i < g.n && lean; ++i) lean = g.q[i].G != nullptr || (i == g.n - 1 && g.Wp != nullptr);      // (after opmse_geometry: Wp is null unless the innermost pair goes packed)
    const long total = nchain + nmse + nupd;
    if (total >= (1L << 31)) return hipErrorInvalidValue;
    auto go = [&](auto LEANT, auto FUSET) -> hipError_t {
        constexpr bool LN = decltype(LEANT)::value, FS = decltype(FUSET)::value;
        if (lds > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tail_kernel<LN, FS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        tail_kernel<LN, FS><<<dim3((unsigned)total), 256, lds, st>>>(g, chain ? *chain : g_tail_chain_none, ug, (int)nchain, (int)(nchain + nmse));
        return hipGetLastError();
    };
    if (lean) return fuse ? go(std::true_type{}, std::true_type{}) : go(std::true_type{}, std::false_type{});
    return fuse ? go(std::false_type{}, std::true_type{}) : go(std::false_type{}, std::false_type{});
}

}  // namespace aefft

#if defined(AEFFT_X_WGTIME) && AEFFT_X_WGTIME
extern "C" int aefft_debug_wgtime_opform(void* p) { return aefft::wgtime_set_tu(p); }
#endif
