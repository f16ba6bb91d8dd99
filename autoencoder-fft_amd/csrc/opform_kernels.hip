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
#include "device_util.h"
#include <algorithm>

namespace aefft {

// bin of the small grid [Nx][Ny/2+1] -> the bin of the big grid [NxB][NyB/2+1] it is cropped from / zero-padded to
// (pool_fft's index map, fft_backproplib.cu:102-111 and 117-152; compositions of it have the same form)
__device__ __forceinline__ long map_up(long s, int Nx, int Ny, int NxB, int NyB)
{
    const int nyr = Ny / 2 + 1, NyrB = NyB / 2 + 1;
    const int i = (int)(s / nyr), j = (int)(s - (long)i * nyr);
    const int bi = i < Nx / 2 ? i : (i == Nx / 2 ? NxB / 2 : i + NxB - Nx);
    const int bj = j < nyr - 1 ? j : NyrB - 1;
    return (long)bi * NyrB + bj;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
__device__ __forceinline__ void cfma2(float2& acc, float2 a, float2 b) { acc.x = fmaf(a.x, b.x, acc.x); acc.x = fmaf(-a.y, b.y, acc.x); acc.y = fmaf(a.x, b.y, acc.y); acc.y = fmaf(a.y, b.x, acc.y); }
__device__ __forceinline__ void cfmac(float2& acc, float2 a, float2 b) { acc.x = fmaf(a.x, b.x, acc.x); acc.x = fmaf(a.y, b.y, acc.x); acc.y = fmaf(a.y, b.x, acc.y); acc.y = fmaf(-a.x, b.y, acc.y); }   // += a * conj(b)

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
// second moments of the batch: M^[i][j][u] = sum_b x^_b[i][u] conj(x^_b[j][u]),  x^ = [x_0 .. x_{D0-1}, 0.., 1]
// Workgroup = 64 bins x 4 frame slices; the slices are summed in slice order (deterministic).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void moment_kernel(const float2* __restrict__ Xf, float2* __restrict__ M, int B, int D0, long P0)
{
    __shared__ float2 red[3][9][64];
    const long u = (long)blockIdx.x * 64 + threadIdx.x;
    const int sl = threadIdx.y;
    const long uc = u < P0 ? u : P0 - 1;
    // upper triangle without the constant (3,3): (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3)
    float2 acc[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) acc[e] = make_float2(0.f, 0.f);
    const int nb = (B + 3) / 4;                                       // frames per slice
    const int b0 = sl * nb, b1 = min(B, b0 + nb);
    for (int b = b0; b < b1; b += 4) {
        float2 x[4][3];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int bb = min(b + f, b1 - 1);
                x[f][d] = d < D0 ? Xf[((long)bb * D0 + d) * P0 + uc] : make_float2(0.f, 0.f);
            }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (b + f >= b1) break;
            const float2 x0 = x[f][0], x1 = x[f][1], x2 = x[f][2];
            cfmac(acc[0], x0, x0); cfmac(acc[1], x0, x1); cfmac(acc[2], x0, x2); acc[3].x += x0.x; acc[3].y += x0.y;
            cfmac(acc[4], x1, x1); cfmac(acc[5], x1, x2); acc[6].x += x1.x; acc[6].y += x1.y;
            cfmac(acc[7], x2, x2); acc[8].x += x2.x; acc[8].y += x2.y;
        }
    }
    if (sl > 0) {
#pragma unroll
        for (int e = 0; e < 9; ++e) red[sl - 1][e][threadIdx.x] = acc[e];
    }
    __syncthreads();
    if (sl > 0 || u >= P0) return;
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
        for (int e = 0; e < 9; ++e) { const float2 v = red[s2][e][threadIdx.x]; acc[e].x += v.x; acc[e].y += v.y; }
    const int ei[9] = {0, 0, 0, 0, 1, 1, 1, 2, 2}, ej[9] = {0, 1, 2, 3, 1, 2, 3, 2, 3};
#pragma unroll
    for (int e = 0; e < 9; ++e) {
        float2 v = acc[e];
        if (ei[e] == ej[e]) v.y = 0.f;
        M[(long)(ei[e] * OPC + ej[e]) * P0 + u] = v;
        if (ei[e] != ej[e]) M[(long)(ej[e] * OPC + ei[e]) * P0 + u] = make_float2(v.x, -v.y);
    }
    M[(long)(3 * OPC + 3) * P0 + u] = make_float2((float)B, 0.f);
}
hipError_t launch_moment(const float2* Xf, float2* Mhat, int B, int D0, long P0, hipStream_t st)
{
    if (B < 1 || D0 < 1 || D0 > OPC - 1 || P0 < 1) return hipErrorInvalidValue;
    moment_kernel<<<dim3((unsigned)((P0 + 63) / 64)), dim3(64, 4), 0, st>>>(Xf, Mhat, B, D0, P0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// S_l and es_l of every pair in one launch.  Thread = (bin s of pair l's grid, row a): E[a][.] = O^[.][a][t] - A[.][a][s],
// U[a][.] = E M^[u], then S[a][b] = sum_k U[a][k] conj(A[k][b][s]) for every b.  Lanes run along the bins (coalesced planes).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgrad_kernel(const SgradGroup g)
{
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
    const OpPair& q = g.q[p];
    const int dD = q.dD;
    const int rgroups = (dD + 3) / 4;
    const int blk = blockIdx.x - g.start[p];
    const int bx = blk / rgroups, by = blk - bx * rgroups;
    const long s = (long)bx * 64 + threadIdx.x;
    const int a = by * 4 + threadIdx.y;
    if (s >= q.P || a >= dD) return;
    const long u = map_up(s, q.Nx, q.Ny, g.Nx0, g.Ny0);
    const long t = crop_dest(s, q.Nx, q.Ny, q.NxO, q.NyO);                  // the bin of O^'s grid that lands on s, or -1
    float2 E[OPC];
#pragma unroll
    for (int j = 0; j < OPC; ++j) {
        const float2 av = q.A[((long)j * dD + a) * q.P + s];
        float2 ov = make_float2(0.f, 0.f);
        if (t >= 0) ov = q.O[((long)j * dD + a) * q.PO + t];
        E[j] = make_float2(ov.x - av.x, ov.y - av.y);
    }
    float2 U[OPC];
#pragma unroll
    for (int k = 0; k < OPC; ++k) {
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < OPC; ++j) cfma2(acc, E[j], g.Mhat[(long)(j * OPC + k) * g.P0 + u]);
        U[k] = acc;
    }
    if (s == 0) { q.es[2 * a] = U[OPC - 1].x; q.es[2 * a + 1] = U[OPC - 1].y; }
    for (int b0 = 0; b0 < dD; b0 += 4) {
        float2 av[4][OPC];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
#pragma unroll
            for (int k = 0; k < OPC; ++k) av[bb][k] = q.A[((long)k * dD + min(b0 + bb, dD - 1)) * q.P + s];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            if (b0 + bb >= dD) break;
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < OPC; ++k) cfmac(acc, U[k], av[bb][k]);
            q.S[((long)a * dD + b0 + bb) * q.P + s] = acc;
        }
    }
}
hipError_t launch_sgrad_group(SgradGroup& g, hipStream_t st)
{
    if (g.n < 1 || g.n > 8) return hipErrorInvalidValue;
    long total = 0;
    for (int i = 0; i < g.n; ++i) {
        g.start[i] = (int)total;
        total += ((g.q[i].P + 63) / 64) * ((g.q[i].dD + 3) / 4);
    }
    if (total >= (1L << 31)) return hipErrorInvalidValue;
    g.start[g.n] = (int)total;
    sgrad_kernel<<<dim3((unsigned)total), dim3(64, 4), 0, st>>>(g);
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
    const OpMsePair& q = g.q[p];
    const int dD = q.dD, dM = q.dM;
    float2* As = sh;
    float2* Ts = sh + (size_t)OPC * dD * BT;
    const long blk = blockIdx.x - g.start[p];
    const int bl = threadIdx.x % BT, ry = threadIdx.x / BT;
    const long s = blk * BT + bl;
    const bool ok = s < q.P;
    const long sc = ok ? s : q.P - 1;
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
    __syncthreads();
    const float NN = (float)q.Nx * (float)q.Ny;
    const float idM = 1.0f / (float)dM, idD = 1.0f / (float)dD;
    for (int m = ry; m < dM; m += RT) {
        float2 t[OPC];
#pragma unroll
        for (int k = 0; k < OPC; ++k) t[k] = make_float2(0.f, 0.f);
        const float2* Cp = q.C + (long)m * dD * q.P + sc;
        for (int d0 = 0; d0 < dD; d0 += 8) {
            float2 c[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) c[w] = Cp[(long)min(d0 + w, dD - 1) * q.P];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                if (d0 + w >= dD) break;
#pragma unroll
                for (int k = 0; k < OPC; ++k) cfma2(t[k], c[w], As[(k * dD + d0 + w) * BT + bl]);
            }
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
        const long u = map_up(sc, q.Nx, q.Ny, g.Nx0, g.Ny0);
        float2 Mh[OPC][OPC];
#pragma unroll
        for (int i = 0; i < OPC; ++i)
#pragma unroll
            for (int j = 0; j < OPC; ++j) Mh[i][j] = g.Mhat[(long)(i * OPC + j) * g.P0 + u];
        for (int a = ry; a < dD; a += RT) {
            float2 r[OPC];
#pragma unroll
            for (int k = 0; k < OPC; ++k) r[k] = make_float2(0.f, 0.f);
            const float2* Fp = q.F + (long)a * dM * q.P + sc;
            for (int m0 = 0; m0 < dM; m0 += 8) {
                float2 f[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) f[w] = Fp[(long)min(m0 + w, dM - 1) * q.P];
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    if (m0 + w >= dM) break;
#pragma unroll
                    for (int k = 0; k < OPC; ++k) cfma2(r[k], f[w], Ts[((m0 + w) * OPC + k) * BT + bl]);
                }
            }
#pragma unroll
            for (int k = 0; k < OPC; ++k) {
                const float2 av = As[(k * dD + a) * BT + bl];
                r[k] = make_float2(av.x - r[k].x * idD, av.y - r[k].y * idD);
            }
            if (s == 0) r[OPC - 1].x -= q.p[a] * NN;
            // sum_{k,k'} r[k] M^[k][k'] conj(r[k'])  (real)
#pragma unroll
            for (int k = 0; k < OPC; ++k) {
                float2 v = make_float2(0.f, 0.f);
#pragma unroll
                for (int k2 = 0; k2 < OPC; ++k2) cfmac(v, Mh[k][k2], r[k2]);
                part += r[k].x * v.x - r[k].y * v.y;
            }
        }
        const int nyr = q.Ny / 2 + 1;
        const int j = (int)(sc % nyr);
        part *= !ok ? 0.f : ((j > 0 && j < nyr - 1) ? 2.f : 1.f);          // Hermitian half-plane: interior columns count twice
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    float* red = reinterpret_cast<float*>(Ts + (size_t)dM * OPC * BT);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (red[0] + red[1]) + (red[2] + red[3]);
        if (tot != 0.f) atomicAdd(q.slots + (blockIdx.x % MSE_SLOTS) * MSE_SLOT_STRIDE, tot * q.scale);
    }
}

__global__ __launch_bounds__(256) void opmse_kernel(const OpMseGroup g)
{
    extern __shared__ float2 sh[];                                   // As[OPC][dD][BT] | Ts[dM][OPC][BT] | red[4]
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
    const int bt = g.bt[p];                                          // uniform per workgroup
    if (bt == 16) opmse_body<16>(g, p, sh);
    else if (bt == 8) opmse_body<8>(g, p, sh);
    else opmse_body<4>(g, p, sh);
}

hipError_t launch_opmse_group(OpMseGroup& g, hipStream_t st)
{
    if (g.n < 1 || g.n > 8) return hipErrorInvalidValue;
    long total = 0;
    size_t lds = 0;
    for (int i = 0; i < g.n; ++i) {
        const OpMsePair& q = g.q[i];
        // bins per workgroup: whole 128-byte lines when the pair still yields >= 128 workgroups and its tiles fit 64 KB of LDS
        int bt = 16;
        while (bt > 4 && ((q.P + bt - 1) / bt < 128 || (size_t)OPC * (q.dD + q.dM) * bt * sizeof(float2) > 60 * 1024)) bt >>= 1;
        const size_t need = (size_t)OPC * (q.dD + q.dM) * bt * sizeof(float2) + 64;
        if (need > 150 * 1024) return hipErrorInvalidValue;
        g.bt[i] = bt;
        lds = std::max(lds, need);
        g.start[i] = (int)total;
        total += (q.P + bt - 1) / bt;
    }
    if (total >= (1L << 31)) return hipErrorInvalidValue;
    g.start[g.n] = (int)total;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(opmse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    opmse_kernel<<<dim3((unsigned)total), 256, lds, st>>>(g);
    return hipGetLastError();
}

}  // namespace aefft
