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
#include "internal.h"

namespace aefft {

__device__ __forceinline__ void cfma(float2& acc, float2 a, float2 b)
{
    acc.x += a.x * b.x - a.y * b.y;
    acc.y += a.x * b.y + a.y * b.x;
}

template <int TR, int TC, bool CA, bool CB>
__global__ __launch_bounds__(256) void contract_kernel(const Contract q)
{
    const long pair = (long)blockIdx.x * 256 + threadIdx.x;      // index of the float4 (two bins)
    if (pair * 2 >= q.P) return;
    const int r0 = blockIdx.y * TR, c0 = blockIdx.z * TC;
    const float4* A4 = reinterpret_cast<const float4*>(q.A);
    const float4* B4 = reinterpret_cast<const float4*>(q.B);
    long aoff[TR], boff[TC];
    int rr[TR];
#pragma unroll
    for (int i = 0; i < TR; ++i) { rr[i] = (r0 + i < q.R) ? r0 + i : q.R - 1; aoff[i] = (rr[i] * q.a_r) / 2 + pair; }
#pragma unroll
    for (int j = 0; j < TC; ++j) { const int cc = (c0 + j < q.C) ? c0 + j : q.C - 1; boff[j] = (cc * q.b_c) / 2 + pair; }
    const long a_k2 = q.a_k / 2, b_k2 = q.b_k / 2;

    float2 acc0[TR][TC], acc1[TR][TC];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) acc0[i][j] = acc1[i][j] = make_float2(0.f, 0.f);

    const bool dc_thread = (pair == 0);
    for (int k = 0; k < q.K; ++k) {
        float4 a[TR], b[TC];
#pragma unroll
        for (int i = 0; i < TR; ++i) {
            a[i] = A4[aoff[i] + k * a_k2];
            if (CA) { a[i].y = -a[i].y; a[i].w = -a[i].w; }
        }
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            b[j] = B4[boff[j] + k * b_k2];
            if (q.preDivB != 0.f) { b[j].x /= q.preDivB; b[j].y /= q.preDivB; b[j].z /= q.preDivB; b[j].w /= q.preDivB; }
            if (CB) { b[j].y = -b[j].y; b[j].w = -b[j].w; }
        }
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
            for (int j = 0; j < TC; ++j) {
                cfma(acc0[i][j], make_float2(a[i].x, a[i].y), make_float2(b[j].x, b[j].y));
                cfma(acc1[i][j], make_float2(a[i].z, a[i].w), make_float2(b[j].z, b[j].w));
            }
        if (k == 0 && q.bias && q.biasAfterFirst && dc_thread) {
#pragma unroll
            for (int i = 0; i < TR; ++i)
#pragma unroll
                for (int j = 0; j < TC; ++j) acc0[i][j].x += q.bias[rr[i]] * q.biasScale;
        }
    }
    float4* O4 = reinterpret_cast<float4*>(q.Out);
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            if (r0 + i >= q.R || c0 + j >= q.C) continue;
            float2 v0 = acc0[i][j], v1 = acc1[i][j];
            if (q.bias && !q.biasAfterFirst && dc_thread) v0.x += q.bias[rr[i]] * q.biasScale;
            if (q.postDiv != 0.f) { v0.x /= q.postDiv; v0.y /= q.postDiv; v1.x /= q.postDiv; v1.y /= q.postDiv; }
            O4[((r0 + i) * q.o_r + (c0 + j) * q.o_c) / 2 + pair] = make_float4(v0.x, v0.y, v1.x, v1.y);
        }
}

template <int TR, int TC> static hipError_t contract_tile(const Contract& q, hipStream_t st)
{
    const long pairs = q.P / 2;
    dim3 grid((unsigned)((pairs + 255) / 256), (unsigned)((q.R + TR - 1) / TR), (unsigned)((q.C + TC - 1) / TC));
    if (q.conjA && q.conjB) contract_kernel<TR, TC, true, true><<<grid, 256, 0, st>>>(q);
    else if (q.conjA) contract_kernel<TR, TC, true, false><<<grid, 256, 0, st>>>(q);
    else if (q.conjB) contract_kernel<TR, TC, false, true><<<grid, 256, 0, st>>>(q);
    else contract_kernel<TR, TC, false, false><<<grid, 256, 0, st>>>(q);
    return hipGetLastError();
}

hipError_t launch_contract(const Contract& q, hipStream_t st)
{
    // every plane offset must keep float4 alignment: P and all strides even (P = Nx*(Ny/2+1), Nx even)
    if ((q.P & 1) || (q.a_r & 1) || (q.a_k & 1) || (q.b_k & 1) || (q.b_c & 1) || (q.o_r & 1) || (q.o_c & 1)) return hipErrorInvalidValue;
    if (q.R <= 0 || q.C <= 0 || q.K <= 0) return hipErrorInvalidValue;
    if (q.C == 1) return contract_tile<4, 1>(q, st);
    if (q.R == 1) return contract_tile<1, 4>(q, st);
    if (q.C < 4) return contract_tile<4, 2>(q, st);
    return contract_tile<4, 4>(q, st);
}

// ------------------------------------------------------------------------------------------
// spectral pooling (fft_backproplib.cu:87-157).  Destination fully written (zeros where the
// reference relies on its cudaMemset, :990).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_kernel(const float2* __restrict__ in, float2* __restrict__ out, long planes,
                                                     int Nx, int Ny, int Nxs, int Nys)
{
    const int Nyr = Ny / 2 + 1, Nyrs = Nys / 2 + 1;
    const long total = planes * Nxs * (long)Nyrs;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long d = idx / ((long)Nxs * Nyrs);
        const int rem = (int)(idx - d * (long)Nxs * Nyrs);
        const int i = rem / Nyrs, j = rem % Nyrs;
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
        if (si >= 0 && sj >= 0) v = in[(d * Nx + si) * (long)Nyr + sj];
        out[idx] = v;
    }
}

hipError_t launch_resize(const float2* in, float2* out, long planes, int Nx, int Ny, int Nxs, int Nys, hipStream_t st)
{
    const long total = planes * Nxs * (long)(Nys / 2 + 1);
    if (total <= 0) return hipSuccess;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    resize_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(in, out, planes, Nx, Ny, Nxs, Nys);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// E = O - T, and the spectral MSE (fft_backproplib.cu:480-498 + 1188-1190):
//   mse = sum_bins |T-O|^2 / n_bin / (2*dM*Nx*Ny),  n_bin = dD*Nx*Ny, halved for columns 0<j<Nyr-1.
// For a batch the mean over frames is accumulated: *mse_acc += partial * scale.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void diff_mse_kernel(const float2* __restrict__ T, const float2* __restrict__ O,
                                                       float2* __restrict__ E, float* __restrict__ mse_acc, long total,
                                                       int Nyr, float nfull, float scale)
{
    float part = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const float2 t = T[idx], o = O[idx];
        const float dx = o.x - t.x, dy = o.y - t.y;
        if (E) E[idx] = make_float2(dx, dy);
        const int j = (int)(idx % Nyr);
        float n = nfull;
        if (j > 0 && j < Nyr - 1) n /= 2;
        part += (dx * dx + dy * dy) / n;
    }
    if (!mse_acc) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    __shared__ float wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(mse_acc, (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * scale);
}

hipError_t launch_diff_mse(const float2* T, const float2* O, float2* E, float* mse_acc, int B, int ch, int Nx, int Ny, float scale, hipStream_t st)
{
    const int Nyr = Ny / 2 + 1;
    const long total = (long)B * ch * Nx * Nyr;
    if (total <= 0) return hipSuccess;
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    diff_mse_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(T, O, E, mse_acc, total, Nyr, (float)ch * Nx * Ny, scale);
    return hipGetLastError();
}

// db[m] = (1/B) sum_b Re G_b[m](0) * norm / Norm ; dp[d] = (1/B) sum_b Re E_b[d](0) * norm / Norm
__global__ void bias_grad_kernel(const float2* __restrict__ G, const float2* __restrict__ E, float* __restrict__ db,
                                 float* __restrict__ dp, int B, int dM, int dD, long P, float norm, float Norm)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < dM) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += G[((long)b * dM + i) * P].x * norm / Norm;
        db[i] = s / (float)B;
    } else if (i < dM + dD) {
        const int d = i - dM;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += E[((long)b * dD + d) * P].x * norm / Norm;
        dp[d] = s / (float)B;
    }
}

hipError_t launch_bias_grad(const float2* G, const float2* E, float* db, float* dp, int B, int dM, int dD, long P,
                            float norm, float Norm, hipStream_t st)
{
    const int n = dM + dD;
    bias_grad_kernel<<<dim3((n + 63) / 64), 64, 0, st>>>(G, E, db, dp, B, dM, dD, P, norm, Norm);
    return hipGetLastError();
}

}  // namespace aefft
