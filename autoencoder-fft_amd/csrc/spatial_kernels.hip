// Coordinate-space ("spatial mode") kernels for gfx950.
//
//   conv_spatial_kernel : zero-padded direct convolution, Conv_gpu / conv_parallel semantics
//                         (backproplib.cu:70-111,114-182): tap offset ik = -2*ak-1+k with
//                         ak = ((Nk-1)/2-1)/2, input pre-divided by dM (:134), +b[m], identity act.
//                         The same kernel with lo=1, ak=(Nk-1)/2-1 and no division reproduces the
//                         CPU Conv (netlib.cpp:318-358).
//   back-conv + weight-gradient correlation: backprop_gpu (backproplib.cu:291-418).  The
//                         reference launches one kernel + 2-4 thrust::reduce per weight element
//                         (dM*dD*Nk*Nl round trips) and recomputes the back-convolution through f
//                         for every element (O(K^4)).  Here it is two stages over the whole batch:
//                           g[m][i'][j']  = sum_{d1,k1,l1} s0[d1][i'+ik1][j'+il1] * f[d1][m][k1][l1]
//                           dC[m][d][k][l]= sum_{i',j'} g[m][i'][j'] * in[d][i'-ik][j'-il] / Norm
//                           dF[d][m][k][l]= sum_{i,j}   s0[d][i][j]  * hin[m][i-ik][j-il]   / Norm
//                           dB[m] = sum g[m] / Norm ; dP[d] = sum s0[d] / Norm,   s0 = out - in
//                         with the reference's range tests on every shifted index (lo = 0: '>=0',
//                         backproplib.cu:209,213; lo = 1: '>0', netlib.cpp:412,416).
#include "internal.h"

namespace aefft {

__global__ __launch_bounds__(256) void conv_spatial_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           const float* __restrict__ c, const float* __restrict__ b,
                                                           int B, int dD, int dM, int Nx, int Ny, int Nk, int Nl, int ak, int al,
                                                           float div, int lo)
{
    const long total = (long)B * dM * Nx * Ny;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= total) return;
    const int j = (int)(n % Ny), i = (int)((n / Ny) % Nx);
    const int m = (int)((n / ((long)Nx * Ny)) % dM);
    const long bb = n / ((long)Nx * Ny * dM);
    const float* inb = in + bb * dD * (long)Nx * Ny;
    float h = 0.f;
    for (int d = 0; d < dD; ++d) {
        for (int k = 0; k < Nk; ++k) {
            const int ii = i - (-2 * ak - 1 + k);
            if (ii < lo || ii >= Nx) continue;
            for (int l = 0; l < Nl; ++l) {
                const int jj = j - (-2 * al - 1 + l);
                if (jj < lo || jj >= Ny) continue;
                float x = inb[((long)d * Nx + ii) * Ny + jj];
                if (div != 1.f) x = x / div;
                h += c[((m * dD + d) * Nk + k) * Nl + l] * x;
            }
        }
    }
    out[n] = h + b[m];
}

hipError_t launch_conv_spatial(const float* in, float* out, const float* c, const float* b, int B, int dD, int dM,
                               int Nx, int Ny, int Nk, int Nl, int ak, int al, float in_scale_div, int lo, hipStream_t st)
{
    const long total = (long)B * dM * Nx * Ny;
    if (total <= 0) return hipSuccess;
    conv_spatial_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(in, out, c, b, B, dD, dM, Nx, Ny, Nk, Nl, ak, al, in_scale_div, lo);
    return hipGetLastError();
}

// g[b][m][i'][j'] : back-convolution of the error through f
__global__ __launch_bounds__(256) void backconv_kernel(const SpatialGradArgs a)
{
    const long total = (long)a.B * a.dM * a.Nx * a.Ny;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= total) return;
    const int jp = (int)(n % a.Ny), ip = (int)((n / a.Ny) % a.Nx);
    const int m = (int)((n / ((long)a.Nx * a.Ny)) % a.dM);
    const long bb = n / ((long)a.Nx * a.Ny * a.dM);
    float g = 0.f;
    if (ip >= a.lo && jp >= a.lo) {
        const long base = bb * a.dD * (long)a.Nx * a.Ny;
        for (int d1 = 0; d1 < a.dD; ++d1)
            for (int k1 = 0; k1 < a.Nk; ++k1) {
                const int i = ip + (-2 * a.ak - 1 + k1);
                if (i < 0 || i >= a.Nx) continue;
                for (int l1 = 0; l1 < a.Nl; ++l1) {
                    const int j = jp + (-2 * a.al - 1 + l1);
                    if (j < 0 || j >= a.Ny) continue;
                    const long q = base + ((long)d1 * a.Nx + i) * a.Ny + j;
                    g += (a.out[q] - a.in[q]) * a.f[((d1 * a.dM + m) * a.Nk + k1) * a.Nl + l1];
                }
            }
    }
    a.ws[n] = g;
}

__device__ __forceinline__ float block_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __shared__ float w[4];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = v;
    __syncthreads();
    return w[0] + w[1] + w[2] + w[3];
}

// one block per weight element: blockIdx.x over [dM*dD*Nk*Nl] (c-gradient) then the same count (f-gradient)
__global__ __launch_bounds__(256) void wgrad_kernel(const SpatialGradArgs a)
{
    const int kl = a.Nk * a.Nl, nw = a.dM * a.dD * kl;
    const bool isF = (int)blockIdx.x >= nw;
    const int e = isF ? blockIdx.x - nw : blockIdx.x;
    const int l = e % a.Nl, k = (e / a.Nl) % a.Nk;
    int m, d;
    if (!isF) { d = (e / kl) % a.dD; m = e / (kl * a.dD); } else { m = (e / kl) % a.dM; d = e / (kl * a.dM); }
    const int ik = -2 * a.ak - 1 + k, il = -2 * a.al - 1 + l;
    const long plane = (long)a.Nx * a.Ny;
    float s = 0.f;
    for (long t = threadIdx.x; t < a.B * plane; t += 256) {
        const long bb = t / plane;
        const int i = (int)((t % plane) / a.Ny), j = (int)(t % a.Ny);
        const int ii = i - ik, jj = j - il;
        if (ii < a.lo || ii >= a.Nx || jj < a.lo || jj >= a.Ny) continue;
        if (!isF) {
            s += a.ws[(bb * a.dM + m) * plane + t % plane] * a.in[(bb * a.dD + d) * plane + (long)ii * a.Ny + jj];
        } else {
            const long q = (bb * a.dD + d) * plane + t % plane;
            s += (a.out[q] - a.in[q]) * a.hin[(bb * a.dM + m) * plane + (long)ii * a.Ny + jj];
        }
    }
    s = block_sum(s);
    if (threadIdx.x == 0) (isF ? a.gf : a.gc)[e] = s / a.Norm / (float)a.B;
}

// gb[m] = sum g[m] / Norm ; gp[d] = sum s0[d] / Norm    (blockIdx.x over dM + dD)
__global__ __launch_bounds__(256) void bgrad_kernel(const SpatialGradArgs a)
{
    const long plane = (long)a.Nx * a.Ny;
    const bool isP = (int)blockIdx.x >= a.dM;
    const int ch = isP ? blockIdx.x - a.dM : blockIdx.x;
    float s = 0.f;
    for (long t = threadIdx.x; t < a.B * plane; t += 256) {
        const long bb = t / plane, r = t % plane;
        if (!isP) s += a.ws[(bb * a.dM + ch) * plane + r];
        else { const long q = (bb * a.dD + ch) * plane + r; s += a.out[q] - a.in[q]; }
    }
    s = block_sum(s);
    if (threadIdx.x == 0) (isP ? a.gp : a.gb)[ch] = s / a.Norm / (float)a.B;
}

hipError_t launch_spatial_grad(const SpatialGradArgs& a, hipStream_t st)
{
    const long total = (long)a.B * a.dM * a.Nx * a.Ny;
    backconv_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int nw = a.dM * a.dD * a.Nk * a.Nl;
    wgrad_kernel<<<dim3(2 * nw), 256, 0, st>>>(a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    bgrad_kernel<<<dim3(a.dM + a.dD), 256, 0, st>>>(a);
    return hipGetLastError();
}

}  // namespace aefft
