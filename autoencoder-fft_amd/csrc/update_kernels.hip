// Coordinate-space side of the weight update for gfx950:
//   pad / shrink of the centred-wrapped Nk x Nl kernel support  (fft_backproplib.cu:535-600)
//   clipped-gradient + momentum update                          (fft_backproplib.cu:605-652, 657-704)
//   kernel-distance gradient of the multiobjective mode         (fft_backproplib.cu:709-753)
// These tensors are tiny (dM*dD*Nk*Nl floats); the kernels are launch-latency bound.
#include "internal.h"
#include "update_device.h"

namespace aefft {

// tap k of Nk -> row of the padded plane: (k - Nk/2) mod Nx   (fft_backproplib.cu:544-563)
__device__ __forceinline__ int tap_row(int k, int Nk, int Nx) { return k >= Nk / 2 ? k - Nk / 2 : k + Nx - Nk / 2; }
// inverse: padded row i -> tap index, or -1
__device__ __forceinline__ int row_tap(int i, int Nk, int Nx)
{
    if (i + Nk / 2 < Nk) return i + Nk / 2;
    if (i >= Nx - Nk / 2) return i - (Nx - Nk / 2);
    return -1;
}

// memset + pad_k in one pass: every element of the padded plane is written.
template <typename I>
__global__ __launch_bounds__(256) void pad_kernel(const float* __restrict__ ck, float* __restrict__ cpad, I planes,
                                                  int Nx, int Ny, int Nk, int Nl)
{
    const I psz = (I)Nx * Ny;
    const I total = planes * psz;
    for (I idx = (I)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (I)gridDim.x * 256) {
        const I pl = idx / psz;
        const int rem = (int)(idx - pl * psz);
        const int i = rem / Ny, j = rem - i * Ny;
        const int k = row_tap(i, Nk, Nx), l = row_tap(j, Nl, Ny);
        cpad[idx] = (k >= 0 && l >= 0) ? ck[(pl * Nk + k) * Nl + l] : 0.f;
    }
}

hipError_t launch_pad(const float* ck, float* cpad, long planes, int Nx, int Ny, int Nk, int Nl, hipStream_t st)
{
    if (Nk > Nx || Nl > Ny) return hipErrorInvalidValue;
    const long total = planes * Nx * (long)Ny;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (total < (1L << 31)) pad_kernel<unsigned><<<dim3((unsigned)blocks), 256, 0, st>>>(ck, cpad, (unsigned)planes, Nx, Ny, Nk, Nl);
    else pad_kernel<long><<<dim3((unsigned)blocks), 256, 0, st>>>(ck, cpad, planes, Nx, Ny, Nk, Nl);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void shrink_kernel(const float* __restrict__ cpad, float* __restrict__ ck, long planes,
                                                     int Nx, int Ny, int Nk, int Nl, float scale)
{
    const long total = planes * Nk * Nl;
    const long idk = (long)blockIdx.x * 256 + threadIdx.x;
    if (idk >= total) return;
    const long pl = idk / (Nk * Nl);
    const int rem = (int)(idk - pl * Nk * Nl);
    const int k = rem / Nl, l = rem % Nl;
    ck[idk] = cpad[(pl * Nx + tap_row(k, Nk, Nx)) * (long)Ny + tap_row(l, Nl, Ny)] * scale;
}

hipError_t launch_shrink(const float* cpad, float* ck, long planes, int Nx, int Ny, int Nk, int Nl, float scale, hipStream_t st)
{
    const long total = planes * Nk * Nl;
    shrink_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(cpad, ck, planes, Nx, Ny, Nk, Nl, scale);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void update_kernel(const UpdateArgs a) { update_body(a, blockIdx.x); }

// every pair's update in one launch (the pairs are independent; each launch otherwise costs ~4.7 us for ~1 us of work)
__global__ __launch_bounds__(256) void update_group_kernel(const UpdateGroup g)
{
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
    update_body(g.a[p], blockIdx.x - g.start[p]);
}

hipError_t launch_update_group(UpdateGroup& g, hipStream_t st)
{
    if (g.n < 1 || g.n > 8) return hipErrorInvalidValue;
    int total = 0;
    for (int i = 0; i < g.n; ++i) {
        const UpdateArgs& a = g.a[i];
        const int n = a.dM * a.dD * a.Nk * a.Nl;
        if (n < a.dM || n < a.dD) return hipErrorInvalidValue;
        g.start[i] = total; total += (n + 255) / 256;
    }
    g.start[g.n] = total;
    update_group_kernel<<<dim3(total), 256, 0, st>>>(g);
    return hipGetLastError();
}

hipError_t launch_update(const UpdateArgs& a, hipStream_t st)
{
    const int n = a.dM * a.dD * a.Nk * a.Nl;
    if (n < a.dM || n < a.dD) return hipErrorInvalidValue;   // the reference's idk<dM / idk<dD trick needs this too
    update_kernel<<<dim3((n + 255) / 256), 256, 0, st>>>(a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// gradient_diff (fft_backproplib.cu:709-753).  The reference recomputes the squared kernel
// distance den(m,d;m1,d1) inside every (k,l) thread; here it is computed once per kernel pair
// (same k1,l1 summation order, so identical floats) into `den_ws` = 2*(dM*dD)^2 floats, then
// each (m,d,k,l) thread sums (w - w1)/den over the partners in the reference's (m1,d1) order.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kdist_kernel(const float* __restrict__ c, const float* __restrict__ f,
                                                    float* __restrict__ den, int dM, int dD, int kl)
{
    const long np = (long)dM * dD;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= np * np) return;
    const int a = (int)(idx / np), b = (int)(idx % np);
    const int m = a / dD, d = a % dD, m1 = b / dD, d1 = b % dD;
    float dc = 0.f, df = 0.f;
    const float* ca = c + (long)a * kl;
    const float* cb = c + (long)b * kl;
    const float* fa = f + ((long)d * dM + m) * kl;
    const float* fb = f + ((long)d1 * dM + m1) * kl;
    for (int t = 0; t < kl; ++t) {
        const float x = ca[t] - cb[t], y = fa[t] - fb[t];
        dc += x * x; df += y * y;
    }
    den[idx] = dc;
    den[np * np + idx] = df;
}

__global__ __launch_bounds__(256) void gradient_diff_kernel(const float* __restrict__ c, const float* __restrict__ f,
                                                            const float* __restrict__ b, const float* __restrict__ p,
                                                            const float* __restrict__ den, float* __restrict__ cd,
                                                            float* __restrict__ fd, float* __restrict__ bd,
                                                            float* __restrict__ pd, int dM, int dD, int Nk, int Nl)
{
    const int kl = Nk * Nl;
    const long np = (long)dM * dD;
    const long idk = (long)blockIdx.x * 256 + threadIdx.x;
    if (idk >= np * kl) return;
    const int a = (int)(idk / kl), r = (int)(idk % kl);
    const int m = a / dD, d = a % dD;
    const float cw = c[idk], fw = f[((long)d * dM + m) * kl + r];
    float sc = 0.f, sf = 0.f, sb = 0.f, sp = 0.f;
    for (int m1 = 0; m1 < dM; ++m1) {
        for (int d1 = 0; d1 < dD; ++d1) {
            if (m1 != m && d1 != d) {
                const long pb = (long)m1 * dD + d1;
                sc += (cw - c[pb * kl + r]) / den[(long)a * np + pb];
                sf += (fw - f[((long)d1 * dM + m1) * kl + r]) / den[np * np + (long)a * np + pb];
            }
            if (m1 == 0 && d1 != d) sp += 1.f / (p[d] - p[d1]);
        }
        if (m1 != m) sb += 1.f / (b[m] - b[m1]);
    }
    cd[idk] = sc;
    fd[((long)d * dM + m) * kl + r] = sf;
    // every thread of a given m (d) writes the same value; keep a single writer
    if (d == 0 && r == 0) bd[m] = sb;
    if (m == 0 && r == 0) pd[d] = sp;
}

hipError_t launch_gradient_diff(const float* c, const float* f, const float* b, const float* p, float* cd, float* fd,
                                float* bd, float* pd, float* den_ws, int dM, int dD, int Nk, int Nl, hipStream_t st)
{
    const long np = (long)dM * dD;
    const int kl = Nk * Nl;
    kdist_kernel<<<dim3((unsigned)((np * np + 255) / 256)), 256, 0, st>>>(c, f, den_ws, dM, dD, kl);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    gradient_diff_kernel<<<dim3((unsigned)((np * kl + 255) / 256)), 256, 0, st>>>(c, f, b, p, den_ws, cd, fd, bd, pd, dM, dD, Nk, Nl);
    return hipGetLastError();
}

}  // namespace aefft
