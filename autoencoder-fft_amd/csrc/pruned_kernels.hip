// Kernel-support-pruned transforms for the weight side of the training loop (gfx950).
//
// The reference pads each Nk x Nl kernel to Nx x Ny and runs a full batched R2C over dM*dD
// planes (fft_backproplib.cu:1274-1282), and runs a full unnormalised C2R over the dM*dD gradient
// spectra only to keep Nk x Nl samples of each (fft_backproplib.cu:1219-1226).  Both are linear maps
// with a tiny support, evaluated here directly:
//
//   kspec_kernel : K[i][j] = sum_{k,l} c[k][l] * e^{-2 pi i (i*kap_k/Nx + j*lam_l/Ny)},
//                  kap_k = k - Nk/2, lam_l = l - Nl/2  (the centred-wrapped tap positions of
//                  kernel_pad / pad_k, fft_backproplib.cu:1034-1058,579-598).  Write-only over the spectrum.
//   kgrad_kernel : g[k][l] = scale * sum_{i,j} w_j * Re( D[i][j] * e^{+2 pi i (i*kap_k/Nx + j*lam_l/Ny)} ),   (separable, two steps)
//                  w_j = 1 on the self-conjugate columns j = 0, Ny/2 and 2 elsewhere: the unnormalised
//                  C2R sampled at the kernel support (== shrink_k(C2R(D)), imaginary parts of
//                  self-conjugate bins ignored).  Read-only over the spectrum.
//
// Row and column phase tables (Nx*Nk and Nyr*Nl complex) are built per workgroup in LDS from the
// global twiddle table, so each bin costs one 8-byte global access plus LDS reads.
#include "internal.h"
#include <algorithm>

namespace aefft {

__device__ __forceinline__ float2 phase(const float2* tw, int pos, int off, int N, float sign)
{
    // e^{sign * -2 pi i * pos*off / N}; pos in [0,N), off may be negative; N is a power of two, so the
    // two's-complement AND is the non-negative residue (a 64-bit '%' costs hundreds of cycles on the GPU)
    const int r = (pos * off) & (N - 1);
    float2 w = tw[r * (TW_N / N)];
    w.y *= sign;
    return w;
}

// Thread <-> (plane, column j); the column factor v_k = sum_l c[k][l] e^{-2 pi i j lam_l / Ny} is row independent, so
// it is formed once per thread (25 real x complex FMAs) and every output row then costs 5 complex multiplies with
// the row phases (LDS broadcast reads).  Consecutive threads own consecutive columns: coalesced stores.
template <int NK, int NL>
__global__ __launch_bounds__(320) void kspec_kernel(const float* __restrict__ kern, float2* __restrict__ K,
                                                    const float2* __restrict__ tw, long planes, int Nx, int Ny,
                                                    int rows_per_chunk, int ppb)
{
    extern __shared__ float2 lds[];
    const int Nyr = Ny / 2 + 1;
    float2* rowph = lds;                                 // [rows_per_chunk][NK]
    const int i0 = blockIdx.y * rows_per_chunk;
    const int nrows = min(rows_per_chunk, Nx - i0);
    for (int t = threadIdx.x; t < nrows * NK; t += blockDim.x) rowph[t] = phase(tw, i0 + t / NK, t % NK - NK / 2, Nx, 1.f);
    __syncthreads();
    const int pl = threadIdx.x / Nyr, j = threadIdx.x - pl * Nyr;
    const long plane = (long)blockIdx.x * ppb + pl;
    if (pl >= ppb || plane >= planes) return;
    float2 v[NK];
    {
        float2 cp[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) cp[l] = phase(tw, j, l - NL / 2, Ny, 1.f);
        const float* c = kern + plane * NK * NL;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            v[k] = make_float2(0.f, 0.f);
#pragma unroll
            for (int l = 0; l < NL; ++l) { const float w = c[k * NL + l]; v[k].x += w * cp[l].x; v[k].y += w * cp[l].y; }
        }
    }
    float2* dst = K + (plane * Nx + i0) * (long)Nyr + j;
    for (int i = 0; i < nrows; ++i) {
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float2 rp = rowph[i * NK + k];
            acc.x += v[k].x * rp.x - v[k].y * rp.y;
            acc.y += v[k].x * rp.y + v[k].y * rp.x;
        }
        dst[(long)i * Nyr] = acc;
    }
}

// Thread <-> (plane, column j) over a chunk of RB rows:
//   t_k[j]   = sum_i D[i][j] * e^{+2 pi i i*kap_k/Nx}                  (row phases: LDS broadcast reads; D read once, coalesced)
//   g[k][l] += w_j * Re( t_k[j] * e^{+2 pi i j*lam_l/Ny} )              (summed over the plane's columns through LDS, fixed order)
// Output: part[plane][chunk][NK*NL] (summed over chunks by ksum_kernel when chunks > 1).  Deterministic.
template <int NK, int NL>
__global__ __launch_bounds__(320) void kgrad_kernel(const float2* __restrict__ D, float* __restrict__ part,
                                                    const float2* __restrict__ tw, long planes, int Nx, int Ny, int RB, int ppb, float scale)
{
    extern __shared__ float2 lds[];
    const int Nyr = Ny / 2 + 1;
    const int nthr = blockDim.x;
    float2* rowph = lds;                                            // [RB][NK]
    float* contrib = reinterpret_cast<float*>(rowph + RB * NK);     // [NK*NL][nthr]
    const int chunk = blockIdx.y, nchunks = gridDim.y;
    const int i0 = chunk * RB;
    for (int t = threadIdx.x; t < RB * NK; t += nthr) rowph[t] = phase(tw, i0 + t / NK, t % NK - NK / 2, Nx, -1.f);
    __syncthreads();
    const int pl = threadIdx.x / Nyr, j = threadIdx.x - pl * Nyr;
    const long plane = (long)blockIdx.x * ppb + pl;
    const bool active = pl < ppb && plane < planes;
    float2 t[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) t[k] = make_float2(0.f, 0.f);
    if (active) {
        const float2* src = D + (plane * Nx + i0) * (long)Nyr + j;
        // batches of 8 rows: all 8 loads are issued before the first use (hipcc does not pipeline loads across
        // loop iterations by itself, and this loop is otherwise one memory round trip per row)
        int i = 0;
        for (; i + 8 <= RB; i += 8) {
            float2 d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) d[u] = src[(long)(i + u) * Nyr];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const float2 rp = rowph[(i + u) * NK + k];
                    t[k].x += d[u].x * rp.x - d[u].y * rp.y;
                    t[k].y += d[u].x * rp.y + d[u].y * rp.x;
                }
        }
        for (; i < RB; ++i) {
            const float2 d = src[(long)i * Nyr];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const float2 rp = rowph[i * NK + k];
                t[k].x += d.x * rp.x - d.y * rp.y;
                t[k].y += d.x * rp.y + d.y * rp.x;
            }
        }
        const float wj = (j == 0 || j == Ny / 2) ? 1.f : 2.f;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const float2 cp = phase(tw, j, l - NL / 2, Ny, -1.f);
#pragma unroll
            for (int k = 0; k < NK; ++k) contrib[(k * NL + l) * nthr + threadIdx.x] = wj * (t[k].x * cp.x - t[k].y * cp.y);
        }
    }
    __syncthreads();
    for (int it = threadIdx.x; it < ppb * NK * NL; it += nthr) {
        const int p2 = it / (NK * NL), kl = it - p2 * (NK * NL);
        const long pln = (long)blockIdx.x * ppb + p2;
        if (pln >= planes) continue;
        float g = 0.f;
        for (int jj = 0; jj < Nyr; ++jj) g += contrib[kl * nthr + p2 * Nyr + jj];
        part[(pln * nchunks + chunk) * (NK * NL) + kl] = g * scale;
    }
}

// g[e] = sum_chunk part[plane][chunk][tap]
__global__ __launch_bounds__(256) void ksum_kernel(const float* __restrict__ part, float* __restrict__ g, long n, int taps, int nchunks)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const long plane = e / taps; const int t = (int)(e % taps);
    float s = 0.f;
    for (int c = 0; c < nchunks; ++c) s += part[(plane * nchunks + c) * taps + t];
    g[e] = s;
}

static size_t kspec_lds(int rows, int Nk) { return sizeof(float2) * (size_t)rows * Nk; }
// planes per workgroup (one thread per column) and rows per workgroup (>= 8, fewer when that would leave the
// chip short of workgroups)
static void kgrad_geom(long planes, int Nx, int Ny, int* RB, int* ppb, int* threads)
{
    const int Nyr = Ny / 2 + 1;
    *ppb = std::max(1, 256 / Nyr);
    *threads = ((*ppb * Nyr + 63) / 64) * 64;
    const long pblocks = (planes + *ppb - 1) / *ppb;
    int chunks = 1;       // row chunks cost a second (ksum) launch: only when one workgroup per plane group would starve the chip
    if (pblocks < 96) while (pblocks * chunks < 256 && Nx / (chunks * 2) >= 16) chunks *= 2;
    *RB = Nx / chunks;
}
static size_t kgrad_lds(long planes, int Nx, int Ny, int Nk, int Nl)
{
    int RB, ppb, thr;
    kgrad_geom(planes, Nx, Ny, &RB, &ppb, &thr);
    return sizeof(float2) * (size_t)RB * Nk + sizeof(float) * (size_t)Nk * Nl * thr;
}
size_t kgrad_partial_floats(long planes, int Nx, int Ny, int Nk, int Nl)
{
    int RB, ppb, thr;
    kgrad_geom(planes, Nx, Ny, &RB, &ppb, &thr);
    return (size_t)planes * (Nx / RB) * Nk * Nl;
}

bool pruned_supported(int Nk, int Nl, int Nx, int Ny)
{
    if (!((Nk == 3 && Nl == 3) || (Nk == 5 && Nl == 5) || (Nk == 7 && Nl == 7))) return false;
    if (Nx > TW_N || Ny > TW_N || (TW_N % Nx) || (TW_N % Ny) || Ny / 2 + 1 > 320) return false;
    return kgrad_lds(1, Nx, Ny, Nk, Nl) <= 150 * 1024 && kgrad_lds(1L << 20, Nx, Ny, Nk, Nl) <= 150 * 1024;
}

template <int NK, int NL>
static hipError_t run_kspec(const float* k, float2* K, const float2* tw, long planes, int Nx, int Ny, hipStream_t st)
{
    const int Nyr = Ny / 2 + 1;
    if (Nyr > 320) return hipErrorInvalidValue;          // one thread per column (callers fall back to pad + R2C above 512)
    const int ppb = std::max(1, 256 / Nyr);              // planes per workgroup
    const int threads = ((ppb * Nyr + 63) / 64) * 64;
    const long pblocks = (planes + ppb - 1) / ppb;
    int chunks = 1;                                       // split the rows until the chip has ~1024 workgroups (>= 8 rows each)
    while (pblocks * chunks < 1024 && Nx / (chunks * 2) >= 8) chunks *= 2;
    const int rows = (Nx + chunks - 1) / chunks;
    kspec_kernel<NK, NL><<<dim3((unsigned)pblocks, chunks), threads, kspec_lds(rows, NK), st>>>(k, K, tw, planes, Nx, Ny, rows, ppb);
    return hipGetLastError();
}

template <int NK, int NL>
static hipError_t run_kgrad(const float2* D, float* g, float* part, const float2* tw, long planes, int Nx, int Ny, float scale, hipStream_t st)
{
    int RB, ppb, thr;
    kgrad_geom(planes, Nx, Ny, &RB, &ppb, &thr);
    const int chunks = Nx / RB;
    const size_t lds = kgrad_lds(planes, Nx, Ny, NK, NL);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kgrad_kernel<NK, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    float* dst = chunks == 1 ? g : part;
    kgrad_kernel<NK, NL><<<dim3((unsigned)((planes + ppb - 1) / ppb), chunks), thr, lds, st>>>(D, dst, tw, planes, Nx, Ny, RB, ppb, scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || chunks == 1) return e;
    const long n = planes * NK * NL;
    ksum_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(part, g, n, NK * NL, chunks);
    return hipGetLastError();
}

hipError_t launch_kspec(const float* k, float2* K, const float2* tw, long planes, int Nx, int Ny, int Nk, int Nl, hipStream_t st)
{
    if (!pruned_supported(Nk, Nl, Nx, Ny) || !tw) return hipErrorInvalidValue;
    if (planes <= 0) return hipSuccess;
    if (Nk == 3) return run_kspec<3, 3>(k, K, tw, planes, Nx, Ny, st);
    if (Nk == 5) return run_kspec<5, 5>(k, K, tw, planes, Nx, Ny, st);
    return run_kspec<7, 7>(k, K, tw, planes, Nx, Ny, st);
}

hipError_t launch_kgrad(const float2* D, float* g, float* part, const float2* tw, long planes, int Nx, int Ny, int Nk, int Nl, float scale, hipStream_t st)
{
    if (!pruned_supported(Nk, Nl, Nx, Ny) || !tw) return hipErrorInvalidValue;
    if (planes <= 0) return hipSuccess;
    if (Nk == 3) return run_kgrad<3, 3>(D, g, part, tw, planes, Nx, Ny, scale, st);
    if (Nk == 5) return run_kgrad<5, 5>(D, g, part, tw, planes, Nx, Ny, scale, st);
    return run_kgrad<7, 7>(D, g, part, tw, planes, Nx, Ny, scale, st);
}

}  // namespace aefft
