// Device-side pieces of the operator form that more than one kernel file hosts (opform_kernels.hip, pruned_kernels.hip).
#pragma once
#include "internal.h"

namespace aefft {

// bin of the small grid [Nx][Ny/2+1] -> the bin of the big grid [NxB][NyB/2+1] it is cropped from / zero-padded to
// (pool_fft's index map, fft_backproplib.cu:102-111 and 117-152; compositions of it have the same form)
// (32-bit arithmetic: a plane has at most 2048 * 1025 bins, and 64-bit division costs hundreds of cycles)
__device__ __forceinline__ long map_up(long s, int Nx, int Ny, int NxB, int NyB)
{
    const unsigned nyr = Ny / 2 + 1, NyrB = NyB / 2 + 1;
    const unsigned i = (unsigned)s / nyr, j = (unsigned)s - i * nyr;
    const unsigned bi = i < (unsigned)Nx / 2 ? i : (i == (unsigned)Nx / 2 ? (unsigned)NxB / 2 : i + NxB - Nx);
    const unsigned bj = j < nyr - 1 ? j : NyrB - 1;
    return (long)(bi * NyrB + bj);
}
// the bin of the small grid [Nxs][Nys/2+1] that lands on bin `bin` of the big grid [Nx][Ny/2+1], or -1 (crop_dest in 32 bits)
__device__ __forceinline__ int crop_dest32(long bin, int Nx, int Ny, int Nxs, int Nys)
{
    const int Nyr = Ny / 2 + 1, Nyrs = Nys / 2 + 1;
    const int i = (int)((unsigned)bin / (unsigned)Nyr), j = (int)((unsigned)bin - (unsigned)i * Nyr);
    int di = -1, dj = -1;
    if (i < Nxs / 2) di = i;
    else if (i == Nx / 2) di = Nxs / 2;
    else if (i > Nx - Nxs / 2) di = i - Nx + Nxs;
    if (j < Nyrs - 1) dj = j;
    else if (j == Nyr - 1) dj = Nyrs - 1;
    return (di >= 0 && dj >= 0) ? di * Nyrs + dj : -1;
}

__device__ __forceinline__ float2 phase_tw(const float2* tw, int pos, int off, int N)
{
    // e^{-2 pi i pos*off / N}; N a power of two (pruned_kernels.hip `phase`)
    return tw[((pos * off) & (N - 1)) * (TW_N / N)];
}

// Wp[t][e]: for support bin t the elements of C_0 .. C_{L-1}, F_{L-1} .. F_0 (chain order) at the bins of their grids that t
// maps to.  Workgroup = 256 consecutive elements x TB support bins; thread = one element: its Nk*Nl taps stay in registers,
// per bin the column factor v_k = sum_l c[k][l] e^{-2 pi i j lam_l / Ny} and then sum_k v_k e^{-2 pi i i kap_k / Nx}
// (the association of kspec_body).  Stores are coalesced along e.
template <int NK>
__device__ __forceinline__ void kspec_packed_body(const PackArgs& g, int bx, int by)
{
    constexpr int TB = 8, KK = NK * NK;
    __shared__ float2 ph[TB][2][NK];                               // [bin][row/col][tap] phases on this tensor's grid
    __shared__ float taps[256 * KK];                               // the workgroup's 256 elements x Nk*Nk taps (coalesced copy)
    const PackSeg sd = g.seg[g.blk_seg[bx]];               // (uniform: a workgroup's elements belong to ONE tensor)
    const int l0 = g.blk_start[bx];                        // first element of the block inside the tensor
    const int nel = min(256, sd.n - l0);
    const int t0 = by * TB;
    const bool wk = threadIdx.x < 256;                               // (hosted by kernels with larger workgroups: the extra threads only meet the barrier)
    for (int i = threadIdx.x; i < TB * 2 * NK; i += blockDim.x) {
        const int k = i % NK, rc = (i / NK) & 1, b = i / (2 * NK);
        const int t = min(t0 + b, (int)g.Pc - 1);
        const long s = map_up(t, g.NxC, g.NyC, g.Nx[sd.lev], g.Ny[sd.lev]);
        const int nyr = g.Ny[sd.lev] / 2 + 1;
        const int bi = (int)((unsigned)s / (unsigned)nyr), bj = (int)((unsigned)s - (unsigned)bi * nyr);
        ph[b][rc][k] = rc == 0 ? phase_tw(g.tw, bi, k - NK / 2, g.Nx[sd.lev]) : phase_tw(g.tw, bj, k - NK / 2, g.Ny[sd.lev]);
    }
    if (wk) {
        const float* src = sd.k + (long)l0 * KK;                   // nel * KK consecutive floats
        const int nf = nel * KK;
        float v[KK];                                               // every load of the copy in flight at once: one round trip
#pragma unroll
        for (int w = 0; w < KK; ++w) v[w] = src[min(w * 256 + (int)threadIdx.x, nf - 1)];
#pragma unroll
        for (int w = 0; w < KK; ++w) { const int f = w * 256 + threadIdx.x; if (f < nf) taps[f] = v[w]; }
    }
    __syncthreads();
    if (!wk || (int)threadIdx.x >= nel) return;
    const int e = sd.off + l0 + threadIdx.x;
    float c[NK * NK];
#pragma unroll
    for (int i = 0; i < NK * NK; ++i) c[i] = taps[threadIdx.x * KK + i];
    for (int b = 0; b < TB && t0 + b < g.Pc; ++b) {
        float2 cp[NK], rp[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) { rp[k] = ph[b][0][k]; cp[k] = ph[b][1][k]; }
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            float2 v = make_float2(0.f, 0.f);
#pragma unroll
            for (int l = 0; l < NK; ++l) { v.x += c[k * NK + l] * cp[l].x; v.y += c[k * NK + l] * cp[l].y; }
            acc.x += v.x * rp[k].x - v.y * rp[k].y;
            acc.y += v.x * rp[k].y + v.y * rp[k].x;
        }
        g.Wp[(long)(t0 + b) * g.E + e] = acc;
    }
}


}  // namespace aefft
