// Device-side pieces of the operator form that more than one kernel file hosts (opform_kernels.hip, pruned_kernels.hip).
#pragma once
#include "internal.h"
#include "update_device.h"

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

// the two halves of map_up: row i / column j of the small grid -> row / column of the big grid
__device__ __forceinline__ int map_up_row(int i, int Nx, int NxB) { return i < Nx / 2 ? i : (i == Nx / 2 ? NxB / 2 : i + NxB - Nx); }
__device__ __forceinline__ int map_up_col(int j, int Ny, int NyB) { return j < Ny / 2 ? j : NyB / 2; }

// complex helpers on the native 2-vector (v_pk_* instructions)
__device__ __forceinline__ float2 pk_scale(float s, float2 a) { return make_float2(s, s) * a; }
__device__ __forceinline__ float2 pk_cmul(float2 a, float2 b)
{
    const float2 ax = make_float2(a.x, a.x), ay = make_float2(a.y, a.y), bs = make_float2(-b.y, b.x);
    return ax * b + ay * bs;
}

// Wp[t][e]: for support bin t the elements of C_0 .. C_{L-1}, F_{L-1} .. F_0 (chain order) at the bins of their grids that t
// maps to.  Workgroup = 256 consecutive elements x ONE column j of the support x PACK_RG of its rows; thread = one element: its
// Nk*Nl taps stay in registers, the column factor v_k = sum_l c[k][l] e^{-2 pi i j lam_l / Ny} is formed once (the association of
// kspec_body) and every row then costs Nk complex multiplies: sum_k v_k e^{-2 pi i i kap_k / Nx}.  Stores are coalesced along e.
constexpr int PACK_RG = 16;                     // (8: 25.2 us, 16: 22.9, 32: 29.5 for the merged launch at cfg3, with the 1024-workgroup target below)
static inline int pack_yblocks(const PackArgs& g) { return (g.NyC / 2 + 1) * ((g.NxC + PACK_RG - 1) / PACK_RG); }
// (LDS: `smem`, kspec_packed_lds(NK) bytes of the hosting kernel's dynamic region -- a static allocation would ADD to the dynamic size the
// host launch asks for on behalf of its other workgroups: 52 KB instead of 26 KB per workgroup at cfg3)
constexpr size_t kspec_packed_lds(int NK) { return sizeof(float2) * (size_t)(PACK_RG + 1) * (NK / 2 > 0 ? NK / 2 : 1) + sizeof(float) * 256 * (size_t)(NK * NK); }
template <int NK>
__device__ __forceinline__ void kspec_packed_body(const PackArgs& g, int bx, int by, void* smem)
{
    constexpr int RG = PACK_RG, KK = NK * NK, H = NK / 2;
    static_assert(NK % 2 == 1, "symmetric tap offsets");
    float2 (*ph)[H > 0 ? H : 1] = reinterpret_cast<float2 (*)[H > 0 ? H : 1]>(smem);      // [RG + 1][H]: row phases of the RG rows, then the column's: offsets
                                                                   // 1 .. H (the phase of a negative offset is the conjugate: kspec_body)
    float* taps = reinterpret_cast<float*>(ph + RG + 1);           // [256 * KK]: the workgroup's 256 elements x Nk*Nk taps (coalesced copy)
    const PackSeg sd = g.seg[g.blk_seg[bx]];               // (uniform: a workgroup's elements belong to ONE tensor)
    const int l0 = g.blk_start[bx];                        // first element of the block inside the tensor
    const int nel = min(256, sd.n - l0);
    const int nyrc = g.NyC / 2 + 1;
    const int j = by % nyrc, i0 = (by / nyrc) * RG;
    const bool wk = threadIdx.x < 256;                               // (hosted by kernels with larger workgroups: the extra threads only meet the barrier)
    if (wk) {
        const float* src = sd.k + (long)l0 * KK;                   // nel * KK consecutive floats
        const int nf = nel * KK;
        float v[KK];                                               // every load of the copy in flight at once: one round trip
#pragma unroll
        for (int w = 0; w < KK; ++w) v[w] = src[min(w * 256 + (int)threadIdx.x, nf - 1)];
        if (g.upd) {                                               // (uniform) the taps as the pending update will leave them (TapUpd)
            const float* gs = sd.g + (long)l0 * KK;
            const float* ds = sd.D + (long)l0 * KK;
            float gg[KK], dd[KK];
#pragma unroll
            for (int w = 0; w < KK; ++w) { const int f = min(w * 256 + (int)threadIdx.x, nf - 1); gg[w] = gs[f]; dd[w] = ds[f]; }
#pragma unroll
            for (int w = 0; w < KK; ++w) v[w] += -clip_step(gg[w] * g.upd_gscale, dd[w], g.upd_del, g.upd_alpha);
        }
        float2 pv = make_float2(0.f, 0.f);                         // (the phase gathers ride in the same round trip)
        if (threadIdx.x < (RG + 1) * H) {
            const int k = threadIdx.x % H, r = threadIdx.x / H;
            const int NxB = g.Nx[sd.lev], NyB = g.Ny[sd.lev];
            pv = r < RG ? phase_tw(g.tw, map_up_row(min(i0 + r, g.NxC - 1), g.NxC, NxB), k + 1, NxB)
                        : phase_tw(g.tw, map_up_col(j, g.NyC, NyB), k + 1, NyB);
        }
#pragma unroll
        for (int w = 0; w < KK; ++w) { const int f = w * 256 + threadIdx.x; if (f < nf) taps[f] = v[w]; }
        if (threadIdx.x < (RG + 1) * H) ph[threadIdx.x / H][threadIdx.x % H] = pv;
    }
    __syncthreads();
    if (!wk || (int)threadIdx.x >= nel) return;
    const int e = sd.off + l0 + threadIdx.x;
    float2 v0, sv[H > 0 ? H : 1], dv[H > 0 ? H : 1];
    {
        const float* c = taps + threadIdx.x * KK;
        float2 v[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            v[k] = make_float2(c[k * NK + H], 0.f);
#pragma unroll
            for (int l = 0; l < H; ++l) {
                const float2 cp = ph[RG][l];
                const float cpl = c[k * NK + H + 1 + l], cmi = c[k * NK + H - 1 - l];
                v[k].x = fmaf(cpl + cmi, cp.x, v[k].x);
                v[k].y = fmaf(cpl - cmi, cp.y, v[k].y);
            }
        }
        v0 = v[H];
#pragma unroll
        for (int k = 0; k < H; ++k) { sv[k] = v[H + 1 + k] + v[H - 1 - k]; dv[k] = v[H + 1 + k] - v[H - 1 - k]; }
    }
    float2* dst = g.Wp + ((long)i0 * nyrc + j) * g.E + e;
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        if (i0 + r >= g.NxC) break;
        float2 acc = v0;
#pragma unroll
        for (int k = 0; k < H; ++k) {
            const float2 rp = ph[r][k];
            acc.x = fmaf(rp.x, sv[k].x, acc.x); acc.x = fmaf(-rp.y, dv[k].y, acc.x);
            acc.y = fmaf(rp.y, dv[k].x, acc.y); acc.y = fmaf(rp.x, sv[k].y, acc.y);
        }
        dst[(long)r * nyrc * g.E] = acc;
    }
}


}  // namespace aefft
