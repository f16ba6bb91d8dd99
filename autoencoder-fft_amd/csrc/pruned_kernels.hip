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
#include "device_util.h"
#include "opform_device.h"
#include "update_device.h"
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
// The transform half of kspec_body: thread <-> column j of ONE plane whose NK*NL taps lie at `c` (LDS); rows i0 .. i0+nrows of the
// plane at `dst` ([Nx][Nyr], j added by the caller).  (NxB, NyB): the grid the phases are taken on -- row i / column j of the
// [Nx][Ny/2+1] output are the images map_up_row / map_up_col of pool_fft's crop in it (the spectrum sampled where the NEXT pair's
// grid lands: the chain's planar tiles read nothing else); NxB == Nx: the plane's own grid.
// the column phases of thread column j (offsets 1 .. NL/2): gathers from the global twiddle table -- a memory round trip, so the callers ask for
// them BEFORE they stage the taps (the G' workgroups had it behind their tap products: 1.5-2 us of every such workgroup)
template <int NL>
__device__ __forceinline__ void kspec_col_phases(const float2* __restrict__ tw, int j, int Ny, int NyB, float2 (&cp)[NL / 2 > 0 ? NL / 2 : 1])
{
    const int jB = map_up_col(j, Ny, NyB);
#pragma unroll
    for (int l = 0; l < NL / 2; ++l) cp[l] = phase_tw(tw, jB, l + 1, NyB);
}
template <int NK, int NL>
__device__ __forceinline__ void kspec_rows(const float* __restrict__ c, float2* __restrict__ dst, const float2 (&cp)[NL / 2 > 0 ? NL / 2 : 1], const float2* __restrict__ rowph,
                                           int Nyr, int nrows)
{
    constexpr int H = NK / 2, HL = NL / 2;
    float2 v0, sv[H > 0 ? H : 1], dv[H > 0 ? H : 1];
    {
        float2 v[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            v[k] = make_float2(c[k * NL + HL], 0.f);
#pragma unroll
            for (int l = 0; l < HL; ++l) {
                const float cpl = c[k * NL + HL + 1 + l], cmi = c[k * NL + HL - 1 - l];
                v[k].x = fmaf(cpl + cmi, cp[l].x, v[k].x);
                v[k].y = fmaf(cpl - cmi, cp[l].y, v[k].y);
            }
        }
        v0 = v[H];
#pragma unroll
        for (int k = 0; k < H; ++k) { sv[k] = v[H + 1 + k] + v[H - 1 - k]; dv[k] = v[H + 1 + k] - v[H - 1 - k]; }
    }
    // (4 rows per trip: the phase reads of the next rows are in flight while a row's FMAs issue -- a rolled loop is one LDS round trip per row)
#pragma unroll 4
    for (int i = 0; i < nrows; ++i) {
        float2 acc = v0;
#pragma unroll
        for (int k = 0; k < H; ++k) {
            const float2 rp = rowph[i * H + k];
            acc.x = fmaf(rp.x, sv[k].x, acc.x); acc.x = fmaf(-rp.y, dv[k].y, acc.x);
            acc.y = fmaf(rp.y, dv[k].x, acc.y); acc.y = fmaf(rp.x, sv[k].y, acc.y);
        }
        dst[(long)i * Nyr] = acc;
    }
}

// G' = F'.C' / (dM dD) of one pair (the collapsed operator the post-update MSE needs, fft_backproplib.cu:1460-1463) is the spectrum
// of the (2NK-1)^2-tap kernel  gsp[d'][d] = scale * sum_m f'[d'][m] (*) c'[m][d]  (weight_kernels.hip).  Here a workgroup that
// transforms the planes (d', d0 .. d0+np) forms their taps itself, from the pair's taps READ THROUGH the pending update (TapUpd):
// f'[d'][.] and c'[.][d0..] staged in LDS, thread <-> (plane, output row tx, slice of m) owns the T outputs of that row (per (m, k)
// one row of f' and one of c' feed NK*NK FMAs in registers), slices summed in slice order.  taps_s [np][T*T].
template <int NK, int MU = 2>
__device__ __forceinline__ void gtaps_stage(const GtapSrc& gs, const TapUpd& upd, int dp, int d0, int np, float* __restrict__ taps_s, float* __restrict__ work)
{
    constexpr int T = 2 * NK - 1, KK = NK * NK;
    const int NT = blockDim.x, tid = threadIdx.x;
    const int dM = gs.dM, dD = gs.dD;
    // cs: plane stride padded to an odd number of floats (planes dM*KK apart share a bank: the tiles' rows would collide np-fold)
    const int CS = dM * KK + ((dM * KK) & 1 ? 0 : 1);
    float* fs = work;                                    // [dM][KK]
    float* cs = fs + dM * KK;                            // [np][CS]: c'[m][d0+i] at i*CS + m*KK
    float* part = work;                                  // [slices][np*T][T]: takes the place of fs | cs once every slice has its sums in registers
    const unsigned nk = (unsigned)(dM * dD * KK);        // f follows c (c|f, dck|dfk, Dc|Df: each pair contiguous); element offsets fit 32 bits
    const int nf = dM * KK, nc = np * dM * KK;
#ifndef AEFFT_X_GT_U
#define AEFFT_X_GT_U 16
#endif
    constexpr int U = AEFFT_X_GT_U;                      // loads per array and batch (registers: the launch's other workgroups pay for every one)
    for (int t0 = 0; t0 < nf + nc; t0 += U * NT) {
        float w[U], gq[U], dq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t0 + u * NT >= nf + nc) break;                       // uniform: whole load instructions are skipped
            const int t = min(t0 + u * NT + tid, nf + nc - 1);
            unsigned idx;
            if (t < nf) idx = nk + (unsigned)(dp * dM * KK + t);
            else {
                const int t2 = t - nf;
                const int m = t2 / (np * KK), r = t2 - m * (np * KK);
                idx = (unsigned)((m * dD + d0) * KK + r);
            }
            w[u] = gs.c[idx];
            if (upd.g) { gq[u] = upd.g[idx]; dq[u] = upd.D[idx]; } else { gq[u] = 0.f; dq[u] = 0.f; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u * NT + tid;
            if (t >= nf + nc) continue;
            const float v = upd.g ? w[u] - clip_step(gq[u] * upd.gscale, dq[u], upd.del, upd.alpha) : w[u];
            if (t < nf) fs[t] = v;
            else {
                const int t2 = t - nf;
                const int m = t2 / (np * KK), r = t2 - m * (np * KK);
                const int i = r / KK, rr = r - i * KK;
                cs[i * CS + m * KK + rr] = v;
            }
        }
    }
    __syncthreads();
    AEFFT_WGSTAMP(3, 0);
    const int items = np * T;
    int nsl = NT / items;
    if (nsl > dM) nsl = dM;
    if (nsl < 1) nsl = 1;
    // (items * nsl <= NT: one (plane, row, slice) per thread)
    const int t = tid;
    const bool work_thr = t < items * nsl;
    const int sl = t / items, it = t - sl * items;
    float acc[T];
#pragma unroll
    for (int y = 0; y < T; ++y) acc[y] = 0.f;
    if (work_thr) {
        const int i = it / T, tx = it - i * T;
        // (MU: the m loop's unroll factor.  Unrolled by 2 its LDS reads are hoisted into 153 registers for the spectra launch -- 3 workgroups per CU;
        // rolled 128 -- 4, which the launches with G' problems want (cfg3-P2 19.1 -> 18.5 us) and the planar-spectra launch of cfg3-P1 does NOT
        // (5.7 GB of plane writes: 1 375 us at 153 registers, 1 460-1 750 at 128): the launcher picks the instantiation, run_kspec_group)
#pragma unroll MU
        for (int m = sl; m < dM; m += nsl) {
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int k2 = tx - k;
                const bool ok = k2 >= 0 && k2 < NK;
                const float* fr = fs + m * KK + k * NK;
                const float* cr = cs + i * CS + m * KK + (ok ? k2 : 0) * NK;
                float f5[NK], c5[NK];
#pragma unroll
                for (int l = 0; l < NK; ++l) { f5[l] = ok ? fr[l] : 0.f; c5[l] = cr[l]; }
#pragma unroll
                for (int l = 0; l < NK; ++l)
#pragma unroll
                    for (int l2 = 0; l2 < NK; ++l2) acc[l + l2] = fmaf(f5[l], c5[l2], acc[l + l2]);
            }
        }
    }
    __syncthreads();                                     // every read of fs | cs is done
    AEFFT_WGSTAMP(3, 1);
    if (work_thr) {
#pragma unroll
        for (int y = 0; y < T; ++y) part[(sl * items + it) * T + y] = acc[y];
    }
    __syncthreads();
    AEFFT_WGSTAMP(3, 2);
    for (int t2 = tid; t2 < items * T; t2 += NT) {
        float a = part[t2];
        for (int s2 = 1; s2 < nsl; ++s2) a += part[s2 * items * T + t2];
        taps_s[t2] = a * gs.scale;                       // (t2 = (plane*T + tx)*T + ty: the [np][T*T] layout)
    }
}

template <int NK, int NL>
__device__ __forceinline__ void kspec_body(const float* __restrict__ kern, float2* __restrict__ K,
                                                    const float2* __restrict__ tw, long planes, int Nx, int Ny,
                                                    int rows_per_chunk, int ppb, int bx, int by, float2* lds, const TapUpd& upd = TapUpd{},
                                                    int NxB = 0, int NyB = 0)
{
    if (NxB == 0) { NxB = Nx; NyB = Ny; }
    // The tap offsets are symmetric (kap = -H .. H, lam = -HL .. HL) and the phase of -kap is the conjugate of kap's, so a pair of
    // taps shares its phase:  c+ e + c- conj(e) = (c+ + c-) ex + i (c+ - c-) ey  (column factor, real taps: 2 FMAs per pair),
    // v+ p + v- conj(p) = (px sx - py dy) + i (py dx + px sy)  with s = v+ + v-, d = v+ - v-  (row sum: 4 FMAs per pair instead of 8,
    // one phase read instead of two).  s and d are formed once per thread: the column factor does not depend on the row.
    constexpr int H = NK / 2;
    static_assert(NK % 2 == 1 && NL % 2 == 1, "symmetric tap offsets");
    const int Nyr = Ny / 2 + 1;
    float2* rowph = lds;                                 // [rows_per_chunk][H]: offsets 1 .. H
    float* taps_s = reinterpret_cast<float*>(rowph + rows_per_chunk * H);      // [ppb][NK*NL]: the planes' taps, staged once per workgroup
    const int i0 = by * rows_per_chunk;
    const int nrows = min(rows_per_chunk, Nx - i0);
    const int pl = threadIdx.x / Nyr, j = threadIdx.x - pl * Nyr;
    float2 cp[NL / 2 > 0 ? NL / 2 : 1];
    kspec_col_phases<NL>(tw, j, Ny, NyB, cp);
    const bool one_trip = nrows * H <= (int)blockDim.x;          // (the row phases: requested here, stored behind the taps' loads -- one round trip for both)
    float2 rp0 = make_float2(0.f, 0.f);
    if (one_trip) { if ((int)threadIdx.x < nrows * H) rp0 = phase_tw(tw, map_up_row(i0 + threadIdx.x / H, Nx, NxB), threadIdx.x % H + 1, NxB); }
    else for (int t = threadIdx.x; t < nrows * H; t += blockDim.x) rowph[t] = phase_tw(tw, map_up_row(i0 + t / H, Nx, NxB), t % H + 1, NxB);
    {
        // (through the pending update when there is one -- uniform --: w - clip_step(g, D), TapUpd)
        const long e0 = (long)bx * ppb * (NK * NL);
        const long ne = min((long)ppb, planes - (long)bx * ppb) * (NK * NL);
        for (int t0 = 0; t0 < ne; t0 += 3 * (int)blockDim.x) {          // (batches of independent loads: one round trip, not one per element)
            float w[3], gq[3], dq[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const long t = min((long)t0 + u * blockDim.x + threadIdx.x, ne - 1);
                w[u] = kern[e0 + t];
                if (upd.g) { gq[u] = upd.g[e0 + t]; dq[u] = upd.D[e0 + t]; }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const long t = (long)t0 + u * blockDim.x + threadIdx.x;
                if (upd.g) w[u] += -clip_step(gq[u] * upd.gscale, dq[u], upd.del, upd.alpha);
                if (t < ne) taps_s[t] = w[u];
            }
        }
    }
    if (one_trip && (int)threadIdx.x < nrows * H) rowph[threadIdx.x] = rp0;
    __syncthreads();
    AEFFT_WGSTAMP(3, 0);
    const long plane = (long)bx * ppb + pl;
    if (pl >= ppb || plane >= planes) return;
    kspec_rows<NK, NL>(taps_s + pl * (NK * NL), K + (plane * Nx + i0) * (long)Nyr + j, cp, rowph, Nyr, nrows);
}

// the G' form of kspec_body: workgroup = (d', tile of ppb d's) x row chunk; the taps come from gtaps_stage
template <int NK, int MU>
__device__ __forceinline__ void gspec_gbody(const GtapSrc& gs, float2* __restrict__ G, const float2* __restrict__ tw, int Nx, int Ny,
                                            int rows_per_chunk, int ppb, int bx, int by, float2* lds, const TapUpd& upd)
{
    constexpr int T = 2 * NK - 1, TT = T * T, H = T / 2;
    const int Nyr = Ny / 2 + 1;
    float2* rowph = lds;                                                     // [rows_per_chunk][H]
    float* taps_s = reinterpret_cast<float*>(rowph + rows_per_chunk * H);    // [ppb][TT]
    float* work = taps_s + ppb * TT;
    const int i0 = by * rows_per_chunk;
    const int nrows = min(rows_per_chunk, Nx - i0);
    const int pl = threadIdx.x / Nyr, j = threadIdx.x - pl * Nyr;
    float2 cp[H];
    kspec_col_phases<T>(tw, j, Ny, Ny, cp);
    // the row phases: requested here, stored BEHIND the tap stage's loads (a load -> LDS store in front of them is a round trip of its own:
    // the first barrier of these workgroups came at 5.8-7 us against 3.8 for the plain transform's)
    const bool one_trip = nrows * H <= (int)blockDim.x;
    float2 rp0 = make_float2(0.f, 0.f);
    if (one_trip) { if ((int)threadIdx.x < nrows * H) rp0 = phase_tw(tw, i0 + threadIdx.x / H, threadIdx.x % H + 1, Nx); }
    else for (int t = threadIdx.x; t < nrows * H; t += blockDim.x) rowph[t] = phase_tw(tw, i0 + t / H, t % H + 1, Nx);
    const int tiles = (gs.dD + ppb - 1) / ppb;
    const int dp = bx / tiles, d0 = (bx - dp * tiles) * ppb;
    const int np = min(ppb, gs.dD - d0);
    gtaps_stage<NK, MU>(gs, upd, dp, d0, np, taps_s, work);
    if (one_trip && (int)threadIdx.x < nrows * H) rowph[threadIdx.x] = rp0;
    __syncthreads();
    AEFFT_WGSTAMP(3, 3);
    if (pl >= np) return;
    const long plane = (long)dp * gs.dD + d0 + pl;
    kspec_rows<T, T>(taps_s + pl * TT, G + (plane * Nx + i0) * (long)Nyr + j, cp, rowph, Nyr, nrows);
}

template <int NK, int NL>
__global__ __launch_bounds__(320) void kspec_kernel(const float* __restrict__ kern, float2* __restrict__ K,
                                                    const float2* __restrict__ tw, long planes, int Nx, int Ny,
                                                    int rows_per_chunk, int ppb)
{
    extern __shared__ float2 lds[];
    kspec_body<NK, NL>(kern, K, tw, planes, Nx, Ny, rows_per_chunk, ppb, blockIdx.x, blockIdx.y, lds);
}

// all pairs' kernel spectra in one launch: problem p owns workgroups [start[p], start[p+1]), plane groups fastest
#ifndef AEFFT_X_KSPEC_W
#define AEFFT_X_KSPEC_W 1
#endif
template <int NK, int NL, int MU>
__global__ __launch_bounds__(320, AEFFT_X_KSPEC_W) void kspec_group_kernel(const PrunedGroup g, const float2* __restrict__ tw, const PackArgs pk, const BiasUpdGroup bu, const int nbias_start)
{
    AEFFT_WGTIME(3);
    extern __shared__ float2 lds[];
    if ((int)blockIdx.x >= nbias_start) {
        // last of all: the bias half of a fused update (nothing in this launch reads b or p), one workgroup per pair
        const BiasUpd& a = bu.a[blockIdx.x - nbias_start];
        if (threadIdx.x == 0 && a.zero) *a.zero = 0.f;
        for (int i = threadIdx.x; i < a.dM; i += blockDim.x) { const float Db = clip_step(a.db[i] * bu.gscale, a.Db[i], bu.del, bu.alpha); a.b[i] += -Db; a.Db[i] = Db; }
        for (int i = threadIdx.x; i < a.dD; i += blockDim.x) { const float Dp = clip_step(a.dp[i] * bu.gscale, a.Dp[i], bu.del, bu.alpha); a.p[i] += -Dp; a.Dp[i] = Dp; }
        return;
    }
    if ((int)blockIdx.x >= g.start[g.n]) {
        // trailing workgroups: the bin-major copy of the spectra for the operator chain (opform_device.h), from the same taps
        if constexpr (NK == NL && (NK == 3 || NK == 5)) {
            const int lin = blockIdx.x - g.start[g.n];
            kspec_packed_body<NK>(pk, lin % pk.nblk, lin / pk.nblk, lds);
        }
        return;
    }
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
    const int lin = blockIdx.x - g.start[p];
    const PrunedProb& q = g.q[p];
    if constexpr (NK == NL && (NK == 3 || NK == 5)) {
        if (g.gsrc[p].f) {                                // (uniform) a G' problem: the (2NK-1)^2 taps are formed in the workgroup
            gspec_gbody<NK, MU>(g.gsrc[p], static_cast<float2*>(q.dst), tw, q.Nx, q.Ny, g.rows[p], g.ppb[p], lin % g.pblocks[p], lin / g.pblocks[p], lds, g.upd[p]);
            return;
        }
    }
    kspec_body<NK, NL>(static_cast<const float*>(q.src), static_cast<float2*>(q.dst), tw, q.planes, q.Nx, q.Ny, g.rows[p], g.ppb[p],
                       lin % g.pblocks[p], lin / g.pblocks[p], lds, g.upd[p], q.NxB, q.NyB);
}

// Thread <-> (plane, column j) over a chunk of RB rows:
//   t_k[j]   = sum_i D[i][j] * e^{+2 pi i i*kap_k/Nx}                  (row phases: LDS broadcast reads; D read once, coalesced)
//   g[k][l] += w_j * Re( t_k[j] * e^{+2 pi i j*lam_l/Ny} )              (summed over the plane's columns through LDS, fixed order)
// Output: part[plane][chunk][NK*NL] (summed over chunks by ksum_kernel when chunks > 1).  Deterministic.
template <int NK, int NL>
__device__ __forceinline__ void kgrad_body(const float2* __restrict__ D, float* __restrict__ part,
                                                    const float2* __restrict__ tw, long planes, int Nx, int Ny, int RB, int ppb, float scale,
                                                    int bx, int chunk, int nchunks, float2* lds)
{
    const int Nyr = Ny / 2 + 1;
    const int nthr = blockDim.x;
    float2* rowph = lds;                                            // [RB][NK]
    float* contrib = reinterpret_cast<float*>(rowph + RB * NK);     // [NK*NL][nthr]
    const int i0 = chunk * RB;
    for (int t = threadIdx.x; t < RB * NK; t += nthr) rowph[t] = phase(tw, i0 + t / NK, t % NK - NK / 2, Nx, -1.f);
    __syncthreads();
    const int pl = threadIdx.x / Nyr, j = threadIdx.x - pl * Nyr;
    const long plane = (long)bx * ppb + pl;
    const bool active = pl < ppb && plane < planes;
    float2 t[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) t[k] = make_float2(0.f, 0.f);
    if (active) {
        const float2* src = D + (plane * Nx + i0) * (long)Nyr + j;
        // batches of 8 rows: all 8 loads are issued before the first use (hipcc does not pipeline loads across
        // loop iterations by itself, and this loop is otherwise one memory round trip per row)
        int i = 0;
        for (; i + 16 <= RB; i += 16) {
            float2 d[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) d[u] = src[(long)(i + u) * Nyr];
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const float2 rp = rowph[(i + u) * NK + k];
                    t[k].x += d[u].x * rp.x - d[u].y * rp.y;
                    t[k].y += d[u].x * rp.y + d[u].y * rp.x;
                }
        }
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
        const long pln = (long)bx * ppb + p2;
        if (pln >= planes) continue;
        float g = 0.f;
        for (int jj = 0; jj < Nyr; ++jj) g += contrib[kl * nthr + p2 * Nyr + jj];
        part[(pln * nchunks + chunk) * (NK * NL) + kl] = g * scale;
    }
}

template <int NK, int NL>
__global__ __launch_bounds__(320) void kgrad_kernel(const float2* __restrict__ D, float* __restrict__ part,
                                                    const float2* __restrict__ tw, long planes, int Nx, int Ny, int RB, int ppb, float scale)
{
    extern __shared__ float2 lds[];
    kgrad_body<NK, NL>(D, part, tw, planes, Nx, Ny, RB, ppb, scale, blockIdx.x, blockIdx.y, gridDim.y, lds);
}

// Grouped form: all pairs' pruned inverse transforms in ONE launch.  A workgroup of NT threads owns ppb planes x one chunk of
// CR = S*RB rows; thread <-> (row slice s, plane, column j) walks RB rows (one batch of 16 loads when the problem is chunked),
// the slices are combined in LDS in a fixed order.  Row chunks leave their partial sums side by side ([plane][chunk][tap]) and
// the NEXT kernel adds them in chunk order while staging (an in-launch combine by the last workgroup to arrive was measured at
// 3.6x the whole kernel: its agent-scope release writes back the XCD's L2, which is full of freshly written S).
// The global loads of the first batch are issued BEFORE the phase tables are built: the table gathers ride in their shadow.
template <int NK, int NL, int NT>
__device__ __forceinline__ void kgrad_sliced_body(const float2* __restrict__ D, float* __restrict__ g, const float2* __restrict__ tw,
                                                  long planes, int Nx, int Ny, int ppb, int S, int RB, int nchunks, float scale,
                                                  int bx, int chunk, float2* lds)
{
    constexpr int TS = NT + 1;                                      // row stride of the t arrays (+1: the NK rows land in different banks)
    constexpr int TT = NK * NL, H = NK / 2;
    static_assert(NK % 2 == 1, "symmetric offsets");
    const int Nyr = Ny / 2 + 1;
    const int CR = S * RB;                                          // rows of this chunk (the last one may hold fewer)
    const int r0 = chunk * CR, crow = min(CR, Nx - r0);
    float2* rowph = lds;                                            // [CR][H]    e^{+2 pi i i kap / Nx}, kap = 1 .. H
    float2* colph = rowph + CR * H;                                 // [Nyr][NL]  e^{+2 pi i j lam_l / Ny}
    float* tre = reinterpret_cast<float*>(colph + Nyr * NL);        // [NK][TS]   per-thread row sums t_k (real part)
    float* tim = tre + NK * TS;                                     // [NK][TS]
    const int per = ppb * Nyr;
    const int s = threadIdx.x / per, rem = threadIdx.x - s * per;
    const int pl = rem / Nyr, j = rem - pl * Nyr;
    const long plane = (long)bx * ppb + pl;
    const bool active = s < S && plane < planes;
    const int i0 = s * RB, nrows = active ? max(0, min(RB, crow - i0)) : 0;
    const float2* src = D + (plane * Nx + r0 + i0) * (long)Nyr + j;
    float2 d[16];
    if (nrows > 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) d[u] = ld_stream(&src[(long)min(u, nrows - 1) * Nyr]);      // (S is read once)
    }
    // phase tables: batches of independent gathers (a rolled load -> store loop is one L2 round trip per entry)
    // (row phases of the H = NK/2 positive offsets only: the phase of -kap is the conjugate)
    for (int t0 = 0; t0 < CR * H; t0 += NT * 4) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int t = min(t0 + u * NT + (int)threadIdx.x, CR * H - 1); v[u] = phase(tw, (r0 + t / H) & (Nx - 1), t % H + 1, Nx, -1.f); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int t = t0 + u * NT + threadIdx.x; if (t < CR * H) rowph[t] = v[u]; }
    }
    // (column phases of the offsets 0 .. HLc only, stored at [j][HLc + lam]: the column stage reads nothing else)
    constexpr int HLc = NL / 2, NLc = HLc + 1;
    for (int t0 = 0; t0 < Nyr * NLc; t0 += NT * 3) {
        float2 v[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) { const int t = min(t0 + u * NT + (int)threadIdx.x, Nyr * NLc - 1); v[u] = phase(tw, t / NLc, t % NLc, Ny, -1.f); }
#pragma unroll
        for (int u = 0; u < 3; ++u) { const int t = t0 + u * NT + threadIdx.x; if (t < Nyr * NLc) colph[(t / NLc) * NL + HLc + t % NLc] = v[u]; }
    }
    __syncthreads();
    AEFFT_WGSTAMP(1, 0);
    // t_k = sum_i d_i e^{+i th_k(i)} for the offsets kap = k - H.  The offsets come in conjugate pairs: d*p and d*conj(p) share their
    // four products, so per pair the sums A = sum dx px, B = sum dy py, C = sum dx py, E = sum dy px are accumulated (4 FMAs per
    // element instead of 8) and t_{+kap} = (A - B, C + E), t_{-kap} = (A + B, E - C) at the end; kap = 0 is the plain sum.
    float2 t[NK];
    float2 s0 = make_float2(0.f, 0.f);
    float pa[H > 0 ? H : 1], pb[H > 0 ? H : 1], pc[H > 0 ? H : 1], pe[H > 0 ? H : 1];
#pragma unroll
    for (int k = 0; k < H; ++k) pa[k] = pb[k] = pc[k] = pe[k] = 0.f;
    if (nrows > 0) {
        const float2* rp0 = rowph + i0 * H;
        for (int i = 0; i < nrows; i += 16) {
            if (i > 0) {
#pragma unroll
                for (int u = 0; u < 16; ++u) d[u] = ld_stream(&src[(long)min(i + u, nrows - 1) * Nyr]);
            }
            if (i + 16 > nrows) {                         // tail batch: rows past the slice were read clamped, they count zero
#pragma unroll
                for (int u = 0; u < 16; ++u) if (i + u >= nrows) d[u] = make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float2* rp = rp0 + min(i + u, nrows - 1) * H;
                s0.x += d[u].x; s0.y += d[u].y;
#pragma unroll
                for (int k = 0; k < H; ++k) {
                    const float2 r = rp[k];
                    pa[k] = fmaf(d[u].x, r.x, pa[k]); pb[k] = fmaf(d[u].y, r.y, pb[k]);
                    pc[k] = fmaf(d[u].x, r.y, pc[k]); pe[k] = fmaf(d[u].y, r.x, pe[k]);
                }
            }
        }
    }
    t[H] = s0;
#pragma unroll
    for (int k = 0; k < H; ++k) {
        t[H + 1 + k] = make_float2(pa[k] - pb[k], pc[k] + pe[k]);
        t[H - 1 - k] = make_float2(pa[k] + pb[k], pe[k] - pc[k]);
    }
    if (active) {
#pragma unroll
        for (int k = 0; k < NK; ++k) { tre[k * TS + threadIdx.x] = t[k].x; tim[k * TS + threadIdx.x] = t[k].y; }
    }
    __syncthreads();
    AEFFT_WGSTAMP(1, 1);
    if (active && s == 0) {                               // row slices -> slice 0, in slice order
        // (every read of a slice step issued before the first add: a rolled loop over k and s2 is one LDS round trip per element -- 2.5 us of
        // this kernel's 15)
        float a[NK], b[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) { a[k] = tre[k * TS + rem]; b[k] = tim[k * TS + rem]; }
        for (int s2 = 1; s2 < S; ++s2) {
            float a2[NK], b2[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) { a2[k] = tre[k * TS + s2 * per + rem]; b2[k] = tim[k * TS + s2 * per + rem]; }
#pragma unroll
            for (int k = 0; k < NK; ++k) { a[k] += a2[k]; b[k] += b2[k]; }
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) { tre[k * TS + rem] = a[k]; tim[k * TS + rem] = b[k]; }
    }
    __syncthreads();
    AEFFT_WGSTAMP(1, 2);
    // g[k][l] = sum_j w_j Re( t_k[j] e^{+2 pi i j lam_l / Ny} ).  The column offsets come in conjugate pairs as well: a task is
    // (plane, k, |lam|) and accumulates P = sum w tre cx and Q = sum w tim cy, g[+lam] = P - Q, g[-lam] = P + Q.  The columns of a
    // plane are split over JS threads per task (a 129-column plane would otherwise be one 129-step serial LDS chain), partial sums
    // combined in slice order.
    constexpr int HL = NL / 2;
    static_assert(NL % 2 == 1 && NLc == HL + 1, "symmetric offsets");
    // A task is (plane, k, slice of the plane's columns) and accumulates P[lam] = sum w tre cx, Q[lam] = sum w tim cy for ALL |lam| at once: t_k[j]
    // is read once per column instead of once per (column, lam) -- the stage is bound by its LDS reads (one task per (k, |lam|): 2.4-5 us of the
    // kernel's 15) -- and the reads of four columns are issued before their FMAs.
    const int nout = ppb * TT, ntask = ppb * NK;
    int JS = NT / ntask;
    if (JS < 1) JS = 1;
    if (JS > 16) JS = 16;
    const int jlen = (Nyr + JS - 1) / JS;
    float* part = tim + NK * TS;                          // [ntask][JS][NLc][2], after the t arrays
    for (int it = threadIdx.x; it < ntask * JS; it += NT) {
        const int o = it / JS, js = it - o * JS;
        const int p2 = o / NK, k = o - p2 * NK;
        float pp[NLc], qq[NLc];
#pragma unroll
        for (int lam = 0; lam < NLc; ++lam) pp[lam] = qq[lam] = 0.f;
        const int j0 = js * jlen, j1 = min(Nyr, j0 + jlen);
        const float* tr0 = tre + k * TS + p2 * Nyr;
        const float* ti0 = tim + k * TS + p2 * Nyr;
        for (int jb = j0; jb < j1; jb += 4) {
            float tr[4], ti[4]; float2 cp[4][NLc];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = min(jb + u, j1 - 1);
                tr[u] = tr0[jj]; ti[u] = ti0[jj];
#pragma unroll
                for (int lam = 0; lam < NLc; ++lam) cp[u][lam] = colph[jj * NL + HL + lam];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = jb + u;
                if (jj < j1) {
                    const float wj = (jj == 0 || jj == Ny / 2) ? 1.f : 2.f;
                    const float a = wj * tr[u], b = wj * ti[u];
#pragma unroll
                    for (int lam = 0; lam < NLc; ++lam) { pp[lam] = fmaf(a, cp[u][lam].x, pp[lam]); qq[lam] = fmaf(b, cp[u][lam].y, qq[lam]); }
                }
            }
        }
#pragma unroll
        for (int lam = 0; lam < NLc; ++lam) { part[(it * NLc + lam) * 2] = pp[lam]; part[(it * NLc + lam) * 2 + 1] = qq[lam]; }
    }
    __syncthreads();
    AEFFT_WGSTAMP(1, 3);
    for (int o = threadIdx.x; o < nout; o += NT) {
        const int p2 = o / TT, kl = o - p2 * TT;
        const long pln = (long)bx * ppb + p2;
        if (pln >= planes) continue;
        const int k = kl / NL, l = kl - k * NL;
        const int lam = l >= HL ? l - HL : HL - l;
        const float sg = l >= HL ? -1.f : 1.f;
        const float* pt = part + ((p2 * NK + k) * JS * NLc + lam) * 2;
        float a = 0.f;
#pragma unroll 4
        for (int js = 0; js < JS; ++js) a += pt[js * NLc * 2] + sg * pt[js * NLc * 2 + 1];
        g[(pln * nchunks + chunk) * TT + kl] = a * scale;       // [plane][chunk][tap]: the consumer adds the chunks in order
    }
}

template <int NK, int NL, int NT>
#ifndef AEFFT_X_KGRAD_W
#define AEFFT_X_KGRAD_W 1
#endif
__global__ __launch_bounds__(NT, AEFFT_X_KGRAD_W) void kgrad_group_kernel(const PrunedGroup g, const float2* __restrict__ tw, const BiasGradGroup bg)
{
    AEFFT_WGTIME(1);
    extern __shared__ float2 lds[];
    if ((int)blockIdx.x >= g.start[g.n]) {
        // trailing workgroups: the independent DC-bin terms (db, dp, es) of every pair ride along instead of costing a launch
        const int blk = blockIdx.x - g.start[g.n];
        int p = 0;
#pragma unroll
        for (int i = 1; i < 8; ++i) if (i < bg.n && blk >= bg.start[i]) p = i;
        const BiasGradArgs& a = bg.a[p];
        bias_grad_body(a.O, a.T, a.F, a.b, a.df, a.db, a.dp, a.B, a.dM, a.dD, a.P, a.norm, a.Norm, bg.fix[p], blk - bg.start[p], lds, a.PO, a.es_out, a.es_in);
        return;
    }
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
    const PrunedProb& q = g.q[p];
    const int lin = blockIdx.x - g.start[p];
    kgrad_sliced_body<NK, NL, NT>(static_cast<const float2*>(q.src), static_cast<float*>(q.dst), tw, q.planes, q.Nx, q.Ny, g.ppb[p], g.rows[p],
                                  g.rb[p], g.chunks[p], q.scale, lin % g.pblocks[p], lin / g.pblocks[p], lds);
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

static size_t kspec_lds(int rows, int Nk, int ppb) { return sizeof(float2) * (size_t)rows * (Nk / 2) + sizeof(float) * (size_t)ppb * Nk * Nk; }   // row phases | the planes' taps
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
    kspec_kernel<NK, NL><<<dim3((unsigned)pblocks, chunks), threads, kspec_lds(rows, NK, ppb), st>>>(k, K, tw, planes, Nx, Ny, rows, ppb);
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

// the taps of G' for every plane (d', d) of the group's pairs: one workgroup per plane, gtaps_stage, stored [plane][T*T]
template <int NK>
__global__ __launch_bounds__(320) void gtaps_group_kernel(const GtapsGroup g)
{
    constexpr int T = 2 * NK - 1, TT = T * T;
    extern __shared__ float gt_lds[];
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
    const GtapSrc& gs = g.gs[p];
    const int lin = blockIdx.x - g.start[p];
    const int dp = lin / gs.dD, d0 = lin - dp * gs.dD;
    float* taps_s = gt_lds;
    float* work = gt_lds + TT;
    const TapUpd none{};
    gtaps_stage<NK>(gs, none, dp, d0, 1, taps_s, work);
    __syncthreads();
    for (int t = threadIdx.x; t < TT; t += blockDim.x) g.out[p][(long)lin * TT + t] = taps_s[t];
}

hipError_t launch_gtaps_group(GtapsGroup& g, int Nk, hipStream_t st)
{
    if (g.n < 1 || g.n > 8 || (Nk != 3 && Nk != 5)) return hipErrorInvalidValue;
    const int T = 2 * Nk - 1;
    int total = 0; size_t lds = 0;
    for (int i = 0; i < g.n; ++i) {
        g.start[i] = total; total += g.gs[i].dD * g.gs[i].dD;
        lds = std::max(lds, sizeof(float) * ((size_t)T * T + std::max((size_t)(g.gs[i].dM * Nk * Nk + 1) * 2, (size_t)320 * T)));
    }
    g.start[g.n] = total;
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        const hipError_t e = Nk == 3 ? hipFuncSetAttribute(reinterpret_cast<const void*>(gtaps_group_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                     : hipFuncSetAttribute(reinterpret_cast<const void*>(gtaps_group_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (Nk == 3) gtaps_group_kernel<3><<<dim3(total), 320, lds, st>>>(g);
    else gtaps_group_kernel<5><<<dim3(total), 320, lds, st>>>(g);
    return hipGetLastError();
}

static PackArgs g_pack_none{};
static BiasUpdGroup g_bu_none{};
template <int NK, int NL> static hipError_t run_kspec_group(PrunedGroup& g, const float2* tw, hipStream_t st, PackArgs* pk = nullptr, const BiasUpdGroup* bu = nullptr)
{
    int total = 0; size_t lds = 0;
    for (int p = 0; p < g.n; ++p) { const int Nyr = g.q[p].Ny / 2 + 1; g.ppb[p] = std::max(1, 256 / Nyr); }
#ifndef AEFFT_X_GROWS
#define AEFFT_X_GROWS 64
#endif
    int extra = 0;                                          // trailing workgroups: the bin-major copy (kspec_packed_body)
    if (pk && pk->Wp && pk->Nk == NK && NK == NL) { pack_blocks(*pk); extra = pk->nblk * pack_yblocks(*pk); }
    long wg64 = extra;                                      // workgroups of the launch with 64-row chunks everywhere (G' tiles counted at one plane each: an upper bound)
    for (int p = 0; p < g.n; ++p) wg64 += (g.gsrc[p].f ? (long)g.gsrc[p].dD * g.gsrc[p].dD : (g.q[p].planes + g.ppb[p] - 1) / g.ppb[p]) * ((g.q[p].Nx + 63) / 64);
    const int grows = wg64 < 512 ? 16 : AEFFT_X_GROWS;
    for (int p = 0; p < g.n; ++p) {
        const PrunedProb& q = g.q[p];
        // <= 64 rows per workgroup for every problem (a launch-wide chunk count gave the big grids 128-row workgroups on half their
        // lanes: 23.7 us against 21.5 us at cfg3; 32 rows: 22.0, 16 rows: 24.9)
        // (G' problems -- their workgroups stand behind a tap-product stage and are the launch's long pole -- with 32-row chunks are each 2-3 us
        // shorter, but twice as many for the launch's resident slots: 18.8-19.1 vs 18.7-18.9 us at cfg3, step +2 us.  So: 64 rows, and 16 in a launch
        // that leaves most of the chip idle anyway (cfg2: 22 such workgroups): grows)
        const int rmax = g.gsrc[p].f ? grows : 64;
        const int chunks = (q.Nx + rmax - 1) / rmax;
        g.rows[p] = (q.Nx + chunks - 1) / chunks;
        if (g.gsrc[p].f) {
            // G' problem (gspec_gbody): plane groups (d', tile of ppb d's); LDS = row phases | taps | f' | c' tile | slice partials
            if (!(NK == NL && (NK == 3 || NK == 5))) return hipErrorInvalidValue;
            constexpr int T = 2 * NK - 1;
            const GtapSrc& gs = g.gsrc[p];
            auto need = [&](int ppb) { return sizeof(float2) * (size_t)g.rows[p] * (T / 2) + sizeof(float) * ((size_t)ppb * T * T + std::max((size_t)(gs.dM * NK * NK + 1) * (1 + ppb), (size_t)320 * T)); };
            // (a launch's dynamic LDS size applies to EVERY workgroup: the tile shrinks until it fits beside what the ordinary problems need)
            // ... and so that (planes x output rows x slices of m) fill the workgroup with about two m's per slice: the tap products are
            // a dependent stage in front of the transform, more and smaller tiles shorten it (5 planes: 24.8 us for the launch at cfg3, 2: see DESIGN.md)
            int ppb = std::min(std::min(g.ppb[p], gs.dD), std::max(1, (2 * 320) / (T * gs.dM)));
            while (ppb > 1 && need(ppb) > 26 * 1024) --ppb;
            if (need(ppb) > 150 * 1024) return hipErrorInvalidValue;
            g.ppb[p] = ppb;
            g.pblocks[p] = gs.dD * ((gs.dD + ppb - 1) / ppb);
            lds = std::max(lds, need(ppb));
        } else {
            g.pblocks[p] = (int)((q.planes + g.ppb[p] - 1) / g.ppb[p]);
            lds = std::max(lds, kspec_lds(g.rows[p], NK, g.ppb[p]));
        }
        g.start[p] = total; total += g.pblocks[p] * chunks;
    }
    g.start[g.n] = total;
    {
        // HBM-sized write streams (cfg3-P1: 5.7 GB of planar spectra, 1.4 GB of G' planes) want FEW resident workgroups: every one of them is a
        // write stream of its own rows, and the fewer streams the memory controllers see the better they keep their pages -- 2 per CU (a dynamic LDS
        // size of 60 KB asks for that) 5.55 TB/s, 3-4 per CU 5.3, 6 per CU (the 9x9-tap launch at 62 registers) 4.3: 1 375 -> 1 285 us for the two launches
        double wbytes = 0;
        for (int p = 0; p < g.n; ++p) wbytes += (double)g.q[p].planes * g.q[p].Nx * (g.q[p].Ny / 2 + 1) * 8.0;
        if (wbytes > 512e6) lds = std::max(lds, (size_t)60000);
    }
    bool has_g = false;                                   // (which instantiation: see gtaps_stage)
    for (int p = 0; p < g.n; ++p) has_g = has_g || g.gsrc[p].f != nullptr;
    if (lds > 64 * 1024) {
        const hipError_t e = has_g ? hipFuncSetAttribute(reinterpret_cast<const void*>(kspec_group_kernel<NK, NL, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                   : hipFuncSetAttribute(reinterpret_cast<const void*>(kspec_group_kernel<NK, NL, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int threads = 256;                                    // one thread per (plane in group, column)
    for (int p = 0; p < g.n; ++p) threads = std::max(threads, ((g.ppb[p] * (g.q[p].Ny / 2 + 1) + 63) / 64) * 64);
    if (threads > 320) return hipErrorInvalidValue;
    if (extra) lds = std::max(lds, kspec_packed_lds(NK));
    const int nb = bu ? bu->n : 0;
    if (has_g) kspec_group_kernel<NK, NL, 1><<<dim3(total + extra + nb), threads, lds, st>>>(g, tw, extra ? *pk : g_pack_none, nb ? *bu : g_bu_none, total + extra);
    else kspec_group_kernel<NK, NL, 2><<<dim3(total + extra + nb), threads, lds, st>>>(g, tw, extra ? *pk : g_pack_none, nb ? *bu : g_bu_none, total + extra);
    return hipGetLastError();
}

// geometry of one problem of the grouped inverse transform
struct KgGeom { int ppb, S, RB, chunks, pblocks; };
static constexpr int KG_NT = 512, KG_MAXCHUNKS = 32;     // (8 row chunks unless the problem would then leave most of the chip idle: see kgrad_group_geom)
static KgGeom kgrad_group_geom(long planes, int Nx, int Ny, int max_chunks, int nt = KG_NT)
{
    KgGeom o{};
    const int Nyr = Ny / 2 + 1;
    o.ppb = std::max(1, std::min(256, nt) / Nyr);         // ~256 columns per workgroup, the remaining threads become row slices
    if (o.ppb > planes) o.ppb = (int)planes;
    o.S = std::max(1, nt / (o.ppb * Nyr));
    while (o.S > 1 && Nx / o.S < 8) --o.S;
    o.RB = (Nx + o.S - 1) / o.S;
    o.chunks = 1;
    // few planes on a large grid (the outermost pair of a 1024^2 net: 9 planes of 512 x 257 bins): 8 chunks are 72 workgroups of 64
    // serial rows each -- the launch's critical path; up to 32 chunks while the problem stays below ~1000 workgroups
    const long pb = (planes + o.ppb - 1) / o.ppb;
    const int cap = std::min(pb * 8 >= 1024 ? 8 : KG_MAXCHUNKS, max_chunks);      // (the launcher passes max_chunks <= 8 when the LAUNCH is large: run_kgrad_group)
    if (cap > 1 && o.RB > 16) {
        // one 16-row load batch per slice and chunk while the destination has room for the chunks' partial sums
        int cr = 16 * o.S;
        o.chunks = (Nx + cr - 1) / cr;
        if (o.chunks > cap) { o.chunks = cap; cr = (Nx + cap - 1) / cap; }
        o.RB = (cr + o.S - 1) / o.S;
        o.chunks = (Nx + o.S * o.RB - 1) / (o.S * o.RB);
    }
    o.pblocks = (int)((planes + o.ppb - 1) / o.ppb);
    return o;
}
// row chunks the grouped inverse transform would like for this problem: size the destination [planes][chunks][taps]
int kgrad_group_chunks(long planes, int Nx, int Ny) { return kgrad_group_geom(planes, Nx, Ny, KG_MAXCHUNKS).chunks; }

// (NT = 576 for grids of 257 columns -- cfg3-P1: one thread per column leaves 255 of 512 threads without a column, two row slices of 257 columns fill 514
// of 576 -- measured 509 vs 375 us there: no.)
template <int NK, int NL, int NT> static hipError_t run_kgrad_group(PrunedGroup& g, const float2* tw, hipStream_t st, BiasGradGroup* bgp = nullptr)
{
    int total = 0; size_t lds = 0;
    // more than 8 row chunks only where a problem's 64-row workgroups would be the stragglers of a SMALL launch (cfg5's outermost pair);
    // in a launch of thousands of equally long workgroups (cfg3-P1: every pair on the 512^2 grid) they only add partial sums: 411 -> 456 us
    long total8 = 0;
    for (int p = 0; p < g.n; ++p) { const KgGeom k8 = kgrad_group_geom(g.q[p].planes, g.q[p].Nx, g.q[p].Ny, std::min(8, std::max(1, g.chunks[p])), NT); total8 += (long)k8.pblocks * k8.chunks; }
    const int launch_cap = total8 < 2048 ? KG_MAXCHUNKS : 8;
    for (int p = 0; p < g.n; ++p) {
        const PrunedProb& q = g.q[p];
        const int Nyr = q.Ny / 2 + 1;
        if (Nyr > NT) return hipErrorInvalidValue;
        const KgGeom k = kgrad_group_geom(q.planes, q.Nx, q.Ny, std::min(launch_cap, std::max(1, g.chunks[p])), NT);     // in: room at dst; out: chunks used
        g.ppb[p] = k.ppb; g.rows[p] = k.S; g.rb[p] = k.RB; g.chunks[p] = k.chunks; g.pblocks[p] = k.pblocks;
        g.start[p] = total; total += k.pblocks * k.chunks;
        const int ntask = k.ppb * NK, js = std::min(16, std::max(1, NT / ntask));      // (the column stage's tasks and column slices, as kgrad_sliced_body forms them)
        lds = std::max(lds, sizeof(float2) * ((size_t)k.S * k.RB * NK + (size_t)Nyr * NL) + sizeof(float) * (2 * NK * (NT + 1) + 2 * (size_t)(NL / 2 + 1) * ntask * js));      // (part: [tasks x column slices][NL/2+1][2])
    }
    g.start[g.n] = total;
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kgrad_group_kernel<NK, NL, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    BiasGradGroup bg{};
    int extra = 0;
    if (bgp) {
        for (int i = 0; i < bgp->n; ++i) {
            const BiasGradArgs& a = bgp->a[i];
            bgp->fix[i] = (a.dM * a.dD + 255) / 256;
            bgp->start[i] = extra; extra += bgp->fix[i] + (a.dM + 3) / 4;
            lds = std::max(lds, bias_grad_lds(a.B, a.dD));
        }
        bgp->start[bgp->n] = extra;
        bg = *bgp;
    }
    kgrad_group_kernel<NK, NL, NT><<<dim3(total + extra), NT, lds, st>>>(g, tw, bg);
    return hipGetLastError();
}

static bool pruned_group_ok(const PrunedGroup& g, const float2* tw, int Nk, int Nl)
{
    if (g.n < 1 || g.n > 8 || !tw) return false;
    for (int p = 0; p < g.n; ++p) {
        const PrunedProb& q = g.q[p];
        if (q.planes <= 0 || !pruned_supported(Nk, Nl, q.Nx, q.Ny)) return false;
    }
    return true;
}

static bool taps_group_ok(const PrunedGroup& g, const float2* tw)
{
    if (g.n < 1 || g.n > 8 || !tw) return false;
    for (int p = 0; p < g.n; ++p) {
        const PrunedProb& q = g.q[p];
        if (q.planes <= 0 || q.Nx > TW_N || q.Ny > TW_N || (TW_N % q.Nx) || (TW_N % q.Ny) || q.Ny / 2 + 1 > 320) return false;
    }
    return true;
}
hipError_t launch_kgrad_group_taps(PrunedGroup& g, const float2* tw, int T, hipStream_t st, BiasGradGroup* bias)
{
    if (!taps_group_ok(g, tw) || (bias && (bias->n < 1 || bias->n > 8))) return hipErrorInvalidValue;
    if (T == 5) return run_kgrad_group<5, 5, KG_NT>(g, tw, st, bias);
    if (T == 9) return run_kgrad_group<9, 9, KG_NT>(g, tw, st, bias);
    return hipErrorInvalidValue;
}

hipError_t launch_kspec_group(PrunedGroup& g, const float2* tw, int Nk, int Nl, hipStream_t st, PackArgs* packed, const BiasUpdGroup* bias_upd)
{
    if (!pruned_group_ok(g, tw, Nk, Nl)) return hipErrorInvalidValue;
    if (Nk == 3) return run_kspec_group<3, 3>(g, tw, st, packed, bias_upd);
    if (Nk == 5) return run_kspec_group<5, 5>(g, tw, st, packed, bias_upd);
    return run_kspec_group<7, 7>(g, tw, st, nullptr, bias_upd);
}

hipError_t launch_kspec_group_taps(PrunedGroup& g, const float2* tw, int T, hipStream_t st)
{
    if (g.n < 1 || g.n > 8 || !tw || (T != 5 && T != 9)) return hipErrorInvalidValue;
    for (int p = 0; p < g.n; ++p) {
        const PrunedProb& q = g.q[p];
        if (q.planes <= 0 || g.gsrc[p].f || q.Nx > TW_N || q.Ny > TW_N || (TW_N % q.Nx) || (TW_N % q.Ny) || q.Ny / 2 + 1 > 320 || T > q.Nx || T > q.Ny) return hipErrorInvalidValue;
    }
    return T == 5 ? run_kspec_group<5, 5>(g, tw, st) : run_kspec_group<9, 9>(g, tw, st);
}

hipError_t launch_kgrad_group(PrunedGroup& g, const float2* tw, int Nk, int Nl, hipStream_t st)
{
    if (!pruned_group_ok(g, tw, Nk, Nl)) return hipErrorInvalidValue;
    if (Nk == 3) return run_kgrad_group<3, 3, KG_NT>(g, tw, st);
    if (Nk == 5) return run_kgrad_group<5, 5, KG_NT>(g, tw, st);
    return run_kgrad_group<7, 7, KG_NT>(g, tw, st);
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

#if defined(AEFFT_X_WGTIME) && AEFFT_X_WGTIME
extern "C" int aefft_debug_wgtime_pruned(void* p) { return aefft::wgtime_set_tu(p); }
#endif
