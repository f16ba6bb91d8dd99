// Weight-side algebra of the training step in the COORDINATE domain of the Nk x Nl kernels (gfx950).
//
// The kernel spectra are 25-term trigonometric sums, F[d][m][bin] = sum_{k,l} f[d][m][k][l] e^{-i th_kl(bin)}, so the
// reference's frequency-domain weight gradient (gradient_k_io, fft_backproplib.cu:395-475, then C2R and shrink_k,
// :1219-1226)
//     g_c[m][d][k][l] = sum_bin w Re( e^{+i th_kl} * sum_d1 conj(F[d1][m]) S[d1][d] ) / (Norm B)
// factors through Q[a][b][tau] = sum_bin w Re( S[a][b][bin] e^{+i th_tau(bin)} ), the pruned inverse transform of the
// dD x dD planes of S on the (2Nk-1) x (2Nl-1) offsets tau = kl + k'l':
//     g_c[m][d][kl] = sum_d1 sum_k'l' f[d1][m][k'l'] Q[d1][d][kl + k'l'] / (Norm B)
//     g_f[d][m][kl] = ( sum_d1 sum_k'l' c[m][d1][k'l'] Q[d][d1][kl + k'l'] + Re es[d] b[m] Nx Ny ) / (Norm B)
// (same sums, re-associated: float32 rounding only).  The dM*dD-plane gradient spectra dc|df (64 MB written and read
// back per step at cfg3) never exist.  (Likewise the per-bin product G = F.C/(dM dD) used by the post-update MSE is the
// spectrum of the (2Nk-1) x (2Nl-1) kernel sum_m f[d'][m] (*) c[m][d] / (dM dD): gspec_gbody, pruned_kernels.hip.)
#include "internal.h"
#include "device_util.h"
#include "update_device.h"
#include <algorithm>

namespace aefft {

// Workgroup = (which gradient, fixed inner channel a, tile of TM outer channels).  For g_f the Q rows Q[a][d1][.] and for g_c
// the Q columns Q[d1][a][.] (dD x T*T floats) are staged in LDS once, with the tile's weights.  Thread <-> (m in tile, tap
// row k) owns the NK outputs g[m][a][k][0..NK): per (d1, k2) it reads one weight row (NK floats) and one Q row (T floats)
// for NK*NK FMAs (a 1-D valid correlation in registers), i.e. ~0.5 LDS reads per FMA instead of 2.
// Blocks of a problem: [0, dD*mt) -> g_c, [dD*mt, 2*dD*mt) -> g_f, mt = ceil(dM/TM).
constexpr int WG_DB = 32;            // inner channels staged per block (a multiple of the 8 slices)
#ifndef AEFFT_X_WGRAD_W
#define AEFFT_X_WGRAD_W 1
#endif
template <int NK>
__global__ __launch_bounds__(256, AEFFT_X_WGRAD_W) void wgrad_taps_kernel(const WgradGroup g)
{
    AEFFT_WGTIME(2);
    constexpr int T = 2 * NK - 1, KK = NK * NK, TT = T * T, SL = 8, TM = 256 / (NK * SL);    // 6 outer channels x 8 d1 slices per workgroup for 5x5 (4 slices: 18.1 us, 8: 14.2 us at cfg3)
    extern __shared__ float sh[];                       // Qs[dD][TT] | ws[dD][TM][KK] | red[256][NK]
    if ((int)blockIdx.x == g.start[g.n]) {              // (trailing workgroup, only launched for it) the previous step's deferred MSE sums
        mse_finish_body(g.fin_slots, g.fin_out, nullptr, g.fin_tail, g.fin_L, g.fin_scale, sh);
        return;
    }
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
    const WgradProb& q = g.q[p];
    const int dM = q.dM, dD = q.dD;
    const int mt = (dM + TM - 1) / TM;
    int blk = blockIdx.x - g.start[p];
    const bool isf = blk >= dD * mt;
    if (isf) blk -= dD * mt;
    const int a = blk / mt, m0 = (blk - a * mt) * TM;
    // The inner channels d1 go through LDS in blocks of WG_DB: a pair with 64 inner channels (cfg5's 64 -> 128) would otherwise need 64 KB per
    // workgroup, and the launch's LDS size -- the largest pair's -- applies to EVERY workgroup of the grouped launch: 2 per CU instead of 4.
    // Per slice the channels are still visited in ascending order: the sums are those of the single-block form, bit for bit.
    const int DBmax = dD < WG_DB ? dD : WG_DB;
    float* Qs = sh;
    float* ws = sh + DBmax * TT;
    float* red = ws + DBmax * TM * KK;
    const int s = threadIdx.x / (TM * NK), rem = threadIdx.x - s * (TM * NK);
    const int ml = rem / NK, k = rem - ml * NK;
    const int m = m0 + ml;
    float acc[NK];
#pragma unroll
    for (int l = 0; l < NK; ++l) acc[l] = 0.f;
    const int nq = q.nq;
    for (int db = 0; db < dD; db += WG_DB) {
        const int nb = dD - db < WG_DB ? dD - db : WG_DB;
        if (db > 0) __syncthreads();                                  // the previous block's tiles have been consumed
        // staging in batches of independent loads per thread (hipcc keeps load -> wait -> store order inside a rolled loop:
        // one memory round trip per element otherwise)
        auto wload = [&](int t) {
            t = min(t, nb * TM * KK - 1);
            const int d1 = db + t / (TM * KK), r2 = t % (TM * KK);
            const int m2 = min(m0 + r2 / KK, dM - 1), r = r2 % KK;
            return isf ? q.c[((long)m2 * dD + d1) * KK + r] : q.f[((long)d1 * dM + m2) * KK + r];
        };
        // the first 24 loads per thread of the weight slab go out BEFORE the Q loads: both are in flight in the same round trip
        float wv[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            if (u * 256 >= nb * TM * KK) break;                          // uniform
            wv[u] = wload(u * 256 + (int)threadIdx.x);
        }
        if (nq <= 1) {
            for (int t0 = 0; t0 < nb * TT; t0 += 256 * 12) {
                float v[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    if (t0 + u * 256 >= nb * TT) break;                       // uniform
                    const int t = min(t0 + u * 256 + (int)threadIdx.x, nb * TT - 1);
                    const int d1 = db + t / TT, r = t % TT;
                    v[u] = isf ? q.Q[((long)a * dD + d1) * TT + r] : q.Q[((long)d1 * dD + a) * TT + r];
                }
#pragma unroll
                for (int u = 0; u < 12; ++u) { const int t = t0 + u * 256 + threadIdx.x; if (t < nb * TT) Qs[t] = v[u]; }
            }
        } else {
            // the producer left nq row-chunk partial sums per plane ([plane][chunk][tap]): added here in chunk order, 4 elements x 4
            // chunks of independent loads per batch
            for (int t0 = 0; t0 < nb * TT; t0 += 256 * 4) {
                float acc4[4] = {0.f, 0.f, 0.f, 0.f};
                for (int c0 = 0; c0 < nq; c0 += 4) {
                    float v[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int t = min(t0 + u * 256 + (int)threadIdx.x, nb * TT - 1);
                        const int d1 = db + t / TT, r = t % TT;
                        const long pl = isf ? (long)a * dD + d1 : (long)d1 * dD + a;
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[u][c] = q.Q[(pl * nq + min(c0 + c, nq - 1)) * TT + r];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (c0 + c < nq) acc4[u] += v[u][c];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int t = t0 + u * 256 + threadIdx.x; if (t < nb * TT) Qs[t] = acc4[u]; }
            }
        }
        // the rest of the weight slab (a 32-channel block is 19 loads per thread: inside the first batch)
#pragma unroll
        for (int u = 0; u < 24; ++u) { const int t = u * 256 + threadIdx.x; if (t < nb * TM * KK) ws[t] = wv[u]; }
        for (int t0 = 256 * 24; t0 < nb * TM * KK; t0 += 256 * 24) {
            float v[24];
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                if (t0 + u * 256 >= nb * TM * KK) break;                 // uniform: whole load instructions are skipped
                v[u] = wload(t0 + u * 256 + (int)threadIdx.x);
            }
#pragma unroll
            for (int u = 0; u < 24; ++u) { const int t = t0 + u * 256 + threadIdx.x; if (t < nb * TM * KK) ws[t] = v[u]; }
        }
        __syncthreads();
        if (db == 0) AEFFT_WGSTAMP(2, 0);
        if (s < SL && m < dM) {
            for (int d1 = s; d1 < nb; d1 += SL) {
                const float* wb = ws + (d1 * TM + ml) * KK;
                const float* Qb = Qs + d1 * TT + k * T;
#pragma unroll
                for (int k2 = 0; k2 < NK; ++k2) {
                    float w[NK], qr[T];
#pragma unroll
                    for (int l2 = 0; l2 < NK; ++l2) w[l2] = wb[k2 * NK + l2];
#pragma unroll
                    for (int t = 0; t < T; ++t) qr[t] = Qb[k2 * T + t];
#pragma unroll
                    for (int l2 = 0; l2 < NK; ++l2)
#pragma unroll
                        for (int l = 0; l < NK; ++l) acc[l] = fmaf(w[l2], qr[l + l2], acc[l]);
                }
            }
        }
    }
#pragma unroll
    for (int l = 0; l < NK; ++l) red[threadIdx.x * NK + l] = acc[l];
    __syncthreads();
    AEFFT_WGSTAMP(2, 1);
    if (s != 0 || m >= dM) return;
#pragma unroll
    for (int l = 0; l < NK; ++l) {
        float v = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < SL; ++s2) v += red[(s2 * (TM * NK) + rem) * NK + l];     // slice order: deterministic
        acc[l] = v;
    }
    const float bias = isf ? q.es[2 * a] * q.b[m] * q.norm : 0.f;                    // the b0 term of fft.cu:448-455 at the DC bin
    float* dst = isf ? q.gf + ((long)a * dM + m) * KK + k * NK : q.gc + ((long)m * dD + a) * KK + k * NK;
#pragma unroll
    for (int l = 0; l < NK; ++l) dst[l] = (acc[l] + bias) * q.inv_den;
}

hipError_t launch_wgrad_taps_group(WgradGroup& g, int Nk, hipStream_t st)
{
    if (g.n < 1 || g.n > 8 || (Nk != 3 && Nk != 5)) return hipErrorInvalidValue;
    const int KK = Nk * Nk, TT = (2 * Nk - 1) * (2 * Nk - 1), TM = 256 / (Nk * 8);
    int total = 0; size_t lds = 0;
    for (int i = 0; i < g.n; ++i) {
        const WgradProb& q = g.q[i];
        g.start[i] = total; total += 2 * q.dD * ((q.dM + TM - 1) / TM);
        const size_t db = std::min(q.dD, WG_DB);
        lds = std::max(lds, sizeof(float) * (db * TT + db * TM * KK + 256 * Nk));
    }
    g.start[g.n] = total;
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = Nk == 3 ? hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_taps_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                               : hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_taps_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    static_assert(MSE_SLOTS == 256, "the deferred MSE sums run as one workgroup of this launch");
    const int extra = g.fin_slots ? 1 : 0;
    if (Nk == 3) wgrad_taps_kernel<3><<<dim3(total + extra), 256, lds, st>>>(g);
    else wgrad_taps_kernel<5><<<dim3(total + extra), 256, lds, st>>>(g);
    return hipGetLastError();
}

}  // namespace aefft

#if defined(AEFFT_X_WGTIME) && AEFFT_X_WGTIME
extern "C" int aefft_debug_wgtime_weight(void* p) { return aefft::wgtime_set_tu(p); }
#endif
