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
#include "../../include/aefft.h"
#include "internal.h"
#include <algorithm>
#include <cstdlib>

namespace aefft {

// XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin by their linear id, so tiles that are neighbours in the image --
// and share the cache lines of their halos (a 64-pixel tile row with a one-pixel halo touches 4 lines of 128 B for 2 lines of payload) -- land
// in 8 different L2s and each fetches those lines from the fabric.  Remapped, every XCD walks a contiguous range of the (tile, frame)
// sequence: neighbours meet in the same L2 at nearly the same time.  (Launches with one map tile, gridDim.y == 1; otherwise identity.)
__device__ __forceinline__ void xcd_tile_order(int& bx, long& bz)
{
    const long total = (long)gridDim.x * gridDim.z;
    if (gridDim.y != 1 || (total & 7) != 0) return;
    const long lin = blockIdx.x + (long)gridDim.x * blockIdx.z;
    const long logical = (lin & 7) * (total >> 3) + (lin >> 3);
    bx = (int)(logical % gridDim.x); bz = logical / gridDim.x;
}


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


// ------------------------------------------------------------------------------------------
// LDS-tiled forms for the kernel sizes the application uses (3x3, 5x5, 7x7).
//
// dconv_kernel<TM,NK>: direct convolution / back-convolution.  A workgroup owns a 16x16 output tile of TM output maps of
// one frame; per input channel the (16+NK-1)^2 input tile is staged in LDS once (range tests, the out-in subtraction and
// the /dM of Conv_gpu, backproplib.cu:134, applied while staging), then every thread runs NK*NK taps x TM maps out of
// LDS with the weights in scalar registers (uniform indices -> s_load).  Summation order d, k, l as in the reference loops.
// ------------------------------------------------------------------------------------------
struct DConvArgs {
    const float *in, *in2;      // input planes [B][Din][Nx][Ny]; in2 != null: the input is in - in2 (s0 = out - in, backproplib.cu:312)
    const float* w;             // weight of (m, d, k, l) at w[m*w_m + d*w_d + k*Nl + l]
    const float* bias;          // [M] or null
    float* out;                 // [B][M][Nx][Ny]
    int Din, M, Nx, Ny, w_m, w_d;
    int ik0, il0;               // tap k sits at offset ik0 + k (backproplib.cu:85: -2*ak-1+k)
    int sgn;                    // -1: input index = i - offset (conv_parallel); +1: i + offset (back-convolution through f)
    int lo_in, hi_lo;           // valid input indices [lo_in, N); outputs with i < hi_lo or j < hi_lo are 0 (back-conv with lo = 1)
    float div;                  // input divided by this while staging (1: none)
    // fused Pool (netlib.cpp:114-164, scale > 0) in front of the convolution: `in` is the UN-pooled plane [Din][Nx*pool][Ny*pool]
    // and every staged input pixel is max(0, trunc(max of its pool x pool window)); pooled_out (nullable) also receives the pooled
    // layer [B][Din][Nx][Ny] (the training step needs it as the pair's input).  pool == 0: off.
    int pool; float* pooled_out;
    int nt_out;                 // (set by launch_dconv) the output is larger than the Infinity Cache: streaming stores
};

template <int TM, int NK>
__global__ __launch_bounds__(256) void dconv_kernel(const DConvArgs a)
{
    constexpr int TW = 16 + NK - 1, DC = 8;              // DC input channels staged per barrier pair
    __shared__ float tile[DC][TW * TW];
    const int tiles_y = (a.Ny + 15) / 16;
    const int ti = blockIdx.x / tiles_y, tj = blockIdx.x - ti * tiles_y;
    const int m0 = blockIdx.y * TM;
    const long bb = blockIdx.z;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int i = ti * 16 + ty, j = tj * 16 + tx;
    // first input row / column the tile needs
    const int r0 = a.sgn < 0 ? ti * 16 - a.ik0 - (NK - 1) : ti * 16 + a.ik0;
    const int c0 = a.sgn < 0 ? tj * 16 - a.il0 - (NK - 1) : tj * 16 + a.il0;
    const long plane = (long)a.Nx * a.Ny;
    float acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) acc[t] = 0.f;
    for (int d0 = 0; d0 < a.Din; d0 += DC) {
        const int nd = min(DC, a.Din - d0);
        __syncthreads();
        // staging in branch-free batches of 8 loads per thread (a rolled loop is one memory round trip per element under hipcc)
        for (int t0 = 0; t0 < nd * TW * TW; t0 += 256 * 8) {
            float v[8], v2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * 256 + threadIdx.x;
                const int dl = min(t / (TW * TW), nd - 1), q = t % (TW * TW);
                const int r = r0 + q / TW, cc = c0 + q % TW;
                const bool ok = t < nd * TW * TW && r >= a.lo_in && r < a.Nx && cc >= a.lo_in && cc < a.Ny;
                const long idx = ok ? (bb * a.Din + d0 + dl) * plane + (long)r * a.Ny + cc : 0;
                v[u] = a.in[idx];
                v2[u] = a.in2 ? a.in2[idx] : 0.f;
                if (!ok) { v[u] = 0.f; v2[u] = 0.f; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * 256 + threadIdx.x;
                if (t >= nd * TW * TW) break;
                float x = v[u] - v2[u];
                if (a.div != 1.f) x = x / a.div;
                tile[t / (TW * TW)][t % (TW * TW)] = x;
            }
        }
        __syncthreads();
        for (int dl = 0; dl < nd; ++dl) {
            const float* wd = a.w + (long)(d0 + dl) * a.w_d;
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int l = 0; l < NK; ++l) {
                    const float x = tile[dl][(a.sgn < 0 ? ty + (NK - 1 - k) : ty + k) * TW + (a.sgn < 0 ? tx + (NK - 1 - l) : tx + l)];
#pragma unroll
                    for (int t = 0; t < TM; ++t) {
                        const int m = m0 + t < a.M ? m0 + t : a.M - 1;        // uniform: the weight comes through a scalar load
                        acc[t] = fmaf(wd[(long)m * a.w_m + k * NK + l], x, acc[t]);
                    }
                }
        }
    }
    if (i >= a.Nx || j >= a.Ny) return;
    const bool zero = i < a.hi_lo || j < a.hi_lo;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int m = m0 + t;
        if (m >= a.M) break;
        a.out[(bb * a.M + m) * plane + (long)i * a.Ny + j] = zero ? 0.f : acc[t] + (a.bias ? a.bias[m] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------
// dconv4_kernel<TM,NK>: the few-map form (M <= 4: the decoder's last layer, 50 -> 3 maps).  Too few maps for the matrix cores
// (a 32-map MFMA tile would be 90 % padding) and, in dconv_kernel, one LDS read plus TM scalar weight loads per TM FMAs: the
// scalar loads share the LDS wait counter, so every input read waits for them.  Here a thread owns 4 consecutive pixels of a
// row x TM maps: per (channel, tap row) it reads its 4+NK-1 inputs (16-byte aligned vector reads) and the NK*TM weights of the
// row from an LDS slab (broadcast reads), for 4*NK*TM FMAs -- (2 + NK*TM/4) LDS instructions per 4*NK*TM FMAs instead of
// NK*(1 LDS + TM scalar).  Workgroup = 16 rows x 64 columns of one frame.  Same staging, tests and summation order (d, k, l).
// ------------------------------------------------------------------------------------------
// output rows per thread: 2 share their input rows in registers (21 instead of 36 LDS instructions per 72 FMAs) but double the tile, the
// staging registers (209 VGPRs) and halve the workgroups: 171 us against 146 at 32 x 256^2 x (50 -> 3) -- one row it is
constexpr int DCONV4_RW = 1;
template <int TM, int NK, bool SUB>
__global__ __launch_bounds__(256) void dconv4_kernel(const DConvArgs a)
{
    constexpr int PX = 4, RW = DCONV4_RW, TR = 16 * RW, TC = 16 * PX, TWY = TR + NK - 1, TWX = ((TC + NK - 1) + 3) & ~3, DC = 4, TSZ = TWY * TWX;
    constexpr int NLD = (DC * TSZ + 255) / 256;           // staged elements per thread and stage
    constexpr int WS = (NK * TM + 3) & ~3;                // weights of one (channel, tap row): [l][t], padded to whole float4
    __shared__ __attribute__((aligned(16))) float tile[DC][TSZ];
    extern __shared__ __attribute__((aligned(16))) float wl[];     // [Din][NK][WS]: ALL weights of this workgroup's maps, staged once
    const int tiles_y = (a.Ny + TC - 1) / TC;
    int bx = blockIdx.x; long bb = blockIdx.z;
    xcd_tile_order(bx, bb);
    const int ti = bx / tiles_y, tj = bx - ti * tiles_y;
    const int m0 = blockIdx.y * TM;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int i = ti * TR + ty * RW, j0 = tj * TC + tx * PX;               // the thread's RW rows: i, i + 1
    for (int t = threadIdx.x; t < a.Din * NK * NK * TM; t += 256) {      // (maps past M read the last map: their sums are never stored)
        const int tm = t % TM, l = (t / TM) % NK, k = (t / (TM * NK)) % NK, d = t / (TM * NK * NK);
        const int m = min(m0 + tm, a.M - 1);
        // stored at the tile offset the tap reads (sgn < 0: input index = output index - tap offset, i.e. the taps run backwards
        // over the tile), so that the compute loop indexes its registers with compile-time constants
        const int kk = a.sgn < 0 ? NK - 1 - k : k, ll = a.sgn < 0 ? NK - 1 - l : l;
        wl[(d * NK + kk) * WS + ll * TM + tm] = a.w[(long)m * a.w_m + (long)d * a.w_d + k * NK + l];
    }
    const int r0 = a.sgn < 0 ? ti * TR - a.ik0 - (NK - 1) : ti * TR + a.ik0;
    const int c0 = a.sgn < 0 ? tj * TC - a.il0 - (NK - 1) : tj * TC + a.il0;
    const long plane = (long)a.Nx * a.Ny;
    // this thread's staged elements: offset inside a stage's DC planes (the same for every stage) and whether the element exists
    static_assert(NLD <= 64, "one mask bit per staged element");
    int off[NLD]; unsigned long long okm = 0;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int t = u * 256 + threadIdx.x;
        const int dl = min(t / TSZ, DC - 1), q = t % TSZ;
        const int r = r0 + q / TWX, cc = c0 + q % TWX;
        const bool ok = t < DC * TSZ && r >= a.lo_in && r < a.Nx && cc >= a.lo_in && cc < a.Ny;
        off[u] = ok ? dl * (int)plane + r * a.Ny + cc : 0;
        okm |= (ok ? 1ull : 0ull) << u;
    }
    float acc[RW][TM][PX];
#pragma unroll
    for (int rw = 0; rw < RW; ++rw)
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[rw][t][p] = 0.f;
    float v[NLD], v2[SUB ? NLD : 1];
    auto prefetch = [&](int d0) {                          // the next stage's inputs: in flight while this stage computes
        const int nd = min(DC, a.Din - d0);
        const float* base = a.in + (bb * a.Din + d0) * plane;
        const float* base2 = SUB ? a.in2 + (bb * a.Din + d0) * plane : nullptr;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const bool ok = ((okm >> u) & 1ull) && min((u * 256 + (int)threadIdx.x) / TSZ, DC - 1) < nd;
            v[u] = base[ok ? off[u] : 0];
            if (SUB) v2[u] = base2[ok ? off[u] : 0];
            if (!ok) { v[u] = 0.f; if (SUB) v2[u] = 0.f; }
        }
    };
    prefetch(0);
    for (int d0 = 0; d0 < a.Din; d0 += DC) {
        const int nd = min(DC, a.Din - d0);
        __syncthreads();                                  // the previous stage's reads are done
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int t = u * 256 + threadIdx.x;
            if (t < DC * TSZ) {
                float x = SUB ? v[u] - v2[u] : v[u];
                if (a.div != 1.f) x = x / a.div;
                tile[0][t] = x;
            }
        }
        __syncthreads();
        if (d0 + DC < a.Din) prefetch(d0 + DC);
        for (int dl = 0; dl < nd; ++dl) {
            // the RW + NK - 1 input rows of the thread's RW output rows, once per channel (every row serves up to NK tap rows; with one
            // output row per thread the LDS pipe, not the FMAs, set the pace: 36 LDS instructions per 72 FMAs, now 21)
            float x[RW + NK - 1][PX + NK - 1];
#pragma unroll
            for (int ry = 0; ry < RW + NK - 1; ++ry) {
                const float* xr = &tile[dl][(ty * RW + ry) * TWX + tx * PX];     // (k, l: tile offsets, see the weight slab)
                const float4 x4 = *reinterpret_cast<const float4*>(xr);
                x[ry][0] = x4.x; x[ry][1] = x4.y; x[ry][2] = x4.z; x[ry][3] = x4.w;
#pragma unroll
                for (int e = PX; e < PX + NK - 1; ++e) x[ry][e] = xr[e];
            }
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                float w[WS];
#pragma unroll
                for (int e = 0; e < WS; e += 4) {
                    const float4 w4 = *reinterpret_cast<const float4*>(&wl[((d0 + dl) * NK + k) * WS + e]);
                    w[e] = w4.x; w[e + 1] = w4.y; w[e + 2] = w4.z; w[e + 3] = w4.w;
                }
#pragma unroll
                for (int rw = 0; rw < RW; ++rw)
#pragma unroll
                    for (int l = 0; l < NK; ++l)
#pragma unroll
                        for (int t = 0; t < TM; ++t)
#pragma unroll
                            for (int p = 0; p < PX; ++p) acc[rw][t][p] = fmaf(w[l * TM + t], x[rw + k][p + l], acc[rw][t][p]);
            }
        }
    }
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
        const int ir = i + rw;
        if (ir >= a.Nx) return;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int m = m0 + t;
            if (m >= a.M) break;
            const float bs = a.bias ? a.bias[m] : 0.f;
            float* dst = a.out + (bb * a.M + m) * plane + (long)ir * a.Ny + j0;
            float o[PX];
#pragma unroll
            for (int p = 0; p < PX; ++p) o[p] = (ir < a.hi_lo || j0 + p < a.hi_lo) ? 0.f : acc[rw][t][p] + bs;
            if (j0 + PX <= a.Ny && (a.Ny & 3) == 0 && (reinterpret_cast<size_t>(a.out) & 15) == 0) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            else {
#pragma unroll
                for (int p = 0; p < PX; ++p) if (j0 + p < a.Ny) dst[p] = o[p];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// mconv_kernel<MB,NK>: the same convolution as an IMPLICIT GEMM on the matrix cores (v_mfma_f32_32x32x2_f32, exact f32):
//     out[m][pixel] = sum_{(d,k,l)} w[m][(d,k,l)] * in[d][pixel shifted by (k,l)]        M = maps, N = pixels, K = Din*NK*NK
// A = weights (rows = 32 maps of a block, k = 2 consecutive (d,k,l) indices), B = input pixels (cols = 32 pixels of one tile row,
// same 2 k), gathered per lane from the LDS input tile -- the im2col matrix is never materialised.  A workgroup owns an
// 8-row x 32-column pixel tile x 32*MB maps of one frame; wave w owns tile rows 2w, 2w+1, i.e. 2*MB accumulator blocks of 32x32;
// per k-step it reads 2 B operands and MB A operands from LDS for 2*MB MFMAs of 2048 MACs.  The D layout puts the 32 pixels of a
// row on consecutive lanes: every store instruction writes two full 128-byte lines (the 16-pixel tiles of the VALU kernel write
// 64-byte runs, and this layer is bound by its output stream).  Input channels are staged DC at a time together with their weight
// slab wl[(d,k,l)][map] (zero-filled beyond K and M, so padded k-steps and map blocks contribute nothing).  Range tests, the
// out-in subtraction, the /dM of Conv_gpu (backproplib.cu:134) and the optional Pool are applied while staging, as in
// dconv_kernel.  The sums run over (d,k,l) in pairs instead of one at a time: float32 rounding only.
// ------------------------------------------------------------------------------------------
typedef float v16f_s __attribute__((ext_vector_type(16)));

template <int NK> struct MConvCfg { static constexpr int DC = NK == 3 ? 8 : (NK == 5 ? 4 : 2), KK = NK * NK, KC = DC * KK, KCP = (KC + 1) & ~1; };

template <int MB, int NK>
__global__ __launch_bounds__(256) void mconv_kernel(const DConvArgs a)
{
    using Cfg = MConvCfg<NK>;
    constexpr int TR = 8, TC = 32, TWY = TR + NK - 1, TWX = TC + NK - 1, TSZ = TWY * TWX;
    constexpr int DC = Cfg::DC, KK = Cfg::KK, KCP = Cfg::KCP, MT = 32 * MB;
    __shared__ float tile[DC][TSZ];
    __shared__ float wl[KCP][MT];
    __shared__ float bsl[MT];                               // the block's biases (read 16*MB times per lane by the epilogue)
    const int tiles_y = (a.Ny + TC - 1) / TC;
    int bx = blockIdx.x; long bb = blockIdx.z;
    xcd_tile_order(bx, bb);
    const int ti = bx / tiles_y, tj = bx - ti * tiles_y;
    const int m0 = blockIdx.y * MT;
    if ((int)threadIdx.x < MT) bsl[threadIdx.x] = (a.bias && m0 + (int)threadIdx.x < a.M) ? a.bias[m0 + threadIdx.x] : 0.f;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 31, kq = lane >> 5;
    const int r0 = a.sgn < 0 ? ti * TR - a.ik0 - (NK - 1) : ti * TR + a.ik0;
    const int c0 = a.sgn < 0 ? tj * TC - a.il0 - (NK - 1) : tj * TC + a.il0;
    const long plane = (long)a.Nx * a.Ny;
    v16f_s acc[2][MB];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][mb][e] = 0.f;
    for (int d0 = 0; d0 < a.Din; d0 += DC) {
        const int nd = min(DC, a.Din - d0);
        const int nk = nd * KK;
        __syncthreads();
        if (a.pool) {
            const int s = a.pool, Nxi = a.Nx * s, Nyi = a.Ny * s;
            for (int t = threadIdx.x; t < nd * TSZ; t += 256) {
                const int dl = t / TSZ, q = t - dl * TSZ;
                const int r = r0 + q / TWX, cc = c0 + q % TWX;
                float x = 0.f;
                if (r >= 0 && r < a.Nx && cc >= 0 && cc < a.Ny) {
                    const float* src = a.in + ((bb * a.Din + d0 + dl) * (long)Nxi + (long)r * s) * Nyi + (long)cc * s;
                    float mx = 0.f;
                    for (int k = 0; k < s; ++k)
                        for (int l = 0; l < s; ++l) mx = fmaxf(mx, src[(long)k * Nyi + l]);
                    x = (float)(int)mx;                                       // `int smax = 0` accumulator of netlib.cpp:127-136
                    // the tile's own pixels (not the halo) of map block 0 also publish the pooled layer
                    if (a.pooled_out && blockIdx.y == 0 && r >= ti * TR && r < ti * TR + TR && cc >= tj * TC && cc < tj * TC + TC)
                        a.pooled_out[(bb * a.Din + d0 + dl) * plane + (long)r * a.Ny + cc] = x;
                    if (r < a.lo_in || cc < a.lo_in) x = 0.f;                 // the convolution's own range test ('>0' of netlib.cpp:344)
                }
                tile[dl][q] = a.div != 1.f ? x / a.div : x;
            }
        } else {
            for (int t0 = 0; t0 < nd * TSZ; t0 += 256 * 8) {
                float v[8], v2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t0 + u * 256 + threadIdx.x;
                    const int dl = min(t / TSZ, nd - 1), q = t % TSZ;
                    const int r = r0 + q / TWX, cc = c0 + q % TWX;
                    const bool ok = t < nd * TSZ && r >= a.lo_in && r < a.Nx && cc >= a.lo_in && cc < a.Ny;
                    const long idx = ok ? (bb * a.Din + d0 + dl) * plane + (long)r * a.Ny + cc : 0;
                    v[u] = a.in[idx];
                    v2[u] = a.in2 ? a.in2[idx] : 0.f;
                    if (!ok) { v[u] = 0.f; v2[u] = 0.f; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t0 + u * 256 + threadIdx.x;
                    if (t >= nd * TSZ) break;
                    float x = v[u] - v2[u];
                    if (a.div != 1.f) x = x / a.div;
                    tile[t / TSZ][t % TSZ] = x;
                }
            }
        }
        // weight slab of the chunk: wl[(dl,k,l)][map], zero beyond the chunk's K and the layer's M
        const int nkp = (nk + 1) & ~1;
        for (int t0 = 0; t0 < nkp * MT; t0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * 256 + threadIdx.x;
                const int kidx = min(t / MT, KCP - 1), ml = t % MT;
                const bool ok = t < nkp * MT && kidx < nk && m0 + ml < a.M;
                const int dl = kidx / KK, kl = kidx - dl * KK;
                v[u] = ok ? a.w[(long)(m0 + ml) * a.w_m + (long)(d0 + dl) * a.w_d + kl] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int t = t0 + u * 256 + threadIdx.x; if (t < nkp * MT) wl[t / MT][t % MT] = v[u]; }
        }
        __syncthreads();
        for (int ks = 0; ks < nk; ks += 2) {
            const int kidx = min(ks + kq, nk - 1);                      // (beyond nk the weights are zero)
            const int dl = kidx / KK, kl = kidx - dl * KK;
            const int kr = kl / NK, lc = kl - kr * NK;
            const int rr = a.sgn < 0 ? NK - 1 - kr : kr, cc = a.sgn < 0 ? NK - 1 - lc : lc;
            const float* tp = &tile[dl][(2 * wv + rr) * TWX + col + cc];
            float bv[2], av[MB];
            bv[0] = tp[0]; bv[1] = tp[TWX];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = wl[ks + kq][32 * mb + col];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[r][mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb], bv[r], acc[r][mb], 0, 0, 0);
        }
    }
    // D layout (32x32): lane holds column (pixel) lane&31 of rows (maps) (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int j = tj * TC + col;
    if (j >= a.Ny) return;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = ti * TR + 2 * wv + r;
        if (i >= a.Nx) continue;
        const bool zero = i < a.hi_lo || j < a.hi_lo;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ml = 32 * mb + (e & 3) + 8 * (e >> 2) + 4 * kq;
                if (m0 + ml >= a.M) continue;
                const float o = zero ? 0.f : acc[r][mb][e] + bsl[ml];
                float* dst = &a.out[(bb * a.M + m0 + ml) * plane + (long)i * a.Ny + j];
                if (a.nt_out) __builtin_nontemporal_store(o, dst);      // (uniform) a layer larger than the caches: do not displace its own input
                else *dst = o;
            }
    }
}

template <int NK> static hipError_t run_mconv(const DConvArgs& a, int B, hipStream_t st)
{
    const int tiles = ((a.Nx + 7) / 8) * ((a.Ny + 31) / 32);
    if (a.M > 32) mconv_kernel<2, NK><<<dim3(tiles, (a.M + 63) / 64, B), 256, 0, st>>>(a);
    else mconv_kernel<1, NK><<<dim3(tiles, (a.M + 31) / 32, B), 256, 0, st>>>(a);
    return hipGetLastError();
}

template <int NK> static hipError_t run_dconv(const DConvArgs& a, int B, hipStream_t st)
{
    if (a.M <= 4 && a.Ny >= 64 && (long)a.Nx * a.Ny * 4 < (1L << 31) && (size_t)a.Din * NK * ((NK * 4 + 3) & ~3) * 4 <= 32 * 1024 && !flag(AEFFT_F_NOFAST)) {        // few maps, wide planes: the register-blocked form
        const int tiles4 = ((a.Nx + 16 * DCONV4_RW - 1) / (16 * DCONV4_RW)) * ((a.Ny + 63) / 64);      // (dconv4_kernel: 16 * DCONV4_RW rows x 64 columns per workgroup)
#define AEFFT_D4(TMV) { const size_t wb = sizeof(float) * (size_t)a.Din * NK * ((NK * TMV + 3) & ~3);                            \
            if (a.in2) dconv4_kernel<TMV, NK, true><<<dim3(tiles4, 1, B), 256, wb, st>>>(a);                                      \
            else dconv4_kernel<TMV, NK, false><<<dim3(tiles4, 1, B), 256, wb, st>>>(a); }
        if (a.M == 4) AEFFT_D4(4)
        else if (a.M == 3) AEFFT_D4(3)
        else if (a.M == 2) AEFFT_D4(2)
        else AEFFT_D4(1)
#undef AEFFT_D4
        return hipGetLastError();
    }
    const int tiles = ((a.Nx + 15) / 16) * ((a.Ny + 15) / 16);
    if (a.M >= 12) dconv_kernel<16, NK><<<dim3(tiles, (a.M + 15) / 16, B), 256, 0, st>>>(a);
    else if (a.M >= 6) dconv_kernel<8, NK><<<dim3(tiles, (a.M + 7) / 8, B), 256, 0, st>>>(a);
    else dconv_kernel<4, NK><<<dim3(tiles, (a.M + 3) / 4, B), 256, 0, st>>>(a);
    return hipGetLastError();
}
static bool dconv_ok(int Nk, int Nl, int B) { return Nk == Nl && (Nk == 3 || Nk == 5 || Nk == 7) && B <= 65535; }
static hipError_t launch_dconv(const DConvArgs& a0, int Nk, int B, hipStream_t st)
{
    DConvArgs a = a0;
    a.nt_out = (double)B * a.M * a.Nx * a.Ny * 4.0 > 192e6;
    // matrix cores whenever there is a GEMM to speak of (>= 8 maps); AEFFT_F_NOMFMA keeps the VALU tile kernel (and the fused
    // Pool exists only in the matrix-core kernel)
    if ((a.M >= 8 && !flag(AEFFT_F_NOMFMA)) || a.pool) {
        if (Nk == 3) return run_mconv<3>(a, B, st);
        if (Nk == 5) return run_mconv<5>(a, B, st);
        return run_mconv<7>(a, B, st);
    }
    if (Nk == 3) return run_dconv<3>(a, B, st);
    if (Nk == 5) return run_dconv<5>(a, B, st);
    return run_dconv<7>(a, B, st);
}

// ------------------------------------------------------------------------------------------
// wcorr_kernel<TA,TB,NK>: weight-gradient correlations
//     out[a][b][k][l] = sum_{frames, pixels} A[a][i][j] * Bp[b][i - ik][j - il]        (shifted index inside [lo, N))
// (dC: A = back-convolved error g, Bp = in; dF: A = out - in, Bp = hin; backproplib.cu:200-230,340-388 summed over the
// image instead of one launch + reduce per weight element).  A workgroup owns a 16-row band of one frame and a (TA x TB)
// tile of plane pairs: per 16x16 sub-tile the TB shifted planes are staged in LDS, every thread keeps TA*TB*NK*NK partial
// sums for its pixel column, reduced once at the end (wave shuffles, LDS across waves) into part[workgroup][...].  The
// plane sums of A (the bias gradients, :231,389) fall out of the same pass.  wsum_kernel adds the partials in a fixed order.
// ------------------------------------------------------------------------------------------
struct WCorrArgs {
    const float *A, *A2;        // [B][nA][Nx][Ny]; A2 != null: A - A2
    const float* Bp;            // [B][nB][Nx][Ny]
    float* part;                // [B * bands][atiles * btiles][TA*TB*KK + TA]
    int nA, nB, Nx, Ny, ik0, il0, lo;
};

template <int TA, int TB, int NK>
__global__ __launch_bounds__(256) void wcorr_kernel(const WCorrArgs a)
{
    constexpr int PX = 4, CW = 16 * PX;                  // a stage covers 16 rows x 64 columns: PX pixels per thread between barriers
    constexpr int TWR = 16 + NK - 1, TWC = CW + NK - 1, KK = NK * NK, NACC = TA * TB * KK;
    __shared__ float tile[TB][TWR * TWC];
    __shared__ float red[4][NACC + TA];
    const int btiles = (a.nB + TB - 1) / TB;
    const int at = blockIdx.y / btiles, bt = blockIdx.y - at * btiles;
    const int band = blockIdx.x;
    const long bb = blockIdx.z;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int i = band * 16 + ty;
    const long plane = (long)a.Nx * a.Ny;
    float acc[NACC], asum[TA];
#pragma unroll
    for (int e = 0; e < NACC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int t = 0; t < TA; ++t) asum[t] = 0.f;
    const int r0 = band * 16 - a.ik0 - (NK - 1);
    for (int tj = 0; tj < (a.Ny + CW - 1) / CW; ++tj) {
        const int c0 = tj * CW - a.il0 - (NK - 1);
        __syncthreads();
        for (int t0 = 0; t0 < TB * TWR * TWC; t0 += 256 * 8) {          // branch-free batches of 8 loads per thread
            float v[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int t = t0 + w * 256 + threadIdx.x;
                const int u = min(t / (TWR * TWC), TB - 1), q = t % (TWR * TWC);
                const int b = bt * TB + u;
                const int r = r0 + q / TWC, cc = c0 + q % TWC;
                const bool ok = t < TB * TWR * TWC && b < a.nB && r >= a.lo && r < a.Nx && cc >= a.lo && cc < a.Ny;
                v[w] = a.Bp[ok ? (bb * a.nB + b) * plane + (long)r * a.Ny + cc : 0];
                if (!ok) v[w] = 0.f;
            }
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int t = t0 + w * 256 + threadIdx.x;
                if (t < TB * TWR * TWC) tile[t / (TWR * TWC)][t % (TWR * TWC)] = v[w];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int px = 0; px < PX; ++px) {
            const int jl = px * 16 + tx, j = tj * CW + jl;          // consecutive lanes -> consecutive columns
            float av[TA];
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                const int ap = at * TA + t;
                float v = 0.f;
                if (ap < a.nA && i < a.Nx && j < a.Ny) {
                    const long q = (bb * a.nA + ap) * plane + (long)i * a.Ny + j;
                    v = a.A[q];
                    if (a.A2) v -= a.A2[q];
                }
                av[t] = v; asum[t] += v;
            }
#pragma unroll
            for (int u = 0; u < TB; ++u)
#pragma unroll
                for (int k = 0; k < NK; ++k)
#pragma unroll
                    for (int l = 0; l < NK; ++l) {
                        const float x = tile[u][(ty + (NK - 1 - k)) * TWC + jl + (NK - 1 - l)];
#pragma unroll
                        for (int t = 0; t < TA; ++t) acc[(t * TB + u) * KK + k * NK + l] = fmaf(av[t], x, acc[(t * TB + u) * KK + k * NK + l]);
                    }
        }
    }
    // workgroup reduction in a fixed order
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int e = 0; e < NACC + TA; ++e) {
        float v = e < NACC ? acc[e < NACC ? e : 0] : asum[e >= NACC ? e - NACC : 0];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][e] = v;
    }
    __syncthreads();
    float* dst = a.part + ((bb * gridDim.x + band) * gridDim.y + blockIdx.y) * (long)(NACC + TA);
    for (int e = threadIdx.x; e < NACC + TA; e += 256) dst[e] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
}

// out[a][b][kl] = scale * sum over (frame, band) partials; bias sums from the b-tile-0 workgroups
template <int TA, int TB, int NK>
__global__ __launch_bounds__(256) void wsum_kernel(const float* __restrict__ part, float* __restrict__ out, float* __restrict__ osum, int nA, int nB,
                                                   int nparts, float scale)
{
    // one wave per output: lanes stride over the (frame, band) partials, then a shuffle tree (fixed order: deterministic)
    constexpr int KK = NK * NK, NACC = TA * TB * KK;
    const int atiles = (nA + TA - 1) / TA, btiles = (nB + TB - 1) / TB;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int total = nA * nB * KK;
    if (e >= total + nA || (e >= total && !osum)) return;
    long col;
    if (e < total) {
        const int kl = e % KK, b = (e / KK) % nB, ap = e / (KK * nB);
        const int at = ap / TA, t = ap - at * TA, bt = b / TB, u = b - bt * TB;
        col = (long)(at * btiles + bt) * (NACC + TA) + (t * TB + u) * KK + kl;
    } else {
        const int ap = e - total, at = ap / TA, t = ap - at * TA;
        col = (long)(at * btiles) * (NACC + TA) + NACC + t;
    }
    const long stride = (long)atiles * btiles * (NACC + TA);
    float s = 0.f;
    for (int pidx = lane; pidx < nparts; pidx += 64) s += part[pidx * stride + col];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) { if (e < total) out[e] = s * scale; else osum[e - total] = s * scale; }
}

template <int TA, int TB, int NK>
static hipError_t run_wcorr(const WCorrArgs& a, int B, float* out, float* osum, float scale, hipStream_t st)
{
    const int bands = (a.Nx + 15) / 16, atiles = (a.nA + TA - 1) / TA, btiles = (a.nB + TB - 1) / TB;
    wcorr_kernel<TA, TB, NK><<<dim3(bands, atiles * btiles, B), 256, 0, st>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int total = a.nA * a.nB * NK * NK + a.nA;
    wsum_kernel<TA, TB, NK><<<dim3((total + 3) / 4), 256, 0, st>>>(a.part, out, osum, a.nA, a.nB, bands * B, scale);
    return hipGetLastError();
}
// ------------------------------------------------------------------------------------------
// mcorr_kernel<NK,NCB>: the same weight-gradient correlations as ONE GEMM per launch on the matrix cores (32x32x2 f32):
//     out[row a][col (b,k,l)] = sum_{frames, pixels} A[a][pix] * Bv[b][pix shifted by sgn*tap]      M = planes of A, N = nB*NK*NK, K = pixels
// Rows are the UNSHIFTED planes (always the side with the many channels: the back-convolved error g for dC, the hidden layer for
// dF after re-indexing the sum by i' = i - ik), columns the shifted copies of the few planes of the other side -- gathered per lane
// from an LDS tile with halo, so nothing is materialised.  One extra column of ones gives the row sums (dB), one extra row of ones
// the column sums at the zero-shift tap (dP).  A workgroup owns a 16-row band of one frame (x 64 rows x 32*NCB columns of the
// GEMM) and walks it in chunks of 2 image rows x 64 columns: the A tile [64][128 px] goes through LDS, every wave takes a quarter of the chunk's pixels = 16 k-steps of 2*NCB MFMAs.
// The global loads of the next TWO chunks are in flight while the current one is multiplied (two register sets).  Partials [workgroup][64][32*NCB] are summed
// in a fixed order by msum_kernel.
// ------------------------------------------------------------------------------------------
struct MCorrArgs {
    const float* A; int nA, loA;            // rows   [B][nA][Nx][Ny]; pixels with i < loA or j < loA count as 0
    const float *Bp, *Bp2; int nB, loB;     // columns: (Bp - Bp2)[B][nB][Nx][Ny] read at (i + sgn*(ik0+k), j + sgn*(il0+l)), 0 outside [loB, N)
    int sgn, Nx, Ny, ik0, il0;
    int ones_row, ones_col;                 // append a row / a column of ones
    int row0, col0;                         // first GEMM row / column of this launch's tile (multiples of 64 / 32*NCB)
    float* part;                            // [B * bands][64][32*NCB]
};

template <int NK, int NCB, int NRB>
__global__ __launch_bounds__(256) void mcorr_kernel(const MCorrArgs a)
{
    constexpr int KK = NK * NK, CR = 2, CC = 64, CPX = CR * CC, PA = CPX + 4;          // chunk: 2 rows x 64 columns; A tile pitch (16-byte rows)
    constexpr int TWR = CR + NK - 1, TWC = CC + NK - 1, TB = 3;                         // B tile with halo; at most 3 planes staged at a time
    __shared__ __attribute__((aligned(16))) float At[64 * PA];
    __shared__ float Bt[TB][TWR * TWC];
    const int band = blockIdx.x;
    const long bb = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 31, kq = lane >> 5;
    const long plane = (long)a.Nx * a.Ny;
    // per column block: which plane / tap this lane's column is, and its offset inside the B tile
    int cplane[NCB], coff[NCB];
    float cone[NCB];
    const int ncol = a.nB * KK;
    int pmin = 1 << 30, pmax = -1;                                                        // (uniform bounds of the planes this launch touches)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int col = a.col0 + 32 * cb + li;
        cone[cb] = (a.ones_col && col == ncol) ? 1.f : 0.f;
        if (col < ncol) {
            const int b = col / KK, kl = col - b * KK, k = kl / NK, l = kl - k * NK;
            cplane[cb] = b;
            // image index = pixel + sgn*(ik0 + k); tile origin = chunk origin + min shift
            const int sr = a.sgn * (a.ik0 + k), sc = a.sgn * (a.il0 + l);
            const int r0s = a.sgn > 0 ? a.ik0 : -(a.ik0 + NK - 1), c0s = a.sgn > 0 ? a.il0 : -(a.il0 + NK - 1);
            coff[cb] = (sr - r0s) * TWC + (sc - c0s);
        } else { cplane[cb] = -1; coff[cb] = 0; }
    }
    {
        const int c_lo = a.col0, c_hi = min(a.col0 + 32 * NCB, ncol) - 1;
        pmin = c_lo / KK; pmax = c_hi >= c_lo ? c_hi / KK : pmin;         // (a tile holding only the ones column still makes one pass)
    }
    const int r0s = a.sgn > 0 ? a.ik0 : -(a.ik0 + NK - 1), c0s = a.sgn > 0 ? a.il0 : -(a.il0 + NK - 1);
    v16f_s acc[NRB][NCB];                                             // NRB = 1: at most 32 GEMM rows, one row block of the matrix instruction
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rb][cb][e] = 0.f;
    constexpr int BR = 8;                                             // image rows per workgroup (band)
    const int nchunk_c = (a.Ny + CC - 1) / CC, nchunks = (BR / CR) * nchunk_c;
    constexpr int NBL = (TB * TWR * TWC + 255) / 256;
    // loads of a chunk: the A tile (64 planes x 128 px = 2048 float4, 8 per thread) and the B tiles of one plane group
    // (raw, branch-free loads from clamped addresses: every range test and mask is applied when the registers go to LDS -- a test
    //  on the loaded value right after the load would make each load wait for the one before it)
    auto load_A = [&](int ch, float4 (&av)[8]) {
        const int i0 = band * BR + (ch / nchunk_c) * CR, j0 = (ch % nchunk_c) * CC;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int t = w * 256 + threadIdx.x;
            const int row = t / (CPX / 4), q = (t % (CPX / 4)) * 4;              // GEMM row, pixel quad inside the chunk
            const int i = min(i0 + q / CC, a.Nx - 1), j = min(j0 + q % CC, a.Ny - 4);
            if (NRB == 1 && row >= 32) continue;                                 // (uniform per w: 4 rows per 128 threads... rows 32.. are never multiplied)
            const int ga = min(a.row0 + row, a.nA - 1);
            av[w] = *reinterpret_cast<const float4*>(a.A + (bb * a.nA + ga) * plane + (long)i * a.Ny + j);
        }
    };
    auto store_A = [&](int ch, const float4 (&av)[8]) {
        const int i0 = band * BR + (ch / nchunk_c) * CR, j0 = (ch % nchunk_c) * CC;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int t = w * 256 + threadIdx.x;
            const int row = t / (CPX / 4), q = (t % (CPX / 4)) * 4;
            if (NRB == 1 && row >= 32) continue;
            const int i = i0 + q / CC, j = j0 + q % CC;
            const int ga = a.row0 + row;
            const bool ones = a.ones_row && ga == a.nA;
            const float vv[4] = {av[w].x, av[w].y, av[w].z, av[w].w};
            float o[4];
            // (Ny is a multiple of 4 and j of the quad size: a quad is inside the image or outside as a whole; lo <= 1 can only cut its
            // first element.  One test per quad instead of five per element: the staging instructions, not the MFMAs, set this kernel's pace)
            const bool in_img = i < a.Nx && j < a.Ny;
            const bool rowok = ga < a.nA && in_img && i >= a.loA;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = (rowok && j + e >= a.loA) ? vv[e] : 0.f;
                if (ones && in_img) x = 1.f;
                o[e] = x;
            }
            // one 16-byte write per lane (consecutive lanes -> consecutive quads of a row: conflict-free); the MFMA's A-operand reads
            // of 32 rows at one pixel are 4-way conflicted with this pitch, which costs a few cycles beside 128 cycles of MFMA
            *reinterpret_cast<float4*>(At + row * PA + q) = make_float4(o[0], o[1], o[2], o[3]);
        }
    };
    auto load_B = [&](int ch, int p0, float (&bvr)[NBL], float (&bvr2)[NBL]) {
        const int i0 = band * BR + (ch / nchunk_c) * CR, j0 = (ch % nchunk_c) * CC;
#pragma unroll
        for (int w = 0; w < NBL; ++w) {
            const int t = min(w * 256 + (int)threadIdx.x, TB * TWR * TWC - 1);
            const int u = t / (TWR * TWC), q = t - u * (TWR * TWC);
            const int b = min(p0 + u, a.nB - 1);
            const int r = min(max(i0 + r0s + q / TWC, 0), a.Nx - 1), c = min(max(j0 + c0s + q % TWC, 0), a.Ny - 1);
            const long idx = (bb * a.nB + b) * plane + (long)r * a.Ny + c;
            bvr[w] = a.Bp[idx];
            bvr2[w] = a.Bp2 ? a.Bp2[idx] : 0.f;
        }
    };
    auto store_B = [&](int ch, int p0, const float (&bvr)[NBL], const float (&bvr2)[NBL]) {
        const int i0 = band * BR + (ch / nchunk_c) * CR, j0 = (ch % nchunk_c) * CC;
#pragma unroll
        for (int w = 0; w < NBL; ++w) {
            const int t = w * 256 + threadIdx.x;
            if (t < TB * TWR * TWC) {
                const int u = t / (TWR * TWC), q = t - u * (TWR * TWC);
                const int b = p0 + u;
                const int r = i0 + r0s + q / TWC, c = j0 + c0s + q % TWC;
                const bool ok = b <= pmax && b < a.nB && r >= a.loB && r < a.Nx && c >= a.loB && c < a.Ny;
                Bt[u][q] = ok ? bvr[w] - bvr2[w] : 0.f;
            }
        }
    };
    const bool one_group = pmax - pmin < TB;                          // (uniform) the usual case: all B planes staged at once, prefetched with A
    // one chunk: tiles from the register set -> LDS, the loads of chunk ch + 2 into the same set (two chunks in flight), the MFMAs
    auto step = [&](int ch, float4 (&av)[8], float (&bvr)[NBL], float (&bvr2)[NBL]) {
        __syncthreads();                                             // the previous chunk's MFMAs are done with the tiles
        store_A(ch, av);
        if (one_group) store_B(ch, pmin, bvr, bvr2);
        if (ch + 2 < nchunks) { load_A(ch + 2, av); if (one_group) load_B(ch + 2, pmin, bvr, bvr2); }
        for (int p0 = pmin; p0 <= pmax; p0 += TB) {
            if (!one_group) {
                if (p0 > pmin) __syncthreads();
                float tmp[NBL], tmp2[NBL];
                load_B(ch, p0, tmp, tmp2);
                store_B(ch, p0, tmp, tmp2);
            }
            __syncthreads();
            // wave wv: pixels [32 wv, 32 wv + 32) of the chunk, 2 per k-step
#pragma unroll 4
            for (int ks = 0; ks < 16; ++ks) {
                const int px = 32 * wv + 2 * ks + kq;
                const int pr = px / CC, pc = px - pr * CC;
                const float a0 = At[li * PA + px], a1 = NRB > 1 ? At[(32 + li) * PA + px] : 0.f;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const int u = cplane[cb] - p0;
                    float bvv = 0.f;
                    if (cplane[cb] >= 0) { if (u >= 0 && u < TB) bvv = Bt[u][coff[cb] + pr * TWC + pc]; }
                    else if (p0 == pmin) bvv = cone[cb];                 // the ones column counts once, with the first plane group
                    acc[0][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bvv, acc[0][cb], 0, 0, 0);
                    if (NRB > 1) acc[NRB - 1][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bvv, acc[NRB - 1][cb], 0, 0, 0);
                }
            }
        }
    };
    float4 av0[8], av1[8];
    float bv0[NBL], bv1[NBL], bw0[NBL], bw1[NBL];
    load_A(0, av0);
    if (one_group) load_B(0, pmin, bv0, bw0);
    if (nchunks > 1) { load_A(1, av1); if (one_group) load_B(1, pmin, bv1, bw1); }
    for (int ch = 0; ch < nchunks; ch += 2) {
        step(ch, av0, bv0, bw0);
        if (ch + 1 < nchunks) step(ch + 1, av1, bv1, bw1);
    }
    // the four waves' accumulators -> one partial tile, summed in wave order through LDS (At is free now)
    __syncthreads();
    float* red = At;                                                  // [64][32] per column block at a time
    float* dst = a.part + ((bb * gridDim.x + band) * 64) * (long)(32 * NCB);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        for (int w2 = 0; w2 < 4; ++w2) {
            if (wv == w2) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = 32 * rb + (e & 3) + 8 * (e >> 2) + 4 * kq;
                        float* q = red + row * 33 + li;
                        *q = w2 == 0 ? acc[rb][cb][e] : *q + acc[rb][cb][e];
                    }
            }
            __syncthreads();
        }
        for (int t = threadIdx.x; t < 64 * 32; t += 256) dst[(long)(t >> 5) * (32 * NCB) + 32 * cb + (t & 31)] = (NRB == 1 && t >= 32 * 32) ? 0.f : red[(t >> 5) * 33 + (t & 31)];
        __syncthreads();
    }
}

// out = scale * sum over workgroups of part[.][row][col] for the rows / columns of this tile; transposed scatter for dF
__global__ __launch_bounds__(256) void msum_kernel(const float* __restrict__ part, int nparts, int ncols_tile, int row0, int col0,
                                                   int nA, int ncol, int KK, int nB, int transpose, float* __restrict__ out,
                                                   float* __restrict__ row_sums, float* __restrict__ col_sums, int zero_tap, float scale)
{
    // one wave per element of the 64 x ncols_tile tile
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= 64 * ncols_tile) return;
    const int row = e / ncols_tile, colt = e - row * ncols_tile;
    const int ga = row0 + row, col = col0 + colt;
    const bool is_w = ga < nA && col < ncol, is_rs = ga < nA && col == ncol && row_sums, is_cs = ga == nA && col < ncol && col_sums;
    if (!is_w && !is_rs && !is_cs) return;
    float s = 0.f;
    for (int p = lane; p < nparts; p += 64) s += part[((long)p * 64 + row) * ncols_tile + colt];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane != 0) return;
    s *= scale;
    if (is_w) {
        const int b = col / KK, kl = col - b * KK;
        if (transpose) out[((long)b * nA + ga) * KK + kl] = s;          // [b][a][k][l]
        else out[((long)ga * nB + b) * KK + kl] = s;                    // [a][b][k][l]
    } else if (is_rs) row_sums[ga] = s;
    else { const int b = col / KK, kl = col - b * KK; if (kl == zero_tap) col_sums[b] = s; }
}

template <int NK> static hipError_t run_mcorr(MCorrArgs a, int B, float* out, int transpose, float* row_sums, float* col_sums, float scale, hipStream_t st)
{
    constexpr int KK = NK * NK;
    const int bands = (a.Nx + 7) / 8;
    const int nrows = a.nA + (a.ones_row ? 1 : 0), ncols = a.nB * KK + (a.ones_col ? 1 : 0);
    const int zk = -a.ik0, zl = -a.il0;                                // the tap with zero shift (exists for the reference geometries)
    const int zero_tap = (zk >= 0 && zk < NK && zl >= 0 && zl < NK) ? zk * NK + zl : -1;
    if (col_sums && zero_tap < 0) return hipErrorInvalidValue;
    const int ncb_all = (ncols + 31) / 32;
    for (int row0 = 0; row0 < nrows; row0 += 64)
        for (int cb0 = 0; cb0 < ncb_all; cb0 += 3) {
            const int ncb = std::min(3, ncb_all - cb0);
            a.row0 = row0; a.col0 = 32 * cb0;
            const bool one_rb = nrows - row0 <= 32;                 // a single 32-row block of the matrix instruction covers the rows
            if (one_rb) {
                if (ncb == 1) mcorr_kernel<NK, 1, 1><<<dim3(bands, B), 256, 0, st>>>(a);
                else if (ncb == 2) mcorr_kernel<NK, 2, 1><<<dim3(bands, B), 256, 0, st>>>(a);
                else mcorr_kernel<NK, 3, 1><<<dim3(bands, B), 256, 0, st>>>(a);
            } else {
                if (ncb == 1) mcorr_kernel<NK, 1, 2><<<dim3(bands, B), 256, 0, st>>>(a);
                else if (ncb == 2) mcorr_kernel<NK, 2, 2><<<dim3(bands, B), 256, 0, st>>>(a);
                else mcorr_kernel<NK, 3, 2><<<dim3(bands, B), 256, 0, st>>>(a);
            }
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            msum_kernel<<<dim3((64 * 32 * ncb + 3) / 4), 256, 0, st>>>(a.part, bands * B, 32 * ncb, row0, 32 * cb0, a.nA, a.nB * KK, KK, a.nB, transpose, out,
                                                                 row_sums, col_sums, zero_tap, scale);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}
static size_t mcorr_part_floats(int Nx, int B) { return (size_t)B * ((Nx + 7) / 8) * 64 * 96; }

// tile shapes: few A planes x few B planes such that TA*TB*NK*NK accumulators stay in registers
template <int NK> struct WTile;
template <> struct WTile<3> { static constexpr int TA = 2, TB = 3; };
template <> struct WTile<5> { static constexpr int TA = 1, TB = 2; };
template <> struct WTile<7> { static constexpr int TA = 1, TB = 1; };
template <int NK> static size_t wcorr_part_floats(int nA, int nB, int Nx, int B)
{
    constexpr int TA = WTile<NK>::TA, TB = WTile<NK>::TB;
    return (size_t)B * ((Nx + 15) / 16) * ((nA + TA - 1) / TA) * ((nB + TB - 1) / TB) * (TA * TB * NK * NK + TA);
}
size_t spatial_partial_floats(int B, int dD, int dM, int Nx, int Ny, int Nk, int Nl)
{
    if (!dconv_ok(Nk, Nl, B)) return 0;
    size_t x = 0;
    if (Nk == 3) x = std::max(wcorr_part_floats<3>(dM, dD, Nx, B), wcorr_part_floats<3>(dD, dM, Nx, B));
    else if (Nk == 5) x = std::max(wcorr_part_floats<5>(dM, dD, Nx, B), wcorr_part_floats<5>(dD, dM, Nx, B));
    else x = std::max(wcorr_part_floats<7>(dM, dD, Nx, B), wcorr_part_floats<7>(dD, dM, Nx, B));
    x = std::max(x, (size_t)B * ((Nx + 7) / 8) * ((Ny + 255) / 256) * (16 * (9 * 25 + 3)));      // rcorr_kernel's workgroup partials (rc_pw<3>)
    return std::max(x, mcorr_part_floats(Nx, B));
}
template <int NK> static hipError_t launch_wcorr(const WCorrArgs& a, int B, float* out, float* osum, float scale, hipStream_t st)
{
    return run_wcorr<WTile<NK>::TA, WTile<NK>::TB, NK>(a, B, out, osum, scale, st);
}

hipError_t launch_conv_spatial(const float* in, float* out, const float* c, const float* b, int B, int dD, int dM,
                               int Nx, int Ny, int Nk, int Nl, int ak, int al, float in_scale_div, int lo, hipStream_t st,
                               int pool, float* pooled_out)
{
    const long total = (long)B * dM * Nx * Ny;
    if (total <= 0) return hipSuccess;
    if (dconv_ok(Nk, Nl, B) && !flag(AEFFT_F_NOTILEDSPATIAL)) {
        DConvArgs a{};
        a.in = in; a.w = c; a.bias = b; a.out = out;
        a.Din = dD; a.M = dM; a.Nx = Nx; a.Ny = Ny; a.w_m = dD * Nk * Nl; a.w_d = Nk * Nl;
        a.ik0 = -2 * ak - 1; a.il0 = -2 * al - 1; a.sgn = -1; a.lo_in = lo; a.hi_lo = 0; a.div = in_scale_div;
        a.pool = pool; a.pooled_out = pooled_out;
        return launch_dconv(a, Nk, B, st);
    }
    if (pool) return hipErrorInvalidValue;                             // (callers pool separately for shapes the tiled kernels do not serve)
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

// Pool (netlib.cpp:114-164) on the device.  scale > 0: window maximum through the reference's `int smax = 0` accumulator, i.e.
// max(0, trunc(max of the window)) (the running truncation of :127-136 is order-independent: an element only replaces smax
// when it exceeds it, and the window maximum always ends it at trunc(max)); scale < 0: nearest-neighbour up-sampling.
__global__ __launch_bounds__(256) void pool_spatial_kernel(const float* __restrict__ in, float* __restrict__ out, long planes,
                                                           int Nxi, int Nyi, int Nxo, int Nyo, int scale)
{
    const long total = planes * Nxo * Nyo;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= total) return;
    const int jo = (int)(n % Nyo), io = (int)((n / Nyo) % Nxo);
    const long d = n / ((long)Nxo * Nyo);
    const float* src = in + d * Nxi * Nyi;
    if (scale > 0) {
        const int i0 = io * scale, j0 = jo * scale;
        if (i0 >= Nxi || j0 >= Nyi) return;                 // the reference loop never reaches these outputs: left untouched
        float mx = 0.f;
        for (int k = 0; k < scale; ++k)
            for (int l = 0; l < scale; ++l)
                if (i0 + k < Nxi && j0 + l < Nyi) mx = fmaxf(mx, src[(long)(i0 + k) * Nyi + j0 + l]);
        out[n] = (float)(int)mx;
    } else {
        const int s = -scale;
        if (io / s >= Nxi || jo / s >= Nyi) return;
        out[n] = src[(long)(io / s) * Nyi + jo / s];
    }
}

hipError_t launch_pool_spatial(const float* in, float* out, long planes, int Nxi, int Nyi, int Nxo, int Nyo, int scale, hipStream_t st)
{
    const long total = planes * Nxo * Nyo;
    if (total <= 0 || scale == 0) return scale == 0 ? hipErrorInvalidValue : hipSuccess;
    pool_spatial_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(in, out, planes, Nxi, Nyi, Nxo, Nyo, scale);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// SURVEY Appendix B-11 compat mode: the decoder-kernel and encoder-bias gradients exactly as the reference's CUDA source
// computes them (the literal per-element restatement the parity tests check against), overwriting what the kernels above produced:
//   gf : gradient_CFBP reads the hidden layer at flat index (i-ik)*Nx + (j-il), gradient_CF at (i-ik)*Nx + (j-ik)
//        (backproplib.cu:226,283: row stride Nx, column shifted by ik); an index outside the hin buffer is undefined
//        behaviour there and reads 0 here; pixels whose shifted position is out of range keep the value of the PREVIOUS
//        launch (the per-pixel buffer is written only inside the range test, :225,282; launches run in (m, d, k, l) order,
//        the buffer starts zeroed per frame, :335).
//   gb : `dDdB2 = ...` (:220) -- only the last d1 contributes.
// Thread = one pixel of one frame walking all dM*dD*Nk*Nl launches in order with its buffer value in a register;
// per launch the block sum goes to gf with one atomic per wave.  A reference-compat path, not a fast one.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spatial_compat_kernel(const SpatialGradArgs a)
{
    const long plane = (long)a.Nx * a.Ny;
    const long px = (long)blockIdx.x * 256 + threadIdx.x;
    const long bb = blockIdx.y;
    const bool ok = px < plane;
    const int i = ok ? (int)(px / a.Ny) : 0, j = ok ? (int)(px - (long)i * a.Ny) : 0;
    const float* in = a.in + bb * a.dD * plane;
    const float* out = a.out + bb * a.dD * plane;
    const float* hin = a.hin + bb * a.dM * plane;
    const long hsize = (long)a.dM * plane;
    const float sc = 1.0f / a.Norm / (float)a.B;
    float buf = 0.f;
    for (int m = 0; m < a.dM; ++m)
        for (int d = 0; d < a.dD; ++d) {
            const float s0 = ok ? out[d * plane + px] - in[d * plane + px] : 0.f;
            for (int k = 0; k < a.Nk; ++k) {
                const int ik = -2 * a.ak - 1 + k;
                for (int l = 0; l < a.Nl; ++l) {
                    const int il = -2 * a.al - 1 + l;
                    if (ok && i - ik >= 0 && i - ik < a.Nx && j - il >= 0 && j - il < a.Ny) {
                        const long idx = (long)m * plane + (long)(i - ik) * a.Nx + ((k == 0 && l == 0) ? j - il : j - ik);
                        buf = (idx >= 0 && idx < hsize) ? s0 * hin[idx] : 0.f;
                    }
                    float v = buf;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                    if ((threadIdx.x & 63) == 0 && v != 0.f) atomicAdd(&a.gf[((d * a.dM + m) * a.Nk + k) * a.Nl + l], v * sc);
                }
            }
        }
    // gb[m] = sum_pixels s0[dD-1] * sum_{k1,l1 in range} f[dD-1][m][k1][l1]
    const int dl = a.dD - 1;
    const float s0 = ok ? out[dl * plane + px] - in[dl * plane + px] : 0.f;
    for (int m = 0; m < a.dM; ++m) {
        float w = 0.f;
        for (int k1 = 0; k1 < a.Nk; ++k1) {
            const int ik1 = -2 * a.ak - 1 + k1;
            if (i - ik1 < 0 || i - ik1 >= a.Nx) continue;
            for (int l1 = 0; l1 < a.Nl; ++l1) {
                const int il1 = -2 * a.al - 1 + l1;
                if (j - il1 >= 0 && j - il1 < a.Ny) w += a.f[((dl * a.dM + m) * a.Nk + k1) * a.Nl + l1];
            }
        }
        float v = s0 * w;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v != 0.f) atomicAdd(&a.gb[m], v * sc);
    }
}

hipError_t launch_spatial_compat(const SpatialGradArgs& a, hipStream_t st)
{
    const size_t nk = (size_t)a.dM * a.dD * a.Nk * a.Nl;
    hipError_t e = hipMemsetAsync(a.gf, 0, nk * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(a.gb, 0, a.dM * sizeof(float), st);
    if (e != hipSuccess) return e;
    const long plane = (long)a.Nx * a.Ny;
    spatial_compat_kernel<<<dim3((unsigned)((plane + 255) / 256), a.B), 256, 0, st>>>(a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// dC without the back-convolved error (3x3 supports, dD <= 3 input channels: the reference's default first layer).
// With s0 = out - in:  g[m][q] = sum_{d1,t1} f[d1][m][t1] s0[d1][q + t1]  and  dC[m][d][t] = sum_q g[m][q] in[d][q - t];  summed over the
// error pixel p = q + t1 instead,
//     dC[m][d][t] = sum_{d1,t1} f[d1][m][t1] R_t1[d1][d][t1 + t],      R_t1[d1][d][u] = sum_{p : p - t1 is a hidden pixel} s0[d1][p] in[d][p - u]
// -- the correlation of the dD error planes with the dD input planes at the (2Nk-1)^2 composite offsets u: dD*dD*25 numbers per mask.
// "p - t1 is a hidden pixel" only removes border rows / columns of the error (rows 0, 1, Nx-1 at most, for |t1| <= 1 and lo <= 1), so
// the kernel accumulates the correlation per REGION -- row classes {0, 1, Nx-1, the rest} x column sets {all, 0, 1, Ny-1} -- and the
// contraction kernel puts each tap's R together from the regions its mask admits.  Neither the dM-plane error g (419 MB written by a
// back-convolution launch and read back by a 50-row correlation at 32 x 256^2 x 50 maps) nor s0 itself is stored: the step reads `out`
// and `in` once.  dB[m] = sum_q g[m][q] comes from the regions' sums of s0.
//
// rcorr_kernel: workgroup = 8 image rows x 256 columns of one frame; the error tile and the input tile (halo 4) in LDS.  Thread =
// (input plane d, row offset a) x segment (row, half): 5 column offsets x dD error planes = 15 running sums, one new input element and
// dD error elements (broadcast among the lanes of a segment) per pixel.
// ------------------------------------------------------------------------------------------
static bool rcorr_ok(const SpatialGradArgs& a);
struct RCorrArgs { const float *out, *in; float* part; int B, Nx, Ny, ik0, il0, lo; };
constexpr int RC_TR = 8, RC_CW = 256, RC_T = 5, RC_NSEG = 16;
template <int D> constexpr int rc_pw() { return 16 * (D * D * RC_T * RC_T + D); }      // floats of one workgroup's partial: [rc 4][cs 4][d1][d][a][b] | [rc][cs][d1]

template <int D>
__global__ __launch_bounds__(256) void rcorr_kernel(const RCorrArgs g)
{
    constexpr int TR = RC_TR, CW = RC_CW, T = RC_T, XR = TR + T - 1, XC = CW + T - 1, NC = D * T, NV = 4 * D * T + 4 * D;
    extern __shared__ float rc_sh[];
    float* s0t = rc_sh;                                   // [D][TR][CW]
    float* xt = rc_sh + D * TR * CW;                      // [D][XR][XC]
    const int i0 = blockIdx.x * TR, j0 = blockIdx.y * CW;
    const long bb = blockIdx.z, plane = (long)g.Nx * g.Ny;
    // stage: s0 = out - in (zero outside the image), in shifted: tile (r, c) <-> image (i0 - 2 ik0 - 4 + r, j0 - 2 il0 - 4 + c), zero outside [lo, N).
    // Every global load of the workgroup is issued before the first LDS store (clamped addresses, masks applied on the way to LDS): a
    // rolled load -> store loop is one memory round trip per iteration.
    {
        constexpr int NQ = D * TR * (CW / 4) / 256;                      // float4s of the error tile per thread (6 for D = 3)
        static_assert(D * TR * (CW / 4) % 256 == 0, "error tile: whole float4s per thread");
        constexpr int NX = (D * XR * XC + 255) / 256;                    // elements of the input tile per thread (37 for D = 3)
        float4 vo[NQ], vi[NQ];
        float vx[NX];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int t = u * 256 + threadIdx.x;
            const int d = t / (TR * (CW / 4)), rem = t - d * TR * (CW / 4), r = rem / (CW / 4), c = (rem - r * (CW / 4)) * 4;
            const int i = min(i0 + r, g.Nx - 1), j = min(j0 + c, g.Ny - 4);
            const long idx = (bb * D + d) * plane + (long)i * g.Ny + j;
            vo[u] = *reinterpret_cast<const float4*>(g.out + idx);
            vi[u] = *reinterpret_cast<const float4*>(g.in + idx);
        }
        const int xi0 = i0 - 2 * g.ik0 - (T - 1), xj0 = j0 - 2 * g.il0 - (T - 1);
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int t = min(u * 256 + (int)threadIdx.x, D * XR * XC - 1);
            const int d = t / (XR * XC), rem = t - d * XR * XC, r = rem / XC, c = rem - r * XC;
            const int i = min(max(xi0 + r, 0), g.Nx - 1), j = min(max(xj0 + c, 0), g.Ny - 1);
            vx[u] = g.in[(bb * D + d) * plane + (long)i * g.Ny + j];
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int t = u * 256 + threadIdx.x;
            const int d = t / (TR * (CW / 4)), rem = t - d * TR * (CW / 4), r = rem / (CW / 4), c = (rem - r * (CW / 4)) * 4;
            const bool ok = i0 + r < g.Nx && j0 + c < g.Ny;               // (Ny is a multiple of 4: a quad is inside or outside as a whole)
            const float4 v = ok ? make_float4(vo[u].x - vi[u].x, vo[u].y - vi[u].y, vo[u].z - vi[u].z, vo[u].w - vi[u].w) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(s0t + (d * TR + r) * CW + c) = v;
        }
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int t = u * 256 + threadIdx.x;
            if (t < D * XR * XC) {
                const int d = t / (XR * XC), rem = t - d * XR * XC, r = rem / XC, c = rem - r * XC;
                const int i = xi0 + r, j = xj0 + c;
                xt[t] = (i >= g.lo && i < g.Nx && j >= g.lo && j < g.Ny) ? vx[u] : 0.f;
            }
        }
    }
    __syncthreads();
    const int seg = threadIdx.x / NC, combo = threadIdx.x - seg * NC;
    const int d = combo / T, a = combo - d * T;
    const int r = seg >> 1, jb = (seg & 1) * (CW / 2);
    float acc[4][D][T];                                   // column sets: all, column 0, column 1, column Ny-1
    float ss[4][D];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs)
#pragma unroll
        for (int d1 = 0; d1 < D; ++d1) {
            ss[cs][d1] = 0.f;
#pragma unroll
            for (int b = 0; b < T; ++b) acc[cs][d1][b] = 0.f;
        }
    const bool active = seg < RC_NSEG && i0 + r < g.Nx;
    if (active) {
        // x at image (i - u_i, j - u_j), u = 2 ik0 + (a, b): tile row r + (T-1-a), tile column jj + (T-1-b)
        const float* xrow = xt + (d * XR + r + (T - 1 - a)) * XC;
        const float* srow = s0t + r * CW;
        float w[T];
#pragma unroll
        for (int b = 1; b < T; ++b) w[b] = xrow[jb + (T - 1 - b)];
        const int jend = min(jb + CW / 2, g.Ny - j0);
        // (the next pixel's LDS reads are issued before this pixel's FMAs: two workgroups per CU leave little else to hide their latency)
        float wn = xrow[jb + T - 1], svn[D];
#pragma unroll
        for (int d1 = 0; d1 < D; ++d1) svn[d1] = srow[d1 * TR * CW + jb];
        for (int jj = jb; jj < jend; ++jj) {
            w[0] = wn;
            float sv[D];
#pragma unroll
            for (int d1 = 0; d1 < D; ++d1) sv[d1] = svn[d1];
            const int jn = min(jj + 1, jb + CW / 2 - 1);
            wn = xrow[jn + T - 1];
#pragma unroll
            for (int d1 = 0; d1 < D; ++d1) svn[d1] = srow[d1 * TR * CW + jn];
#pragma unroll
            for (int d1 = 0; d1 < D; ++d1) {
#pragma unroll
                for (int b = 0; b < T; ++b) acc[0][d1][b] = fmaf(sv[d1], w[b], acc[0][d1][b]);
                ss[0][d1] += sv[d1];
            }
            const int gj = j0 + jj;
            if (gj <= 1 || gj == g.Ny - 1) {
                const int cs = gj == 0 ? 1 : (gj == 1 ? 2 : 3);
#pragma unroll
                for (int c2 = 1; c2 < 4; ++c2)
                    if (c2 == cs) {
#pragma unroll
                        for (int d1 = 0; d1 < D; ++d1) {
#pragma unroll
                            for (int b = 0; b < T; ++b) acc[c2][d1][b] = fmaf(sv[d1], w[b], acc[c2][d1][b]);
                            ss[c2][d1] += sv[d1];
                        }
                    }
            }
#pragma unroll
            for (int b = T - 1; b > 0; --b) w[b] = w[b - 1];
        }
    }
    __syncthreads();
    // per-thread sums -> LDS [seg][combo][NV], then the workgroup's partial per row class, segments added in order
    float* red = rc_sh;
    if (seg < RC_NSEG) {
        float* q = red + (seg * NC + combo) * NV;
#pragma unroll
        for (int cs = 0; cs < 4; ++cs)
#pragma unroll
            for (int d1 = 0; d1 < D; ++d1) {
#pragma unroll
                for (int b = 0; b < T; ++b) q[(cs * D + d1) * T + b] = acc[cs][d1][b];
                q[4 * D * T + cs * D + d1] = ss[cs][d1];
            }
    }
    __syncthreads();
    constexpr int NR = D * D * T * T, PW = rc_pw<D>();
    float* dst = g.part + ((bb * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y) * (long)PW;
    for (int e = threadIdx.x; e < PW; e += 256) {
        int rc, cs, d1, cmb, off;
        if (e < 16 * NR) {
            rc = e / (4 * NR); int rem = e - rc * 4 * NR;
            cs = rem / NR; rem -= cs * NR;
            d1 = rem / (D * T * T); rem -= d1 * D * T * T;
            const int dd = rem / (T * T); rem -= dd * T * T;
            const int aa = rem / T, b = rem - aa * T;
            cmb = dd * T + aa; off = (cs * D + d1) * T + b;
        } else {
            const int e2 = e - 16 * NR;
            rc = e2 / (4 * D); const int rem = e2 - rc * 4 * D;
            cs = rem / D; d1 = rem - cs * D;
            cmb = 0; off = 4 * D * T + cs * D + d1;                       // (every combo of a segment holds the same s0 sums)
        }
        float v = 0.f;
        for (int sg = 0; sg < RC_NSEG; ++sg) {
            const int i = i0 + (sg >> 1);
            const int cls = i == 0 ? 0 : (i == 1 ? 1 : (i == g.Nx - 1 ? 2 : 3));
            if (cls == rc && i < g.Nx) v += red[(sg * NC + cmb) * NV + off];
        }
        dst[e] = v;
    }
}

// the workgroups' partials summed in RC_NCH chunks (thread = element, consecutive threads read consecutive floats of one partial; the
// partials of a chunk are added in order): tmp[chunk][pw].  dc_from_regions_kernel adds the chunks.
constexpr int RC_NCH = 16;
__global__ __launch_bounds__(256) void rcorr_sum_kernel(const float* __restrict__ part, float* __restrict__ tmp, int nparts, int pw)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= pw) return;
    const int per = (nparts + RC_NCH - 1) / RC_NCH;
    const int p0 = blockIdx.y * per, p1 = min(nparts, p0 + per);
    float s = 0.f;
    int p = p0;
    for (; p + 8 <= p1; p += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long)(p + u) * pw + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < p1; ++p) s += part[(long)p * pw + e];
    tmp[(long)blockIdx.y * pw + e] = s;
}

// dC, dB from the region sums:  R_t1 = sum over the row classes whose row r has lo <= r - ik1 < Nx of (all columns - the columns c in {0, 1, Ny-1}
// with c - il1 outside [lo, Ny));  gc[m][d][k][l] = sum_{d1,k1,l1} f[d1][m][k1][l1] R_(k1,l1)[d1][d][k1+k][l1+l],  gb[m] likewise from the s0 sums
// With c1 (the fused step: the hidden layer is Conv_gpu(in; c1, b1) itself, hin[m][q] = sum_{d2,t2} c1[m][d2][t2] in[d2][q - t2] / div1 + b1[m]):
//     gf[d][m][t] = sum_p s0[d][p] hin[m][p - t]   (p - t a hidden pixel)   = sum_{d2,t2} c1[m][d2][t2] / div1 * R_t[d][d2][t + t2] + b1[m] S_t[d]
// -- the SAME masked correlations R_t (mask: p - t is a hidden pixel; composite offset t + t2) contracted with c1 instead of f, S_t the masked
// s0 sums -- and gp[d] = sum of s0[d] over the whole image (backproplib.cu:271-288 re-associated): the 50-plane hidden layer is not read.
template <int D>
__global__ __launch_bounds__(256) void dc_from_regions_kernel(const float* __restrict__ f, const float* __restrict__ tmp, float* __restrict__ gc,
                                                              float* __restrict__ gb, int dM, int Nx, int Ny, int ik0, int il0, int lo, float scale,
                                                              const float* __restrict__ c1, const float* __restrict__ b1, float inv_div1,
                                                              float* __restrict__ gf, float* __restrict__ gp)
{
    constexpr int NK = 3, KK = 9, T = RC_T, NR = D * D * T * T, PW = rc_pw<D>();
    __shared__ float reg[PW];
    for (int e0 = 0; e0 < PW; e0 += 256 * 2) {
        float v[2][RC_NCH];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < RC_NCH; ++c) v[u][c] = tmp[(long)c * PW + min(e0 + u * 256 + (int)threadIdx.x, PW - 1)];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float sacc = 0.f;
#pragma unroll
            for (int c = 0; c < RC_NCH; ++c) sacc += v[u][c];
            const int e = e0 + u * 256 + threadIdx.x;
            if (e < PW) reg[e] = sacc * scale;
        }
    }
    __syncthreads();
    // R of every tap t1 = (k1, l1) from the regions its mask admits: Rt[t1][d1][d][u] and the masked s0 sums St[t1][d1]
    constexpr int NT = KK * (NR + D);
    __shared__ float Rt[NT];
    const int rowpos[3] = {0, 1, Nx - 1}, colpos[3] = {0, 1, Ny - 1};
    for (int e = threadIdx.x; e < NT; e += 256) {
        const int t1 = e / (NR + D), r2 = e - t1 * (NR + D);
        const int k1 = t1 / NK, l1 = t1 - k1 * NK, ik1 = ik0 + k1, il1 = il0 + l1;
        float R = 0.f;
        for (int rc = 0; rc < 4; ++rc) {
            if (rc < 3 && !(rowpos[rc] - ik1 >= lo && rowpos[rc] - ik1 < Nx)) continue;      // (the rows of class 3 pass for every tap)
            auto at = [&](int cs) { return r2 < NR ? reg[(rc * 4 + cs) * NR + r2] : reg[16 * NR + (rc * 4 + cs) * D + (r2 - NR)]; };
            float v = at(0);
            for (int c = 0; c < 3; ++c) if (!(colpos[c] - il1 >= lo && colpos[c] - il1 < Ny)) v -= at(1 + c);
            R += v;
        }
        Rt[e] = R;
    }
    __syncthreads();
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int nw = dM * D * KK;
    if (idx >= nw + dM) {
        if (!c1) return;
        const int e = idx - (nw + dM);
        if (e < nw) {
            // gf[d][m][k][l]
            const int d = e / (dM * KK), rem = e - d * dM * KK, m = rem / KK, kl = rem - m * KK, k = kl / NK, l = kl - k * NK;
            const float* Rb = Rt + kl * (NR + D);
            float s = 0.f;
#pragma unroll
            for (int d2 = 0; d2 < D; ++d2)
#pragma unroll
                for (int k2 = 0; k2 < NK; ++k2)
#pragma unroll
                    for (int l2 = 0; l2 < NK; ++l2)
                        s = fmaf(c1[((long)(m * D + d2) * NK + k2) * NK + l2], Rb[(d * D + d2) * T * T + (k + k2) * T + (l + l2)], s);
            gf[e] = fmaf(b1[m], Rb[NR + d], s * inv_div1);
        } else if (e < nw + D) {
            // gp[d]: s0 summed over the whole image = every row class, all columns
            const int d = e - nw;
            float s = 0.f;
            for (int rc = 0; rc < 4; ++rc) s += reg[16 * NR + (rc * 4) * D + d];
            gp[d] = s;
        }
        return;
    }
    const bool isb = idx >= nw;
    const int m = isb ? idx - nw : idx / (D * KK);
    const int rr = isb ? 0 : idx - m * D * KK, d = rr / KK, kl = rr - d * KK, k = kl / NK, l = kl - k * NK;
    float s = 0.f;
#pragma unroll
    for (int d1 = 0; d1 < D; ++d1)
#pragma unroll
        for (int k1 = 0; k1 < NK; ++k1)
#pragma unroll
            for (int l1 = 0; l1 < NK; ++l1) {
                const float* Rb = Rt + (k1 * NK + l1) * (NR + D);
                const float R = isb ? Rb[NR + d1] : Rb[(d1 * D + d) * T * T + (k1 + k) * T + (l1 + l)];
                s = fmaf(f[((long)(d1 * dM + m) * NK + k1) * NK + l1], R, s);
            }
    if (isb) gb[m] = s; else gc[idx] = s;
}

// floats of SpatialGradArgs::rq (the region sums)
size_t spatial_rq_floats(int dD, int Nk, int Nl) { (void)Nk; (void)Nl; return (size_t)RC_NCH * 16 * ((size_t)dD * dD * RC_T * RC_T + dD); }

bool spatial_regions_ok(const SpatialGradArgs& a) { return a.part && dconv_ok(a.Nk, a.Nl, a.B) && !flag(AEFFT_F_NOTILEDSPATIAL) && rcorr_ok(a); }
static bool rcorr_ok(const SpatialGradArgs& a)
{
    return a.rq && a.Nk == 3 && a.Nl == 3 && (a.dD == 3 || a.dD == 1) && a.dM >= 8 && a.Nx >= 8 && a.Ny >= 8 && a.Ny % 4 == 0 && a.lo <= 1 && a.ak == 0 && a.al == 0 &&
           !flag(AEFFT_F_NOMFMA) && !flag(AEFFT_F_NORCORR);
}

template <int D> static hipError_t run_rcorr(const SpatialGradArgs& a, float scale, hipStream_t st)
{
    const int ik0 = -2 * a.ak - 1, il0 = -2 * a.al - 1;
    const dim3 grid((a.Nx + RC_TR - 1) / RC_TR, (a.Ny + RC_CW - 1) / RC_CW, a.B);
    const size_t lds = sizeof(float) * std::max<size_t>((size_t)D * RC_TR * RC_CW + (size_t)D * (RC_TR + RC_T - 1) * (RC_CW + RC_T - 1),
                                                        (size_t)RC_NSEG * D * RC_T * (4 * D * RC_T + 4 * D));
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rcorr_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    RCorrArgs g{a.out, a.in, a.part, a.B, a.Nx, a.Ny, ik0, il0, a.lo};
    rcorr_kernel<D><<<grid, 256, lds, st>>>(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int nparts = (int)(grid.x * grid.y * grid.z), pw = rc_pw<D>();
    rcorr_sum_kernel<<<dim3((pw + 255) / 256, RC_NCH), 256, 0, st>>>(a.part, a.rq, nparts, pw);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int total = a.dM * D * 9 + a.dM + (a.c1 ? a.dM * D * 9 + D : 0);
    dc_from_regions_kernel<D><<<dim3((total + 255) / 256), 256, 0, st>>>(a.f, a.rq, a.gc, a.gb, a.dM, a.Nx, a.Ny, ik0, il0, a.lo, scale,
                                                                         a.c1, a.b1, 1.0f / a.div1, a.gf, a.gp);
    return hipGetLastError();
}

hipError_t launch_spatial_grad(const SpatialGradArgs& a, hipStream_t st)
{
    if (a.part && dconv_ok(a.Nk, a.Nl, a.B) && !flag(AEFFT_F_NOTILEDSPATIAL) && rcorr_ok(a)) {
        // dC, dB through the error-input correlation (no back-convolved error); dF, dP as the hidden-layer correlation on the matrix cores
        const float scale = 1.0f / a.Norm / (float)a.B;
        hipError_t e = a.dD == 3 ? run_rcorr<3>(a, scale, st) : run_rcorr<1>(a, scale, st);
        if (e != hipSuccess || a.c1) return e;             // (fused step: dF, dP came out of the region sums too)
        MCorrArgs mf{a.hin, a.dM, a.lo, a.out, a.in, a.dD, 0, +1, a.Nx, a.Ny, -2 * a.ak - 1, -2 * a.al - 1, 1, 0, 0, 0, a.part};
        return run_mcorr<3>(mf, a.B, a.gf, 1, nullptr, a.gp, scale, st);
    }
    if (a.part && dconv_ok(a.Nk, a.Nl, a.B) && !flag(AEFFT_F_NOTILEDSPATIAL)) {
        // g = back-convolution of s0 = out - in through f (zero for i' < lo or j' < lo)
        DConvArgs g{};
        g.in = a.out; g.in2 = a.in; g.w = a.f; g.bias = nullptr; g.out = a.ws;
        g.Din = a.dD; g.M = a.dM; g.Nx = a.Nx; g.Ny = a.Ny; g.w_m = a.Nk * a.Nl; g.w_d = a.dM * a.Nk * a.Nl;
        g.ik0 = -2 * a.ak - 1; g.il0 = -2 * a.al - 1; g.sgn = +1; g.lo_in = 0; g.hi_lo = a.lo; g.div = 1.f;
        hipError_t e = launch_dconv(g, a.Nk, a.B, st);
        if (e != hipSuccess) return e;
        const float scale = 1.0f / a.Norm / (float)a.B;
        if (a.dM >= 8 && a.Ny % 4 == 0 && !flag(AEFFT_F_NOMFMA)) {
            // both correlations as GEMMs on the matrix cores, rows = the dM planes (mcorr_kernel):
            //   dC[m][d][k][l] = sum g[m][i][j] in[d][i-ik][j-il]                  rows g, columns in shifted by -tap, + ones column (dB)
            //   dF[d][m][k][l] = sum s0[d][i][j] hin[m][i-ik][j-il]                 re-indexed by i' = i - ik: rows hin (masked below lo),
            //                  = sum hin[m][i'][j'] s0[d][i'+ik][j'+il]             columns s0 shifted by +tap, + ones row (dP at the zero tap)
            MCorrArgs mc{a.ws, a.dM, 0, a.in, nullptr, a.dD, a.lo, -1, a.Nx, a.Ny, -2 * a.ak - 1, -2 * a.al - 1, 0, 1, 0, 0, a.part};
            MCorrArgs mf{a.hin, a.dM, a.lo, a.out, a.in, a.dD, 0, +1, a.Nx, a.Ny, -2 * a.ak - 1, -2 * a.al - 1, 1, 0, 0, 0, a.part};
            if (a.Nk == 3) { e = run_mcorr<3>(mc, a.B, a.gc, 0, a.gb, nullptr, scale, st); if (e == hipSuccess) e = run_mcorr<3>(mf, a.B, a.gf, 1, nullptr, a.gp, scale, st); }
            else if (a.Nk == 5) { e = run_mcorr<5>(mc, a.B, a.gc, 0, a.gb, nullptr, scale, st); if (e == hipSuccess) e = run_mcorr<5>(mf, a.B, a.gf, 1, nullptr, a.gp, scale, st); }
            else { e = run_mcorr<7>(mc, a.B, a.gc, 0, a.gb, nullptr, scale, st); if (e == hipSuccess) e = run_mcorr<7>(mf, a.B, a.gf, 1, nullptr, a.gp, scale, st); }
            return e;
        }
        WCorrArgs wc{a.ws, nullptr, a.in, a.part, a.dM, a.dD, a.Nx, a.Ny, -2 * a.ak - 1, -2 * a.al - 1, a.lo};       // dC, dB
        WCorrArgs wf{a.out, a.in, a.hin, a.part, a.dD, a.dM, a.Nx, a.Ny, -2 * a.ak - 1, -2 * a.al - 1, a.lo};        // dF, dP
        if (a.Nk == 3) { e = launch_wcorr<3>(wc, a.B, a.gc, a.gb, scale, st); if (e == hipSuccess) e = launch_wcorr<3>(wf, a.B, a.gf, a.gp, scale, st); }
        else if (a.Nk == 5) { e = launch_wcorr<5>(wc, a.B, a.gc, a.gb, scale, st); if (e == hipSuccess) e = launch_wcorr<5>(wf, a.B, a.gf, a.gp, scale, st); }
        else { e = launch_wcorr<7>(wc, a.B, a.gc, a.gb, scale, st); if (e == hipSuccess) e = launch_wcorr<7>(wf, a.B, a.gf, a.gp, scale, st); }
        return e;
    }
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
