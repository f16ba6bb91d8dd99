// Coordinate-space side of the weight update for gfx950:
//   pad / shrink of the centred-wrapped Nk x Nl kernel support  (fft_backproplib.cu:535-600)
//   clipped-gradient + momentum update                          (fft_backproplib.cu:605-652, 657-704)
//   kernel-distance gradient of the multiobjective mode         (fft_backproplib.cu:709-753)
// These tensors are tiny (dM*dD*Nk*Nl floats); the kernels are launch-latency bound.
#include "internal.h"
#include "update_device.h"
#include <algorithm>

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

// out = a + b (the weights of the step before the last update: w_old = w + D, the momentum buffer holds the step that was applied)
__global__ __launch_bounds__(256) void vec_add_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
__global__ __launch_bounds__(256) void u8_to_f32_kernel(float* __restrict__ out, const unsigned char* __restrict__ in, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
hipError_t launch_u8_to_f32(float* out, const unsigned char* in, long n, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    u8_to_f32_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(out, in, n);
    return hipGetLastError();
}
hipError_t launch_vec_add(float* out, const float* a, const float* b, long n, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    vec_add_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(out, a, b, n);
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
// gradient_diff (fft_backproplib.cu:709-753):
//   cd[m][d][r] = sum_{m1 != m, d1 != d} (c[m][d][r] - c[m1][d1][r]) / |c[m][d] - c[m1][d1]|^2      (same for f, b, p)
// The reference recomputes the squared kernel distance inside every (k,l) thread and divides per tap: O((dM dD)^2 (Nk Nl)^2).
// Here a thread owns one kernel a = (m,d) -- its KL taps and KL running sums in registers -- and walks a chunk of partners whose
// taps every thread of the workgroup reads from LDS at the same address (broadcast): per partner KL subtractions, KL FMAs for the
// distance, ONE reciprocal, KL FMAs for the sums.  The partner range is cut into chunks (grid y) so that the small tensors still
// fill the chip; the chunks' partial sums lie side by side and gdiff_finish_kernel adds them in chunk order (deterministic: the
// replicas of a data-parallel run must stay bit-identical).  No (dM dD)^2 distance matrix is stored (537 MB at cfg5's 64->128 pair).
// ------------------------------------------------------------------------------------------
#ifndef AEFFT_X_GDEXP
#define AEFFT_X_GDEXP 1
#endif
#ifndef AEFFT_X_GDKPT
#define AEFFT_X_GDKPT 1
#endif
#ifndef AEFFT_X_GDUNR
#define AEFFT_X_GDUNR 2
#endif
#define GD_UNROLL AEFFT_X_GDUNR
// kernels per thread of the partner loop (workgroup = 256 * kpt rows).  Two, sharing each partner row's LDS reads (148 registers at 5x5, 3 waves per SIMD),
// measured 254 (276-285 with two partners per trip) against 244 us for one kernel per thread and two partners per trip at cfg5: one.
constexpr int gd_kpt(int kl) { return kl <= 25 ? AEFFT_X_GDKPT : 1; }
constexpr int gd_rows(int kl) { return 256 * gd_kpt(kl); }
template <int KL>
__device__ __forceinline__ void gdiff_part_body(const float* __restrict__ c, const float* __restrict__ f, float* __restrict__ part,
                                                int dM, int dD, int chunk, int bx, int by, int bz, int nchunks)
{
    constexpr int KP = (KL + 3) & ~3;                      // LDS row pitch: whole float4s
    extern __shared__ float4 gd_sh4[];
    float* sh = reinterpret_cast<float*>(gd_sh4);
    const int np = dM * dD;
    const bool isf = bz != 0;                              // rows of f are (d,m)-ordered, rows of c (m,d)-ordered
    const float* __restrict__ w = isf ? f : c;
    static_assert(AEFFT_X_GDEXP || gd_kpt(KL) == 1, "the difference forms own one kernel per thread");
    const int i = bx * 256 + threadIdx.x;
    const int j0 = by * chunk, j1 = min(np, j0 + chunk);
    // partner taps -> LDS (coalesced), zero padding of the pitch
    for (int t = threadIdx.x; t < (j1 - j0) * KP; t += 256) {
        const int j = t / KP, r = t - j * KP;
        sh[t] = r < KL ? w[(long)(j0 + j) * KL + r] : 0.f;
    }
#ifndef AEFFT_X_GDPK
#define AEFFT_X_GDPK 1
#endif
#if AEFFT_X_GDEXP
    // |a - b|^2 = |a|^2 + |b|^2 - 2 a.b and  sum_j w_j (a - b_j) = a sum_j w_j - sum_j w_j b_j:  per partner KP/2 packed FMAs for the dot product, one
    // FMA, ONE reciprocal, KP/2 packed FMAs for the weighted partner sum -- two thirds of the instructions of the difference form below.  The partner
    // norms are formed once per chunk and kept in the last padding slot of each LDS row (this thread's padding taps are zero, so the slot drops out
    // of the dot product; the padding lanes of the sums are not stored).  A thread owns GD_KPT kernels (rows i, i + 256, ...: gd_kpt).  Rounding: the squared distance is exact
    // to a few ulp of |a|^2 + |b|^2 instead of of itself; identical kernels still give 1/0 and NaN sums like the reference's 0/0 (the norm and the dot
    // product are the same FMA chain).
    {
        static_assert(KP > KL, "a padding slot for the norm");
        constexpr int KH = KP / 2, GD_KPT = gd_kpt(KL);
        const int nj = j1 - j0;
        __syncthreads();
        for (int t = threadIdx.x; t < nj; t += 256) {
            const float4* row = reinterpret_cast<const float4*>(sh + t * KP);
            float2 n2 = make_float2(0.f, 0.f);
#pragma unroll
            for (int q = 0; q < KP / 4; ++q) { const float4 v = row[q]; n2 = n2 + make_float2(v.x, v.y) * make_float2(v.x, v.y); n2 = n2 + make_float2(v.z, v.w) * make_float2(v.z, v.w); }
            sh[t * KP + KP - 1] = n2.x + n2.y;
        }
        float2 ca2[GD_KPT][KH], acc2[GD_KPT][KH];
        float ni[GD_KPT], sw[GD_KPT];
        int i_f[GD_KPT], i_s[GD_KPT];
        const int fast_n = isf ? dM : dD;                                          // the index that runs fastest along a tensor's rows
#pragma unroll
        for (int u = 0; u < GD_KPT; ++u) {
            const int ii = min(bx * (256 * GD_KPT) + u * 256 + (int)threadIdx.x, np - 1);
            float2 ni2 = make_float2(0.f, 0.f);
#pragma unroll
            for (int h = 0; h < KH; ++h) {
                ca2[u][h] = make_float2(2 * h < KL ? w[(long)ii * KL + 2 * h] : 0.f, 2 * h + 1 < KL ? w[(long)ii * KL + 2 * h + 1] : 0.f);
                acc2[u][h] = make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int h = 0; h < KH; ++h) ni2 = ni2 + ca2[u][h] * ca2[u][h];
            ni[u] = ni2.x + ni2.y; sw[u] = 0.f;
            i_f[u] = ii % fast_n; i_s[u] = ii / fast_n;
        }
        __syncthreads();
        // the partner's (m, d) as running counters (uniform: a division per partner was ~50 scalar instructions of a ~90-instruction iteration); the
        // pair test as a select, not a branch around the LDS reads
        int jf = j0 % fast_n, js = j0 / fast_n;
#pragma unroll GD_UNROLL
        for (int j = j0; j < j1; ++j) {
            const float4* row = reinterpret_cast<const float4*>(sh + (j - j0) * KP);
            float2 b2[KH];
#pragma unroll
            for (int q = 0; q < KP / 4; ++q) { const float4 v = row[q]; b2[2 * q] = make_float2(v.x, v.y); b2[2 * q + 1] = make_float2(v.z, v.w); }
            const float nb = b2[KH - 1].y;                                         // |b|^2 (the row's last slot; ca2's is zero)
#pragma unroll
            for (int u = 0; u < GD_KPT; ++u) {
                float2 dot2 = make_float2(0.f, 0.f);
#pragma unroll
                for (int h = 0; h < KH; ++h) dot2 = dot2 + ca2[u][h] * b2[h];
                const float den = fmaf(-2.f, dot2.x + dot2.y, ni[u] + nb);
                // pairs need m1 != m AND d1 != d (:724)
                const float wgt = (jf != i_f[u] && js != i_s[u]) ? __builtin_amdgcn_rcpf(den) : 0.f;
                const float2 w2 = make_float2(wgt, wgt);
                sw[u] += wgt;
#pragma unroll
                for (int h = 0; h < KH; ++h) acc2[u][h] = acc2[u][h] + b2[h] * w2;
            }
            if (++jf == fast_n) { jf = 0; ++js; }
        }
#pragma unroll
        for (int u = 0; u < GD_KPT; ++u) {
            const int i = bx * (256 * GD_KPT) + u * 256 + (int)threadIdx.x;
            if (i >= np) continue;
            float* dst = part + (((long)bz * nchunks + by) * np + i) * KL;
#pragma unroll
            for (int h = 0; h < KH; ++h) {
                if (2 * h < KL) dst[2 * h] = fmaf(ca2[u][h].x, sw[u], -acc2[u][h].x);
                if (2 * h + 1 < KL) dst[2 * h + 1] = fmaf(ca2[u][h].y, sw[u], -acc2[u][h].y);
            }
        }
        return;
    }
#endif
#if AEFFT_X_GDPK
    // taps in PAIRS (native 2-vectors: the back end selects v_pk_add_f32 / v_pk_fma_f32, two taps per lane and issue slot): per partner
    // KP/2 packed subtractions, KP/2 packed FMAs for the distance, one reciprocal, KP/2 packed FMAs for the sums; the pitch's padding taps
    // are zero on both sides and drop out
    {
        constexpr int KH = KP / 2;
        float2 ca2[KH], acc2[KH];
        const int ii = min(i, np - 1);
#pragma unroll
        for (int h = 0; h < KH; ++h) {
            ca2[h] = make_float2(2 * h < KL ? w[(long)ii * KL + 2 * h] : 0.f, 2 * h + 1 < KL ? w[(long)ii * KL + 2 * h + 1] : 0.f);
            acc2[h] = make_float2(0.f, 0.f);
        }
        const int mi = isf ? ii % dM : ii / dD, di = isf ? ii / dM : ii % dD;
        __syncthreads();
        for (int j = j0; j < j1; ++j) {
            const int mj = isf ? j % dM : j / dD, dj = isf ? j / dM : j % dD;       // (uniform)
            const float4* row = reinterpret_cast<const float4*>(sh + (j - j0) * KP);
            float2 diff[KH];
            float2 den2 = make_float2(0.f, 0.f);
#pragma unroll
            for (int q = 0; q < KP / 4; ++q) {
                const float4 v = row[q];
                diff[2 * q] = ca2[2 * q] - make_float2(v.x, v.y);
                diff[2 * q + 1] = ca2[2 * q + 1] - make_float2(v.z, v.w);
            }
#pragma unroll
            for (int h = 0; h < KH; ++h) den2 = den2 + diff[h] * diff[h];
            const float den = den2.x + den2.y;
            // pairs need m1 != m AND d1 != d (:724); coinciding kernels give 0 * inf = NaN like the reference's 0 / 0
            if (mj != mi && dj != di) {
                const float wgt = __builtin_amdgcn_rcpf(den);
                const float2 w2 = make_float2(wgt, wgt);
#pragma unroll
                for (int h = 0; h < KH; ++h) acc2[h] = acc2[h] + diff[h] * w2;
            }
        }
        if (i >= np) return;
        float* dst = part + (((long)bz * nchunks + by) * np + i) * KL;
#pragma unroll
        for (int h = 0; h < KH; ++h) {
            if (2 * h < KL) dst[2 * h] = acc2[h].x;
            if (2 * h + 1 < KL) dst[2 * h + 1] = acc2[h].y;
        }
        return;
    }
#endif
    float ca[KL], acc[KL];
    const int ii = min(i, np - 1);
#pragma unroll
    for (int r = 0; r < KL; ++r) { ca[r] = w[(long)ii * KL + r]; acc[r] = 0.f; }
    const int mi = isf ? ii % dM : ii / dD, di = isf ? ii / dM : ii % dD;
    __syncthreads();
    for (int j = j0; j < j1; ++j) {
        const int mj = isf ? j % dM : j / dD, dj = isf ? j / dM : j % dD;       // (uniform)
        const float4* row = reinterpret_cast<const float4*>(sh + (j - j0) * KP);
        float diff[KP];
#pragma unroll
        for (int q = 0; q < KP / 4; ++q) { const float4 v = row[q]; diff[4 * q] = v.x; diff[4 * q + 1] = v.y; diff[4 * q + 2] = v.z; diff[4 * q + 3] = v.w; }
        float den = 0.f;
#pragma unroll
        for (int r = 0; r < KL; ++r) { diff[r] = ca[r] - diff[r]; den = fmaf(diff[r], diff[r], den); }
        // pairs need m1 != m AND d1 != d (:724); coinciding kernels give 0 * inf = NaN like the reference's 0 / 0
        const float wgt = (mj != mi && dj != di) ? __builtin_amdgcn_rcpf(den) : 0.f;
        if (mj != mi && dj != di) {
#pragma unroll
            for (int r = 0; r < KL; ++r) acc[r] = fmaf(diff[r], wgt, acc[r]);
        }
    }
    if (i >= np) return;
    float* dst = part + (((long)bz * nchunks + by) * np + i) * KL;
#pragma unroll
    for (int r = 0; r < KL; ++r) dst[r] = acc[r];
}

template <int KL>
__global__ __launch_bounds__(256) void gdiff_part_kernel(const float* __restrict__ c, const float* __restrict__ f, float* __restrict__ part,
                                                         int dM, int dD, int chunk)
{
    gdiff_part_body<KL>(c, f, part, dM, dD, chunk, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y);
}

// every pair's partial sums in one launch (the small pairs fill the chip beside the large one): workgroup -> (pair, tensor, chunk, row tile)
template <int KL>
__global__ __launch_bounds__(256) void gdiff_part_group_kernel(const GdiffGroup g)
{
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
    const GdiffProb& q = g.q[p];
    int blk = blockIdx.x - g.start[p];
    const int rt = (q.dM * q.dD + gd_rows(KL) - 1) / gd_rows(KL);
    const int bx = blk % rt; blk /= rt;
    const int by = blk % q.nchunks, bz = blk / q.nchunks;
    gdiff_part_body<KL>(q.c, q.f, q.part, q.dM, q.dD, q.chunk, bx, by, bz, q.nchunks);
}

// chunk partials -> cd | fd (each in its tensor's own layout), and the bias terms bd[m] = sum_{m1 != m} 1/(b[m]-b[m1]), pd likewise
__device__ __forceinline__ void gdiff_finish_body(const float* __restrict__ part, const float* __restrict__ b, const float* __restrict__ p,
                                                  float* __restrict__ cd, float* __restrict__ fd, float* __restrict__ bd, float* __restrict__ pd,
                                                  int dM, int dD, int kl, int nchunks, long blk)
{
    const long n = (long)dM * dD * kl;
    const long idx = blk * 256 + threadIdx.x;
    if (idx < 2 * n) {
        const int z = idx >= n;
        const long e = idx - z * n;
        float s = 0.f;
        for (int y = 0; y < nchunks; ++y) s += part[((long)z * nchunks + y) * n + e];
        (z ? fd : cd)[e] = s;
        return;
    }
    const long t = idx - 2 * n;
    if (t < dM) {
        float s = 0.f;
        for (int m1 = 0; m1 < dM; ++m1) if (m1 != (int)t) s += 1.f / (b[t] - b[m1]);
        bd[t] = s;
    } else if (t < dM + dD) {
        const int d = (int)t - dM;
        float s = 0.f;
        for (int d1 = 0; d1 < dD; ++d1) if (d1 != d) s += 1.f / (p[d] - p[d1]);
        pd[d] = s;
    }
}

__global__ __launch_bounds__(256) void gdiff_finish_kernel(const float* __restrict__ part, const float* __restrict__ b, const float* __restrict__ p,
                                                           float* __restrict__ cd, float* __restrict__ fd, float* __restrict__ bd, float* __restrict__ pd,
                                                           int dM, int dD, int kl, int nchunks)
{
    gdiff_finish_body(part, b, p, cd, fd, bd, pd, dM, dD, kl, nchunks, blockIdx.x);
}

__global__ __launch_bounds__(256) void gdiff_finish_group_kernel(const GdiffGroup g, int kl)
{
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.fstart[i]) p = i;
    const GdiffProb& q = g.q[p];
    gdiff_finish_body(q.part, q.b, q.p, q.cd, q.fd, q.bd, q.pd, q.dM, q.dD, kl, q.nchunks, (long)blockIdx.x - g.fstart[p]);
}

// Other supports (non-square, or not 3x3 / 5x5 / 7x7): the distance matrix den[(m,d)][(m1,d1)] once per kernel pair (2 (dM dD)^2 floats), then one
// thread per (m,d,k,l) over the partners in the reference's (m1,d1) order.
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

// partner chunk and chunk count of the launch for dM*dD kernels of kl taps: at most 32 chunks while the chunk's taps fit in 64 KB of LDS
// (5x5: 512 partners); beyond that (dM*dD > 16384 at 5x5) the chunk stays at the cap and the chunk count grows -- the finish kernel adds any
// number of partial sums in chunk order
static void gdiff_geom(long np, int kl, int* chunk, int* nchunks)
{
    const int pitch = (kl + 3) & ~3;
    int cap = 128;
    while ((size_t)cap * 2 * pitch * sizeof(float) <= 64 * 1024) cap *= 2;
    int ch = 128;
    while ((np + ch - 1) / ch > 32 && ch < cap) ch *= 2;
    *chunk = ch; *nchunks = (int)((np + ch - 1) / ch);
}

size_t gradient_diff_ws_floats(int dM, int dD, int Nk, int Nl)
{
    const long np = (long)dM * dD;
    int chunk, nchunks;
    const int kl = Nk * Nl;
    gdiff_geom(np, kl, &chunk, &nchunks);
    if (kl != 9 && kl != 25 && kl != 49) return (size_t)2 * np * np;      // (the generic kernels' distance matrix)
    return (size_t)2 * nchunks * np * kl;
}

hipError_t launch_gradient_diff(const float* c, const float* f, const float* b, const float* p, float* cd, float* fd,
                                float* bd, float* pd, float* part_ws, int dM, int dD, int Nk, int Nl, hipStream_t st)
{
    const long np = (long)dM * dD;
    const int kl = Nk * Nl;
    if (kl != 9 && kl != 25 && kl != 49) {
        kdist_kernel<<<dim3((unsigned)((np * np + 255) / 256)), 256, 0, st>>>(c, f, part_ws, dM, dD, kl);
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return e0;
        gradient_diff_kernel<<<dim3((unsigned)((np * kl + 255) / 256)), 256, 0, st>>>(c, f, b, p, part_ws, cd, fd, bd, pd, dM, dD, Nk, Nl);
        return hipGetLastError();
    }
    int chunk, nchunks;
    gdiff_geom(np, kl, &chunk, &nchunks);
    const dim3 grid((unsigned)((np + gd_rows(kl) - 1) / gd_rows(kl)), (unsigned)nchunks, 2);
    const size_t lds = sizeof(float) * (size_t)chunk * ((kl + 3) & ~3);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    if (kl == 9) gdiff_part_kernel<9><<<grid, 256, lds, st>>>(c, f, part_ws, dM, dD, chunk);
    else if (kl == 25) gdiff_part_kernel<25><<<grid, 256, lds, st>>>(c, f, part_ws, dM, dD, chunk);
    else gdiff_part_kernel<49><<<grid, 256, lds, st>>>(c, f, part_ws, dM, dD, chunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long total = 2 * np * kl + dM + dD;
    gdiff_finish_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(part_ws, b, p, cd, fd, bd, pd, dM, dD, kl, nchunks);
    return hipGetLastError();
}

// gradient_diff of up to 8 pairs with the same 3x3 / 5x5 / 7x7 support in two launches (GdiffProb::part: gradient_diff_ws_floats each)
hipError_t launch_gradient_diff_group(GdiffGroup& g, int Nk, int Nl, hipStream_t st)
{
    const int kl = Nk * Nl;
    if (g.n < 1 || g.n > 8 || (kl != 9 && kl != 25 && kl != 49)) return hipErrorInvalidValue;
    long total = 0, ftotal = 0;
    size_t lds = 0;
    for (int i = 0; i < g.n; ++i) {
        GdiffProb& q = g.q[i];
        const long np = (long)q.dM * q.dD;
        gdiff_geom(np, kl, &q.chunk, &q.nchunks);
        g.start[i] = (int)total; total += ((np + gd_rows(kl) - 1) / gd_rows(kl)) * q.nchunks * 2;
        g.fstart[i] = (int)ftotal; ftotal += (2 * np * kl + q.dM + q.dD + 255) / 256;
        lds = std::max(lds, sizeof(float) * (size_t)q.chunk * ((kl + 3) & ~3));
    }
    g.start[g.n] = (int)total; g.fstart[g.n] = (int)ftotal;
    if (lds > 64 * 1024 || total >= (1L << 31) || ftotal >= (1L << 31)) return hipErrorInvalidValue;
    if (kl == 9) gdiff_part_group_kernel<9><<<dim3((unsigned)total), 256, lds, st>>>(g);
    else if (kl == 25) gdiff_part_group_kernel<25><<<dim3((unsigned)total), 256, lds, st>>>(g);
    else gdiff_part_group_kernel<49><<<dim3((unsigned)total), 256, lds, st>>>(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    gdiff_finish_group_kernel<<<dim3((unsigned)ftotal), 256, 0, st>>>(g, kl);
    return hipGetLastError();
}

}  // namespace aefft
