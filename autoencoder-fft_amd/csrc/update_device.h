// Device side of the clipped-momentum update (fft_backproplib.cu:605-652), shared by update_kernels.hip and the launches that carry
// parts of it (pruned_kernels.hip: taps updated on the fly in front of the spectra; spectral_kernels.hip: the in-place write).
#pragma once
#include "internal.h"

namespace aefft {

// D <- (1-alpha)*del*g/max(10,|g|) + alpha*D ; w <- w - D      (fft_backproplib.cu:616-618)
// (explicit rounding steps: the same value must come out of every kernel that evaluates it -- kspec forms the new taps from
// (w, g, D) before update_weights_part stores them)
__device__ __forceinline__ float clip_step(float g, float D, float del, float alpha)
{
    const float ag = fabsf(g);
    const float q = __fdiv_rn(__fmul_rn(__fmul_rn(1.0f - alpha, del), g), (10.f < ag) ? ag : 10.f);
    return __fadd_rn(q, __fmul_rn(alpha, D));
}

// the kernel taps of block `blk` (256 elements): c, f and their momentum, in place
__device__ __forceinline__ void update_weights_part(const UpdateArgs& a, int blk)
{
    const int n = a.dM * a.dD * a.Nk * a.Nl;
    const int idk = blk * 256 + threadIdx.x;
    if (idk >= n) return;
    const bool multi = a.cd != nullptr;
    if (!a.sym) {
        float gc = a.dck[idk] * a.gscale, gf = a.dfk[idk] * a.gscale;
        if (a.ddc) a.ddc[idk] = gc;
        if (a.ddf) a.ddf[idk] = gf;
        if (multi) { gc = a.w0 * gc - a.w1 * a.cd[idk]; gf = a.w0 * gf - a.w1 * a.fd[idk]; }
        const float Dc = clip_step(gc, a.Dc[idk], a.del, a.alpha);
        a.c[idk] += -Dc; a.Dc[idk] = Dc;
        const float Df = clip_step(gf, a.Df[idk], a.del, a.alpha);
        a.f[idk] += -Df; a.Df[idk] = Df;
    } else {
        // tied weights: g = g_c[m][d] + g_f[d][m] with the caller's doubled Norm (backproplib.cu:533,466);
        // c <- c - D ; f[d][m] <- c[m][d] (:621-622).  FFT mode: build-defined.
        const int kl = a.Nk * a.Nl;
        const int m = idk / (a.dD * kl), d = (idk / kl) % a.dD, r = idk % kl;
        const int idf = (d * a.dM + m) * kl + r;
        float g = (a.dck[idk] + a.dfk[idf]) * a.gscale;
        if (a.ddc) a.ddc[idk] = g;
        if (multi) g = a.w0 * g - a.w1 * 0.5f * (a.cd[idk] + a.fd[idf]);
        const float Dc = clip_step(g, a.Dc[idk], a.del, a.alpha);
        const float cn = a.c[idk] - Dc;
        a.c[idk] = cn; a.Dc[idk] = Dc;
        a.f[idf] = cn;
    }
}

// the biases b, p (blocks 0 ..: 256 elements each) and the pair's MSE accumulator
__device__ __forceinline__ void update_bias_part(const UpdateArgs& a, int blk)
{
    const int idk = blk * 256 + threadIdx.x;
    if (idk == 0 && a.zero) *a.zero = 0.f;
    const bool multi = a.cd != nullptr;
    if (idk < a.dM) {
        float g = a.db[idk] * a.gscale;
        if (a.ddb) a.ddb[idk] = g;
        if (multi) g = a.w0 * g - a.w1 * a.bd[idk];
        const float Db = clip_step(g, a.Db[idk], a.del, a.alpha);
        a.b[idk] += -Db; a.Db[idk] = Db;
    }
    if (idk < a.dD) {
        float g = a.dp[idk] * a.gscale;
        if (a.ddp) a.ddp[idk] = g;
        if (multi) g = a.w0 * g - a.w1 * a.pd[idk];
        const float Dp = clip_step(g, a.Dp[idk], a.del, a.alpha);
        a.p[idk] += -Dp; a.Dp[idk] = Dp;
    }
}


__device__ __forceinline__ void update_body(const UpdateArgs& a, int blk)
{
    update_weights_part(a, blk);
    if (blk * 256 < max(a.dM, a.dD)) update_bias_part(a, blk);        // (launch_update checks n >= dM, dD: the blocks exist)
}

// sums (and clears) the MSE slot accumulators of one step: out[l] += sum, copy[l] = out[l]; the packed buffer's tail (copy2): what the
// caller's all-reduce left there (the sum over ranks of the PREVIOUS step's MSEs) is kept, scaled to the global mean, in the L floats
// behind the tail before this step's local value takes its place.  One workgroup of MSE_SLOTS threads; ws: MSE_SLOTS/64 floats of LDS.
__device__ __forceinline__ void mse_finish_body(float* __restrict__ slots, float* __restrict__ out, float* __restrict__ copy, float* __restrict__ copy2,
                                                int L, float prev_scale, float* ws)
{
    for (int l = 0; l < L; ++l) {
        float* s = slots + ((long)l * MSE_SLOTS + threadIdx.x) * MSE_SLOT_STRIDE;
        float v = *s;
        *s = 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = out[l];
            for (int w = 0; w < MSE_SLOTS / 64; ++w) t += ws[w];
            out[l] = t;
            if (copy) copy[l] = t;
            if (copy2) { copy2[L + l] = copy2[l] * prev_scale; copy2[l] = t; }
        }
        __syncthreads();
    }
}

}  // namespace aefft
