// dev tool (GPU box): the data exchange between the radix-8 passes of a 512-point complex FFT held by ONE wave64 (8 elements
// per lane), done two ways --
//   lds   : write the pass's outputs to LDS, read the next pass's inputs back (what fft_kernels.hip does; wave-local barrier only)
//   xlane : no LDS storage: a lane-dependent rotation of the 8 registers (3 stages of conditional swaps), 8 ds_bpermute rounds per
//           component, and the inverse rotation on the receiving side (the "wave64 butterfly shuffle" form of the exchange)
// Both kernels share loads, butterflies, twiddles and stores; REP repeats the transform in registers so that the exchange cost
// shows next to (REP = 1) and without (REP = 16) the global-memory time.  Prints per-variant time and the max difference.
//   hipcc --offload-arch=gfx950 -O3 -o build_x/fft_exchange_probe tools/fft_exchange_probe.hip && build_x/fft_exchange_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }      // * (-i)

// forward 8-point DFT in place, natural order out
__device__ __forceinline__ void dft8(float2* a)
{
    const float h = 0.70710678118654752f;
    float2 b0 = cadd(a[0], a[4]), b4 = csub(a[0], a[4]), b1 = cadd(a[1], a[5]), b5 = csub(a[1], a[5]);
    float2 b2 = cadd(a[2], a[6]), b6 = csub(a[2], a[6]), b3 = cadd(a[3], a[7]), b7 = csub(a[3], a[7]);
    b5 = cmul(b5, make_float2(h, -h)); b6 = mul_mi(b6); b7 = cmul(b7, make_float2(-h, -h));
    float2 c0 = cadd(b0, b2), c2 = csub(b0, b2), c1 = cadd(b1, b3), c3 = mul_mi(csub(b1, b3));
    float2 c4 = cadd(b4, b6), c6 = csub(b4, b6), c5 = cadd(b5, b7), c7 = mul_mi(csub(b5, b7));
    a[0] = cadd(c0, c1); a[4] = csub(c0, c1); a[2] = cadd(c2, c3); a[6] = csub(c2, c3);
    a[1] = cadd(c4, c5); a[5] = csub(c4, c5); a[3] = cadd(c6, c7); a[7] = csub(c6, c7);
}

// rotate the 8 registers by r (0..7, per lane): out[i] = a[(i + r) & 7], as a 3-stage barrel shifter of conditional moves
__device__ __forceinline__ void rot8(float2* a, int r)
{
#pragma unroll
    for (int s = 1; s < 8; s <<= 1) {
        const bool on = (r & s) != 0;
        float2 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = a[(i + s) & 7];
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i].x = on ? t[i].x : a[i].x; a[i].y = on ? t[i].y : a[i].y; }
    }
}

// dest (lane t, reg tt) <- src (lane srcl(t, tt), reg srcr(t)).  Every source lane's 8 registers go to 8 different lanes, so the
// exchange runs in 8 rounds; in round k lane t fetches its register tt = (k + srcr(t)) & 7 from lane srcl(t, tt), which offers its
// register (high(src) - k) & 7 ... arranged by pre-rotating the source registers and post-rotating the received ones.
//   exchange 1 (after the stride-1 pass):  srcl = 8*tt + t/8,     srcr = t%8      (source lane s sends reg u to lane 8*(s%8)+u, reg s/8)
//   exchange 2 (after the stride-8 pass):  srcl = t%8 + 8*tt,     srcr = t/8      (source lane s sends reg u to lane s%8 + 8*u, reg s/8)
template <int WHICH> __device__ __forceinline__ void exchange_xlane(float2* a, int t)
{
    // source side: after rot8 by (s/8), register index j holds original register (j + s/8) & 7; in round k the lane offers j = (8 - k) & 7,
    // i.e. original register (s/8 - k) & 7
    rot8(a, t >> 3);
    float2 r[8];
    const int sr = WHICH == 1 ? (t & 7) : (t >> 3);          // the source register this lane wants (same in every round)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // this round lane t fetches destination register tt = (k + sr) & 7: the source lane is srcl(t, tt), whose s/8 = tt, and it offers
        // original register (tt - k) & 7 = sr  -- the wanted one
        const int tt = (k + sr) & 7;
        const int src = WHICH == 1 ? 8 * tt + (t >> 3) : (t & 7) + 8 * tt;
        const float2 v = a[(8 - k) & 7];
        r[k].x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v.x)));
        r[k].y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v.y)));
    }
    // receiving side: r[k] belongs in register (k + sr) & 7  ->  a[i] = r[(i - sr) & 7]: rotate by (8 - sr) & 7
    rot8(r, (8 - sr) & 7);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = r[i];
}

// the same exchanges through LDS: pass outputs to their natural positions, next pass reads position t + 64*tt
template <int WHICH> __device__ __forceinline__ void exchange_lds(float2* a, int t, float2* s)
{
    // (pad: one float2 per 32 to spread the strided writes over the banks, like pad_idx of fft_kernels.hip)
    auto pad = [](int i) { return i + (i >> 5); };
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int pos = WHICH == 1 ? 8 * t + u : (t & 7) + 64 * (t >> 3) + 8 * u;
        s[pad(pos)] = a[u];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) a[tt] = s[pad(t + 64 * tt)];
    __builtin_amdgcn_wave_barrier();
}

template <bool XLANE, int REP>
__global__ __launch_bounds__(256) void fft512_kernel(const float2* __restrict__ in, float2* __restrict__ out, long ntr)
{
    __shared__ float2 sh[4][512 + 16];
    const int w = threadIdx.x >> 6, t = threadIdx.x & 63;
    const long tr = (long)blockIdx.x * 4 + w;
    if (tr >= ntr) return;
    float2 a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = in[tr * 512 + t + 64 * k];
    // twiddles of the two twiddled passes: W_512^(t*u) and W_64^((t/8)*u), u = 1..7
    float2 w1[8], w2[8];
#pragma unroll
    for (int u = 1; u < 8; ++u) {
        float sn, cs;
        sincospif(-2.0f * (float)(t * u) / 512.0f, &sn, &cs); w1[u] = make_float2(cs, sn);
        sincospif(-2.0f * (float)((t >> 3) * u) / 64.0f, &sn, &cs); w2[u] = make_float2(cs, sn);
    }
#pragma unroll 1
    for (int rep = 0; rep < REP; ++rep) {
        dft8(a);
#pragma unroll
        for (int u = 1; u < 8; ++u) a[u] = cmul(a[u], w1[u]);
        if (XLANE) exchange_xlane<1>(a, t); else exchange_lds<1>(a, t, sh[w]);
        dft8(a);
#pragma unroll
        for (int u = 1; u < 8; ++u) a[u] = cmul(a[u], w2[u]);
        if (XLANE) exchange_xlane<2>(a, t); else exchange_lds<2>(a, t, sh[w]);
        dft8(a);
        if (REP > 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { a[u].x *= (1.0f / 512.0f); a[u].y *= (1.0f / 512.0f); }
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) out[tr * 512 + t + 64 * u] = a[u];
}

template <bool XLANE, int REP> static float run(const float2* in, float2* out, long ntr, int iters)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid((unsigned)((ntr + 3) / 4));
    for (int i = 0; i < 3; ++i) fft512_kernel<XLANE, REP><<<grid, 256>>>(in, out, ntr);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) fft512_kernel<XLANE, REP><<<grid, 256>>>(in, out, ntr);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main()
{
    const long ntr = 96L * 512;                    // the column transforms of one cfg3 input batch: 32 frames x 3 channels x 512 columns
    const size_t n = (size_t)ntr * 512;
    std::vector<float2> h(n);
    srand(1);
    for (auto& v : h) v = make_float2((float)(rand() % 256), (float)(rand() % 256));
    float2 *din, *d0, *d1;
    CHECK(hipMalloc(&din, n * sizeof(float2))); CHECK(hipMalloc(&d0, n * sizeof(float2))); CHECK(hipMalloc(&d1, n * sizeof(float2)));
    CHECK(hipMemcpy(din, h.data(), n * sizeof(float2), hipMemcpyHostToDevice));
    const float tl1 = run<false, 1>(din, d0, ntr, 50), tx1 = run<true, 1>(din, d1, ntr, 50);
    std::vector<float2> r0(n), r1(n);
    CHECK(hipMemcpy(r0.data(), d0, n * sizeof(float2), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r1.data(), d1, n * sizeof(float2), hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < n; ++i) { md = fmax(md, fmax(fabs(r0[i].x - r1[i].x), fabs(r0[i].y - r1[i].y))); mx = fmax(mx, fabs(r0[i].x)); }
    // reference check of transform 0 against a direct DFT (double)
    double err = 0;
    for (int k = 0; k < 512; k += 37) {
        double re = 0, im = 0;
        for (int j = 0; j < 512; ++j) { const double ph = -2.0 * M_PI * j * k / 512.0; re += h[j].x * cos(ph) - h[j].y * sin(ph); im += h[j].x * sin(ph) + h[j].y * cos(ph); }
        err = fmax(err, fmax(fabs(re - r0[k].x), fabs(im - r0[k].y)));
    }
    const float tl16 = run<false, 16>(din, d0, ntr, 20), tx16 = run<true, 16>(din, d1, ntr, 20);
    const double bytes = 2.0 * n * sizeof(float2);
    printf("{\"probe\": \"512-point complex FFT per wave64, %ld transforms (%.0f MB in + out)\", \"lds_exchange_us\": %.2f, \"xlane_exchange_us\": %.2f, "
           "\"lds_exchange_x16_us\": %.2f, \"xlane_exchange_x16_us\": %.2f, \"per_transform_pass_cost_us\": {\"lds\": %.3f, \"xlane\": %.3f}, "
           "\"GBps_rep1\": {\"lds\": %.0f, \"xlane\": %.0f}, \"max_abs_diff_between_variants\": %.3g, \"max_abs_value\": %.3g, \"max_err_vs_direct_dft\": %.3g}\n",
           ntr, bytes / 1e6, tl1, tx1, tl16, tx16, (tl16 - tl1) / 15.0, (tx16 - tx1) / 15.0, bytes / tl1 / 1e3, bytes / tx1 / 1e3, md, mx, err);
    return (md <= 1e-3 * mx && err <= 1e-3 * mx) ? 0 : 1;
}
