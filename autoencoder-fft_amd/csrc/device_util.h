// Device-side helpers shared by the contraction kernels (spectral_kernels.hip, contract_mfma.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace aefft {

// Streaming accesses: bytes that are read once (a step's input frames, an intermediate its only consumer walks through) or written
// and not read again by the step (the reconstruction) carry the non-temporal hint, so that they do not displace what IS re-read
// from the L2s and the Infinity Cache.  Measured at cfg3: the input row pass 36.2 -> 30.6 us, the output row pass 27.0 -> 24.9 us,
// and every kernel in between a few percent faster (whole step 227 -> 209 us, side stream off).
typedef float nt_v4f __attribute__((ext_vector_type(4)));
typedef float nt_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld_stream(const float4* p) { const nt_v4f v = __builtin_nontemporal_load(reinterpret_cast<const nt_v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float2 ld_stream(const float2* p) { const nt_v2f v = __builtin_nontemporal_load(reinterpret_cast<const nt_v2f*>(p)); return make_float2(v.x, v.y); }
__device__ __forceinline__ void st_stream(float4* p, float4 a) { const nt_v4f v = {a.x, a.y, a.z, a.w}; __builtin_nontemporal_store(v, reinterpret_cast<nt_v4f*>(p)); }
__device__ __forceinline__ void st_stream(float2* p, float2 a) { const nt_v2f v = {a.x, a.y}; __builtin_nontemporal_store(v, reinterpret_cast<nt_v2f*>(p)); }
__device__ __forceinline__ void st_stream(float* p, float a) { __builtin_nontemporal_store(a, p); }

// destination bin of source bin (i, j) under the spectral crop [Nx][Ny/2+1] -> [Nxs][Nys/2+1] (inverse of fft.cu:102-111), or -1
__device__ __forceinline__ long crop_dest(long bin, int Nx, int Ny, int Nxs, int Nys)
{
    const int Nyr = Ny / 2 + 1, Nyrs = Nys / 2 + 1;
    const int i = (int)(bin / Nyr), j = (int)(bin - (long)i * Nyr);
    int di = -1, dj = -1;
    if (i < Nxs / 2) di = i;
    else if (i == Nx / 2) di = Nxs / 2;
    else if (i > Nx - Nxs / 2) di = i - Nx + Nxs;
    if (j < Nyrs - 1) dj = j;
    else if (j == Nyr - 1) dj = Nyrs - 1;
    return (di >= 0 && dj >= 0) ? (long)di * Nyrs + dj : -1;
}

struct BlockId { int bx, by, bz; bool ok; };
__device__ __forceinline__ BlockId xcd_decode(int gx, int gy, int gz)
{
    const int lin = blockIdx.x;
    BlockId b;
    if (gx < 64) {
        // too few bin tiles to give every XCD an equal share: plain order (bin tile fastest), all XCDs busy
        b.bx = lin % gx;
        const int rest = lin / gx;
        b.by = rest % gy; b.bz = rest / gy;
        b.ok = b.bz < gz;
        return b;
    }
    // bin tile slowest: the gy*gz workgroups that re-read the same A / B bins run back to back on ONE XCD, so the
    // re-reads hit that XCD's L2 instead of going back to HBM (decisive when the tensors exceed the caches)
    const int xcd = lin & 7, slot = lin >> 3;
    const int per = gy * gz;
    b.bx = (slot / per) * 8 + xcd;
    const int rest = slot - (slot / per) * per;
    b.by = rest % gy; b.bz = rest / gy;
    b.ok = b.bx < gx && b.bz < gz;
    return b;
}
__device__ __forceinline__ BlockId xcd_decode_lin(int lin, int gx, int gy, int gz)
{
    BlockId b;
    if (gx < 64) {
        b.bx = lin % gx;
        const int rest = lin / gx;
        b.by = rest % gy; b.bz = rest / gy;
        b.ok = b.bz < gz;
        return b;
    }
    const int xcd = lin & 7, slot = lin >> 3;
    const int per = gy * gz;
    b.bx = (slot / per) * 8 + xcd;
    const int rest = slot - (slot / per) * per;
    b.by = rest % gy; b.bz = rest / gy;
    b.ok = b.bx < gx && b.bz < gz;
    return b;
}

static inline unsigned xcd_grid(long gx, int gy, int gz) { return gx < 64 ? (unsigned)(gx * gy * gz) : (unsigned)(((gx + 7) / 8) * 8 * gy * gz); }

typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef int v4i_t __attribute__((ext_vector_type(4)));

template <int VEC> struct BufLoad;
template <> struct BufLoad<1> {
    typedef float2 T;
    static __device__ __forceinline__ T ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
    { v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0); return make_float2(__int_as_float(v.x), __int_as_float(v.y)); }
};
template <> struct BufLoad<2> {
    typedef float4 T;
    static __device__ __forceinline__ T ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
    { v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0); return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w)); }
};


// DC-bin terms of gradient_k_io (see bias_grad_kernel, spectral_kernels.hip)
static inline size_t bias_grad_lds(int B, int dD) { return sizeof(float2) * ((size_t)dD + (size_t)B * dD); }
__device__ __forceinline__ void bias_grad_body(const float2* __restrict__ O, const float2* __restrict__ T,
                                                        const float2* __restrict__ F, const float* __restrict__ b,
                                                        float2* __restrict__ df, float* __restrict__ db, float* __restrict__ dp,
                                                        int B, int dM, int dD, long P, float norm, float Norm, int fix_blocks, int blk, float2* es, long PO, float* es_out, const float* es_in = nullptr)
{
    // written for 256 threads; in a larger workgroup (the fused kgrad launch) the extra threads only take part in the barriers
    const bool wk = threadIdx.x < 256;
    // es[d] = sum over frames, in FRAME ORDER (deterministic: 100-iteration bursts amplify any run-to-run rounding difference).
    // LDS: es[dD] complex, then the B*dD per-frame DC differences (callers size the dynamic LDS with bias_grad_lds()).
    float* esf = reinterpret_cast<float*>(es);
    float2* val = es + dD;
    // (frame, channel) pairs spread over the threads so the B*dD DC-bin loads are all in flight at once
    if (es_in) {                                  // operator form: es comes from sgrad_kernel
        if (wk) for (int d = threadIdx.x; d < 2 * dD; d += 256) esf[d] = es_in[d];
    } else {
        if (wk) for (int idx = threadIdx.x; idx < B * dD; idx += 256) {
            const float2 o = O[(long)idx * PO], t = T[(long)idx * P];
            val[idx] = make_float2(o.x - t.x, o.y - t.y);
        }
        __syncthreads();
        if (wk) for (int d = threadIdx.x; d < dD; d += 256) {
            float sx = 0.f, sy = 0.f;
            for (int b2 = 0; b2 < B; ++b2) { const float2 v = val[b2 * dD + d]; sx += v.x; sy += v.y; }
            esf[2 * d] = sx; esf[2 * d + 1] = sy;
        }
    }
    __syncthreads();
    if (!wk) return;
    const float den = Norm * (float)B;
    if (blk == 0 && es_out) for (int d = threadIdx.x; d < 2 * dD; d += 256) es_out[d] = esf[d];
    if (blk < fix_blocks) {
        const int t = blk * 256 + threadIdx.x;
        if (t < dD * dM && df) {
            const int d = t / dM, m = t - d * dM;
            const float2 e = es[d];
            const float bb = b[m] * norm;
            float2* p = df + ((long)d * dM + m) * P;
            float2 v = *p;
            v.x += e.x * bb / den; v.y += e.y * bb / den;
            *p = v;
        }
        if (t < dD) dp[t] = es[t].x * norm / den;
        return;
    }
    // db: one wave per output map m, lanes over d1
    const int m = (blk - fix_blocks) * 4 + (threadIdx.x >> 6);
    if (m >= dM) return;
    float s = 0.f;
    for (int d1 = threadIdx.x & 63; d1 < dD; d1 += 64) {
        const float2 e = es[d1], f = F[((long)d1 * dM + m) * P];
        s += e.x * f.x + e.y * f.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) db[m] = s * norm / den;
}


// Experiment builds only (tools/mkx.sh ... -DAEFFT_X_WGTIME=1, tools/wgtime.py): every workgroup of the instrumented kernels leaves the wall-clock time
// (100 MHz) at which it started and ended in a debug buffer -- which workgroups of a grouped launch are the long pole.  Compiled out of the product.
#if defined(AEFFT_X_WGTIME) && AEFFT_X_WGTIME
static __device__ unsigned long long* g_wgtime = nullptr;   // [kernel slot][WGT_MAX][2], null = off; one copy per translation unit (no relocatable device code)
static inline int wgtime_set_tu(void* p) { unsigned long long* q = (unsigned long long*)p; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wgtime), &q, sizeof q); }
constexpr int WGT_MAX = 8192;
struct WgTimer {
    int slot; unsigned long long t0;
    __device__ __forceinline__ WgTimer(int s) : slot(s), t0(wall_clock64()) {}
    __device__ __forceinline__ ~WgTimer()
    {
        if (threadIdx.x == 0 && threadIdx.y == 0 && g_wgtime && blockIdx.x < WGT_MAX && blockIdx.y == 0) {
            unsigned long long* p = g_wgtime + ((size_t)slot * WGT_MAX + blockIdx.x) * 2;
            p[0] = t0; p[1] = wall_clock64();
        }
    }
};
#define AEFFT_WGTIME(slot) WgTimer wg_timer_(slot)
// stage stamps inside a workgroup (thread 0, after the barrier that ends the stage): [5 slots][WGT_MAX][8] behind the start / end pairs
#define AEFFT_WGSTAMP(slot, k) do { if (threadIdx.x == 0 && g_wgtime && blockIdx.x < WGT_MAX) g_wgtime[(size_t)5 * WGT_MAX * 2 + ((size_t)(slot) * WGT_MAX + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define AEFFT_WGTIME(slot)
#define AEFFT_WGSTAMP(slot, k) do { } while (0)
#endif
}  // namespace aefft
