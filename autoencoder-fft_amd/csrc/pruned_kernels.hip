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

template <int NK, int NL>
__global__ __launch_bounds__(256) void kspec_kernel(const float* __restrict__ kern, float2* __restrict__ K,
                                                    const float2* __restrict__ tw, long planes, int Nx, int Ny,
                                                    int rows_per_chunk, int ppb)
{
    extern __shared__ float2 lds[];
    const int Nyr = Ny / 2 + 1;
    float2* rowph = lds;                                 // [rows_per_chunk][NK]
    float2* colph = lds + rows_per_chunk * NK;           // [Nyr][NL]
    float* taps = reinterpret_cast<float*>(colph + Nyr * NL);   // [ppb][NK*NL]
    const long plane0 = (long)blockIdx.x * ppb;
    const int np = (int)min((long)ppb, planes - plane0);
    const int i0 = blockIdx.y * rows_per_chunk;
    const int nrows = min(rows_per_chunk, Nx - i0);
    // the phase tables depend on (Nx, Ny) only: built once per workgroup, shared by its ppb planes
    for (int t = threadIdx.x; t < nrows * NK; t += 256) rowph[t] = phase(tw, i0 + t / NK, t % NK - NK / 2, Nx, 1.f);
    for (int t = threadIdx.x; t < Nyr * NL; t += 256) colph[t] = phase(tw, t / NL, t % NL - NL / 2, Ny, 1.f);
    for (int t = threadIdx.x; t < np * NK * NL; t += 256) taps[t] = kern[plane0 * NK * NL + t];
    __syncthreads();
    const int nb = nrows * Nyr;
    for (int idx = threadIdx.x; idx < np * nb; idx += 256) {
        const int pl = idx / nb, bin = idx - pl * nb;
        const int i = bin / Nyr, j = bin - i * Nyr;
        const float* c = taps + pl * NK * NL;
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            float2 v = make_float2(0.f, 0.f);
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const float2 cp = colph[j * NL + l];
                const float w = c[k * NL + l];
                v.x += w * cp.x; v.y += w * cp.y;
            }
            const float2 rp = rowph[i * NK + k];
            acc.x += v.x * rp.x - v.y * rp.y;
            acc.y += v.x * rp.y + v.y * rp.x;
        }
        K[((plane0 + pl) * Nx + i0) * (long)Nyr + bin] = acc;
    }
}

// Separable evaluation on a chunk of RB rows of one plane:
//   t[i][l]   = sum_j  w_j * D[i][j] * e^{+2 pi i j*lam_l/Ny}            (RB*NL complex values; j split in JS slices)
//   g[k][l]  += Re sum_i t[i][l] * e^{+2 pi i i*kap_k/Nx}                 (partial over the chunk's rows)
// No cross-lane reduction of NK*NL accumulators per thread: step 1 has one thread per (i,l,slice),
// step 2 one thread per (k,l).  Output: part[plane][chunk][NK*NL] (summed by ksum_kernel when chunks > 1).
template <int NK, int NL>
__global__ __launch_bounds__(256) void kgrad_kernel(const float2* __restrict__ D, float* __restrict__ part,
                                                    const float2* __restrict__ tw, int Nx, int Ny, int RB, int JS, float scale)
{
    extern __shared__ float2 lds[];
    const int Nyr = Ny / 2 + 1;
    float2* rows = lds;                          // [RB][Nyr]
    float2* colph = rows + RB * Nyr;             // [Nyr][NL]  w_j * e^{+...}
    float2* tpart = colph + Nyr * NL;            // [JS][RB*NL]
    float2* rowph = tpart + JS * RB * NL;        // [RB][NK]   e^{+...}
    const long plane = blockIdx.x;
    const int chunk = blockIdx.y, nchunks = gridDim.y;
    const int i0 = chunk * RB;
    const float2* src = D + (plane * Nx + i0) * (long)Nyr;
    for (int t = threadIdx.x; t < RB * Nyr; t += 256) rows[t] = src[t];
    for (int t = threadIdx.x; t < Nyr * NL; t += 256) {
        const int j = t / NL;
        const float2 w = phase(tw, j, t % NL - NL / 2, Ny, -1.f);
        const float wj = (j == 0 || j == Ny / 2) ? 1.f : 2.f;
        colph[t] = make_float2(w.x * wj, w.y * wj);
    }
    for (int t = threadIdx.x; t < RB * NK; t += 256) rowph[t] = phase(tw, i0 + t / NK, t % NK - NK / 2, Nx, -1.f);
    __syncthreads();
    const int nitem = RB * NL * JS;
    for (int it = threadIdx.x; it < nitem; it += 256) {
        const int js = it / (RB * NL), il = it % (RB * NL);
        const int i = il / NL, l = il % NL;
        const int per = (Nyr + JS - 1) / JS;
        const int j0 = js * per, j1 = min(Nyr, j0 + per);
        float2 acc = make_float2(0.f, 0.f);
        for (int j = j0; j < j1; ++j) {
            const float2 d = rows[i * Nyr + j], cp = colph[j * NL + l];
            acc.x += d.x * cp.x - d.y * cp.y;
            acc.y += d.x * cp.y + d.y * cp.x;
        }
        tpart[it] = acc;
    }
    __syncthreads();
    if (threadIdx.x < NK * NL) {
        const int k = threadIdx.x / NL, l = threadIdx.x % NL;
        float g = 0.f;
        for (int i = 0; i < RB; ++i) {
            float2 t = make_float2(0.f, 0.f);
            for (int js = 0; js < JS; ++js) { const float2 p = tpart[js * RB * NL + i * NL + l]; t.x += p.x; t.y += p.y; }
            const float2 rp = rowph[i * NK + k];
            g += t.x * rp.x - t.y * rp.y;
        }
        part[(plane * nchunks + chunk) * (NK * NL) + threadIdx.x] = g * scale;
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

static size_t kspec_lds(int rows, int Ny, int Nk, int Nl, int ppb) { return sizeof(float2) * ((size_t)rows * Nk + (size_t)(Ny / 2 + 1) * Nl) + sizeof(float) * Nk * Nl * ppb; }
// rows per workgroup: as many as fit ~32 KB of LDS, fewer when that would leave the chip short of workgroups
static void kgrad_geom(long planes, int Nx, int Ny, int Nl, int* RB, int* JS)
{
    const int Nyr = Ny / 2 + 1;
    int rb = 1;
    while (rb * 2 <= Nx && (size_t)rb * 2 * Nyr * sizeof(float2) <= 32 * 1024) rb *= 2;
    while (rb > 1 && planes * (Nx / rb) < 512) rb /= 2;
    int js = 256 / (rb * Nl);
    if (js < 1) js = 1;
    if (js > Nyr) js = Nyr;
    *RB = rb; *JS = js;
}
static size_t kgrad_lds(long planes, int Nx, int Ny, int Nk, int Nl)
{
    int RB, JS;
    kgrad_geom(planes, Nx, Ny, Nl, &RB, &JS);
    const int Nyr = Ny / 2 + 1;
    return sizeof(float2) * ((size_t)RB * Nyr + (size_t)Nyr * Nl + (size_t)JS * RB * Nl + (size_t)RB * Nk);
}
size_t kgrad_partial_floats(long planes, int Nx, int Ny, int Nk, int Nl)
{
    int RB, JS;
    kgrad_geom(planes, Nx, Ny, Nl, &RB, &JS);
    return (size_t)planes * (Nx / RB) * Nk * Nl;
}

bool pruned_supported(int Nk, int Nl, int Nx, int Ny)
{
    if (!((Nk == 3 && Nl == 3) || (Nk == 5 && Nl == 5) || (Nk == 7 && Nl == 7))) return false;
    if (Nx > TW_N || Ny > TW_N || (TW_N % Nx) || (TW_N % Ny)) return false;
    return kgrad_lds(1, Nx, Ny, Nk, Nl) <= 150 * 1024 && kgrad_lds(1L << 20, Nx, Ny, Nk, Nl) <= 150 * 1024;
}

template <int NK, int NL>
static hipError_t run_kspec(const float* k, float2* K, const float2* tw, long planes, int Nx, int Ny, hipStream_t st)
{
    // ~2048 bins of output per workgroup: small planes are grouped (shared phase tables), large planes are
    // split into row chunks; at least ~1024 workgroups when the problem allows it
    const int Nyr = Ny / 2 + 1;
    int chunks = 1, ppb = 1;
    while ((long)(Nx / chunks) * Nyr > 2048 && Nx / (chunks * 2) >= 1) chunks *= 2;
    while (chunks == 1 && (long)ppb * 2 * Nx * Nyr <= 2048 && planes / (ppb * 2) >= 1024) ppb *= 2;
    const int rows = (Nx + chunks - 1) / chunks;
    const size_t lds = kspec_lds(rows, Ny, NK, NL, ppb);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kspec_kernel<NK, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    kspec_kernel<NK, NL><<<dim3((unsigned)((planes + ppb - 1) / ppb), chunks), 256, lds, st>>>(k, K, tw, planes, Nx, Ny, rows, ppb);
    return hipGetLastError();
}

template <int NK, int NL>
static hipError_t run_kgrad(const float2* D, float* g, float* part, const float2* tw, long planes, int Nx, int Ny, float scale, hipStream_t st)
{
    int RB, JS;
    kgrad_geom(planes, Nx, Ny, NL, &RB, &JS);
    const int chunks = Nx / RB;
    const size_t lds = kgrad_lds(planes, Nx, Ny, NK, NL);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kgrad_kernel<NK, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    float* dst = chunks == 1 ? g : part;
    kgrad_kernel<NK, NL><<<dim3((unsigned)planes, chunks), 256, lds, st>>>(D, dst, tw, Nx, Ny, RB, JS, scale);
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
