// Device-side helpers shared by the contraction kernels (spectral_kernels.hip, contract_mfma.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace aefft {

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


}  // namespace aefft
