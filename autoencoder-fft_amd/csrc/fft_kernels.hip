// Batched 2-D real<->complex FFT for gfx950, written from scratch (no rocFFT/hipFFT).
//
// Replaces the reference's cuFFT call sites (fft_backproplib.cu:779-796 R2C, 821-829 C2R,
// 885-910 / 937-946 kernel transforms, 1208-1220 / 1281-1282 in the training loop) and, when
// asked, fuses the spectral pooling index remap of fft_backproplib.cu:87-157 (`resize`) into the
// transform so cropped bins are never computed/stored and zero-padded bins never read.
//
// Structure (power-of-two sizes 8..2048):
//   R2C  = row pass   : two real rows are packed as one complex row (a + i*b), one N-point
//                       Stockham FFT in LDS, Hermitian split into the two half-spectra; the
//                       DC and Nyquist columns (both real-valued) are packed into ONE complex
//                       column so the intermediate is exactly N/2 columns wide (power of two,
//                       16-byte aligned rows).
//          column pass: tiles of CW columns x Nx rows staged through LDS (transposing load),
//                       Nx-point FFTs, column 0 unpacked into DC / Nyquist on the way out.
//   C2R  = the mirror image (column pass first), with the self-conjugate columns Hermitian-
//          symmetrised on load so that imaginary parts of self-conjugate bins are ignored
//          (pocketfft / numpy.irfft2 semantics; cuFFT leaves this unspecified).
//
// One transform of length N is computed by N/8 threads holding 8 complex values each:
// radix-8 Stockham (decimation in frequency, auto-sort) passes with a final radix-4/2 pass,
// data exchanged through LDS between passes (padded index n + n/8 against bank conflicts).
#include "internal.h"
#include "device_util.h"
#include <hip/hip_ext.h>
#include <math.h>
#include <map>
#include <vector>

namespace aefft {

__device__ float2 g_tw[TW_N];

hipError_t upload_twiddles(hipStream_t st)
{
    static std::vector<float2> host;
    if (host.empty()) {
        host.resize(TW_N);
        for (int k = 0; k < TW_N; ++k) {
            double a = -2.0 * M_PI * (double)k / (double)TW_N;
            host[k] = make_float2((float)cos(a), (float)sin(a));
        }
    }
    hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_tw), host.data(), sizeof(float2) * TW_N, 0, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st);
}

const float2* twiddle_table()
{
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tw)) != hipSuccess) return nullptr;
    return static_cast<const float2*>(p);
}

bool fft_size_supported(int n) { return n >= 8 && n <= 2048 && (n & (n - 1)) == 0; }

// ------------------------------------------------------------------------------------------
// complex helpers
// ------------------------------------------------------------------------------------------
// complex arithmetic on the native 2-vector, two floats per lane and issue slot (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).  The row and
// column kernels are co-limited by their VALU instruction stream (rocprofv3 --pmc: VALUBusy 51-55 %), and what hipcc makes of a
// complex product or of a rotation by +-i written on float2 is the packed arithmetic PLUS v_mov / v_xor instructions that build the swapped and
// negated operand (92 of the 363 vector instructions of the 512-point row pass).  The VOP3P modifiers do that inside the arithmetic instruction --
// op_sel / op_sel_hi pick which half of each source feeds the low / high result, neg_lo / neg_hi negate it -- so the products and the
// rotate-and-add forms below are written as the instructions themselves: a complex product is two instructions, a +- i b is one.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f tov(float2 a) { return __builtin_bit_cast(v2f, a); }
__device__ __forceinline__ float2 tof(v2f a) { return __builtin_bit_cast(float2, a); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return a + b; }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return a - b; }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    v2f t, r;
    const v2f av = tov(a), bv = tov(b);
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(av), "v"(bv));                                        // (a.x b.x, a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));       // + (-a.y b.y, a.y b.x)
    return tof(r);
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// multiply by exp(DIR*i*pi/2): -i for the forward transform, +i for the inverse
template <int DIR> __device__ __forceinline__ float2 mul_i(float2 a)
{
    return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}
// a + mul_i<DIR>(b) and a - mul_i<DIR>(b) in one instruction each
template <int DIR> __device__ __forceinline__ float2 add_muli(float2 a, float2 b)
{
    v2f r;
    const v2f av = tov(a), bv = tov(b);
    if (DIR < 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(av), "v"(bv));             // (a.x + b.y, a.y - b.x)
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(av), "v"(bv));                     // (a.x - b.y, a.y + b.x)
    return tof(r);
}
template <int DIR> __device__ __forceinline__ float2 sub_muli(float2 a, float2 b) { return add_muli<-DIR>(a, b); }
template <int DIR> __device__ __forceinline__ float2 twid(int idx)
{
    float2 w = g_tw[idx];
    if (DIR > 0) w.y = -w.y;
    return w;
}

template <int R, int DIR> struct Dft;
template <int DIR> struct Dft<2, DIR> {
    static __device__ __forceinline__ void run(float2* a)
    {
        float2 t = a[0];
        a[0] = cadd(t, a[1]);
        a[1] = csub(t, a[1]);
    }
};
template <int DIR> struct Dft<4, DIR> {
    static __device__ __forceinline__ void run(float2* a)
    {
        const float2 t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]);
        const float2 t2 = cadd(a[1], a[3]), d = csub(a[1], a[3]);
        a[0] = cadd(t0, t2); a[2] = csub(t0, t2);
        a[1] = add_muli<DIR>(t1, d); a[3] = sub_muli<DIR>(t1, d);
    }
};
template <int DIR> struct Dft<8, DIR> {
    static __device__ __forceinline__ void run(float2* a)
    {
        float2 e[4] = {a[0], a[2], a[4], a[6]};
        float2 o[4] = {a[1], a[3], a[5], a[7]};
        Dft<4, DIR>::run(e);
        Dft<4, DIR>::run(o);
        const float c = 0.70710678118654752440f;
        // o[u] *= w8^u, w8 = exp(DIR*i*pi/4)
        // w8 = (1 -+ i)/sqrt2, w8^3 = (-1 -+ i)/sqrt2:  o*w8 = c*(o + mul_i(o)),  o*w8^3 = -c*(o - mul_i(o)),  o*w8^2 = mul_i(o) (folded into the sums)
        const float2 cc = make_float2(c, c), nc = make_float2(-c, -c);
        const float2 o1 = cc * add_muli<DIR>(o[1], o[1]);
        const float2 o3 = nc * sub_muli<DIR>(o[3], o[3]);
        a[0] = cadd(e[0], o[0]); a[4] = csub(e[0], o[0]);
        a[1] = cadd(e[1], o1);   a[5] = csub(e[1], o1);
        a[2] = add_muli<DIR>(e[2], o[2]); a[6] = sub_muli<DIR>(e[2], o[2]);
        a[3] = cadd(e[3], o3);   a[7] = csub(e[3], o3);
    }
};

__host__ __device__ constexpr int pad_idx(int n) { return n + (n >> 3); }
__host__ __device__ constexpr int pad_len(int n) { return n + (n >> 3) + 2; }

// One Stockham pass of radix R at stride S (= product of earlier radices) on a transform of
// length N held in LDS at `s` (padded indexing); `t` is this thread's index inside the
// transform, 0..N/8-1.  Butterfly ib reads elements ib + tt*(N/R) and writes q + S*(R*p + u)
// with p = ib / S, q = ib % S, twiddle W_n^(p*u), n = N/S.  In place: read, barrier, write, barrier.
// A transform of N <= 512 points is worked on by N/8 <= 64 threads, i.e. inside ONE wave (callers map transform = tid / T):
// its LDS reads and writes are issued in program order by that wave and the LDS serves a wave's requests in order, so the
// exchanges between radix passes need no workgroup barrier -- only a scheduling fence for the compiler.  The waves of a
// workgroup then run their transforms out of step instead of meeting at 2 barriers per pass.
template <int N> __device__ __forceinline__ void pass_sync()
{
    if constexpr (N / 8 <= 64) __builtin_amdgcn_wave_barrier();
    else __syncthreads();
}

template <int N, int R, int S, int DIR>
__device__ __forceinline__ void fft_pass(float2* s, int t, float2 w1)
{
    constexpr int T = N / 8;
    constexpr int E = 8 / R;
    constexpr int n = N / S;
    float2 a[8];
#pragma unroll
    for (int v = 0; v < E; ++v) {
        const int ib = t + T * v;
#pragma unroll
        for (int tt = 0; tt < R; ++tt) a[v * R + tt] = s[pad_idx(ib + tt * (N / R))];
    }
    pass_sync<N>();
#pragma unroll
    for (int v = 0; v < E; ++v) {
        const int ib = t + T * v;
        const int p = ib / S, q = ib % S;
        Dft<R, DIR>::run(&a[v * R]);
        if constexpr (n > R) {
            // W_n^(p*u), u = 1..7, from the pass's base twiddle w1 = W_n^p (FftTw: an exact table entry) by at most three
            // multiplications each: w2 = w1^2, w3 = w2 w1, w4 = w2^2, w5 = w4 w1, w6 = w3^2, w7 = w4 w3  (a twiddled pass
            // always has R = 8 and one butterfly per thread; relative error of a power <= ~4e-7, inside the FFT's own round-off)
            static_assert(R == 8 && E == 1, "twiddled passes are radix 8");
            const float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
            const float2 w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3);
            a[1] = cmul(a[1], w1); a[2] = cmul(a[2], w2); a[3] = cmul(a[3], w3); a[4] = cmul(a[4], w4);
            a[5] = cmul(a[5], w5); a[6] = cmul(a[6], w6); a[7] = cmul(a[7], w7);
        }
#pragma unroll
        for (int u = 0; u < R; ++u) s[pad_idx(q + S * (R * p + u))] = a[v * R + u];
    }
    if constexpr (S * R == N) __syncthreads();      // last pass: the result is read across transforms
    else pass_sync<N>();
}

// Base twiddles of the (at most three) twiddled passes of this thread's butterflies: pass i has stride S = 8^i and needs
// W_{N/S}^p = W_N^(p*S), p = t / S.  Loaded from the global table at the top of a kernel, in the shadow of its data loads
// (no per-workgroup LDS copy of the table: 4 KB of LDS and 2 table loads + 14 LDS reads per thread less at N = 512).
template <int N, int DIR> struct FftTw {
    float2 w[3];
    __device__ __forceinline__ void load(int t)
    {
        constexpr int S[3] = {1, 8, 64};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            w[i] = make_float2(1.f, 0.f);
            if (N / S[i] > 8) {
                float2 v = g_tw[((t / S[i]) * S[i]) * (TW_N / N)];
                if (DIR > 0) v.y = -v.y;
                w[i] = v;
            }
        }
    }
};

template <int N, int S, int I, int DIR> struct Passes {
    static __device__ __forceinline__ void run(float2* s, int t, const FftTw<N, DIR>& tw)
    {
        constexpr int n = N / S;
        constexpr int R = n >= 8 ? 8 : n;
        fft_pass<N, R, S, DIR>(s, t, tw.w[I < 3 ? I : 2]);
        Passes<N, S * R, I + 1, DIR>::run(s, t, tw);
    }
};
template <int N, int I, int DIR> struct Passes<N, N, I, DIR> {
    static __device__ __forceinline__ void run(float2*, int, const FftTw<N, DIR>&) {}
};

// Caller must have issued __syncthreads() after filling `s`; on return the result is in `s`
// (natural order) and visible to the whole workgroup.
template <int N, int DIR> __device__ __forceinline__ void fft_lds(float2* s, int t, const FftTw<N, DIR>& tw) { Passes<N, 1, 0, DIR>::run(s, t, tw); }

// ------------------------------------------------------------------------------------------
// row pass, forward: real rows -> packed half spectra
// ------------------------------------------------------------------------------------------
template <int N> struct RowCfg {
    static constexpr int T = N / 8;
    static constexpr int NT = T > 256 ? T : 256;
    static constexpr int G = NT / T;          // row PAIRS per workgroup
    static constexpr int PL = pad_len(N);
};

// U8: the frames are 8-bit pixels (what a camera delivers; the reference's application converts them to floats on the host, netlib.cpp:37-51):
// a quarter of the launch's reads, the conversion is exact (v_cvt_f32_ubyteN), everything behind the load is the float kernel.
template <int N, bool U8>
__global__ __launch_bounds__(RowCfg<N>::NT) void r2c_rows_kernel(const void* __restrict__ in_v, float2* __restrict__ mid,
                                                                   long npairs, int Wc)
{
    using Cfg = RowCfg<N>;
    constexpr int T = Cfg::T, NT = Cfg::NT, G = Cfg::G, PL = Cfg::PL;
    extern __shared__ float2 s[];
    const int tid = threadIdx.x;
    const int g = tid / T, t = tid % T;
    const long pair0 = (long)blockIdx.x * G;
    FftTw<N, -1> tws;
    tws.load(t);

    // A thread loads the SAME four columns of both rows of a pair (two 16-byte loads per position, every load of the tile issued before the
    // first LDS store: hipcc does not hoist loads across the iterations of a load->use loop) and writes four complete complex elements
    // a + i*b -- consecutive in the padded layout, i.e. two ds_write2_b64 -- instead of eight 4-byte halves at stride 2.
    constexpr int NPOS = G * (N / 4) / NT;                      // positions per thread (= 2 when NT = G*N/8)
    static_assert((G * (N / 4)) % NT == 0, "tile must be a whole number of positions per thread");
    const int live = npairs - pair0 < G ? (int)(npairs - pair0) : G;      // row pairs of this workgroup that exist (uniform)
    float4 va[NPOS], vb[NPOS];
    if constexpr (U8) {
        const unsigned* src = reinterpret_cast<const unsigned*>(static_cast<const unsigned char*>(in_v) + pair0 * 2 * N);      // four pixels per word
        unsigned wa[NPOS], wb[NPOS];
#pragma unroll
        for (int q = 0; q < NPOS; ++q) {
            const int pos = tid + q * NT;
            const int gg = pos / (N / 4), c4 = pos % (N / 4);
            const int idx = gg < live ? gg * (2 * N / 4) + c4 : c4;
            wa[q] = __builtin_nontemporal_load(&src[idx]);
            wb[q] = __builtin_nontemporal_load(&src[idx + N / 4]);
        }
#pragma unroll
        for (int q = 0; q < NPOS; ++q) {
            va[q] = make_float4((float)(wa[q] & 255u), (float)((wa[q] >> 8) & 255u), (float)((wa[q] >> 16) & 255u), (float)(wa[q] >> 24));
            vb[q] = make_float4((float)(wb[q] & 255u), (float)((wb[q] >> 8) & 255u), (float)((wb[q] >> 16) & 255u), (float)(wb[q] >> 24));
        }
    } else {
        const float4* src = reinterpret_cast<const float4*>(static_cast<const float*>(in_v) + pair0 * 2 * N);
#pragma unroll
        for (int q = 0; q < NPOS; ++q) {
            const int pos = tid + q * NT;
            const int gg = pos / (N / 4), c4 = pos % (N / 4);
            const int idx = gg < live ? gg * (2 * N / 4) + c4 : c4;           // (32-bit lane offsets; rows that do not exist re-read pair 0)
            va[q] = ld_stream(&src[idx]);                                     // (frames are read once)
            vb[q] = ld_stream(&src[idx + N / 4]);
        }
    }
#pragma unroll
    for (int q = 0; q < NPOS; ++q) {
        const int pos = tid + q * NT;
        const int gg = pos / (N / 4), c4 = pos % (N / 4);
        if (gg < live) {
            float2* z = s + gg * PL + pad_idx(4 * c4);                    // (4 c4 + i and 4 c4 share their padding block for i < 4)
            z[0] = make_float2(va[q].x, vb[q].x); z[1] = make_float2(va[q].y, vb[q].y);
            z[2] = make_float2(va[q].z, vb[q].z); z[3] = make_float2(va[q].w, vb[q].w);
        }
    }
    __syncthreads();
    fft_lds<N, -1>(s + g * PL, t, tws);

    // Hermitian split of Z = FFT(a + i*b) into the two rows' packed half spectra: a thread takes bins k, k+1 of BOTH rows (the four
    // elements Z[k], Z[k+1], Z[N-k], Z[N-k-1] are read once) and stores 16 bytes to each.  Wc is a power of two: shifts, not the
    // ~20-instruction expansion of a division by a run-time value.
    const int half = Wc / 2, hs = 31 - __builtin_clz(half);
    float2* const obase = mid + pair0 * 2 * Wc;
    for (int it = tid; it < G * half; it += NT) {
        const int gg = it >> hs, k = (it & (half - 1)) * 2;
        if (gg >= live) continue;
        const float2* z = s + gg * PL;
        const float2 zk1 = z[pad_idx(k + 1)], zn1 = z[pad_idx(N - k - 1)];
        float2 a0, b0;
        if (k == 0) {
            const float2 z0 = z[pad_idx(0)], zh = z[pad_idx(N / 2)];      // DC in .x and Nyquist in .y of the packed column
            a0 = make_float2(z0.x, zh.x); b0 = make_float2(z0.y, zh.y);
        } else {
            const float2 zk = z[pad_idx(k)], zn = z[pad_idx(N - k)];
            a0 = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)); b0 = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        }
        const float2 a1 = make_float2(0.5f * (zk1.x + zn1.x), 0.5f * (zk1.y - zn1.y)), b1 = make_float2(0.5f * (zk1.y + zn1.y), -0.5f * (zk1.x - zn1.x));
        float2* dst = obase + (gg * 2 * Wc + k);                              // (uniform base + 32-bit lane offset)
        *reinterpret_cast<float4*>(dst) = make_float4(a0.x, a0.y, a1.x, a1.y);
        *reinterpret_cast<float4*>(dst + Wc) = make_float4(b0.x, b0.y, b1.x, b1.y);
    }
}

// row pass, inverse: packed half spectra (Wc columns, the rest zero) -> two real rows.
// SPARSE (Wc <= N/16: the training step's reconstruction, whose spectrum lives on the coarsest grid -- 16 of 256 columns at cfg3): of the N
// inputs of a row pair's transform only Z[0 .. Wc), Z[N/2] and Z(N - Wc .. N) are non-zero, i.e. butterfly t of the FIRST radix-8 pass
// (inputs t + tt*N/8) has ONE non-zero input -- tt = 0 for t < Wc, tt = 7 for t > N/8 - Wc -- plus the Nyquist element (tt = 4) at t = 0.  The
// thread loads that element itself and forms the pass's outputs  X[u] = a0 + (-1)^u a4 + a7 conj(w8)^u  in registers: no staging of the tile
// through LDS, no first-pass LDS reads, a third of the first pass's arithmetic; the passes at strides 8 and 64 follow unchanged.
template <int N, bool SPARSE>
__global__ __launch_bounds__(RowCfg<N>::NT) void c2r_rows_kernel(const float2* __restrict__ mid, float* __restrict__ out,
                                                                   long npairs, int Wc, float scale)
{
    using Cfg = RowCfg<N>;
    constexpr int T = Cfg::T, NT = Cfg::NT, G = Cfg::G, PL = Cfg::PL;
    extern __shared__ float2 s[];
    const int tid = threadIdx.x;
    const int g = tid / T, t = tid % T;
    FftTw<N, +1> tws;
    tws.load(t);
    const long pair0 = (long)blockIdx.x * G;
    if constexpr (SPARSE) {
        static_assert(N >= 128, "sparse head: three radix-8 passes or more");
        const int live = npairs - pair0 < G ? (int)(npairs - pair0) : G;
        const float2* const ibase = mid + pair0 * 2 * Wc;
        // this butterfly's non-zero input: column k of the pair's two packed rows A, B
        const bool lo = t < Wc, hi = t > T - Wc;                              // (disjoint: Wc <= T/2)
        const int k = lo ? t : (hi ? T - t : 0);
        const bool ok = g < live && (lo || hi);
        const int off = ok ? g * 2 * Wc + k : 0;
        float2 A = ibase[off], B = ibase[off + Wc];
        if (!ok) { A = make_float2(0.f, 0.f); B = A; }
        const float2 zz = make_float2(0.f, 0.f);
        // t = 0: DC of both rows in .x, Nyquist in .y of the packed column (imaginary parts ignored)
        const float2 a0 = lo ? (t == 0 ? make_float2(A.x, B.x) : make_float2(A.x - B.y, A.y + B.x)) : zz;      // Z[t]
        const float2 a4 = t == 0 ? make_float2(A.y, B.y) : zz;                                                 // Z[N/2]
        const float2 a7 = hi ? make_float2(A.x + B.y, -A.y + B.x) : zz;                                        // Z[N - k] = conj(A) + i conj(B)
        // X[u] = a0 + (-1)^u a4 + a7 * exp(-i pi u / 4)   (inverse transform: w8 = exp(+i pi/4), input 7 carries w8^(7u) = conj(w8)^u)
        const float c = 0.70710678118654752440f;
        const float2 e = cadd(a0, a4), o = csub(a0, a4);
        const float2 r1 = make_float2(c * (a7.x + a7.y), c * (a7.y - a7.x));   // a7 * exp(-i pi/4)
        const float2 r2 = make_float2(a7.y, -a7.x);                            // a7 * (-i)
        const float2 r3 = make_float2(c * (a7.y - a7.x), -c * (a7.x + a7.y));  // a7 * exp(-3 i pi/4)
        float2 x[8] = {cadd(e, a7), cadd(o, r1), cadd(e, r2), cadd(o, r3), csub(e, a7), csub(o, r1), csub(e, r2), csub(o, r3)};
        // twiddles W_N^(t u) of the first pass (fft_pass: p = t, S = 1), then the outputs to their Stockham slots 8 t + u
        const float2 w1 = tws.w[0];
        const float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
        const float2 w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3);
        x[1] = cmul(x[1], w1); x[2] = cmul(x[2], w2); x[3] = cmul(x[3], w3); x[4] = cmul(x[4], w4);
        x[5] = cmul(x[5], w5); x[6] = cmul(x[6], w6); x[7] = cmul(x[7], w7);
        float2* z = s + g * PL;
#pragma unroll
        for (int u = 0; u < 8; ++u) z[pad_idx(8 * t + u)] = x[u];
        pass_sync<N>();
        Passes<N, 8, 1, +1>::run(z, t, tws);
    } else {
    // Z[k] = A[k] + i*B[k], Z[N-k] = conj(A[k]) + i*conj(B[k]); k handled in pairs (k, k+1), k even < N/2
    constexpr int NIT = G * (N / 4) / NT;                       // = 2 when NT = G*N/8
    static_assert((G * (N / 4)) % NT == 0, "tile must be a whole number of items per thread");
    float4 av[NIT], bv[NIT];
    const int live = npairs - pair0 < G ? (int)(npairs - pair0) : G;      // row pairs of this workgroup that exist (uniform)
    const float2* const ibase = mid + pair0 * 2 * Wc;                     // (uniform base + 32-bit lane offsets; masked lanes read element 0)
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
        const int it = tid + q * NT;
        const int gg = it / (N / 4), k = (it % (N / 4)) * 2;
        const bool ok = gg < live && k < Wc;
        const int off = ok ? gg * 2 * Wc + k : 0;
        av[q] = ld_stream(reinterpret_cast<const float4*>(ibase + off));
        bv[q] = ld_stream(reinterpret_cast<const float4*>(ibase + off + Wc));
        if (!ok) { av[q] = make_float4(0.f, 0.f, 0.f, 0.f); bv[q] = av[q]; }
    }
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
        const int it = tid + q * NT;
        const int gg = it / (N / 4), k = (it % (N / 4)) * 2;
        if (gg >= live) continue;
        float2* z = s + gg * PL;
        const float4 a = av[q], b = bv[q];
        if (k == 0) {
            z[pad_idx(0)] = make_float2(a.x, b.x);          // DC of both rows (imaginary parts ignored)
            z[pad_idx(N / 2)] = make_float2(a.y, b.y);      // Nyquist, carried in .y of the packed column
        } else {
            z[pad_idx(k)] = make_float2(a.x - b.y, a.y + b.x);
            z[pad_idx(N - k)] = make_float2(a.x + b.y, -a.y + b.x);
        }
        z[pad_idx(k + 1)] = make_float2(a.z - b.w, a.w + b.z);
        z[pad_idx(N - k - 1)] = make_float2(a.z + b.w, -a.w + b.z);
    }
    __syncthreads();
    fft_lds<N, +1>(s + g * PL, t, tws);
    }

    // four consecutive complex elements per lane (read once, adjacent in the padded layout): their real parts are 16 bytes of row A, their
    // imaginary parts 16 bytes of row B
    constexpr int NQ = G * (N / 4) / NT;
    float* const orow = out + pair0 * 2 * N;
    const int live = npairs - pair0 < G ? (int)(npairs - pair0) : G;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int idx = tid + q * NT;
        const int gg = idx / (N / 4), n = (idx % (N / 4)) * 4;
        if (gg < live) {
            const float2* z = s + gg * PL + pad_idx(n);
            const float2 z0 = z[0], z1 = z[1], z2 = z[2], z3 = z[3];
            st_stream(reinterpret_cast<float4*>(orow + (gg * 2) * N + n), make_float4(z0.x * scale, z1.x * scale, z2.x * scale, z3.x * scale));     // (the images are not read again)
            st_stream(reinterpret_cast<float4*>(orow + (gg * 2 + 1) * N + n), make_float4(z0.y * scale, z1.y * scale, z2.y * scale, z3.y * scale));
        }
    }
}

// ------------------------------------------------------------------------------------------
// column passes
// ------------------------------------------------------------------------------------------
// source row of destination row i when the spectrum is cropped from Nx to Nxs rows (fft.cu:102-104)
__device__ __forceinline__ int crop_row(int i, int Nx, int Nxs)
{
    if (Nxs == Nx || i < Nxs / 2) return i;
    if (i == Nxs / 2) return Nx / 2;
    return i + Nx - Nxs;
}
// source row (of Nxi) feeding destination row r (of Nx) under zero-pad up-sampling (fft.cu:119-133); -1 = zero
__device__ __forceinline__ int padsrc_row(int r, int Nx, int Nxi)
{
    if (Nx == Nxi) return r;
    if (r < Nxi / 2) return r;
    if (r > Nx - Nxi / 2) return r - Nx + Nxi;
    if (r == Nx / 2) return Nxi / 2;
    return -1;
}

// forward: mid [planes][N][Wc] -> out [planes][Nxs][Wc+1], FFT along x (length N)
template <int N, int CW>
__global__ __launch_bounds__(CW* N / 8) void fwd_cols_kernel(const float2* __restrict__ mid, float2* __restrict__ out,
                                                              int Wc, int Nxs)
{
    constexpr int T = N / 8, NT = CW * T, PL = pad_len(N);
    extern __shared__ float2 s[];
    const int tid = threadIdx.x;
    FftTw<N, -1> tws;
    tws.load(tid % T);
    const long plane = blockIdx.x;
    const int c0 = blockIdx.y * CW;

    const float2* src = mid + plane * N * (long)Wc + c0;
    constexpr int NLD = N * (CW / 2) / NT;                      // = 4 for every instantiation (NT = CW*N/8)
    float4 val[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int it = tid + k * NT;
        const float4* ptr = reinterpret_cast<const float4*>(src + ((it / (CW / 2)) * Wc + 2 * (it % (CW / 2))));     // (32-bit lane offset)
        // mid dies here: a streaming load -- when the tile's row segment is a whole 128-byte line.  Narrower tiles (1024-point columns: 8 columns,
        // 64 bytes) share every line with the neighbouring tile of the plane, which the launch order puts on the same XCD a dozen workgroups later:
        // a cached load lets it hit the L2 (PMC at cfg5: 362 MB fetched for 201 MB read with streaming loads; 88-94 -> 76-77 us.  Pairing the two
        // tiles as direct neighbours on their XCD on top of that: 76-84 us, nothing)
        if constexpr (CW * sizeof(float2) >= 128) val[k] = ld_stream(ptr);
        else val[k] = *ptr;
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int it = tid + k * NT;
        const int r = it / (CW / 2), c2 = it % (CW / 2);
        s[(2 * c2) * PL + pad_idx(r)] = make_float2(val[k].x, val[k].y);
        s[(2 * c2 + 1) * PL + pad_idx(r)] = make_float2(val[k].z, val[k].w);
    }
    __syncthreads();
    fft_lds<N, -1>(s + (tid / T) * PL, tid % T, tws);

    const int Nyrs = Wc + 1;
    float2* dst = out + plane * Nxs * (long)Nyrs;
    for (int it = tid; it < Nxs * CW; it += NT) {
        const int i = it / CW, c = it % CW;
        const int si = crop_row(i, N, Nxs);
        const float2 z = s[c * PL + pad_idx(si)];
        const int col = c0 + c;
        if (col == 0) {
            // column 0 carries DC + i*Nyquist of the row pass: split by Hermitian symmetry along x
            const float2 zn = s[pad_idx((N - si) % N)];
            dst[i * Nyrs] = make_float2(0.5f * (z.x + zn.x), 0.5f * (z.y - zn.y));
            dst[i * Nyrs + Wc] = make_float2(0.5f * (z.y + zn.y), -0.5f * (z.x - zn.x));
        } else {
            dst[i * Nyrs + col] = z;
        }
    }
}

// Operator-form input of the reconstruction's inverse transform (opform_kernels.hip): the spectrum of plane (b, d) is not
// stored but evaluated where it is read, O_b[d][t] = A[OPC-1][d][t] + sum_j A[j][d][t] x_b[j][u(t)] (u: the bin of the input grid
// that bin t of this grid maps to) -- 4 loads and 3 complex FMAs per element instead of a launch that writes the planes out.
__device__ __forceinline__ float2 opin_load(const OpIn& o, long plane, int t, int Nxi, int Nyi)
{
    const int b = (int)(plane / o.D0), d = (int)(plane - (long)b * o.D0);
    const long Pc = (long)Nxi * (Nyi / 2 + 1), P0 = (long)o.Nx0 * (o.Ny0 / 2 + 1);
    const unsigned nyr = Nyi / 2 + 1, NyrB = o.Ny0 / 2 + 1;
    const unsigned i = (unsigned)t / nyr, j = (unsigned)t - i * nyr;
    const unsigned bi = i < (unsigned)Nxi / 2 ? i : (i == (unsigned)Nxi / 2 ? (unsigned)o.Nx0 / 2 : i + o.Nx0 - Nxi);
    const unsigned bj = j < nyr - 1 ? j : NyrB - 1;
    const long u = (long)bi * NyrB + bj;
    float2 acc = o.A[((long)(OPIN_COLS - 1) * o.D0 + d) * Pc + t];
    for (int jj = 0; jj < o.D0; ++jj) {
        const float2 a = o.A[((long)jj * o.D0 + d) * Pc + t], x = o.Xf[((long)b * o.D0 + jj) * P0 + u];
        acc.x += a.x * x.x - a.y * x.y; acc.y += a.x * x.y + a.y * x.x;
    }
    return acc;
}

// inverse: in [planes][Nxi][Wc+1] (rows zero-padded to N) -> mid [planes][N][Wc], inverse FFT along x
template <int N, int CW>
__global__ __launch_bounds__(CW* N / 8) void inv_cols_kernel(const float2* __restrict__ in, float2* __restrict__ mid,
                                                              int Wc, int Nxi, const OpIn op)
{
    constexpr int T = N / 8, NT = CW * T, PL = pad_len(N);
    extern __shared__ float2 s[];
    const int tid = threadIdx.x;
    FftTw<N, +1> tws;
    tws.load(tid % T);
    const long plane = blockIdx.x;
    const int c0 = blockIdx.y * CW;
    const int Nyri = Wc + 1;
    const float2* src = in + plane * Nxi * (long)Nyri;
    auto ld = [&](long t) { return op.A ? opin_load(op, plane, (int)t, Nxi, 2 * Wc) : src[t]; };

    // Only Nxi of the N rows carry data (the rest is the zero padding of the spectral up-sampling): zero the tile, then ALL loads
    // of the Nxi x CW source elements in one batch (a rolled load -> store loop is one memory round trip per iteration), then the
    // scatter to the padded rows.  The two self-conjugate columns of column 0 are parked in LDS and symmetrised afterwards.
    float2* dcs = s + CW * PL;                                        // [Nxi] column 0 (DC) and [Nxi] Nyquist column, when c0 == 0
    float2* nys = dcs + Nxi;
    if (Nxi < N) {
        for (int it = tid; it < CW * PL; it += NT) s[it] = make_float2(0.f, 0.f);
        __syncthreads();
    }
    constexpr int NLD = 8;                                          // = N * CW / NT
    float2 v[NLD], w[NLD];
    const int nsrc = Nxi * CW;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        if (k * NT < nsrc) {                                        // uniform
            const int it = min(tid + k * NT, nsrc - 1);
            const int sr = it / CW, c = it % CW;
            v[k] = ld((long)sr * Nyri + c0 + c);
            w[k] = (c0 + c == 0) ? ld((long)sr * Nyri + Wc) : make_float2(0.f, 0.f);
        }
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int it = tid + k * NT;
        if (it < nsrc) {
            const int sr = it / CW, c = it % CW;
            if (c0 + c == 0) { dcs[sr] = v[k]; nys[sr] = w[k]; }
            else {
                const int r = Nxi == N ? sr : (sr < Nxi / 2 ? sr : (sr == Nxi / 2 ? N / 2 : sr + N - Nxi));    // inverse of padsrc_row
                s[c * PL + pad_idx(r)] = v[k];
            }
        }
    }
    if (c0 == 0) {
        __syncthreads();
        for (int r = tid; r < N; r += NT) {
            // Hermitian-symmetrise the two self-conjugate columns (imaginary parts of self-conjugate bins are thereby
            // ignored) and pack them as DC + i*Nyquist
            const int sr = padsrc_row(r, N, Nxi), sm = padsrc_row((N - r) % N, N, Nxi);
            const float2 zz = make_float2(0.f, 0.f);
            const float2 d0 = sr >= 0 ? dcs[sr] : zz, n0 = sr >= 0 ? nys[sr] : zz, d1 = sm >= 0 ? dcs[sm] : zz, n1 = sm >= 0 ? nys[sm] : zz;
            const float2 dc = make_float2(0.5f * (d0.x + d1.x), 0.5f * (d0.y - d1.y));
            const float2 ny = make_float2(0.5f * (n0.x + n1.x), 0.5f * (n0.y - n1.y));
            s[pad_idx(r)] = make_float2(dc.x - ny.y, dc.y + ny.x);
        }
    }
    __syncthreads();
    fft_lds<N, +1>(s + (tid / T) * PL, tid % T, tws);

    float2* dst = mid + plane * N * (long)Wc + c0;
    for (int it = tid; it < N * (CW / 2); it += NT) {
        const int r = it / (CW / 2), c2 = it % (CW / 2);
        const float2 a = s[(2 * c2) * PL + pad_idx(r)], b = s[(2 * c2 + 1) * PL + pad_idx(r)];
        *reinterpret_cast<float4*>(dst + (long)r * Wc + 2 * c2) = make_float4(a.x, a.y, b.x, b.y);
    }
}

// ------------------------------------------------------------------------------------------
// sizes that are not powers of two
// ------------------------------------------------------------------------------------------
// The reference hands any {Nx, Ny} to cufftPlanMany (fft_backproplib.cu:773-779, 885, 1208) and any integer Pooling_scale to pool_fft
// (:980-984).  Here an n-point DFT with n not a power of two is re-expressed by Bluestein's identity  j k = (j^2 + k^2 - (k - j)^2) / 2  as a
// circular convolution of length M = 2^q >= 2n - 1 with the chirp w[j] = exp(DIR i pi j^2 / n):
//     X[k] = w[k] * sum_j (x[j] w[j]) conj(w)[k - j]
// and runs on the power-of-two LDS Stockham passes above (forward transform of length M, product with the chirp's spectrum, inverse
// transform).  One kernel transforms ROWS (contiguous); the 2-D transforms go rows -> transpose -> rows -> transpose.  Even n in 8..1024.
// MODE 0: complex rows -> complex rows;  1: real rows -> half spectra (n/2+1);  2: half spectra -> real rows (Hermitian extension on load,
// imaginary parts of the self-conjugate bins ignored -- pocketfft / numpy.irfft semantics, as the power-of-two path).
__device__ __forceinline__ float2 chirp(int j, int n, int dir)
{
    // exp(dir * i * pi * j^2 / n), the phase reduced exactly in integers: j^2 mod 2n
    const unsigned q = ((unsigned)j * (unsigned)j) % (2u * (unsigned)n);
    float sn, cs;
    sincospif((float)q / (float)n, &sn, &cs);
    return make_float2(cs, dir < 0 ? -sn : sn);
}

template <int M> struct BluCfg {
    static constexpr int T = M / 8;
    static constexpr int NT = T > 256 ? T : 256;
    static constexpr int G = NT / T;
    static constexpr int PL = pad_len(M);
};

// spectrum of the chirp filter b[j] = conj(w[j]) (|j| < n, wrapped into M points), once per (n, M, DIR): bhat[M]
template <int M>
__global__ __launch_bounds__(BluCfg<M>::T) void bluestein_setup_kernel(float2* __restrict__ bhat, int n, int dir)
{
    constexpr int T = BluCfg<M>::T;
    extern __shared__ float2 s[];
    const int t = threadIdx.x;
    FftTw<M, -1> tws;
    tws.load(t);
    for (int j = t; j < M; j += T) {
        const int jj = j < n ? j : (M - j < n ? M - j : -1);
        float2 v = make_float2(0.f, 0.f);
        if (jj >= 0) { v = chirp(jj, n, dir); v.y = -v.y; }
        s[pad_idx(j)] = v;
    }
    __syncthreads();
    fft_lds<M, -1>(s, t, tws);
    for (int j = t; j < M; j += T) bhat[j] = s[pad_idx(j)];
}

template <int M, int MODE>
__global__ __launch_bounds__(BluCfg<M>::NT) void bluestein_rows_kernel(const void* __restrict__ in_, void* __restrict__ out_, const float2* __restrict__ bhat,
                                                                        long nrows, int n, int dir, float scale)
{
    using Cfg = BluCfg<M>;
    constexpr int T = Cfg::T, NT = Cfg::NT, G = Cfg::G, PL = Cfg::PL;
    extern __shared__ float2 s[];
    const int tid = threadIdx.x, g = tid / T, t = tid % T;
    const long row = (long)blockIdx.x * G + g;
    const bool live = row < nrows;
    FftTw<M, -1> twf;
    FftTw<M, +1> twi;
    twf.load(t); twi.load(t);
    float2* z = s + g * PL;
    const int nh = n / 2 + 1;
    for (int j = t; j < M; j += T) {
        float2 v = make_float2(0.f, 0.f);
        if (live && j < n) {
            if (MODE == 0) v = reinterpret_cast<const float2*>(in_)[row * n + j];
            else if (MODE == 1) v = make_float2(reinterpret_cast<const float*>(in_)[row * n + j], 0.f);
            else {
                const float2* X = reinterpret_cast<const float2*>(in_) + row * nh;
                if (j < nh) { v = X[j]; if (j == 0 || 2 * j == n) v.y = 0.f; }
                else { v = X[n - j]; v.y = -v.y; }
            }
            v = cmul(v, chirp(j, n, dir));
        }
        z[pad_idx(j)] = v;
    }
    __syncthreads();
    fft_lds<M, -1>(z, t, twf);
    for (int j = t; j < M; j += T) z[pad_idx(j)] = cmul(z[pad_idx(j)], bhat[j]);
    __syncthreads();
    fft_lds<M, +1>(z, t, twi);
    if (!live) return;
    const float sc = scale / (float)M;
    const int nout = MODE == 1 ? nh : n;
    for (int k = t; k < nout; k += T) {
        float2 v = cmul(z[pad_idx(k)], chirp(k, n, dir));
        if (MODE == 2) reinterpret_cast<float*>(out_)[row * n + k] = v.x * sc;
        else reinterpret_cast<float2*>(out_)[row * nout + k] = make_float2(v.x * sc, v.y * sc);
    }
}

// [planes][R][C] -> [planes][C][R]
__global__ __launch_bounds__(256) void transpose_c_kernel(const float2* __restrict__ in, float2* __restrict__ out, int R, int C)
{
    __shared__ float2 tile[32][33];
    const long plane = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;
    for (int k = ty; k < 32; k += 8) if (r0 + k < R && c0 + tx < C) tile[k][tx] = in[(plane * R + r0 + k) * C + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8) if (c0 + k < C && r0 + tx < R) out[(plane * C + c0 + k) * R + r0 + tx] = tile[tx][k];
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------
size_t fft_mid_elems(long planes, int Nx, int Wc) { return (size_t)planes * Nx * Wc; }

template <typename K> static hipError_t allow_lds(K kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int N> static hipError_t run_r2c_rows(const void* in, float2* mid, long npairs, int Wc, hipStream_t st, bool in_u8)
{
    using Cfg = RowCfg<N>;
    const size_t lds = sizeof(float2) * (Cfg::G * Cfg::PL);
    hipError_t e = in_u8 ? allow_lds(r2c_rows_kernel<N, true>, lds) : allow_lds(r2c_rows_kernel<N, false>, lds);
    if (e != hipSuccess) return e;
    const long blocks = (npairs + Cfg::G - 1) / Cfg::G;
    if (in_u8) r2c_rows_kernel<N, true><<<dim3((unsigned)blocks), dim3(Cfg::NT), lds, st>>>(in, mid, npairs, Wc);
    else r2c_rows_kernel<N, false><<<dim3((unsigned)blocks), dim3(Cfg::NT), lds, st>>>(in, mid, npairs, Wc);
    return hipGetLastError();
}
template <int N> static hipError_t run_c2r_rows(const float2* mid, float* out, long npairs, int Wc, float scale, hipStream_t st)
{
    using Cfg = RowCfg<N>;
    const size_t lds = sizeof(float2) * (Cfg::G * Cfg::PL);
    const long blocks = (npairs + Cfg::G - 1) / Cfg::G;
    if constexpr (N >= 128) {
        if (Wc <= N / 16) {                          // few non-zero columns (the reconstruction of a pooled network): sparse first pass
            hipError_t e = allow_lds(c2r_rows_kernel<N, true>, lds);
            if (e != hipSuccess) return e;
            c2r_rows_kernel<N, true><<<dim3((unsigned)blocks), dim3(Cfg::NT), lds, st>>>(mid, out, npairs, Wc, scale);
            return hipGetLastError();
        }
    }
    hipError_t e = allow_lds(c2r_rows_kernel<N, false>, lds);
    if (e != hipSuccess) return e;
    c2r_rows_kernel<N, false><<<dim3((unsigned)blocks), dim3(Cfg::NT), lds, st>>>(mid, out, npairs, Wc, scale);
    return hipGetLastError();
}
template <int N, int CW> static hipError_t run_fwd_cols(const float2* mid, float2* out, long planes, int Wc, int Nxs, hipStream_t st, hipEvent_t done)
{
    const size_t lds = sizeof(float2) * (CW * pad_len(N));
    hipError_t e = allow_lds(fwd_cols_kernel<N, CW>, lds);
    if (e != hipSuccess) return e;
    // `done`: recorded by this dispatch's own completion signal (no marker packet behind it on the stream: a side stream forks here)
    if (done) hipExtLaunchKernelGGL((fwd_cols_kernel<N, CW>), dim3((unsigned)planes, Wc / CW), dim3(CW * N / 8), lds, st, nullptr, done, 0, mid, out, Wc, Nxs);
    else fwd_cols_kernel<N, CW><<<dim3((unsigned)planes, Wc / CW), dim3(CW * N / 8), lds, st>>>(mid, out, Wc, Nxs);
    return hipGetLastError();
}
static OpIn g_opin_none{};
template <int N, int CW> static hipError_t run_inv_cols(const float2* in, float2* mid, long planes, int Wc, int Nxi, hipStream_t st, const OpIn& op = g_opin_none)
{
    const size_t lds = sizeof(float2) * (CW * pad_len(N) + 2 * Nxi);
    hipError_t e = allow_lds(inv_cols_kernel<N, CW>, lds);
    if (e != hipSuccess) return e;
    inv_cols_kernel<N, CW><<<dim3((unsigned)planes, Wc / CW), dim3(CW * N / 8), lds, st>>>(in, mid, Wc, Nxi, op);
    return hipGetLastError();
}

// column tile width: at most 16, at most Wc, and CW*N/8 <= 1024 threads
template <int N, bool FWD> static hipError_t cols_dispatch(const float2* a, float2* b, long planes, int Wc, int Nother, hipStream_t st, const OpIn& op = g_opin_none, hipEvent_t done = nullptr)
{
    constexpr int CWMAX = (8192 / N) < 16 ? (8192 / N) : 16;
    int cw = CWMAX;
    while (cw > Wc) cw >>= 1;
    // few tiles (the reconstruction's compact spectrum: Wc = 16 columns x B*D planes = 96 workgroups): narrower tiles until the grid
    // covers the chip (10.7 -> 9.3 us)
    if (!FWD) while (cw > 4 && planes * (Wc / cw) < 256) cw >>= 1;
#define AEFFT_CW_CASE(C)                                                                                   \
    if constexpr (C <= CWMAX) {                                                                             \
        if (cw == C) return FWD ? run_fwd_cols<N, C>(a, b, planes, Wc, Nother, st, done) : run_inv_cols<N, C>(a, b, planes, Wc, Nother, st, op); \
    }
    AEFFT_CW_CASE(16) AEFFT_CW_CASE(8) AEFFT_CW_CASE(4)
#undef AEFFT_CW_CASE
    return hipErrorInvalidValue;
}

#define AEFFT_N_SWITCH(n, CALL)                       \
    switch (n) {                                      \
    case 8: { constexpr int NN = 8; CALL; }           \
    case 16: { constexpr int NN = 16; CALL; }         \
    case 32: { constexpr int NN = 32; CALL; }         \
    case 64: { constexpr int NN = 64; CALL; }         \
    case 128: { constexpr int NN = 128; CALL; }       \
    case 256: { constexpr int NN = 256; CALL; }       \
    case 512: { constexpr int NN = 512; CALL; }       \
    case 1024: { constexpr int NN = 1024; CALL; }     \
    case 2048: { constexpr int NN = 2048; CALL; }     \
    default: e = hipErrorInvalidValue;                \
    }

// ---- any even size in 8..1024 (Bluestein) ----
bool fft_size_supported_any(int n) { return fft_size_supported(n) || (n >= 8 && n <= 1024 && (n & 1) == 0); }
static int blu_m(int n) { int m = 16; while (m < 2 * n - 1) m *= 2; return m; }
// the chirp spectra, once per (device, n, DIR); they live for the life of the process (a few KB each)
struct BluKey { int dev, n, dir; bool operator<(const BluKey& o) const { return dev != o.dev ? dev < o.dev : (n != o.n ? n < o.n : dir < o.dir); } };
#define AEFFT_M_SWITCH(m, CALL)                      \
    switch (m) {                                      \
    case 16: { constexpr int MM = 16; CALL; }         \
    case 32: { constexpr int MM = 32; CALL; }         \
    case 64: { constexpr int MM = 64; CALL; }         \
    case 128: { constexpr int MM = 128; CALL; }       \
    case 256: { constexpr int MM = 256; CALL; }       \
    case 512: { constexpr int MM = 512; CALL; }       \
    case 1024: { constexpr int MM = 1024; CALL; }     \
    case 2048: { constexpr int MM = 2048; CALL; }     \
    default: e = hipErrorInvalidValue;                \
    }
template <int M> static hipError_t run_blu_setup(float2* bhat, int n, int dir, hipStream_t st)
{
    bluestein_setup_kernel<M><<<1, BluCfg<M>::T, sizeof(float2) * pad_len(M), st>>>(bhat, n, dir);
    return hipGetLastError();
}
static hipError_t blu_table(int n, int dir, hipStream_t st, const float2** out)
{
    static std::map<BluKey, float2*> cache;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const BluKey key{dev, n, dir};
    auto it = cache.find(key);
    if (it == cache.end()) {
        const int M = blu_m(n);
        float2* p = nullptr;
        e = hipMalloc(&p, sizeof(float2) * M);
        if (e != hipSuccess) return e;
        AEFFT_M_SWITCH(M, e = run_blu_setup<MM>(p, n, dir, st); break)
        if (e == hipSuccess) e = hipStreamSynchronize(st);      // (once per size: the table must be complete before another stream uses it)
        if (e != hipSuccess) { (void)hipFree(p); return e; }
        it = cache.emplace(key, p).first;
    }
    *out = it->second;
    return hipSuccess;
}
template <int M, int MODE> static hipError_t run_blu_rows(const void* in, void* out, const float2* bhat, long nrows, int n, int dir, float scale, hipStream_t st)
{
    using Cfg = BluCfg<M>;
    const size_t lds = sizeof(float2) * (size_t)Cfg::G * Cfg::PL;
    hipError_t e = allow_lds(bluestein_rows_kernel<M, MODE>, lds);
    if (e != hipSuccess) return e;
    const long blocks = (nrows + Cfg::G - 1) / Cfg::G;
    if (blocks >= (1L << 31)) return hipErrorInvalidValue;
    bluestein_rows_kernel<M, MODE><<<dim3((unsigned)blocks), dim3(Cfg::NT), lds, st>>>(in, out, bhat, nrows, n, dir, scale);
    return hipGetLastError();
}
static hipError_t blu_rows(int mode, const void* in, void* out, long nrows, int n, int dir, float scale, hipStream_t st)
{
    const float2* bhat = nullptr;
    hipError_t e = blu_table(n, dir, st, &bhat);
    if (e != hipSuccess) return e;
    const int M = blu_m(n);
    if (mode == 0) { AEFFT_M_SWITCH(M, e = (run_blu_rows<MM, 0>(in, out, bhat, nrows, n, dir, scale, st)); break) }
    else if (mode == 1) { AEFFT_M_SWITCH(M, e = (run_blu_rows<MM, 1>(in, out, bhat, nrows, n, dir, scale, st)); break) }
    else { AEFFT_M_SWITCH(M, e = (run_blu_rows<MM, 2>(in, out, bhat, nrows, n, dir, scale, st)); break) }
    return e;
}
static hipError_t transpose_c(const float2* in, float2* out, long planes, int R, int C, hipStream_t st)
{
    if (planes > 65535) return hipErrorInvalidValue;
    transpose_c_kernel<<<dim3((C + 31) / 32, (R + 31) / 32, (unsigned)planes), 256, 0, st>>>(in, out, R, C);
    return hipGetLastError();
}
// complex elements each of the two workspaces of the any-size transforms needs
size_t fft_any_ws_elems(long planes, int Nx, int Ny) { return (size_t)planes * Nx * (Ny / 2 + 1); }
// unnormalised 2-D R2C of any even size: in [planes][Nx][Ny] -> out [planes][Nx][Ny/2+1]; w1, w2: fft_any_ws_elems complex each
hipError_t launch_r2c_any(const float* in, float2* out, float2* w1, float2* w2, long planes, int Nx, int Ny, hipStream_t st)
{
    if (!fft_size_supported_any(Nx) || !fft_size_supported_any(Ny) || Nx > 1024 || Ny > 1024) return hipErrorInvalidValue;
    if (planes <= 0) return hipSuccess;
    const int Nyr = Ny / 2 + 1;
    hipError_t e = blu_rows(1, in, w1, planes * Nx, Ny, -1, 1.f, st);                     // rows: real -> half spectra  [planes][Nx][Nyr]
    if (e == hipSuccess) e = transpose_c(w1, w2, planes, Nx, Nyr, st);                     //                             [planes][Nyr][Nx]
    if (e == hipSuccess) e = blu_rows(0, w2, w1, planes * Nyr, Nx, -1, 1.f, st);           // the x axis as rows
    if (e == hipSuccess) e = transpose_c(w1, out, planes, Nyr, Nx, st);                    //                             [planes][Nx][Nyr]
    return e;
}
// 2-D C2R (scale applied), any even size: in [planes][Nx][Ny/2+1] -> out [planes][Nx][Ny]
hipError_t launch_c2r_any(const float2* in, float* out, float2* w1, float2* w2, long planes, int Nx, int Ny, float scale, hipStream_t st)
{
    if (!fft_size_supported_any(Nx) || !fft_size_supported_any(Ny) || Nx > 1024 || Ny > 1024) return hipErrorInvalidValue;
    if (planes <= 0) return hipSuccess;
    const int Nyr = Ny / 2 + 1;
    hipError_t e = transpose_c(in, w1, planes, Nx, Nyr, st);                               // [planes][Nyr][Nx]
    if (e == hipSuccess) e = blu_rows(0, w1, w2, planes * Nyr, Nx, +1, 1.f, st);
    if (e == hipSuccess) e = transpose_c(w2, w1, planes, Nyr, Nx, st);                     // [planes][Nx][Nyr]
    if (e == hipSuccess) e = blu_rows(2, w1, out, planes * Nx, Ny, +1, scale, st);
    return e;
}

// `in` non-null: run the row pass (in -> mid); `out` non-null: run the column pass (mid -> out).
hipError_t launch_r2c(const void* in, float2* out, float2* mid, long planes, int Nx, int Ny, int Nxs, int Nys, hipStream_t st, hipEvent_t done, bool in_u8)
{
    if (!fft_size_supported(Nx) || !fft_size_supported(Ny) || Nxs > Nx || Nys > Ny || Nys < 8 || Nxs < 2 || (Nys & (Nys - 1)) || (Nxs & 1))
        return hipErrorInvalidValue;                     // (Nys a power of two: the row pass indexes its packed columns with shifts)
    if (planes <= 0) return hipSuccess;
    const int Wc = Nys / 2;
    const long npairs = planes * Nx / 2;
    hipError_t e = hipSuccess;
    if (in) {
        AEFFT_N_SWITCH(Ny, e = run_r2c_rows<NN>(in, mid, npairs, Wc, st, in_u8); break)
        if (e != hipSuccess) return e;
    }
    if (out) {
        AEFFT_N_SWITCH(Nx, e = (cols_dispatch<NN, true>(mid, out, planes, Wc, Nxs, st, g_opin_none, done)); break)
    }
    return e;
}

// `in` non-null: run the column pass (in -> mid); `out` non-null: run the row pass (mid -> out).
hipError_t launch_c2r(const float2* in, float* out, float2* mid, long planes, int Nxi, int Nyi, int Nx, int Ny, float scale, hipStream_t st, const OpIn* opin)
{
    const OpIn op = opin ? *opin : g_opin_none;
    if (!fft_size_supported(Nx) || !fft_size_supported(Ny) || Nxi > Nx || Nyi > Ny || Nyi < 8 || Nxi < 2 || (Nyi & 1) || (Nxi & 1))
        return hipErrorInvalidValue;
    if ((Nxi == Nx) != (Nyi == Ny)) return hipErrorInvalidValue;   // pad both axes or none
    if (planes <= 0) return hipSuccess;
    const int Wc = Nyi / 2;
    const long npairs = planes * Nx / 2;
    hipError_t e = hipSuccess;
    if (in || op.A) {
        AEFFT_N_SWITCH(Nx, e = (cols_dispatch<NN, false>(in, mid, planes, Wc, Nxi, st, op)); break)
        if (e != hipSuccess) return e;
    }
    if (out) {
        AEFFT_N_SWITCH(Ny, e = run_c2r_rows<NN>(mid, out, npairs, Wc, scale, st); break)
    }
    return e;
}

}  // namespace aefft
