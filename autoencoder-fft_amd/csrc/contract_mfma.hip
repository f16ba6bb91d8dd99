// Per-bin complex channel contraction on the matrix cores (gfx950, v_mfma_f32_4x4x1_16b_f32).
//
//   Out[r][c][bin] = alpha * sum_k opA(A[r][k][bin]) * opB(B[k][c][bin])          (Contract, internal.h)
//
// is one small complex GEMM PER FREQUENCY BIN (conv_k, fft_backproplib.cu:162-189, and the re-associated halves of
// gradient_k_io, :395-475).  The 4x4x1 MFMA executes 16 independent 4x4 rank-1 updates per instruction, one per
// "block"; here block = bin.  Operand lane 4*blk + i carries row i (A) / column i (B) of bin blk, result register v
// of lane 4*blk + j is D[v][j] of bin blk (layout verified on the device, tools/mfma_probe.hip).  With the tensors
// kept in the reference's [channel][bin] order this needs NO transposition: for a fixed row the 16 bins of a wave
// are 128 contiguous bytes.  A complex rank-1 update is four real MFMAs (re += ar*br, re -= ai*bi, im += ar*bi,
// im += ai*br); conjugation only changes which of (ai, -ai) feeds the second and fourth.
//
// Work decomposition: a wave owns 16*VEC consecutive bins and a (4*TRB) x (4*TCB) tile of (r, c); per k it issues
// TRB + TCB buffer loads (8*VEC bytes per lane, K stride in a scalar register) for 4*VEC*TRB*TCB MFMAs of 256 MACs.
// Against the scalar-FMA register-tile kernel (spectral_kernels.hip) that is 2x the arithmetic rate and, more
// important, 4-8x fewer operand bytes through the vector memory path per MAC -- the resource those launches saturate.
#include "internal.h"
#include "device_util.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace aefft {

typedef float v4f_t __attribute__((ext_vector_type(4)));

template <int VEC, int TRB, int TCB, bool DIFF>
__device__ __forceinline__ void contract_mfma_body(const Contract& q, int bx, int rt, int ct, int ks, int KS, float* red)
{
    using L = BufLoad<VEC>;
    using V = typename L::T;
    // Two lane layouts.  MEMORY layout (loads, stores, everything per-bin): lane = 16*sub + blk, so 16 consecutive lanes
    // touch 128*VEC contiguous bytes of one plane.  MFMA layout: lane = 4*blk + sub.  Operands are permuted from the
    // first to the second after the load, results back before the store (ds_bpermute, no LDS allocation): with the
    // MFMA layout used directly every lane quad straddles four planes and the vector memory path runs 3-4x slower (measured).
    const int lane = threadIdx.x, blk = lane & 15, sub = lane >> 4;
    const int to_mfma = (16 * (lane & 3) + (lane >> 2)) * 4;    // MFMA-layout lane l takes the value loaded by memory-layout lane 16*(l&3) + (l>>2)
    const int to_mem = (4 * (lane & 15) + (lane >> 4)) * 4;     // memory-layout lane l takes the result held by MFMA-layout lane 4*(l&15) + (l>>4)
    const long grp = (long)bx * 16 + blk;                       // bin group of this lane (VEC bins)
    const int r0 = rt * 4 * TRB, c0 = ct * 4 * TCB;
    const bool tile_ok = r0 < q.R && c0 < q.C;                  // uniform per wave
    const bool binok = grp * VEC < q.P;
    const long gcl = binok ? grp : (q.P - 1) / VEC;             // loads of out-of-range lanes are clamped, never stored
    long bgrp = gcl, agrp = gcl, ogrp = grp;
    bool live = true;
    if (VEC == 1 && q.gdNx) {                                   // small-grid bin -> its bin on the big grid (inverse of fft.cu:102-111)
        const int Nyr = q.gdNy / 2 + 1, Nyrs = q.gdNys / 2 + 1;
        const int di = (int)(gcl / Nyrs), dj = (int)(gcl - (long)di * Nyrs);
        const int si = di < q.gdNxs / 2 ? di : (di == q.gdNxs / 2 ? q.gdNx / 2 : di - q.gdNxs + q.gdNx);
        const int sj = dj < Nyrs - 1 ? dj : Nyr - 1;
        const long big = (long)si * Nyr + sj;
        if (q.gdMask & 1) agrp = big;
        if (q.gdMask & 2) bgrp = big;
        if (q.gdMask & 4) ogrp = big;
    }
    if (VEC == 1 && q.upNx) {                                   // B operand read through the zero-pad index map (fft.cu:117-152)
        const int Nyr = q.upNy / 2 + 1, Nyrs = q.upNys / 2 + 1;
        const int i = (int)(gcl / Nyr), j = (int)(gcl - (long)i * Nyr);
        int si = -1, sj = -1;
        if (i < q.upNxs / 2) si = i;
        else if (i > q.upNx - q.upNxs / 2) si = i - q.upNx + q.upNxs;
        else if (i == q.upNx / 2) si = q.upNxs / 2;
        if (j < Nyrs - 1) sj = j;
        else if (j == Nyr - 1) sj = Nyrs - 1;
        live = si >= 0 && sj >= 0;
        bgrp = live ? (long)si * Nyrs + sj : 0;
    }
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)q.A, 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DIFF ? q.A2 : q.A), 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)q.B, 0, 0xFFFFFFFFu, 0x00020000);
    unsigned aoff[TRB], boff[TCB];
#pragma unroll
    for (int t = 0; t < TRB; ++t) { const int rr = (r0 + 4 * t + sub < q.R) ? r0 + 4 * t + sub : q.R - 1; aoff[t] = (unsigned)((rr * q.a_r + agrp * VEC) * 8); }
#pragma unroll
    for (int u = 0; u < TCB; ++u) { const int cc = (c0 + 4 * u + sub < q.C) ? c0 + 4 * u + sub : q.C - 1; boff[u] = (unsigned)((cc * q.b_c + bgrp * VEC) * 8); }
    const unsigned a_ks = (unsigned)(q.a_k * 8), b_ks = (unsigned)(q.b_k * 8);

    v4f_t re[VEC][TRB][TCB], im[VEC][TRB][TCB];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int t = 0; t < TRB; ++t)
#pragma unroll
            for (int u = 0; u < TCB; ++u) { re[v][t][u] = v4f_t{0.f, 0.f, 0.f, 0.f}; im[v][t][u] = v4f_t{0.f, 0.f, 0.f, 0.f}; }

    // conj(a)*b form when exactly one operand is conjugated; a*conj(b) = conj(conj(a)*b) and conj(a)*conj(b) = conj(a*b)
    // are finished by negating the imaginary part in the epilogue.
    const bool cj = q.conjA != q.conjB, negIm = q.conjB;
    if (tile_ok) {
        constexpr int LD = (TRB * (DIFF ? 2 : 1) + TCB) * VEC;             // 8-byte registers pairs per k
        constexpr int UNR = LD <= 4 ? 8 : (LD <= 8 ? 4 : 2);
        const int kq = (q.K + KS - 1) / KS;
        const int kbeg = ks * kq, kend = (kbeg + kq < q.K) ? kbeg + kq : q.K;
        auto group = [&](int k0, auto NU) {
            constexpr int U = decltype(NU)::value;            // straight-line: all loads of U iterations, then their MFMAs
            V a[U][TRB], b[U][TCB], a2[DIFF ? U : 1][DIFF ? TRB : 1];
#pragma unroll
            for (int s = 0; s < U; ++s) {
                const unsigned sa = (k0 + s) * a_ks, sb = (k0 + s) * b_ks;
#pragma unroll
                for (int t = 0; t < TRB; ++t) a[s][t] = L::ld(ra, aoff[t], sa);
#pragma unroll
                for (int u = 0; u < TCB; ++u) b[s][u] = L::ld(rb, boff[u], sb);
                if (DIFF) {
#pragma unroll
                    for (int t = 0; t < TRB; ++t) a2[s][t] = L::ld(ra2, aoff[t], sa);
                }
            }
#pragma unroll
            for (int s = 0; s < U; ++s)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float brp[TCB], bip[TCB];
#pragma unroll
                    for (int u = 0; u < TCB; ++u) {
                        const float* bf = reinterpret_cast<const float*>(&b[s][u]);
                        float br = bf[2 * v], bi = bf[2 * v + 1];
                        if (VEC == 1) { br = live ? br : 0.f; bi = live ? bi : 0.f; }
                        brp[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(to_mfma, __float_as_int(br)));
                        bip[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(to_mfma, __float_as_int(bi)));
                    }
#pragma unroll
                    for (int t = 0; t < TRB; ++t) {
                        const float* af = reinterpret_cast<const float*>(&a[s][t]);
                        float ar = af[2 * v], ai = af[2 * v + 1];
                        if (DIFF) { const float* a2f = reinterpret_cast<const float*>(&a2[s][t]); ar -= a2f[2 * v]; ai -= a2f[2 * v + 1]; }
                        ar = __int_as_float(__builtin_amdgcn_ds_bpermute(to_mfma, __float_as_int(ar)));
                        ai = __int_as_float(__builtin_amdgcn_ds_bpermute(to_mfma, __float_as_int(ai)));
                        const float sre = cj ? ai : -ai, sim = cj ? -ai : ai;
#pragma unroll
                        for (int u = 0; u < TCB; ++u) {
                            const float br = brp[u], bi = bip[u];
                            re[v][t][u] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar, br, re[v][t][u], 0, 0, 0);
                            im[v][t][u] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar, bi, im[v][t][u], 0, 0, 0);
                            re[v][t][u] = __builtin_amdgcn_mfma_f32_4x4x1f32(sre, bi, re[v][t][u], 0, 0, 0);
                            im[v][t][u] = __builtin_amdgcn_mfma_f32_4x4x1f32(sim, br, im[v][t][u], 0, 0, 0);
                        }
                    }
                }
        };
        int k0 = kbeg;
        for (; k0 + UNR <= kend; k0 += UNR) group(k0, std::integral_constant<int, UNR>{});
        // remainder in halving groups (4, 2, 1) instead of one iteration -- one memory round trip -- at a time
        if (UNR >= 8 && k0 + 4 <= kend) { group(k0, std::integral_constant<int, 4>{}); k0 += 4; }
        if (UNR >= 4 && k0 + 2 <= kend) { group(k0, std::integral_constant<int, 2>{}); k0 += 2; }
        for (; k0 < kend; ++k0) group(k0, std::integral_constant<int, 1>{});
    }
    if (KS > 1) {
        // split-K: waves 1..KS-1 park their partial tiles in LDS ([slice][value][lane]: conflict-free), wave 0 sums them in slice order
        constexpr int NV = VEC * TRB * TCB * 8;
        if (ks > 0) {
#pragma unroll
            for (int v = 0; v < VEC; ++v)
#pragma unroll
                for (int t = 0; t < TRB; ++t)
#pragma unroll
                    for (int u = 0; u < TCB; ++u)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int e = (((v * TRB + t) * TCB + u) * 4 + i) * 2;
                            red[((ks - 1) * NV + e) * 64 + lane] = re[v][t][u][i];
                            red[((ks - 1) * NV + e + 1) * 64 + lane] = im[v][t][u][i];
                        }
        }
        __syncthreads();
        if (ks > 0) return;
        for (int s2 = 0; s2 < KS - 1; ++s2)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
#pragma unroll
                for (int t = 0; t < TRB; ++t)
#pragma unroll
                    for (int u = 0; u < TCB; ++u)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int e = (((v * TRB + t) * TCB + u) * 4 + i) * 2;
                            re[v][t][u][i] += red[(s2 * NV + e) * 64 + lane];
                            im[v][t][u][i] += red[(s2 * NV + e + 1) * 64 + lane];
                        }
    }
    // results back to the memory layout: lane (sub, blk) = column c0 + 4u + sub of bin blk, rows r0 + 4t + i in register i
    if (tile_ok) {
#pragma unroll
        for (int v = 0; v < VEC; ++v)
#pragma unroll
            for (int t = 0; t < TRB; ++t)
#pragma unroll
                for (int u = 0; u < TCB; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        re[v][t][u][i] = __int_as_float(__builtin_amdgcn_ds_bpermute(to_mem, __float_as_int(re[v][t][u][i])));
                        im[v][t][u][i] = __int_as_float(__builtin_amdgcn_ds_bpermute(to_mem, __float_as_int(im[v][t][u][i])));
                    }
    }
    const bool store_ok = tile_ok && binok;
    const int col_base = c0 + sub;

    if (q.mse.acc) {
        // tile = pair-local reconstruction O_b[d'] (rows d' = r, cols b = c) compared with X_b[d'] = B[k = r][c]; nothing is stored
        float part = 0.f;
        if (tile_ok) {
            float2 beta[TRB][4];
#pragma unroll
            for (int t = 0; t < TRB; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) beta[t][i] = make_float2(0.f, 0.f);
            if (bx == 0) {                                      // this wave holds the DC bin: wave-parallel bias terms of conv_k o conv_k
#pragma unroll
                for (int t = 0; t < TRB; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int rr = (r0 + 4 * t + i < q.R) ? r0 + 4 * t + i : q.R - 1;
                        float sx = 0.f, sy = 0.f;
                        for (int m = lane; m < q.mse.dM; m += 64) {
                            const float2 f = q.mse.F[((long)rr * q.mse.dM + m) * q.P];
                            const float bb = q.mse.b[m];
                            sx = fmaf(f.x, bb, sx); sy = fmaf(f.y, bb, sy);
                        }
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) { sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); }
                        beta[t][i] = make_float2((sx / (float)q.R + q.mse.p[rr]) * q.mse.norm, sy / (float)q.R * q.mse.norm);
                    }
            }
            float w[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const unsigned j = (unsigned)((gcl * VEC + v) % q.mse.Nyr);
                w[v] = !binok ? 0.f : ((j > 0 && j < (unsigned)q.mse.Nyr - 1) ? 2.f : 1.f);   // Hermitian half-plane: interior columns count twice
            }
#pragma unroll
            for (int t = 0; t < TRB; ++t)
#pragma unroll
                for (int u = 0; u < TCB; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = r0 + 4 * t + i, col = col_base + 4 * u;
                        if (row >= q.R) continue;                                            // uniform
                        const V tv = L::ld(rb, boff[u], (unsigned)row * b_ks);
                        const float* tf = reinterpret_cast<const float*>(&tv);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            float ox = re[v][t][u][i], oy = negIm ? -im[v][t][u][i] : im[v][t][u][i];
                            if (v == 0 && grp == 0) { ox += beta[t][i].x; oy += beta[t][i].y; }
                            const float ex = tf[2 * v] - ox, ey = tf[2 * v + 1] - oy;
                            part = fmaf(col < q.C ? w[v] : 0.f, ex * ex + ey * ey, part);
                        }
                    }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        const float sc = q.mse.scale / q.mse.nfull;
        float* slot = q.mse.acc + (blockIdx.x % MSE_SLOTS) * MSE_SLOT_STRIDE;
        if (KS > 1) { if (lane == 0 && part != 0.f) atomicAdd(slot, part * sc); return; }
        if (lane == 0) red[threadIdx.y] = part;
        __syncthreads();
        if (lane == 0 && threadIdx.y == 0) {
            float s = 0.f;
            for (int y = 0; y < (int)blockDim.y; ++y) s += red[y];
            if (s != 0.f) atomicAdd(slot, s * sc);
        }
        return;
    }

    if (!store_ok) return;
    const float bmul = q.preDivB != 0.f ? 1.0f / q.preDivB : 1.0f;
    const float omul = q.postDiv != 0.f ? 1.0f / q.postDiv : 1.0f;
    long cdst[VEC];                      // fused down-sampling: destination bin of each of this lane's bins (or -1)
#pragma unroll
    for (int v = 0; v < VEC; ++v) cdst[v] = q.Out2 ? crop_dest(grp * VEC + v, q.dnNx, q.dnNy, q.dnNxs, q.dnNys) : -1;
    const long Ps = (long)q.dnNxs * (q.dnNys / 2 + 1);
    const long orp = q.Out2 ? q.o_r / q.P : 0, ocp = q.Out2 ? q.o_c / q.P : 0;
    V* Op = reinterpret_cast<V*>(q.Out);
#pragma unroll
    for (int t = 0; t < TRB; ++t)
#pragma unroll
        for (int u = 0; u < TCB; ++u) {
            const int col = col_base + 4 * u;
            if (col >= q.C) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = r0 + 4 * t + i;
                if (row >= q.R) continue;                                                    // uniform
                V o;
                float* of = reinterpret_cast<float*>(&o);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float x = re[v][t][u][i] * bmul, y = (negIm ? -im[v][t][u][i] : im[v][t][u][i]) * bmul;
                    if (v == 0 && q.bias && grp == 0 && (q.biasColP1 == 0 || col == q.biasColP1 - 1)) x += q.bias[row] * q.biasScale;
                    of[2 * v] = x * omul; of[2 * v + 1] = y * omul;
                }
                V* dst = Op + (row * q.o_r + col * q.o_c) / VEC + ogrp;
                if (q.accumulate) {
                    const V prev = *dst;
                    const float* pf = reinterpret_cast<const float*>(&prev);
#pragma unroll
                    for (int e2 = 0; e2 < 2 * VEC; ++e2) of[e2] += pf[e2];
                }
                *dst = o;
                if (q.Out2) {
                    const long plane = (long)row * orp + (long)col * ocp;
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        if (cdst[v] >= 0) q.Out2[plane * Ps + cdst[v]] = make_float2(of[2 * v], of[2 * v + 1]);
                }
            }
        }
}

// XCD-aware workgroup order (see xcd_decode): bin tile bx goes to XCD bx % 8 and the gy workgroups that re-read the
// same bins run back to back there.  With fewer than `xmin` bin tiles the XCDs cannot be balanced: plain order.
__device__ __forceinline__ BlockId mfma_decode(int lin, int gx, int gy, int xmin)
{
    BlockId b; b.bz = 0;
    if (gx < xmin) { b.bx = lin % gx; b.by = lin / gx; b.ok = b.by < gy; return b; }
    const int xcd = lin & 7, slot = lin >> 3;
    b.bx = (slot / gy) * 8 + xcd;
    b.by = slot - (slot / gy) * gy;
    b.ok = b.bx < gx;
    return b;
}
static inline long mfma_grid(long gx, int gy, int xmin) { return gx < xmin ? gx * gy : ((gx + 7) / 8) * 8 * gy; }

// One launch serves up to 8 problems (ContractN): each owns a contiguous, 8-aligned range of linear workgroup ids so
// that the XCD-aware decode (bin tile slowest, see xcd_decode) keeps working inside its range.  A workgroup is 4 waves:
// four row tiles of one (bin tile, column tile), or -- split-K -- the four K quarters of one tile.
template <int VEC, int TRB, int TCB, bool DIFF>
__global__ __launch_bounds__(256) void contract_mfma_kernel(const ContractN g)
{
    extern __shared__ float red[];
    const int lin = blockIdx.x;
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < g.n && lin >= g.start[i]) p = i;
    const BlockId b = mfma_decode(lin - g.start[p], g.gx[p], g.gy[p], g.xmin);
    if (!b.ok) return;                                   // uniform per workgroup
    const int KS = g.ks[p];
    const int ks = KS > 1 ? threadIdx.y : 0;
    const int tile = KS > 1 ? b.by : b.by * blockDim.y + threadIdx.y;      // (row tile, column tile) linearised, row tile fastest
    const int rtiles = g.gz[p];
    const Contract q = g.q[p];                             // one bulk scalar load of the descriptor instead of a load per field use
    contract_mfma_body<VEC, TRB, TCB, DIFF>(q, b.bx, tile % rtiles, tile / rtiles, ks, KS, red);
}

template <int VEC, int TRB, int TCB, bool DIFF> static hipError_t contract_mfma_tile(ContractN& g, hipStream_t st)
{
    const int by = 4;
    long total = 0;
    size_t lds = 16;
    for (int p = 0; p < g.n; ++p) {
        const Contract& q = g.q[p];
        const long groups = (q.P + VEC - 1) / VEC;
        const int rtiles = (q.R + 4 * TRB - 1) / (4 * TRB), ctiles = (q.C + 4 * TCB - 1) / (4 * TCB);
        g.gx[p] = (int)((groups + 15) / 16);
        g.gy[p] = g.ks[p] > 1 ? rtiles * ctiles : (rtiles * ctiles + by - 1) / by;
        g.gz[p] = rtiles;                                   // (the grid itself is 2-D: bin tiles x tile groups)
        g.start[p] = (int)total;
        total += (mfma_grid(g.gx[p], g.gy[p], g.xmin) + 7) / 8 * 8;
        if (g.ks[p] > 1) lds = std::max(lds, sizeof(float) * (g.ks[p] - 1) * VEC * TRB * TCB * 8 * 64);
    }
    if (total >= (1L << 31)) return hipErrorInvalidValue;
    g.start[g.n] = (int)total;
    contract_mfma_kernel<VEC, TRB, TCB, DIFF><<<dim3((unsigned)total), dim3(64, by), lds, st>>>(g);
    return hipGetLastError();
}

// true when every problem of the launch can run on the MFMA kernel
static bool mfma_eligible(const Contract& q)
{
    if (q.R <= 0 || q.C <= 0 || q.K <= 0 || q.P <= 0) return false;
    const double Pbig = q.gdNx ? (double)q.gdNx * (q.gdNy / 2 + 1) : (double)q.P;
    if (q.gdNx && (q.gdNxs > q.gdNx || q.gdNys > q.gdNy || !q.gdMask)) return false;
    const double a = ((double)(q.R - 1) * q.a_r + (double)(q.K - 1) * q.a_k + Pbig) * 8.0;
    const double b = ((double)(q.C - 1) * q.b_c + (double)(q.K - 1) * q.b_k + Pbig) * 8.0;
    if (a >= 4.0e9 || b >= 4.0e9) return false;                     // 32-bit buffer offsets
    if (q.bias && !q.biasAfterFirst) return false;
    if (q.mse.acc && (q.R != q.K || q.conjA || q.conjB || q.A2 || q.upNx || q.Out2)) return false;
    if (q.Out2 && (q.o_r % q.P || q.o_c % q.P)) return false;
    if (q.gdNx && (q.upNx || q.Out2 || q.mse.acc || q.P != (long)q.gdNxs * (q.gdNys / 2 + 1))) return false;
    return true;
}

hipError_t launch_contract_mfma(ContractN& g, hipStream_t st)
{
    if (g.n < 1 || g.n > 8) return hipErrorInvalidValue;
    bool even = true, diff = g.q[0].A2 != nullptr;
    int Rmax = 0, Cmax = 0, Kmin = 1 << 30;
    long Pmin = 1L << 60;
    for (int p = 0; p < g.n; ++p) {
        const Contract& q = g.q[p];
        if (!mfma_eligible(q) || (q.A2 != nullptr) != diff) return hipErrorInvalidValue;
        even = even && !((q.P & 1) || (q.a_r & 1) || (q.a_k & 1) || (q.b_k & 1) || (q.b_c & 1) || (q.o_r & 1) || (q.o_c & 1)) && !q.upNx && !q.gdNx;
        Rmax = std::max(Rmax, q.R); Cmax = std::max(Cmax, q.C); Kmin = std::min(Kmin, q.K); Pmin = std::min(Pmin, q.P);
    }
    // Tile choice, from a sweep of every contraction of the cfg3 step on MI355X (tools_sweep.py): these launches are
    // bound by per-wave fixed costs and latency, not by the MFMA rate, so small tiles (many waves) win everywhere;
    // 16-byte loads pay once there are enough waves without them; split-K pays for long K over few, small outputs.
    long w2 = 0;
    for (int p = 0; p < g.n; ++p) { const Contract& q = g.q[p]; w2 += ((q.P + 31) / 32) * ((q.R + 3) / 4) * ((q.C + 3) / 4); }
    int vec = (even && w2 >= 2048) ? 2 : 1;
    bool anysplit = false;
    for (int p = 0; p < g.n; ++p) {
        const Contract& q = g.q[p];
        g.ks[p] = (q.K >= 32 && (long)q.R * q.C <= 1024 && q.P <= 4096) ? 4 : 1;
        anysplit = anysplit || g.ks[p] > 1;
    }
    int trb = 1, tcb = (Cmax >= 8 && !anysplit && !diff) ? 2 : 1;
    // HBM-sized launches (cfg3 without pooling): 8 x 8 tiles halve the operand re-reads that the L2 has to absorb (-10 % step time)
    if (w2 >= 32768 && Rmax >= 8 && Cmax >= 8) { trb = 2; tcb = 2; }
    g.xmin = 64;
#define AEFFT_MT(V, R_, C_) if (vec == V && trb == R_ && tcb == C_) return diff ? contract_mfma_tile<V, R_, C_, true>(g, st) : contract_mfma_tile<V, R_, C_, false>(g, st);
    AEFFT_MT(2, 1, 1) AEFFT_MT(2, 2, 1) AEFFT_MT(2, 1, 2) AEFFT_MT(2, 2, 2) AEFFT_MT(2, 4, 1) AEFFT_MT(2, 1, 4) AEFFT_MT(2, 4, 2) AEFFT_MT(2, 2, 4)
    AEFFT_MT(1, 1, 1) AEFFT_MT(1, 2, 1) AEFFT_MT(1, 1, 2) AEFFT_MT(1, 2, 2) AEFFT_MT(1, 4, 1) AEFFT_MT(1, 1, 4) AEFFT_MT(1, 4, 2) AEFFT_MT(1, 2, 4) AEFFT_MT(1, 4, 4)
#undef AEFFT_MT
    return hipErrorInvalidValue;
}

}  // namespace aefft
