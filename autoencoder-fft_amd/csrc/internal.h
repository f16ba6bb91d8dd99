// Internal launch interface between the HIP kernel files and the C-ABI layer (aefft_capi.hip).
// Nothing here is exported; the public boundary is include/aefft.h and the three C++ headers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace aefft {

// Development switches (include/aefft.h AEFFT_F_*, aefft_ctx_set_flags): one process-wide word, written by the setter (and once
// from the AEFFT_FLAGS environment variable when the first context is created), read by the launchers.  No getenv afterwards.
extern unsigned dev_flags;
inline bool flag(unsigned f) { return (dev_flags & f) != 0; }

// ---- fft_kernels.hip -------------------------------------------------------------------
// Twiddle table W[k] = exp(-2*pi*i*k/TW_N), uploaded once per device.
constexpr int TW_N = 4096;
hipError_t upload_twiddles(hipStream_t st);
bool fft_size_supported(int n);

// Batched 2-D R2C with optional fused spectral crop (== fft.cu:764 `fft` followed by
// fft.cu:87 `resize` down-sampling to Nxs x Nys).  in [planes][Nx][Ny] real ->
// out [planes][Nxs][Nys/2+1] complex.  `mid` is a workspace of planes*Nx*(Nys/2) complex.
// in_u8: `in` holds 8-bit pixels instead of floats (the row pass converts on load).
hipError_t launch_r2c(const void* in, float2* out, float2* mid, long planes, int Nx, int Ny,
                      int Nxs, int Nys, hipStream_t st, hipEvent_t done = nullptr /* recorded by the column pass's own completion signal */, bool in_u8 = false);
// Batched 2-D C2R with optional fused spectral zero-pad (== `resize` up-sampling from
// Nxi x Nyi, then cufftExecC2R, then * scale).  in [planes][Nxi][Nyi/2+1] -> out [planes][Nx][Ny].
// opin (nullable): the input spectra are not stored but evaluated from an operator (see inv_cols_kernel): plane (b, d) at bin t is
// A[OPIN_COLS-1][d][t] + sum_{j<D0} A[j][d][t] * Xf[b][j][u(t)], A [OPIN_COLS][D0][Nxi*(Nyi/2+1)], Xf [B][D0][Nx0*(Ny0/2+1)]
constexpr int OPIN_COLS = 4;
constexpr int OPMSE_PACKED_STEPS = 16;  // steps of the innermost pair's two stages in the post-update MSE (opmse_packed)
constexpr int CH_MAXSTEPS = 40;       // most steps of a per-bin chain item (chain_geometry's table; cfg5: 26)
constexpr int CH_VMAX = 128;          // most rows of a matrix the packed-record kernels (chain, innermost-pair MSE) take
struct OpIn { const float2* A; const float2* Xf; int D0, Nx0, Ny0; };
hipError_t launch_c2r(const float2* in, float* out, float2* mid, long planes, int Nxi, int Nyi,
                      int Nx, int Ny, float scale, hipStream_t st, const OpIn* opin = nullptr);
size_t fft_mid_elems(long planes, int Nx, int Wc);   // complex elements needed in `mid`
// sizes that are not powers of two (cufftPlanMany takes any size, fft.cu:773-779): even n in 8..1024 through Bluestein's chirp-z form on the
// power-of-two LDS passes; rows -> transpose -> rows -> transpose.  w1, w2: workspaces of fft_any_ws_elems complex each.
bool fft_size_supported_any(int n);
size_t fft_any_ws_elems(long planes, int Nx, int Ny);
hipError_t launch_r2c_any(const float* in, float2* out, float2* w1, float2* w2, long planes, int Nx, int Ny, hipStream_t st);
hipError_t launch_c2r_any(const float2* in, float* out, float2* w1, float2* w2, long planes, int Nx, int Ny, float scale, hipStream_t st);
// sizes that are not powers of two (cufftPlanMany takes any size, fft.cu:773-779): even n in 8..1024 through Bluestein's chirp-z form on the
// power-of-two LDS passes; rows -> transpose -> rows -> transpose.  w1, w2: workspaces of fft_any_ws_elems complex each.
bool fft_size_supported_any(int n);
size_t fft_any_ws_elems(long planes, int Nx, int Ny);
hipError_t launch_r2c_any(const float* in, float2* out, float2* w1, float2* w2, long planes, int Nx, int Ny, hipStream_t st);
hipError_t launch_c2r_any(const float2* in, float* out, float2* w1, float2* w2, long planes, int Nx, int Ny, float scale, hipStream_t st);
// the same from a SMALL stored spectrum (the reconstruction's compact support) in one launch: the column pass as a direct Nxi-term sum
// inside the row-pass workgroups, no `mid` (c2r_small_kernel)
bool c2r_small_supported(int Nxi, int Nyi, int Nx, int Ny);
hipError_t launch_c2r_small(const float2* in, float* out, long planes, int Nxi, int Nyi, int Nx, int Ny, float scale, hipStream_t st);

// ---- spectral_kernels.hip --------------------------------------------------------------
// Per-bin complex contraction  Out[r][c][bin] = alpha * sum_k opA(A[r][k][bin]) * opB(B[k][c][bin])
// (+ bias[r]*biasScale added to Re at bin 0), the one primitive behind conv_k (fft.cu:162) and
// both halves of gradient_k_io (fft.cu:395).  Strides are in complex elements.
struct Contract {
    const float2* A; long a_r, a_k;     // A[r][k] plane base = A + r*a_r + k*a_k
    const float2* A2;                   // optional: the A operand is (A - A2), same strides (E = O - T fused, fft.cu:417-424)
    const float2* B; long b_k, b_c;     // B[k][c] plane base = B + k*b_k + c*b_c
    float2* Out;     long o_r, o_c;     // Out[r][c] plane base
    int R, C, K;                        // rows, cols, contraction length
    long P;                             // bins per (A / Out) plane
    bool conjA, conjB;
    float preDivB;                      // !=0: B elements are divided by this BEFORE the product (conv_k: in/dM, fft.cu:176-177)
    float postDiv;                      // !=0: the sum is divided by this (gradient_k_io: /Norm, fft.cu:440-441)
    const float* bias; float biasScale; // Re(Out[r][c][0]) += bias[r]*biasScale; null = none
    bool biasAfterFirst;                // true: added right after the k==0 term (fft.cu:183-184); false: after the sum (fft.cu:454)
    int biasColP1;                      // 0: the bias goes to every column c; k+1: to column k only (operator form: the affine column)
    // Virtual spectral up-sampling of the B operand (fft.cu:117-152 fused into the consumer): B planes
    // are [upNxs][upNys/2+1] spectra that are zero-padded on the fly to the [upNx][upNy/2+1] grid of A/Out.
    // Destination bins outside the padded support only receive the bias term (everything else is 0).
    int upNx, upNy, upNxs, upNys;       // upNx == 0: no remap
    // Fused spectral down-sampling of the OUTPUT (fft.cu:98-113): besides Out, every output bin that survives the
    // crop from [dnNx][dnNy/2+1] to [dnNxs][dnNys/2+1] is also written to Out2 (same [r][c] plane order, small planes).
    float2* Out2; int dnNx, dnNy, dnNxs, dnNys;   // Out2 == null: off
    // Small-grid iteration (matrix-core kernel only): P counts the bins of the SMALL grid [gdNxs][gdNys/2+1]; bin s of it
    // corresponds to bin map(s) of the [gdNx][gdNy/2+1] grid (the index map of pool_fft, fft.cu:102-111 / 117-152).
    // gdMask selects which tensors live on the BIG grid and are addressed at map(s): 1 = A (and A2), 2 = B, 4 = Out.
    // Uses: conv_k + down-sampling without the discarded bins (A|B big, Out small); the decoder on the support of the
    // up-sampled spectra (A big); gradient terms of an up-sampled operand (B, Out big).  gdNx == 0: off.
    int gdNx, gdNy, gdNxs, gdNys, gdMask;
    bool accumulate;                    // Out += result (the element is owned by one lane: no race)
    // MSE epilogue instead of a store (mse.acc != null; needs R == K): with A = G[d'][d] = (F.C)/(dM*dD) of one pair and
    // B = X (the pair's input spectra) the tile holds the pair-local reconstruction O_b[d'] = sum_d G[d'][d] X_b[d] + beta[d']
    // at the DC bin (beta = Nx*Ny*(p[d'] + sum_m F[d'][m](0,0) b[m] / dD): the two bias terms of conv_k o conv_k,
    // fft.cu:183-184), and *mse.acc += scale * sum |X_b[d'] - O_b[d']|^2 / n_bin   (mse_fft, fft.cu:480-498,1188-1190).
    // mse.acc -> MSE_SLOTS accumulators, MSE_SLOT_STRIDE floats apart (zero before the first use; launch_mse_finish sums and clears them).
    struct Mse { float* acc; const float2* F; const float* b; const float* p; int dM, Nyr; float nfull, scale, norm; } mse;
};
enum { MSE_SLOTS = 256, MSE_SLOT_STRIDE = 16 };
struct BetaArgs { float* beta; const float2* F; const float *b, *p; int dM, dD; long P; };     // beta == null: off
hipError_t launch_mse_finish(float* slots /*[L][MSE_SLOTS*MSE_SLOT_STRIDE]*/, float* out /*[L], accumulated*/, float* copy /*[L] nullable*/, int L, hipStream_t st,
                             const BetaArgs* beta = nullptr, float* copy2 = nullptr /*[2L] nullable: the tail of the packed gradient buffer -- [l] <- out[l] after [L + l] <- [l] * prev_scale*/,
                             float prev_scale = 1.0f);
struct Contract2 { Contract q[2]; int n; };   // up to two independent contractions in one launch (grid.z is split)
hipError_t launch_contract2(const Contract2& qq, hipStream_t st);
// Up to 8 independent contractions of one class in ONE launch (the four pairs' S / dc,df / re-forward convs):
// cls 0 = plain conv_k, 1 = S (a*conj(b), fused subtraction), 2 = first nA problems conj(a)*b (dc), the rest a*conj(b) (df).
struct ContractN { Contract q[8]; int n, nA; int gx[8], gy[8], gz[8], start[9], ks[8], xmin; };
// matrix-core (MFMA 4x4x1, block = bin) kernel for any mix of classes; hipErrorInvalidValue = not served, use the scalar kernels
hipError_t launch_contract_mfma(ContractN& g, hipStream_t st);
hipError_t launch_contract_group(ContractN& g, int cls, hipStream_t st);
hipError_t launch_contract(const Contract& q, hipStream_t st);

hipError_t launch_resize(const float2* in, float2* out, long planes, int Nx, int Ny, int Nxs, int Nys, hipStream_t st);
hipError_t launch_magnitude(const float2* X, float* mag, long planes, int ch, int Nx, int Ny, int shift, hipStream_t st);   // fft.cu:27-63
// E = O - T and MSE (fft.cu:480-498): *mse_acc += scale * sum_bins |T-O|^2/n_bin.
hipError_t launch_diff_mse(const float2* T, const float2* O, float2* E /*nullable*/, float* mse_acc /*1 float, pre-zeroed, nullable*/,
                           float* es /*[2*ch] floats, pre-zeroed, nullable: sum_b E_b[d](0,0)*/, int B, int ch, int Nx, int Ny, float scale, hipStream_t st);
// db, dp and the DC-bin bias term of df from es[d] = sum_b (O_b[d] - T_b[d])(0,0)  (fft.cu:448-455,463-473)
hipError_t launch_bias_grad(const float2* O, const float2* T, const float2* F, const float* b, float2* df, float* db, float* dp,
                            int B, int dM, int dD, long P, float norm, float Norm, hipStream_t st);

struct BiasGradArgs { const float2 *O, *T, *F; const float* b; float2* df; float *db, *dp; int B, dM, dD; long P; float norm, Norm; long PO; /* plane stride of O (== P unless O is stored on its support only) */
                      float* es_out; /* nullable: [2*dD] floats, es[d] = sum_b (O_b[d] - T_b[d])(0,0) */
                      const float* es_in; /* nullable: es already known (operator form, sgrad_kernel): O and T are not read */ };
struct BiasGradGroup { BiasGradArgs a[8]; int n; int start[9], fix[8]; };
hipError_t launch_bias_grad_group(BiasGradGroup& g, hipStream_t st);

// ---- pruned_kernels.hip ----------------------------------------------------------------
// Kernel-support-pruned transforms: only Nk x Nl taps are non-zero going forward / needed coming back,
// so pad+R2C and C2R+shrink become direct DFT evaluations (fft.cu:1219-1226 and :1274-1282 fused).
bool pruned_supported(int Nk, int Nl, int Nx, int Ny);
hipError_t launch_kspec(const float* k, float2* K, const float2* tw, long planes, int Nx, int Ny, int Nk, int Nl, hipStream_t st);
hipError_t launch_kgrad(const float2* D, float* g, float* part /*workspace: kgrad_partial_floats()*/, const float2* tw, long planes, int Nx, int Ny, int Nk, int Nl, float scale, hipStream_t st);
size_t kgrad_partial_floats(long planes, int Nx, int Ny, int Nk, int Nl);
// The same transforms for up to 8 problems with equal (Nk, Nl) in ONE launch (kgrad: no row chunks, so no ksum pass).
struct PrunedProb { const void* src; void* dst; long planes; int Nx, Ny; float scale;
                    int NxB, NyB; /* forward transform only: the grid the phases are taken on (0: the plane's own); rows / columns of the
                                     [Nx][Ny/2+1] output are then the images of pool_fft's crop in it (map_up_row / map_up_col) */ };
// forward transform only: the problem's planes are G' = F'.C'/(dM dD) [dD][dD] of a pair, their (2Nk-1)^2 taps formed inside the launch from
// c | f (read through the problem's TapUpd); f == null: an ordinary problem (taps at PrunedProb::src)
struct GtapSrc { const float* c; const float* f; int dM, dD; float scale; };
// the taps of G' = F.C / (dM dD) of up to 8 pairs, [dD*dD][(2Nk-1)^2] each, in one launch (HBM-sized grids: every plane is transformed by several
// row-chunk workgroups, which would each form the same taps again -- 238 of the 538 us of the G' launch at cfg3-P1); their spectra are then an
// ordinary pruned transform of (2Nk-1)^2-tap kernels: launch_kspec_group_taps
struct GtapsGroup { GtapSrc gs[8]; float* out[8]; int n; int start[9]; };
hipError_t launch_gtaps_group(GtapsGroup& g, int Nk, hipStream_t st);
struct PrunedGroup;
hipError_t launch_kspec_group_taps(PrunedGroup& g, const float2* tw, int T, hipStream_t st);      // spectra of stored T x T-tap kernels, T = 5 or 9
// forward pruned transform only: the taps are read THROUGH the pending clipped-momentum update (w - clip_step(g*gscale, D)), which
// another launch stores afterwards (update_device.h) -- g == null: taps as stored
struct TapUpd { const float* g; const float* D; float del, alpha, gscale; };
struct BiasUpd { float *b, *p, *Db, *Dp; const float *db, *dp; float* zero; int dM, dD; };
struct BiasUpdGroup { BiasUpd a[8]; int n; float del, alpha, gscale; };      // n == 0: none
struct PrunedGroup {
    PrunedProb q[8]; int n; int start[9], ppb[8], rows[8], pblocks[8];
    TapUpd upd[8];
    GtapSrc gsrc[8];
    // grouped inverse transform only: rows per slice; chunks[p] in: row chunks dst has room for ([planes][chunks][taps], 0/1 = none),
    // out: the chunks the launch used (the consumer adds them in order)
    int rb[8], chunks[8];
};
int kgrad_group_chunks(long planes, int Nx, int Ny);
struct PackArgs;
hipError_t launch_kspec_group(PrunedGroup& g, const float2* tw, int Nk, int Nl, hipStream_t st, PackArgs* packed = nullptr /* the bin-major copy rides along as extra workgroups */,
                              const BiasUpdGroup* bias_upd = nullptr /* the bias half of a fused update as trailing workgroups */);
hipError_t launch_kgrad_group(PrunedGroup& g, const float2* tw, int Nk, int Nl, hipStream_t st);
// the inverse transform on a T x T support with T = 5 or 9 (the offsets kl + k'l' of 3x3 / 5x5 kernels, weight_kernels.hip)
hipError_t launch_kgrad_group_taps(PrunedGroup& g, const float2* tw, int T, hipStream_t st, BiasGradGroup* bias = nullptr /* fused: the DC-bin terms as extra workgroups */);

// ---- weight_kernels.hip ----------------------------------------------------------------
struct WgradProb { const float *c, *f, *Q, *es, *b; float *gc, *gf; int dM, dD; float inv_den, norm; int nq; };   // Q [dD][dD][nq][T*T] (nq row-chunk partial sums, added here), es [2*dD]
struct WgradGroup { WgradProb q[8]; int n; int start[9];
                    // nullable: the slot sums of the PREVIOUS step's post-update MSE (launch_mse_finish deferred by aefft_net_step_apply(mse_d = NULL)) as one trailing workgroup
                    float *fin_slots, *fin_out, *fin_tail; int fin_L; float fin_scale; };
hipError_t launch_wgrad_taps_group(WgradGroup& g, int Nk, hipStream_t st);

const float2* twiddle_table();   // device address of the table uploaded by upload_twiddles()

// ---- opform_kernels.hip ----------------------------------------------------------------
// Operator form of the training step.  The FFT-mode network is linear (identity activation,
// backproplib.cu:38-51; conv_k and pool_fft are linear maps), so every per-frame spectrum of a step is an affine function of
// the frame's own input spectrum x_b (D0 <= 3 channels on pair 0's grid):  X_l,b = A_l [x_b; 1],  O_l,b = O^_l [x_b; 1],
// with per-bin operators A_l [dD_l x OPC] (column j < D0: response to the unit input on channel j, column OPC-1: response
// to the zero input = the bias terms).  The operators come out of the ordinary forward run on OPC "basis frames"; the batch
// enters only through the input transform, the second moments M^[u] = sum_b [x_b;1][x_b;1]^H (OPC x OPC per bin of grid 0)
// and the reconstruction.  Same sums as gradient_k_io / mse_fft (fft_backproplib.cu:395-498), batch contracted first.
constexpr int OPC = 4;
hipError_t launch_basis_fill(float2* A0 /*[OPC][D0][P0]*/, int D0, long P0, hipStream_t st);
struct OpPair {
    const float2* A;        // A_l    [OPC][dD][P]      (the pair's input on the basis frames)
    const float2* O;        // O^_l   [OPC][dD][PO]     (its decoder output on the basis frames, stored on the grid [NxO][NyO/2+1])
    float2* S;              // out:   [dD][dD][P]       S = sum_b (O_b - X_b) X_b^H  (mk_S layout)
    float* es;              // out:   [2*dD]            es[d] = sum_b (O_b - X_b)[d](0,0)
    int dD, Nx, Ny, NxO, NyO; long P, PO;
};
// (msgrad_kernel: the batch moments are formed from the input spectra Xf [B][D0][P0] inside the launch and stored to Mout [OPC*OPC][P0] in their
// CENTRED form -- layout in opform_kernels.hip)
struct SgradGroup { OpPair q[8]; int n; int start[9], bt[8]; const float2* Xf; float2* Mout; int B, D0, Nx0, Ny0; long P0; };
hipError_t launch_msgrad_group(SgradGroup& g, hipStream_t st);
// O_0,b = O^_0 [x_b; 1] on the grid O^_0 is stored on: Of [B][D0][PO]
hipError_t launch_recon_expand(const float2* O0, const float2* Xf, float2* Of, int B, int D0, int Nx0, int Ny0, int NxO, int NyO, hipStream_t st);
// X_l,b = A_l [x_b; 1] for get_layer-style exports: out [B][dD][P] on the grid [Nx][Ny/2+1] of A
hipError_t launch_op_expand(const float2* A, const float2* Xf, float2* out, int B, int D0, int dD, int Nx0, int Ny0, int Nx, int Ny, hipStream_t st);
struct OpMsePair {
    const float2 *A, *C, *F;   // A_l [OPC][dD][P]; the UPDATED kernel spectra C [dM][dD][P], F [dD][dM][P]
    const float2* G;           // nullable: G' = F.C/(dM dD) [dD][dD][P] of the updated weights -- then C and F are not read (opmse_gbody) ...
    const float2* Fdc; long fdc_stride;   // ... except F at the DC bin: element (a, m) at Fdc[(a*dM + m) * fdc_stride]
    const float *b, *p;        // updated biases
    float* slots;              // MSE_SLOTS accumulators (launch_mse_finish sums them)
    int dD, dM, Nx, Ny; long P;
    float scale;               // 1 / (2 dM Nx Ny B) / (dD Nx Ny)
};
struct OpMseGroup { OpMsePair q[8]; int n; int start[9], bt[8], base; const float2* Mhat; int Nx0, Ny0; long P0;
                    const float2* Wp; int E, offC, offF; /* nullable: bin-major record of the UPDATED spectra; offsets of the innermost pair's C, F in it */
                    unsigned pst_off[OPMSE_PACKED_STEPS], pst_desc[OPMSE_PACKED_STEPS]; int pst_n; /* (filled by the launcher) that pair's two stages as a step list */ };
struct ChainArgs;
struct UpdateGroup;
// the MSE of every pair; optionally in the same launch: the operator chain of the NEXT step (chain, reading the just-written Wp / Cc, writing its own
// operator buffers) and the tap half of a fused update (weights_upd) -- tail_kernel
hipError_t launch_opmse_group(OpMseGroup& g, hipStream_t st, ChainArgs* chain = nullptr, const UpdateGroup* weights_upd = nullptr);
// the network on the basis frames in one launch (chain_kernel): per pair the spectra, biases and the operator outputs
struct ChainLevel { const float2 *C, *F; const float *b, *p; float2 *A /*[OPC][dD][P]*/, *O /*[OPC][dD][Pc]*/; int dD, dM, Nx, Ny; long P;
                    const float2* Cc; /* nullable: C sampled at the bins the NEXT level's grid lands on, [dM][dD][P of the next level] (then C is not read) */ };
struct ChainArgs {
    ChainLevel lv[8]; int L, D0;
    long Pc;
    const float2* Wp; int E;   // bin-major copy of every matrix a coarsest-grid bin needs (kspec_packed_kernel)
    int tile_start[9];         // (filled by launch_chain) first workgroup of the planar tiles of grid j
    int vt_elems;              // (filled by launch_chain) elements of one V tile of the planar part
    int tile_bt[8];            // (filled by launch_chain) bins per planar tile of grid j (8, 16 or 32)
    // (filled by launch_chain) the per-bin item's step list (chain_item_pipelined, opform_kernels.hip)
    unsigned st_off[CH_MAXSTEPS], st_desc[CH_MAXSTEPS]; int st_n;
};
hipError_t launch_chain(ChainArgs& g, hipStream_t st, hipEvent_t done = nullptr /* recorded by the dispatch itself */);
// Wp[t][E]: per bin t of the coarsest grid the elements of C_0 .. C_{L-1}, F_{L-1} .. F_0 at the bins t maps to, from the taps
struct PackSeg { const float* k; int n, lev, off; const float *g, *D; };   // taps [n][Nk*Nk] of one tensor, its pair, its element offset in a record; gradient / momentum of the same taps (TapUpd)
struct PackArgs { PackSeg seg[16]; int nseg, L, E, Nk; int Nx[8], Ny[8]; int NxC, NyC; long Pc; float2* Wp; const float2* tw;
                  unsigned char blk_seg[128]; int blk_start[128]; int nblk; /* (pack_blocks) element blocks: tensor, first element */
                  int upd; float upd_del, upd_alpha, upd_gscale; /* upd != 0: taps read through the pending update (TapUpd) */ };
void pack_blocks(PackArgs& g);
hipError_t launch_kspec_packed(PackArgs& g, hipStream_t st);

// ---- update_kernels.hip ----------------------------------------------------------------
hipError_t launch_pad(const float* ck, float* cpad, long planes, int Nx, int Ny, int Nk, int Nl, hipStream_t st);   // fft.cu:570 (zero-fills)
hipError_t launch_shrink(const float* cpad, float* ck, long planes, int Nx, int Ny, int Nk, int Nl, float scale, hipStream_t st); // fft.cu:535
struct UpdateArgs {
    float *c, *f, *b, *p;                 // weights, updated in place
    const float *dck, *dfk, *db, *dp;     // gradients (coordinate space)
    float *Dc, *Df, *Db, *Dp;             // momentum
    const float *cd, *fd, *bd, *pd;       // multiobjective gradients (null when maxdiff=0)
    int dM, dD, Nk, Nl;
    float del, alpha, w0, w1;
    float gscale;                         // gradients are multiplied by this first (1/world_size for data-parallel means)
    int sym;                              // tied weights: g = g_c[m][d] + g_f[d][m], f[d][m] <- c[m][d]
    float *ddc, *ddf, *ddb, *ddp;         // optional: record the gradients used (adapt_rate, backproplib.cu:33); may be null
    float* zero;                          // optional: one float set to 0 (the pair's MSE accumulator, saves a memset launch)
};
hipError_t launch_update(const UpdateArgs& a, hipStream_t st);
hipError_t launch_vec_add(float* out, const float* a, const float* b, long n, hipStream_t st);      // out = a + b
hipError_t launch_u8_to_f32(float* out, const unsigned char* in, long n, hipStream_t st);             // out = (float)in
struct UpdateGroup { UpdateArgs a[8]; int n; int start[9]; };
hipError_t launch_update_group(UpdateGroup& g, hipStream_t st);                                       // up to 8 pairs, one launch                                      // fft.cu:605 / 657
size_t gradient_diff_ws_floats(int dM, int dD, int Nk, int Nl);      // floats of launch_gradient_diff's workspace (chunk partial sums)
hipError_t launch_gradient_diff(const float* c, const float* f, const float* b, const float* p, float* cd, float* fd,
                                float* bd, float* pd, float* den_ws, int dM, int dD, int Nk, int Nl, hipStream_t st);   // fft.cu:709

struct GdiffProb { const float *c, *f, *b, *p; float *cd, *fd, *bd, *pd, *part; int dM, dD; int chunk, nchunks; /* (filled by the launcher) */ };
struct GdiffGroup { GdiffProb q[8]; int n; int start[9], fstart[9]; };
hipError_t launch_gradient_diff_group(GdiffGroup& g, int Nk, int Nl, hipStream_t st);                 // every pair's fft.cu:709 terms, two launches

// ---- spatial_kernels.hip ---------------------------------------------------------------
hipError_t launch_conv_spatial(const float* in, float* out, const float* c, const float* b, int B, int dD, int dM,
                               int Nx, int Ny, int Nk, int Nl, int ak, int al, float in_scale_div, int lo, hipStream_t st,
                               int pool = 0 /* fused Pool(scale) in front: `in` is [B][dD][Nx*pool][Ny*pool] */, float* pooled_out = nullptr);
struct SpatialGradArgs {
    const float *in, *out, *hin, *f;      // [dD][Nx][Ny], [dD][Nx][Ny], [dM][Nx][Ny], [dD][dM][Nk][Nl]
    float *gc, *gf, *gb, *gp;             // [dM][dD][Nk][Nl], [dD][dM][Nk][Nl], [dM], [dD]
    float *ws;                            // workspace: dM*Nx*Ny (back-conv) floats
    float *part;                          // workspace: spatial_partial_floats() floats (tiled weight-gradient partial sums); null: naive kernels
    float *rq;                            // workspace: spatial_rq_floats() floats (the error-input correlation R and its row sums); null: dC through the back-convolved error
    int B, dD, dM, Nx, Ny, Nk, Nl, ak, al;
    float Norm;
    int lo;                               // 0: GPU boundary test '>=0' (backproplib.cu:95), 1: CPU test '>0' (netlib.cpp:344)
    int tied;                             // backprop_gpu_cc: add the f-gradient into the c-gradient (backproplib.cu:466)
    // fused step (aefft_step_spatial): hin IS Conv_gpu(in; c1, b1) -- then dF, dP follow from the same error-input region sums as dC
    // (dF[d][m][t] = sum_{d2,t2} c1[m][d2][t2]/div1 R_t[d][d2][t+t2] + b1[m] S_t[d]) and the hidden layer is not read by the gradient
    const float *c1 = nullptr, *b1 = nullptr;
    float div1 = 1.f;
};
hipError_t launch_spatial_grad(const SpatialGradArgs& a, hipStream_t st);
bool spatial_regions_ok(const SpatialGradArgs& a);   // the gradient goes through the error-input region sums (what the fused step needs)
hipError_t launch_spatial_compat(const SpatialGradArgs& a, hipStream_t st);   // B-11: gf and gb as the CUDA source computes them (after launch_spatial_grad)
size_t spatial_partial_floats(int B, int dD, int dM, int Nx, int Ny, int Nk, int Nl);
size_t spatial_rq_floats(int dD, int Nk, int Nl);
hipError_t launch_pool_spatial(const float* in, float* out, long planes, int Nxi, int Nyi, int Nxo, int Nyo, int scale, hipStream_t st);   // netlib.cpp:114   // 0: shape not served by the tiled kernels

}  // namespace aefft
