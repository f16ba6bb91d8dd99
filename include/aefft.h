/* aefft.h -- flat C ABI of the MI355X-native FFT-convolution autoencoder training path.
 *
 * This is the drop-in boundary UNDER the reference's C++ operator headers.  The reference
 * (fabrii4/AutoEncoder-FFT) exposes its hot path as C++ free functions over nested std::vector
 * (source/fft_backproplib.h:5-11, source/backproplib.h:5-16, source/netlib.h:4-24); those same
 * functions are re-exported, mangled identically, by include/fft_backproplib.h, backproplib.h and
 * netlib.h of this repo, and are thin marshalling shims over the entry points below.  Every entry
 * point here names the reference function (file:line) whose arithmetic it reproduces.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types.
 *   - "_d" pointers are DEVICE pointers (16-byte aligned), "_h" are host pointers.
 *   - real tensors  : float32, [ch][Nx][Ny]   (x = first index, y contiguous; reference layout)
 *   - spectra       : interleaved complex64 (float pairs), [ch][Nx][Nyr], Nyr = Ny/2+1
 *   - encoder kernel: c[dM][dD][Nk][Nl], bias b[dM]; decoder kernel f[dD][dM][Nk][Nl], bias p[dD]
 *   - batches add an outermost [B] dimension.  B = 1 reproduces the reference call exactly.
 *   - Nx, Ny: powers of two in 8..2048; pooling scales: powers of two -- for the resident network and the per-bin ops.
 *     The transforms and the spectral resize at op level (aefft_r2c, aefft_c2r, aefft_pool, aefft_r2c_pool, aefft_unpool_c2r) also serve
 *     EVEN sizes in 8..1024 that are not powers of two (cufftPlanMany takes any size, fft_backproplib.cu:773-779) and any integer scale,
 *     sized as the reference sizes it (:980-984: int(Nx / l) in float arithmetic; the resized grid must be even).
 *   - every function returns AEFFT_OK (0) or an error code; aefft_last_error() gives the text.
 *     Work is enqueued on the context's stream; nothing blocks unless stated.
 *   - there is NO CPU fallback: every call fails with AEFFT_EHIP when no MI355X is present.
 */
#ifndef AEFFT_H
#define AEFFT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct aefft_ctx aefft_ctx;   /* device + stream + twiddle tables + workspace pool */
typedef struct aefft_net aefft_net;   /* a stacked autoencoder resident on the device       */

enum {
    AEFFT_OK = 0,
    AEFFT_EINVAL = 1,       /* bad argument (size not a power of two, misaligned pointer, ...) */
    AEFFT_EHIP = 2,         /* HIP runtime error / no device */
    AEFFT_ENOMEM = 3,
    AEFFT_ESTATE = 4        /* call order violated (e.g. train before forward) */
};

/* ---- context ------------------------------------------------------------------------------- */
/* create_stream = 0: enqueue on `hip_stream`, a hipStream_t owned by the caller (e.g. torch's
 * current stream; NULL is the legacy default stream).  create_stream = 1: the context creates
 * and owns a non-blocking stream (`hip_stream` ignored). */
int aefft_ctx_create(aefft_ctx** out, int device, void* hip_stream, int create_stream);
void aefft_ctx_destroy(aefft_ctx* ctx);
const char* aefft_last_error(const aefft_ctx* ctx);
/* Optional spatial partition of the chip for pipelined training loops (aefft_net_set_input_ready): side_cus > 0 gives the library's side
 * streams (reconstruction inverse FFT, input prefetch: bandwidth-bound) `side_cus` of the device's compute units and the context's own
 * stream (the latency-bound weight side of the step) the rest, through CU-masked HIP streams; 0 removes the partition.  Only for a context
 * that owns its stream (create_stream = 1) and before its first net is created (AEFFT_ESTATE otherwise): aefft_stream() changes. */
int aefft_ctx_partition(aefft_ctx* ctx, int side_cus);
int aefft_sync(aefft_ctx* ctx);                 /* hipStreamSynchronize on the context stream */
void* aefft_stream(aefft_ctx* ctx);             /* the hipStream_t in use */
/* Development switches: each bit turns ONE optimisation of the training step off (or forces one the shapes would not choose), so
 * that the parity tests can run every fallback against the oracle.  Process-wide (one word for every context); the default is 0.  The
 * environment variable AEFFT_FLAGS (comma-separated names without the AEFFT_F_ prefix, e.g. "NOMFMA,NOGROUP") is read ONCE, when the
 * first context is created; the library never calls getenv after that.  Its switches stay on for the life of the process:
 * aefft_ctx_set_flags sets the word to (AEFFT_FLAGS | flags).  A name in AEFFT_FLAGS that the library does not know makes
 * aefft_ctx_create fail with AEFFT_EINVAL (message on stderr) instead of silently running the default path. */
enum {
    AEFFT_F_NOLAZY = 1 << 0,      /* encoder layers on the full grid even when only their pooled part is consumed */
    AEFFT_F_NOCOMPACT = 1 << 1,   /* decoder outputs on the full grid instead of the support of the up-sampled spectra */
    AEFFT_F_NOQPATH = 1 << 2,     /* weight gradients through the dc|df spectra instead of the pruned transform Q of S */
    AEFFT_F_NOFUSEMSE = 1 << 3,   /* post-update MSE by conv, conv, diff instead of the collapsed operator + epilogue */
    AEFFT_F_NOGROUP = 1 << 4,     /* one launch per pair instead of grouped launches */
    AEFFT_F_NOMFMA = 1 << 5,      /* scalar-FMA contraction kernels instead of the matrix-core kernel */
    AEFFT_F_NOGFWD = 1 << 6,      /* innermost pair by conv, conv instead of the collapsed operator left by the previous step */
    AEFFT_F_NOOVERLAP = 1 << 7,   /* reconstruction inverse FFT on the context stream instead of a side stream (reconstructions below 8 MB or above 256 MB
                                   * stay there anyway) */
    AEFFT_F_NOFUSECROP = 1 << 8,  /* separate resize launches instead of the crop fused into the encoder contraction */
    AEFFT_F_GTAPS = 1 << 9,       /* force G = spectrum of f (*) c (chosen by itself for HBM-sized spectra) */
    AEFFT_F_NOPREFETCH = 1 << 10, /* pipelined mode: input R2C on the context stream */
    AEFFT_F_NODEFER = 1 << 11,    /* pipelined mode: reconstruction not deferred to the end of the gradient half */
    AEFFT_F_NOTILEDSPATIAL = 1 << 12, /* spatial mode: naive kernels instead of the LDS-tiled / matrix-core ones */
    AEFFT_F_NOFAST = 1 << 13,     /* generic scalar contraction kernel instead of the lean one */
    AEFFT_F_NOSPLITK = 1 << 14,   /* no split-K in the scalar contraction */
    AEFFT_F_POISON = 1 << 15,     /* NaN-fill every allocation (uninitialised reads show up in the tests) */
    AEFFT_F_NOOPFORM = 1 << 16,   /* training step per frame (batch contractions) instead of the operator form */
    AEFFT_F_NOCHAIN = 1 << 17,    /* operator form: the network on the basis frames layer by layer instead of one fused launch */
    AEFFT_F_NOFUSEUPD = 1 << 18,  /* operator form: the clipped-momentum update as its own launch instead of riding with the spectra / MSE launches */
    AEFFT_F_NOAHEAD = 1 << 19,    /* operator form: the next step's operator chain as the first launch of that step instead of riding in this step's last launch */
    AEFFT_F_NORCORR = 1 << 20,    /* spatial mode: dC through the back-convolved error (a dM-plane tensor) instead of the error-input correlation R */
    AEFFT_F_NOLAZYMSE = 1 << 21,   /* aefft_net_step_apply(mse_d = NULL) still sums the MSE slots in a launch of its own instead of leaving them to the next step's gradient launch */
    AEFFT_F_SMALLOVERLAP = 1 << 22, /* reconstructions below 8 MB take the side stream as well (the test suite's small nets then run the two-stream path of the large ones) */
    AEFFT_F_CHAINMSE = 1 << 23     /* operator form with the chain: the innermost pair's post-update MSE inside the chain's per-bin items whatever the launch's size
                                   * (by default only in launches of more than ~6 000 workgroups, which are bound by their resident slots) */
};
int aefft_ctx_set_flags(aefft_ctx* ctx, unsigned flags);
unsigned aefft_ctx_get_flags(const aefft_ctx* ctx);
const char* aefft_version(void);

/* ---- op level: one entry point per reference device routine --------------------------------- */

/* fft_backproplib.cu:764-801 `fft`: batched unnormalised 2-D R2C.  x_d [planes][Nx][Ny] -> X_d [planes][Nx][Nyr]. */
int aefft_r2c(aefft_ctx* ctx, const float* x_d, float* X_d, long planes, int Nx, int Ny);
/* fft_backproplib.cu:806-864 `fft_inv` (scale = 1/(Nx*Ny)) and the bare cufftExecC2R of
 * :1219-1220 (scale = 1).  Imaginary parts of self-conjugate bins are ignored. */
int aefft_c2r(aefft_ctx* ctx, const float* X_d, float* x_d, long planes, int Nx, int Ny, float scale);
/* fft_backproplib.cu:975-1002 `pool_fft` + :87-157 `resize`: scale > 1 crops to Nx/scale,
 * scale < -1 zero-pads to Nx*|scale|, no amplitude rescale.  Out-of-place; *Nxs,*Nys receive
 * the new size (may be NULL). */
int aefft_pool(aefft_ctx* ctx, const float* X_d, float* Xs_d, long planes, int Nx, int Ny, int scale, int* Nxs, int* Nys);
/* Fused forms used by the resident network (same arithmetic, fewer bytes):
 * r2c followed by pool(scale>=1), and pool(scale<=-1) followed by c2r. */
int aefft_r2c_pool(aefft_ctx* ctx, const float* x_d, float* Xs_d, long planes, int Nx, int Ny, int scale);
int aefft_unpool_c2r(aefft_ctx* ctx, const float* Xs_d, float* x_d, long planes, int Nxs, int Nys, int scale, float out_scale);

/* fft_backproplib.cu:1018-1064 `kernel_pad` + :869-916 `kfft` (first pass of StoreLoad_cfreq,
 * :1146-1158): k_d [nA][nB][Nk][Nl] -> K_d [nA][nB][Nx][Nyr]. */
int aefft_kernel_spectrum(aefft_ctx* ctx, const float* k_d, float* K_d, int nA, int nB, int Nk, int Nl, int Nx, int Ny);
/* fft_backproplib.cu:1166-1172 `export_cfreq` (= `kfft_inv` :921-970 + `kernel_invpad` :1069-1112). */
int aefft_kernel_export(aefft_ctx* ctx, const float* K_d, float* k_d, int nA, int nB, int Nk, int Nl, int Nx, int Ny);

/* fft_backproplib.cu:1007-1013 `conv_fft` / :162-189 `conv_k`:
 *   O[b][m] = sum_d (X[b][d]/dM) * C[m][d];  Re O[b][m](0,0) += bias[m]*Nx*Ny.
 * X_d [B][dD][P], C_d [dM][dD][P], bias_d [dM], O_d [B][dM][P]. */
int aefft_conv(aefft_ctx* ctx, const float* X_d, const float* C_d, const float* bias_d, float* O_d,
               int B, int dM, int dD, int Nx, int Ny);

/* fft_backproplib.cu:395-475 `gradient_k_io`.  Xin/Xout/O [B][dD][P]; C [dM][dD][P]; F [dD][dM][P];
 * outputs dc [dM][dD][P], df [dD][dM][P], db [dM], dp [dD]; for B > 1 the mean over frames. */
int aefft_gradient(aefft_ctx* ctx, const float* Xin_d, const float* Xout_d, const float* O_d, const float* C_d,
                   const float* F_d, const float* b_d, float* dc_d, float* df_d, float* db_d, float* dp_d,
                   int B, int dM, int dD, int Nx, int Ny);

/* fft_backproplib.cu:1178-1192 `mse_fft` (+ :480-498 `calc_mse`); mean over the B frames.
 * mse_d: one float on the device. */
int aefft_mse(aefft_ctx* ctx, const float* T_d, const float* O_d, float* mse_d, int B, int dM, int dD, int Nx, int Ny);

/* fft_backproplib.cu:1197-1291 host `backprop`: unnormalised C2R of dc/df, shrink_k (:535),
 * backprop_d (:605) or gradient_diff + backprop_double (:709,:657) when maxdiff, pad_k (:570),
 * R2C -> new C, F.  c,f,b,p and the momentum buffers Dc,Df,Db,Dp are updated in place.
 * `del` is the step actually applied (the reference passes 0.1*del0, :1445). */
int aefft_update(aefft_ctx* ctx, float* c_d, float* f_d, float* b_d, float* p_d, float* C_d, float* F_d,
                 const float* dc_d, const float* df_d, const float* db_d, const float* dp_d,
                 float* Dc_d, float* Df_d, float* Db_d, float* Dp_d,
                 int dM, int dD, int Nx, int Ny, int Nk, int Nl, float del, int maxdiff);

/* ---- spatial mode (coordinate space) --------------------------------------------------------- */
/* backproplib.cu:114-182 `Conv_gpu` (+ :70-111 `conv_parallel`): zero-padded direct convolution,
 * tap offset -2*ak-1+k with ak=((Nk-1)/2-1)/2, input divided by dM first (:134), + b[m].
 * in_d [B][dD][Nx][Ny] -> out_d [B][dM][Nx][Ny].  cpu_semantics=1 gives netlib.cpp:318-358 `Conv`
 * instead (ak=(Nk-1)/2-1, boundary test '>0', no division). */
int aefft_conv_spatial(aefft_ctx* ctx, const float* in_d, float* out_d, const float* c_d, const float* b_d,
                       int B, int dD, int dM, int Nx, int Ny, int Nk, int Nl, int cpu_semantics);
/* netlib.cpp:114-164 `Pool` on the device: scale > 0 pools scale x scale windows through the reference's
 * integer accumulator (`int smax = 0`: result = max(0, trunc(window maximum)), also at scale 1); scale < 0 up-samples by
 * nearest neighbour.  in_d [planes][Nxi][Nyi] -> out_d [planes][Nxo][Nyo]; outputs the reference loop never writes stay untouched. */
int aefft_pool_spatial(aefft_ctx* ctx, const float* in_d, float* out_d, long planes, int Nxi, int Nyi, int Nxo, int Nyo, int scale);
/* Pool(scale >= 1) followed by Conv_gpu in ONE launch (autoencoder.cpp:135-150 calls them back to back and
 * bounces the pooled layer through host vectors).  in_d [B][dD][Nx*scale][Ny*scale]; pooled_d (nullable) also receives the pooled
 * layer [B][dD][Nx][Ny] -- the training step needs it as the pair's input; out_d [B][dM][Nx][Ny].  Served for square 3x3 / 5x5 /
 * 7x7 kernels (AEFFT_EINVAL otherwise: pool and convolve separately). */
int aefft_pool_conv_spatial(aefft_ctx* ctx, const float* in_d, float* pooled_d, float* out_d, const float* c_d, const float* b_d,
                            int B, int dD, int dM, int Nx, int Ny, int scale, int Nk, int Nl, int cpu_semantics);
/* backproplib.cu:291-418 `backprop_gpu` (tied=0) / :521-644 `backprop_gpu_cc` (tied=1): back-conv
 * through f + weight-gradient correlation, then the inertia update
 *   d <- (1-alpha)*delmax*g/max(10,|g|) + alpha*d ; w <- w - d     (:392-396)
 * dc..dp are the caller's persistent previous-update buffers, ddc..ddp receive the gradients
 * (adapt_rate, :28-35, is otherwise inert -- Appendix B-12).  All device pointers; for B > 1 the
 * gradient is the mean over frames.  cpu_semantics: 0 = GPU geometry with the index / stale-buffer bugs of gradient_CF and the
 * `dDdB2 =` of gradient_CFBP (Appendix B-11) following the CPU reference (netlib.cpp:425-430) -- the default; 1 = the CPU
 * reference's geometry (ak = (Nk-1)/2-1, range test '>0'); 2 = GPU geometry WITH the B-11 bugs exactly as the CUDA source
 * computes them (backproplib.cu:220,225-227,283; hidden-layer reads outside the buffer, undefined there, read 0): the
 * "identical to the reference CUDA run" switch, slow. */
int aefft_backprop_spatial(aefft_ctx* ctx, const float* in_d, const float* out_d, const float* hin_d,
                           float* c_d, float* b_d, float* f_d, float* p_d,
                           float* dc_d, float* db_d, float* df_d, float* dp_d,
                           float* ddc_d, float* ddb_d, float* ddf_d, float* ddp_d,
                           int B, int dD, int dM, int Nx, int Ny, int Nk, int Nl,
                           float delmax, float alpha, int tied, int cpu_semantics);

/* One spatial-mode training step in ONE call: Conv_gpu (in -> hin), Conv_gpu (hin -> out), backprop_gpu[_cc] -- the sequence
 * autoencoder.cpp:140-148,200 runs per pair (backproplib.cu:114-182, 291-418, 521-644).  Because the hidden layer is then known to be
 * this call's own convolution of `in`, the decoder gradients dF, dP are formed from the same error-input region sums as dC, dB
 *  and the dM-plane hidden layer is read once (by the second convolution) instead of twice; shapes the region
 * route does not serve (kernels other than 3x3, more than 3 input channels, ...) run the plain sequence.  hin_d [B][dM][Nx][Ny] and
 * out_d [B][dD][Nx][Ny] receive the two layers; everything else as aefft_backprop_spatial.  cpu_semantics: 0 or 1. */
int aefft_step_spatial(aefft_ctx* ctx, const float* in_d, float* hin_d, float* out_d,
                       float* c_d, float* b_d, float* f_d, float* p_d,
                       float* dc_d, float* db_d, float* df_d, float* dp_d,
                       float* ddc_d, float* ddb_d, float* ddf_d, float* ddp_d,
                       int B, int dD, int dM, int Nx, int Ny, int Nk, int Nl,
                       float delmax, float alpha, int tied, int cpu_semantics);

/* ---- network level: the resident, batched form of autoenc_fft / backprop_fft ------------------ */
typedef struct {
    int D, Nx, Ny;        /* input frames [B][D][Nx][Ny] */
    int npairs;           /* L encoder/decoder pairs (net_c holds 2L kernels, autoencoder.cpp:115-116,414-417) */
    const int* maps;      /* [L] feature maps dM of each pair */
    const int* Nk;        /* [L] kernel rows */
    const int* Nl;        /* [L] kernel cols */
    const int* scale;     /* [L] pooling scale s>=1 of each pair (decoder mirrors with -s, autoencoder.cpp:119-120) */
    int batch;            /* frames per call on this GPU */
} aefft_net_desc;

int aefft_net_create(aefft_ctx* ctx, const aefft_net_desc* desc, aefft_net** out);
void aefft_net_destroy(aefft_net* net);
/* the descriptor back: number of pairs (negative for a null net); channels / maps / kernel support of pair l (any pointer nullable) */
int aefft_net_npairs(aefft_net* net);
int aefft_net_pair_shape(aefft_net* net, int l, int* dD, int* dM, int* Nk, int* Nl);
/* weights of pair l (host pointers, reference layouts).  set = the caller's net_cfreq.clear():
 * spectra are rebuilt from c,f (fft_backproplib.cu:1148-1158). */
int aefft_net_set_pair(aefft_net* net, int l, const float* c_h, const float* b_h, const float* f_h, const float* p_h);
int aefft_net_get_pair(aefft_net* net, int l, float* c_h, float* b_h, float* f_h, float* p_h);
/* device views of the cached kernel spectra (C [dM][dD][P], F [dD][dM][P]) of pair l */
int aefft_net_pair_spectra(aefft_net* net, int l, float** C_d, float** F_d);
/* the caller-side `net_cfreq` cache (fft_backproplib.cu:1117-1141 store_cfreq / load_cfreq): host
 * copies of the spectra, interleaved floats in the reference layout.  load makes the spectra the
 * source of truth and re-derives c, f from them (export_cfreq, :1166). */
int aefft_net_store_spectra(aefft_net* net, int l, float* C_h, float* F_h);
int aefft_net_load_spectra(aefft_net* net, int l, const float* C_h, const float* b_h, const float* F_h, const float* p_h);

/* fft_backproplib.cu:1331-1376 `autoenc_fft` over a batch.  frames_d [B][D][Nx][Ny];
 * recon_d (nullable) receives layers.back().  All intermediate spectra stay resident. */
int aefft_net_forward(aefft_net* net, const float* frames_d, float* recon_d);
/* fft_l=1 semantics (:1347,1357,1361): coordinate-space copy of reference layer index `layer`
 * (autoencoder.cpp:110-114 ordering, 0..4L) from the last forward.  out_d [B][ch][nx][ny];
 * ch/nx/ny (nullable) receive its shape.  Pass out_d=NULL to query the shape only.
 * After aefft_net_step_grad (with or without the following aefft_net_step_apply) the layers are those of THAT step's forward, i.e. of the
 * weights before the update.  What they are formed from depends on the step form (aefft_net_step_form):
 *   OPERATOR_CHAIN  the resident input spectra and the step's stored operators -- neither the caller's frame buffer nor the current
 *                   weights enter (layer 0, the input itself, is copied from the frame buffer of the last step_grad / forward);
 *   OPERATOR        the per-frame forward of the frames of the last step_grad is re-run with the CURRENT weights: the caller's frame buffer
 *                   must still hold them, and after aefft_net_step_apply the layers are those of the updated weights;
 *   PER_FRAME       the spectra of the step's forward as stored.
 * A HIDDEN layer (even index <= 2L) that the training step did not materialise is formed on request from the pair's input of the step and
 * the encoder of THAT step: in the OPERATOR_CHAIN form after aefft_net_step_apply the pre-update encoder is recovered as w + D (the update was
 * w <- w - D and D stays in the momentum buffer: exact to one rounding), so every layer of one export belongs to the same weight set; in
 * the other two forms from the pair's CURRENT encoder (PER_FRAME before step_apply, OPERATOR as described above). */
int aefft_net_get_layer(aefft_net* net, int layer, float* out_d, int* ch, int* nx, int* ny);

/* fft_l = 1 in one call: EVERY layer 0..4L of the last forward, coordinate space, packed into out_d at the
 * offsets aefft_net_layers_layout reports (floats; offsets_h[4L+1] = total).  Each layer is one inverse transform of the spectrum
 * where it is stored (spectral crop / zero-pad fused); hidden layers the training step skipped are formed first. */
int aefft_net_layers_layout(aefft_net* net, size_t* offsets_h /* [4L+2] */);
int aefft_net_get_layers(aefft_net* net, float* out_d);
/* fft_backproplib.cu:27-63 `magnitude` + `shift_magnitude` (the reference's spectrum display path, dead in its main()):
 * mag[d][i][j] = sqrt(|X[d][i][j]| / (ch*Nx*Ny)) on the half-plane j < Nyr, the mirrored element
 * X[d][Nx-1-i][2*Nyr-1-j] beyond it (the reference's index arithmetic, :53), shift != 0: quadrants swapped so that the zero
 * frequency sits in the centre (:33-36).  X_d [planes][Nx][Nyr] -> mag_d [planes][Nx][Ny]; ch = channels per frame. */
int aefft_magnitude(aefft_ctx* ctx, const float* X_d, float* mag_d, long planes, int ch, int Nx, int Ny, int shift);

/* fft_backproplib.cu:1381-1511 `backprop_fft` burst on pair l, using the spectra of the last
 * forward as in / expout(=in) / out (autoencoder.cpp:169,194): zeroes the momentum (:1420-1423),
 * del = 0.1*del0 (:1445), n_iter iterations (reference: 100, :1446) of gradient -> update ->
 * re-forward -> mse.  mse_h (nullable) receives n_iter+1 values (initial, then one per
 * iteration, :1440,1463); blocks until done when mse_h != NULL. */
int aefft_net_train_pair(aefft_net* net, int l, int n_iter, float del0, int maxdiff, int sym, float* mse_h);

/* One data-parallel training step = forward + ONE loop-body iteration for every pair
 * (SURVEY 8d "one frame fwd+bwd"), split so the caller can all-reduce in between:
 *   step_grad : forward (recon_d nullable: layers.back()), then per pair the batch-mean gradient of the
 *               first loop-body iteration (fft_backproplib.cu:1454-1456: gradient_k_io on the forward's
 *               in / out spectra, C2R, shrink_k) -> packed buffer [dck | dfk | db | dp] per pair, pairs
 *               concatenated.
 *   (caller: all-reduce SUM of aefft_net_grad_buffer over ranks)
 *   step_apply: gradients * grad_scale (1/world_size), update (:1229-1272), new kernel spectra
 *               (:1274-1282), post-update MSE of each pair's own re-forward (:1460-1463).
 *               mse_d (nullable): [L] floats on the device.
 * Same sums as the reference, re-associated; intermediate layers that the
 * reference itself never exports (fft_l = 0) are formed on demand by aefft_net_get_layer.
 * Momentum persists across steps (reset with aefft_net_reset_momentum). */
int aefft_net_step_grad(aefft_net* net, const float* frames_d, float* recon_d);
/* The same calls on 8-BIT frames (frames_d [B][D][Nx][Ny] unsigned char, planar: what a camera delivers -- the reference's application turns each
 * 8-bit pixel into a float on the host, `(float)col[c]`, netlib.cpp:37-51 ImageToSpin_C, called at autoencoder.cpp:125): the input transform
 * converts on load, a quarter of its reads; results are those of the float call on the same pixel values, bit for bit.  Power-of-two frame
 * sizes; 16-byte aligned.  aefft_net_get_layer(0) returns the pixels as floats. */
int aefft_net_step_grad_u8(aefft_net* net, const unsigned char* frames_d, float* recon_d);
int aefft_net_forward_u8(aefft_net* net, const unsigned char* frames_d, float* recon_d);
/* Opt-in input prefetch for pipelined training loops.  enable = 1 asserts that the frames handed to
 * aefft_net_step_grad are COMPLETE in device memory when the call is made (not merely ordered on the
 * context stream, e.g. a loader that synchronises its own copy stream): their R2C then runs on an
 * internal side stream and may overlap the tail of the previous step still queued on the context
 * stream (the input spectra are double-buffered); and the reconstruction written by aefft_net_step_grad is
 * launched at the END of the gradient half (where a data-parallel rank waits for its all-reduce) and is complete
 * on the context stream only after the following aefft_net_step_apply, aefft_sync or any later call on the net.
 * Default 0: everything is ordered on the context stream and recon_d is complete when aefft_net_step_grad's work is. */
int aefft_net_set_input_ready(aefft_net* net, int enable);
/* The packed buffer (device), nfloats = [dck | dfk | db | dp of pair 0] ... [of pair L-1] | mse[L]: the batch-mean gradients of the last
 * aefft_net_step_grad, then ONE float per pair: the post-update MSE of this rank's frames as the PREVIOUS aefft_net_step_apply left it
 * (zero before the first).  A data-parallel caller all-reduces (SUM) the whole buffer: the gradients are then applied with grad_scale =
 * 1/world, and the tail times 1/world is the global-batch MSE of the previous step (the MSE rides in the gradients' message;
 * the post-update MSE of a step needs that step's reduced gradients, so it travels one step behind).  aefft_net_step_apply reads the
 * gradient part only and overwrites the tail; what it finds there it first saves, times its grad_scale, in the L floats BEHIND the
 * buffer (buf_d[nfloats .. nfloats + L), not part of the message): after step_apply of step t+1 they hold the global-batch MSE of step t
 * (with mse_d = NULL, see aefft_net_last_mse: once the sums of step t+1 have been formed, i.e. after the next aefft_net_step_grad). */
int aefft_net_grad_buffer(aefft_net* net, float** buf_d, size_t* nfloats);
/* Which form the NEXT aefft_net_step_grad / _apply of this net runs in (decided by the net's shapes and the development switches; the
 * arithmetic is the reference's in every form, re-associated -- DESIGN.md section 4):
 *   AEFFT_FORM_PER_FRAME       every layer evaluated for every frame (batch contractions on the matrix cores).  Taken for inputs of more
 *                              than 3 channels, kernel supports other than equal square 3x3 / 5x5, channel counts beyond the operator
 *                              kernels' tiles, and under AEFFT_F_NOOPFORM / AEFFT_F_NOQPATH.
 *   AEFFT_FORM_OPERATOR        the network is linear: layers evaluated once per step as per-bin operators on 4 basis frames, the batch
 *                              enters through the input transform, its centred second moments and the reconstruction; layer by layer.
 *   AEFFT_FORM_OPERATOR_CHAIN  ... with the whole operator chain in one launch out of a bin-major copy of the kernel spectra (coarsest grid
 *                              of at most 16384 bins); the next step's chain rides in the last launch of this step.
 * Returns -1 for a null net. */
enum { AEFFT_FORM_PER_FRAME = 0, AEFFT_FORM_OPERATOR = 1, AEFFT_FORM_OPERATOR_CHAIN = 2 };
int aefft_net_step_form(aefft_net* net);
int aefft_net_step_apply(aefft_net* net, float del0, int maxdiff, int sym, float grad_scale, float* mse_d);
/* The per-pair post-update MSEs of the LAST aefft_net_step_apply (fft_backproplib.cu:1463), to mse_d[L] (device), in stream order.
 * aefft_net_step_apply(mse_d = NULL) does not form them in a launch of its own: the per-workgroup partial sums wait in the net and are
 * added up by one extra workgroup of the next aefft_net_step_grad's gradient launch -- in time for the packed buffer's MSE tail and that
 * step's all-reduce -- or by this call, whichever comes first.  A loop that logs the MSE every K steps calls this every K steps. */
int aefft_net_last_mse(aefft_net* net, float* mse_d);
int aefft_net_reset_momentum(aefft_net* net);

/* ---- measurement ----------------------------------------------------------------------------- */
/* Per-kernel HIP event timing on the context stream (bench.py's roofline figure).  With
 * enable=1 every kernel launch is bracketed by hipEvents recorded on the stream it runs on.
 * aefft_prof_read synchronises and returns, for kernel id `kid` (0..aefft_prof_count()-1,
 * named by aefft_prof_name), launches, total milliseconds and total ALGORITHMIC bytes
 * (unique tensors entering + leaving each launch) since the last reset. */
int aefft_prof_enable(aefft_ctx* ctx, int enable);
int aefft_prof_count(void);
const char* aefft_prof_name(int kid);
int aefft_prof_read(aefft_ctx* ctx, int kid, long* launches, double* total_ms, double* algo_bytes);
int aefft_prof_reset(aefft_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* AEFFT_H */
