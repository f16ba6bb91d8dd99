/* Drop-in replacement for the reference's source/fft_backproplib.h (FFT-mode operator API).
 * Same functions, same argument meaning, same mangled symbols; implemented on MI355X by
 * libaefft.so (HIP, include/aefft.h).  Reference behaviour cited per function. */
#ifndef FFTBACKPROPLIB_H
#define FFTBACKPROPLIB_H
#include "aefft_vector_types.h"

/* fft_backproplib.cu:1331-1376.  Whole-autoencoder forward in frequency space.
 *   layers    [4L+1] pre-sized tensors: in, then per encoder (pooled in, conv out), then per decoder
 *             (conv out, up-sampled) (autoencoder.cpp:110-114); layers[0] is the input, layers.back()
 *             receives the reconstruction; with fft_l != 0 every intermediate is filled too (:1347-1361).
 *   net_c     2L kernels: encoders 0..L-1 then mirrored decoders; net_b the matching biases.
 *   net_cfreq caller-invalidated cache of kernel spectra: size() < net_c.size() means "recompute
 *             from net_c and push_back" (:1148-1158), otherwise the cached spectra are used (:1160).
 *   scale     [+s0.. +s(L-1), -s(L-1).. -s0] pooling factors (powers of two). */
void autoenc_fft(aefft_vec::Kernels& layers, aefft_vec::KernelStack& net_c, aefft_vec::BiasStack& net_cfreq,
                 aefft_vec::BiasStack& net_b, std::vector<int>& scale, int fft_l);

/* fft_backproplib.cu:1018-1064.  Centred circular zero-pad of every kernel to Nx x Ny (host helper). */
void kernel_pad(aefft_vec::Kernels& c, aefft_vec::Kernels& c_pad, int Nx, int Ny);

/* fft_backproplib.cu:1381-1511.  One training burst (100 iterations, del = 0.1*del0, momentum 0.9 reset per
 * call) on one encoder/decoder pair in frequency space.  in/expout/out [dD][Nx][Ny]; cfreq/ffreq the
 * cached spectra of c [dM][dD][Nk][Nl] and f [dD][dM][Nk][Nl]; all of cfreq, c, ffreq, f, b, p are updated.
 * Prints "mse fft: .." and "n: i mse: .." lines like the reference (:1441,1464).
 * (The reference header also declares a vector-based conv_fft that it never defines; it is not exported.) */
void backprop_fft(aefft_vec::Maps& in, aefft_vec::Maps& expout, aefft_vec::Maps& out, aefft_vec::Bias& cfreq,
                  aefft_vec::Kernels& c, aefft_vec::Bias& ffreq, aefft_vec::Kernels& f, aefft_vec::Bias& b,
                  aefft_vec::Bias& p, int dM, float del0, int maxdiff);
#endif
