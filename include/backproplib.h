/* Drop-in replacement for the reference's source/backproplib.h (spatial-mode GPU operator API). */
#ifndef BACKPROPLIB_H
#define BACKPROPLIB_H
#include "aefft_vector_types.h"

/* backproplib.cu:291-418.  Spatial-mode training step of one pair: back-convolution through f plus
 * weight-gradient correlation, inertia update d <- (1-alpha)*delmax*g/max(10,|g|) + alpha*d, w -= d.
 * dc..dp persist across calls (previous updates); ddc..ddp receive the gradients; `active` is inert
 * (adapt_rate ends with del=delmax, :34).  Prints "mse: .." (:357). */
void backprop_gpu(aefft_vec::Maps& in, aefft_vec::Maps& out, aefft_vec::Maps& hin,
                  aefft_vec::Kernels& c, aefft_vec::Bias& b, aefft_vec::Kernels& f, aefft_vec::Bias& p,
                  aefft_vec::Kernels& dc, aefft_vec::Bias& db, aefft_vec::Kernels& df, aefft_vec::Bias& dp,
                  aefft_vec::Kernels& ddc, aefft_vec::Bias& ddb, aefft_vec::Kernels& ddf, aefft_vec::Bias& ddp,
                  float delmax, float alpha, int active);

/* backproplib.cu:114-182.  Zero-padded direct convolution, input divided by dM, + bias, identity activation. */
void Conv_gpu(aefft_vec::Maps& in, aefft_vec::Maps& out, aefft_vec::Kernels& c, aefft_vec::Bias& b);

/* backproplib.cu:521-644.  Tied-weight variant: g = dC + dF, Norm doubled, f[d][m] = c[m][d] afterwards. */
void backprop_gpu_cc(aefft_vec::Maps& in, aefft_vec::Maps& out, aefft_vec::Maps& hin,
                     aefft_vec::Kernels& c, aefft_vec::Bias& b, aefft_vec::Kernels& f, aefft_vec::Bias& p,
                     aefft_vec::Kernels& dc, aefft_vec::Bias& db, aefft_vec::Kernels& df, aefft_vec::Bias& dp,
                     aefft_vec::Kernels& ddc, aefft_vec::Bias& ddb, aefft_vec::Kernels& ddf, aefft_vec::Bias& ddp,
                     float delmax, float alpha, int active);

/* backproplib.cu:38-51.  Identity activation and its derivative (used by the CPU path too). */
float act(float x);
float act1(float x);
#endif
