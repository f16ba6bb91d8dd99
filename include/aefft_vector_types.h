/* Shared spellings for the nested-vector tensors of the reference's operator headers
 * (source/netlib.h:4-24, source/backproplib.h:5-16, source/fft_backproplib.h:5-11).
 * Aliases do not change name mangling: functions declared with them export exactly the
 * Itanium symbols autoencoder.cpp imports (libstdc++ std::vector). */
#ifndef AEFFT_VECTOR_TYPES_H
#define AEFFT_VECTOR_TYPES_H
#include <vector>

namespace cv { class Mat; }   /* OpenCV's own declaration wins when its headers are included first */

namespace aefft_vec {
typedef std::vector<float> Bias;                 /* [ch]                         b, p, db, ... */
typedef std::vector<Bias> Plane;                 /* [Nx][Ny]  (x first, y contiguous)          */
typedef std::vector<Plane> Maps;                 /* [ch][Nx][Ny]                 one layer     */
typedef std::vector<Maps> Kernels;               /* [out][in][Nk][Nl]            c, f; also `layers` = [layer][ch][Nx][Ny] */
typedef std::vector<Kernels> KernelStack;        /* [conv][out][in][Nk][Nl]      net_c         */
typedef std::vector<Bias> BiasStack;             /* [conv][ch]  net_b; also net_cfreq = [conv][2*out*in*Nx*Nyr] */
}
#endif
