/* Drop-in replacement for the reference's source/netlib.h (host-side operator API: CPU path,
 * pooling, crop, weight init / file format, image conversion). */
#ifndef NETLIB_H
#define NETLIB_H
#include "aefft_vector_types.h"

/* netlib.cpp:37-111.  cv::Mat <-> [3][Nx][Ny] float tensors (x = column index, i.e. transposed vs image
 * rows).  UI only; functional when the library is built with OpenCV headers, loud runtime error otherwise. */
void ImageToSpin_C(cv::Mat& img, aefft_vec::Maps& spin);
void SpinToImage_C(cv::Mat& img, aefft_vec::Maps& spin);
void SpinToImage_V(cv::Mat& img, aefft_vec::Plane& spin);
void SpinToImage_K(cv::Mat& img, aefft_vec::Plane& spin);

/* netlib.cpp:114-164.  scale > 0: max pooling with integer truncation and clamp at 0 (`int smax=0`);
 * scale < 0: nearest-neighbour up-sampling. */
void Pool(aefft_vec::Maps& in, aefft_vec::Maps& out, int scale);

/* netlib.cpp:167-197.  rand()-uniform weights in [-max, max]. */
void Init_conv(aefft_vec::Kernels& c, aefft_vec::Bias& b, int mS, int dS, int kS, int lS, float max);

/* netlib.cpp:220-272.  Raw float32 dump/load of [m][d][k][l] weights + biases under ./weights/ . */
void SaveLoad_conv(aefft_vec::Kernels& c, aefft_vec::Bias& b, int scale, int L, int io, int write);

/* netlib.cpp:274-289.  Five "name value" lines from New_Layer_Param.txt. */
void LoadParam(int& dM, int& Lk, int& Ll, int& scal, float& rmax);

/* netlib.cpp:292-315.  Centred 1/q crop of the three training tensors. */
void Portion(aefft_vec::Maps& in, aefft_vec::Maps& hin, aefft_vec::Maps& out, aefft_vec::Maps& in_s,
             aefft_vec::Maps& hin_s, aefft_vec::Maps& out_s, int q);

/* netlib.cpp:318-358.  CPU direct convolution (tap offset (Nk-1)/2-1, boundary test '>0', no /dM). */
void Conv(aefft_vec::Maps& in, aefft_vec::Maps& out, aefft_vec::Kernels& c, aefft_vec::Bias& b);

/* netlib.cpp:361-451.  CPU training step of one pair (no inertia, in-loop sequential updates). */
void backprop(aefft_vec::Maps& in, aefft_vec::Maps& out, aefft_vec::Maps& hin, aefft_vec::Kernels& c,
              aefft_vec::Bias& b, aefft_vec::Kernels& f, aefft_vec::Bias& p, float del);
#endif
