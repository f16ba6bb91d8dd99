// TEST INFRASTRUCTURE ONLY -- not product code.
//
// C-callable wrapper around the reference's OWN CPU functions (Conv, backprop, Pool, Portion, Init_conv, SaveLoad_conv, LoadParam;
// /root/reference/source/netlib.cpp:114-451), which oracle/Makefile compiles from the
// reference sources where they lie (nothing is copied into this repo; see the Makefile for the
// exact recipe).  The wrapper only marshals flat float arrays into the nested std::vector
// arguments those functions take.  It is linked into oracle/_ref/libnetlib_ref.so, used to
//   * pin oracle/cpu_ref.c (bit-identical outputs expected) and cross-pin oracle/np_ref.py,
//   * serve as bench.py's cpu_baseline of kind "reference".
#include <vector>
#include <cstddef>

typedef std::vector<float> V1;
typedef std::vector<V1> V2;
typedef std::vector<V2> V3;
typedef std::vector<V3> V4;

// declarations of the reference functions used (signatures as netlib.h:12,20,22,24)
void Pool(V3& in, V3& out, int scale);
void Portion(V3& in, V3& hin, V3& out, V3& in_s, V3& hin_s, V3& out_s, int q);
void Conv(V3& in, V3& out, V4& c, V1& b);
void backprop(V3& in, V3& out, V3& hin, V4& c, V1& b, V4& f, V1& p, float del);
// netlib.h:14,16,18
void Init_conv(V4& c, V1& b, int mS, int dD, int kS, int lS, float max);
void SaveLoad_conv(V4& c, V1& b, int scale, int L, int io, int write);
void LoadParam(int& dM, int& Lk, int& Ll, int& scal, float& rmax);

static V3 to3(const float* a, int A, int B, int C) {
    V3 v(A, V2(B, V1(C)));
    for (int i = 0; i < A; i++) for (int j = 0; j < B; j++) for (int k = 0; k < C; k++)
        v[i][j][k] = a[((size_t)i * B + j) * C + k];
    return v;
}
static void from3(const V3& v, float* a) {
    size_t q = 0;
    for (auto& x : v) for (auto& y : x) for (float z : y) a[q++] = z;
}
static V4 to4(const float* a, int A, int B, int C, int D) {
    V4 v(A);
    for (int i = 0; i < A; i++) v[i] = to3(a + (size_t)i * B * C * D, B, C, D);
    return v;
}
static void from4(const V4& v, float* a) {
    size_t q = 0;
    for (auto& w : v) for (auto& x : w) for (auto& y : x) for (float z : y) a[q++] = z;
}

extern "C" {

void ref_conv(const float* in, float* out, const float* c, const float* b,
              int dD, int dM, int Nx, int Ny, int Nk, int Nl) {
    V3 vin = to3(in, dD, Nx, Ny), vout(dM, V2(Nx, V1(Ny)));
    V4 vc = to4(c, dM, dD, Nk, Nl);
    V1 vb(b, b + dM);
    Conv(vin, vout, vc, vb);
    from3(vout, out);
}

void ref_backprop(const float* in, const float* out, const float* hin, float* c, float* b, float* f, float* p,
                  float del, int dD, int dM, int Nx, int Ny, int Nk, int Nl) {
    V3 vin = to3(in, dD, Nx, Ny), vout = to3(out, dD, Nx, Ny), vh = to3(hin, dM, Nx, Ny);
    V4 vc = to4(c, dM, dD, Nk, Nl), vf = to4(f, dD, dM, Nk, Nl);
    V1 vb(b, b + dM), vp(p, p + dD);
    backprop(vin, vout, vh, vc, vb, vf, vp, del);
    from4(vc, c); from4(vf, f);
    for (int m = 0; m < dM; m++) b[m] = vb[m];
    for (int d = 0; d < dD; d++) p[d] = vp[d];
}

void ref_pool(const float* in, float* out, int D, int Nxi, int Nyi, int Nxo, int Nyo, int scale) {
    V3 vin = to3(in, D, Nxi, Nyi), vout = to3(out, D, Nxo, Nyo);
    Pool(vin, vout, scale);
    from3(vout, out);
}

void ref_portion(const float* in, float* in_s, int ch, int Nx, int Ny, int q) {
    V3 vin = to3(in, ch, Nx, Ny), dummy_h = to3(in, 1, Nx, Ny);
    V3 vs(ch, V2(Nx / q, V1(Ny / q))), vos = vs, vhs(1, V2(Nx / q, V1(Ny / q)));
    Portion(vin, dummy_h, vin, vs, vhs, vos, q);
    from3(vs, in_s);
}

// netlib.cpp:166-197: weights drawn from the C library's rand() stream (the caller seeds it with srand)
void ref_init_conv(float* c, float* b, int mS, int dD, int kS, int lS, float max) {
    V4 vc; V1 vb;
    Init_conv(vc, vb, mS, dD, kS, lS, max);
    from4(vc, c);
    for (int m = 0; m < mS; m++) b[m] = vb[m];
}
// netlib.cpp:220-272: ./weights/C_weights_<L><_in|_out>_D=.._M=.._Lk=.._Ll=.._S=...conv relative to the working directory
void ref_saveload_conv(float* c, float* b, int dM, int dD, int Nk, int Nl, int scale, int L, int io, int write) {
    V4 vc = to4(c, dM, dD, Nk, Nl); V1 vb(b, b + dM);
    SaveLoad_conv(vc, vb, scale, L, io, write);
    from4(vc, c);
    for (int m = 0; m < dM; m++) b[m] = vb[m];
}
// netlib.cpp:274-289: New_Layer_Param.txt in the working directory
void ref_load_param(int* dM, int* Lk, int* Ll, int* scal, float* rmax) { LoadParam(*dM, *Lk, *Ll, *scal, *rmax); }

}  // extern "C"
