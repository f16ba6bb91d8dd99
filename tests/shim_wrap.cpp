// Test-only C wrappers around the C++ operator-API shims exported by libaefft.so (include/netlib.h,
// backproplib.h, fft_backproplib.h), so pytest can drive the nested-std::vector entry points through
// ctypes exactly as autoencoder.cpp would call them.
#include "netlib.h"
#include "backproplib.h"
#include "fft_backproplib.h"
#include <cstddef>
using namespace aefft_vec;

static Maps to3(const float* a, int A, int B, int C) {
    Maps v(A, Plane(B, Bias(C)));
    for (int i = 0; i < A; i++) for (int j = 0; j < B; j++) for (int k = 0; k < C; k++) v[i][j][k] = a[((size_t)i * B + j) * C + k];
    return v;
}
static void from3(const Maps& v, float* a) { size_t q = 0; for (auto& x : v) for (auto& y : x) for (float z : y) a[q++] = z; }
static Kernels to4(const float* a, int A, int B, int C, int D) { Kernels v(A); for (int i = 0; i < A; i++) v[i] = to3(a + (size_t)i * B * C * D, B, C, D); return v; }
static void from4(const Kernels& v, float* a) { size_t q = 0; for (auto& w : v) for (auto& x : w) for (auto& y : x) for (float z : y) a[q++] = z; }

extern "C" {
void w_conv(const float* in, float* out, const float* c, const float* b, int dD, int dM, int Nx, int Ny, int Nk, int Nl, int gpu) {
    Maps vin = to3(in, dD, Nx, Ny), vout(dM, Plane(Nx, Bias(Ny)));
    Kernels vc = to4(c, dM, dD, Nk, Nl); Bias vb(b, b + dM);
    if (gpu) Conv_gpu(vin, vout, vc, vb); else Conv(vin, vout, vc, vb);
    from3(vout, out);
}
void w_backprop_cpu(const float* in, const float* out, const float* hin, float* c, float* b, float* f, float* p, float del,
                    int dD, int dM, int Nx, int Ny, int Nk, int Nl) {
    Maps vin = to3(in, dD, Nx, Ny), vout = to3(out, dD, Nx, Ny), vh = to3(hin, dM, Nx, Ny);
    Kernels vc = to4(c, dM, dD, Nk, Nl), vf = to4(f, dD, dM, Nk, Nl); Bias vb(b, b + dM), vp(p, p + dD);
    backprop(vin, vout, vh, vc, vb, vf, vp, del);
    from4(vc, c); from4(vf, f);
    for (int m = 0; m < dM; m++) b[m] = vb[m];
    for (int d = 0; d < dD; d++) p[d] = vp[d];
}
void w_pool(const float* in, float* out, int D, int Nxi, int Nyi, int Nxo, int Nyo, int scale) {
    Maps vin = to3(in, D, Nxi, Nyi), vout = to3(out, D, Nxo, Nyo);
    Pool(vin, vout, scale); from3(vout, out);
}
void w_portion(const float* in, float* in_s, int ch, int Nx, int Ny, int q) {
    Maps vin = to3(in, ch, Nx, Ny), h = to3(in, 1, Nx, Ny), vs(ch, Plane(Nx / q, Bias(Ny / q))), vo = vs, vh(1, Plane(Nx / q, Bias(Ny / q)));
    Portion(vin, h, vin, vs, vh, vo, q); from3(vs, in_s);
}
// Init_conv / SaveLoad_conv / LoadParam (netlib.h:14,16,18) through the product's exports
void w_init_conv(float* c, float* b, int mS, int dD, int kS, int lS, float max) {
    Kernels vc; Bias vb;
    Init_conv(vc, vb, mS, dD, kS, lS, max);
    from4(vc, c);
    for (int m = 0; m < mS; m++) b[m] = vb[m];
}
void w_saveload_conv(float* c, float* b, int dM, int dD, int Nk, int Nl, int scale, int L, int io, int write) {
    Kernels vc = to4(c, dM, dD, Nk, Nl); Bias vb(b, b + dM);
    SaveLoad_conv(vc, vb, scale, L, io, write);
    from4(vc, c);
    for (int m = 0; m < dM; m++) b[m] = vb[m];
}
void w_load_param(int* dM, int* Lk, int* Ll, int* scal, float* rmax) { LoadParam(*dM, *Lk, *Ll, *scal, *rmax); }
// backprop_gpu / backprop_gpu_cc: all weight-shaped arrays in/out
void w_backprop_gpu(const float* in, const float* out, const float* hin, float* c, float* b, float* f, float* p, float* dc, float* db, float* df,
                    float* dp, float* ddc, float* ddb, float* ddf, float* ddp, float delmax, float alpha, int tied,
                    int dD, int dM, int Nx, int Ny, int Nk, int Nl) {
    Maps vin = to3(in, dD, Nx, Ny), vout = to3(out, dD, Nx, Ny), vh = to3(hin, dM, Nx, Ny);
    Kernels vc = to4(c, dM, dD, Nk, Nl), vf = to4(f, dD, dM, Nk, Nl), vdc = to4(dc, dM, dD, Nk, Nl), vdf = to4(df, dD, dM, Nk, Nl),
            vddc = to4(ddc, dM, dD, Nk, Nl), vddf = to4(ddf, dD, dM, Nk, Nl);
    Bias vb(b, b + dM), vp(p, p + dD), vdb(db, db + dM), vdp(dp, dp + dD), vddb(ddb, ddb + dM), vddp(ddp, ddp + dD);
    if (tied) backprop_gpu_cc(vin, vout, vh, vc, vb, vf, vp, vdc, vdb, vdf, vdp, vddc, vddb, vddf, vddp, delmax, alpha, 1);
    else backprop_gpu(vin, vout, vh, vc, vb, vf, vp, vdc, vdb, vdf, vdp, vddc, vddb, vddf, vddp, delmax, alpha, 1);
    from4(vc, c); from4(vf, f); from4(vdc, dc); from4(vdf, df); from4(vddc, ddc); from4(vddf, ddf);
    for (int m = 0; m < dM; m++) { b[m] = vb[m]; db[m] = vdb[m]; ddb[m] = vddb[m]; }
    for (int d = 0; d < dD; d++) { p[d] = vp[d]; dp[d] = vdp[d]; ddp[d] = vddp[d]; }
}
// one-pair autoencoder through autoenc_fft (fft_l = 1) then a backprop_fft burst, as autoencoder.cpp:132,169,194 does.
// layers_out: concatenation of layers[1..4]; cfreq_io holds [C | F] and doubles as the net_cfreq cache (ncache = 0: recompute).
void w_fft_pair(const float* x, float* layers_out, float* c, float* b, float* f, float* p, float* cfreq_io, int* ncache,
                int D, int dM, int N, int Nk, int s, int fft_l, int do_burst, float del0, int maxdiff) {
    const int n = N / s;
    Kernels layers = {to3(x, D, N, N), Maps(D, Plane(n, Bias(n))), Maps(dM, Plane(n, Bias(n))), Maps(D, Plane(n, Bias(n))), Maps(D, Plane(N, Bias(N)))};
    KernelStack net_c = {to4(c, dM, D, Nk, Nk), to4(f, D, dM, Nk, Nk)};
    BiasStack net_b = {Bias(b, b + dM), Bias(p, p + D)}, net_cfreq;
    const size_t W = (size_t)dM * D * n * (n / 2 + 1) * 2;
    if (*ncache == 2) { net_cfreq.push_back(Bias(cfreq_io, cfreq_io + W)); net_cfreq.push_back(Bias(cfreq_io + W, cfreq_io + 2 * W)); }
    std::vector<int> scale = {s, -s};
    autoenc_fft(layers, net_c, net_cfreq, net_b, scale, fft_l);
    size_t q = 0;
    for (int l = 1; l <= 4; l++) { from3(layers[l], layers_out + q); q += layers[l].size() * layers[l][0].size() * layers[l][0][0].size(); }
    if (do_burst) backprop_fft(layers[1], layers[1], layers[3], net_cfreq[0], net_c[0], net_cfreq[1], net_c[1], net_b[0], net_b[1], dM, del0, maxdiff);
    from4(net_c[0], c); from4(net_c[1], f);
    for (int m = 0; m < dM; m++) b[m] = net_b[0][m];
    for (int d = 0; d < D; d++) p[d] = net_b[1][d];
    for (size_t i = 0; i < W; i++) { cfreq_io[i] = net_cfreq[0][i]; cfreq_io[W + i] = net_cfreq[1][i]; }
    *ncache = (int)net_cfreq.size();
}
}
