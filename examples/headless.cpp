// Headless driver: the orchestration of the reference application's main loop (source/autoencoder.cpp:98-205 forward /
// training dispatch, :245-457 key commands) on a synthetic or pre-recorded video instead of a webcam, through the
// reference's own operator API (include/netlib.h, backproplib.h, fft_backproplib.h) as exported by libaefft.so.
// Nothing here computes: every numeric step is one of those operators.  What it reproduces is the STATE MACHINE --
// which operator runs on which layers/kernels with which flags after which key -- so that the vector-API shims are
// exercised end to end and the weight files of `s` / `l` follow the reference's format (8f-2).
//
//   aefft_headless --size 64 --frames 20 --script "lg5555555555.1..ns1." [--video frames.f32] [--seed 7] [--dump out.f32]
//
// One key of --script is applied after each frame ('.' = none), exactly where the reference polls waitKey
// (autoencoder.cpp:245).  Keys: 1 training on/off, 2/3 q, 4/5 learning rate, 6/7 inertia, 9 active rate, 0 gpu, f fft,
// g fft_l, m multiobjective, z/x active pair, e re-initialise, c clear spectra cache, p tied weights, s/l save/load,
// n add a pair (parameters from ./New_Layer_Param.txt), d delete the innermost pair, i print the structure.
// The display-only keys (q, w: feature-map selection) are accepted and ignored.
#include "netlib.h"
#include "backproplib.h"
#include "fft_backproplib.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

using Plane = std::vector<std::vector<float>>;
using Maps = std::vector<Plane>;
using Kernels = std::vector<Maps>;
using Bias = std::vector<float>;

static Maps make_maps(int ch, int nx, int ny) { return Maps(ch, Plane(nx, std::vector<float>(ny, 0.f))); }

// synthetic frame t: raw 0..255 pixel scale (netlib.cpp:46-48) = white noise + a smooth moving pattern, reproducible
static void synth_frame(Maps& in, int t, unsigned seed)
{
    uint32_t s = 2463534242u ^ (seed * 2654435761u) ^ (uint32_t)(1000 + t) * 40503u;
    const int D = (int)in.size(), Nx = (int)in[0].size(), Ny = (int)in[0][0].size();
    for (int d = 0; d < D; ++d)
        for (int i = 0; i < Nx; ++i)
            for (int j = 0; j < Ny; ++j) {
                s ^= s << 13; s ^= s >> 17; s ^= s << 5;                       // xorshift32
                const float noise = (float)(s >> 8) * (1.0f / 16777216.0f);
                const float smooth = 0.5f + 0.5f * std::sin(6.2831853f * (float)(i + 3 * t) / (float)Nx) * std::cos(6.2831853f * (float)(j + d * 5) / (float)Ny);
                in[d][i][j] = std::floor(std::min(255.0f, 128.0f * noise + 127.0f * smooth));
            }
}

int main(int argc, char** argv)
{
    int N = 64, frames = 8, D = 3;
    unsigned seed = 1;
    std::string script, video, dump;
    for (int a = 1; a < argc; ++a) {
        auto next = [&](const char* what) { if (a + 1 >= argc) { fprintf(stderr, "%s needs a value\n", what); exit(2); } return argv[++a]; };
        if (!strcmp(argv[a], "--size")) N = atoi(next("--size"));
        else if (!strcmp(argv[a], "--frames")) frames = atoi(next("--frames"));
        else if (!strcmp(argv[a], "--script")) script = next("--script");
        else if (!strcmp(argv[a], "--video")) video = next("--video");
        else if (!strcmp(argv[a], "--seed")) seed = (unsigned)atoi(next("--seed"));
        else if (!strcmp(argv[a], "--dump")) dump = next("--dump");
        else { fprintf(stderr, "unknown argument %s\n", argv[a]); return 2; }
    }
    const int Nx = N, Ny = N;
    // first pair from ./New_Layer_Param.txt, as the application does at start-up (autoencoder.cpp:43-45)
    int M = 50, Lk = 0, Ll = 0, s = 1;
    float rmax = 1.f;
    LoadParam(M, Lk, Ll, s, rmax);
    int Nk = 2 * (Lk + 1) + 1, Nl = 2 * (Ll + 1) + 1;

    std::vector<Maps> layers;
    std::vector<Kernels> net_c;
    std::vector<std::vector<float>> net_cfreq;
    std::vector<Bias> net_b;
    std::vector<int> scale;
    Kernels c, f, dc, df, ddc, ddf;
    Bias b(M), p(D), db(M), dp(D), ddb(M), ddp(D);
    // application state (autoencoder.cpp:85-97)
    int sel = 0, q = 1, active = 1, n_l = 0, gpu = 1, sym = 0, fft = 1, fft_l = 0, maxdiff = 0;
    float del = 0.2f, ddel = 0.1f, alpha = 0.9f;

    srand(seed);
    Init_conv(c, b, M, D, Nk, Nl, rmax);
    Init_conv(f, p, D, M, Nk, Nl, rmax);
    auto reset_optimizer = [&](int dM, int dD, int k, int l) {
        Init_conv(dc, db, dM, dD, k, l, 0); Init_conv(df, dp, dD, dM, k, l, 0);
        Init_conv(ddc, ddb, dM, dD, k, l, 0); Init_conv(ddf, ddp, dD, dM, k, l, 0);
    };
    reset_optimizer(M, D, Nk, Nl);
    layers.push_back(make_maps(D, Nx, Ny));
    layers.push_back(make_maps(D, Nx / s, Ny / s));
    layers.push_back(make_maps(M, Nx / s, Ny / s));
    layers.push_back(make_maps(D, Nx / s, Ny / s));
    layers.push_back(make_maps(D, Nx, Ny));
    net_c.push_back(c); net_c.push_back(f);
    net_b.push_back(b); net_b.push_back(p);
    scale.push_back(s); scale.push_back(-s);

    std::ifstream vid;
    if (!video.empty()) { vid.open(video, std::ios::binary); if (!vid) { fprintf(stderr, "cannot open %s\n", video.c_str()); return 2; } }

    for (int t = 0; t < frames; ++t) {
        if (vid.is_open()) {
            for (int d = 0; d < D; ++d) for (int i = 0; i < Nx; ++i) vid.read(reinterpret_cast<char*>(layers[0][d][i].data()), sizeof(float) * Ny);
            if (!vid) { fprintf(stderr, "video ended at frame %d\n", t); return 2; }
        } else synth_frame(layers[0], t, seed);

        // coder-decoder pass (autoencoder.cpp:131-151)
        if (fft == 1) autoenc_fft(layers, net_c, net_cfreq, net_b, scale, fft_l);
        else {
            for (size_t n = 0; n < net_c.size(); ++n) {
                const size_t nl = 2 * n;
                if (n < net_c.size() / 2) { Pool(layers[nl], layers[nl + 1], scale[n]); Conv_gpu(layers[nl + 1], layers[nl + 2], net_c[n], net_b[n]); }
                else { Conv_gpu(layers[nl], layers[nl + 1], net_c[n], net_b[n]); Pool(layers[nl + 1], layers[nl + 2], scale[n]); }
            }
        }
        // training of the active pair (autoencoder.cpp:158-205)
        if (sel == 1) {
            const size_t last = net_c.size() - 1 - n_l;
            Maps& lin = layers[2 * n_l + 1];
            const int dD = (int)lin.size(), dNx = (int)lin[0].size(), dNy = (int)lin[0][0].size(), dM = (int)layers[2 * n_l + 2].size();
            Maps in_s = make_maps(dD, dNx / q, dNy / q), out_s = make_maps(dD, dNx / q, dNy / q), hC_s = make_maps(dM, dNx / q, dNy / q);
            Portion(lin, layers[2 * n_l + 2], layers[layers.size() - 2 - 2 * n_l], in_s, hC_s, out_s, q);
            if (gpu == 1 && fft == 0) {
                if (sym == 0) backprop_gpu(in_s, out_s, hC_s, net_c[n_l], net_b[n_l], net_c[last], net_b[last], dc, db, df, dp, ddc, ddb, ddf, ddp, del, alpha, active);
                else backprop_gpu_cc(in_s, out_s, hC_s, net_c[n_l], net_b[n_l], net_c[last], net_b[last], dc, db, df, dp, ddc, ddb, ddf, ddp, del, alpha, active);
            } else if (gpu == 1 && fft == 1) {
                backprop_fft(in_s, in_s, out_s, net_cfreq[n_l], net_c[n_l], net_cfreq[last], net_c[last], net_b[n_l], net_b[last], dM, del, maxdiff);
                sel = 0;                                   // one burst per key press in FFT mode (:194)
            } else backprop(in_s, out_s, hC_s, net_c[n_l], net_b[n_l], net_c[last], net_b[last], del);
        }

        // key command for this frame (autoencoder.cpp:245-457)
        const char ch = t < (int)script.size() ? script[t] : '.';
        const size_t L = net_c.size() / 2;
        if (ch == '1') sel = (sel + 1) % 2;
        if (ch == '2') q = q + 1;
        if (ch == '3') q = std::max(1, q - 1);
        if (ch == '4') {
            del = del + ddel;
            if (del > 0.1 && del < 1) ddel = 0.1f;
            if (del > 0.01 && del < 0.1) ddel = 0.01f;
            if (del > 0.001 && del < 0.01) ddel = 0.001f;
            if (del > 0.0001 && del < 0.001) ddel = 0.0001f;
            if (del > 1) del = 1;
        }
        if (ch == '5') {
            del = del - ddel;
            if (del > 0.1 && del <= 1) ddel = 0.1f;
            if (del > 0.01 && del <= 0.11) ddel = 0.01f;
            if (del > 0.001 && del <= 0.011) ddel = 0.001f;
            if (del > 0.0001 && del <= 0.0011) ddel = 0.0001f;
            if (del < 0) del = 0;
        }
        if (ch == '6') { alpha += 0.1f; if (alpha > 1) alpha = 1; }
        if (ch == '7') { alpha -= 0.1f; if (alpha < 0) alpha = 0; }
        if (ch == '9') active = (active + 1) % 2;
        if (ch == '0') gpu = (gpu + 1) % 2;
        if (ch == 'f') fft = (fft + 1) % 2;
        if (ch == 'g') fft_l = (fft_l + 1) % 2;
        if (ch == 'm') maxdiff = (maxdiff + 1) % 2;
        if (ch == 'z' || ch == 'x') {
            // the reference evaluates (n_l -/+ 1) % (net_c.size()/2) in size_t arithmetic (:281,297)
            n_l = (int)(((size_t)(ch == 'z' ? n_l + 1 : n_l - 1)) % L);
            reset_optimizer((int)net_c[n_l].size(), (int)net_c[n_l][0].size(), (int)net_c[n_l][0][0].size(), (int)net_c[n_l][0][0][0].size());
            std::cout << "Active layer " << n_l << std::endl;
        }
        // rand() is process-global state that the GPU runtime's threads also draw from: re-seed right before every weight
        // initialisation so that a scripted session is reproducible (the application itself seeds with time(0), :99)
        if (ch == 'e' || ch == 'n') srand(seed * 2654435761u + (unsigned)(t + 1));
        if (ch == 'e') {
            const size_t last = net_c.size() - 1 - n_l;
            const int dM = (int)net_c[n_l].size(), dD = (int)net_c[n_l][0].size(), k = (int)net_c[n_l][0][0].size(), l = (int)net_c[n_l][0][0][0].size();
            int a1, a2, a3, a4; float r;
            LoadParam(a1, a2, a3, a4, r);
            Init_conv(net_c[n_l], net_b[n_l], dM, dD, k, l, r);
            Init_conv(net_c[last], net_b[last], dD, dM, k, l, r);
            net_cfreq.clear();
        }
        if (ch == 'c') net_cfreq.clear();
        if (ch == 'p') {
            sym = (sym + 1) % 2;
            if (sym == 1) {
                const size_t last = net_c.size() - 1 - n_l;
                Kernels& cc = net_c[n_l];
                for (size_t m = 0; m < cc.size(); ++m) for (size_t d = 0; d < cc[0].size(); ++d) net_c[last][d][m] = cc[m][d];
            }
        }
        if (ch == 's' || ch == 'l') {
            const size_t last = net_c.size() - 1 - n_l;
            const int write = ch == 's' ? 1 : 0;
            SaveLoad_conv(net_c[n_l], net_b[n_l], scale[n_l], n_l, 0, write);
            SaveLoad_conv(net_c[last], net_b[last], scale[last], n_l, 1, write);
            if (!write) net_cfreq.clear();
        }
        if (ch == 'n') {
            int dM = 10, lk = 0, ll = 0, scal = 2; float r = 3;
            LoadParam(dM, lk, ll, scal, r);
            const int dNk = 2 * (lk + 1) + 1, dNl = 2 * (ll + 1) + 1;
            size_t n = (layers.size() - 1) / 2;
            const int dD = (int)layers[n].size(), dNx = (int)layers[n][0].size(), dNy = (int)layers[n][0][0].size();
            layers.insert(layers.begin() + n + 1, make_maps(dD, dNx, dNy));                       // out_n
            layers.insert(layers.begin() + n + 1, make_maps(dD, dNx / scal, dNy / scal));         // PhC_n
            layers.insert(layers.begin() + n + 1, make_maps(dM, dNx / scal, dNy / scal));         // hC_n
            layers.insert(layers.begin() + n + 1, make_maps(dD, dNx / scal, dNy / scal));         // Pin_n
            Kernels c_n, f_n; Bias b_n(dM), p_n(dD);
            Init_conv(c_n, b_n, dM, dD, dNk, dNl, r);
            Init_conv(f_n, p_n, dD, dM, dNk, dNl, r);
            n = net_c.size() / 2;
            net_c.insert(net_c.begin() + n, f_n); net_c.insert(net_c.begin() + n, c_n);
            net_b.insert(net_b.begin() + n, p_n); net_b.insert(net_b.begin() + n, b_n);
            scale.insert(scale.begin() + n, -scal); scale.insert(scale.begin() + n, scal);
            reset_optimizer(dM, dD, dNk, dNl);
            n_l = (int)n;
            net_cfreq.clear();
            std::cout << "Added new layer L " << net_c.size() / 2 << std::endl;
        }
        if (ch == 'd' && net_c.size() > 2) {
            size_t n = net_c.size() / 2;
            net_c.erase(net_c.begin() + n - 1, net_c.begin() + n + 1);
            net_b.erase(net_b.begin() + n - 1, net_b.begin() + n + 1);
            scale.erase(scale.begin() + n - 1, scale.begin() + n + 1);
            n = (layers.size() - 1) / 2;
            layers.erase(layers.begin() + n - 1, layers.begin() + n + 3);
            n_l = 0;
            reset_optimizer((int)net_c[0].size(), (int)net_c[0][0].size(), (int)net_c[0][0][0].size(), (int)net_c[0][0][0][0].size());
            net_cfreq.clear();
            std::cout << "Deleted last layer" << std::endl;
        }
        if (ch == 'i') {
            for (size_t n = 0; n < net_c.size(); ++n)
                std::cout << "C=" << n << " M=" << net_c[n].size() << " D=" << net_c[n][0].size() << " Nk=" << net_c[n][0][0].size()
                          << " Nl=" << net_c[n][0][0][0].size() << " S=" << scale[n] << std::endl;
        }
    }

    // what the scripted session left behind
    std::ofstream out;
    if (!dump.empty()) out.open(dump, std::ios::binary);
    for (size_t n = 0; n < net_c.size(); ++n) {
        double sw = 0, aw = 0, sb = 0;
        for (auto& m : net_c[n]) for (auto& d : m) for (auto& k : d) { for (float v : k) { sw += v; aw += std::fabs(v); } if (out.is_open()) out.write(reinterpret_cast<const char*>(k.data()), sizeof(float) * k.size()); }
        for (float v : net_b[n]) sb += v;
        if (out.is_open()) out.write(reinterpret_cast<const char*>(net_b[n].data()), sizeof(float) * net_b[n].size());
        printf("conv %zu: M=%zu D=%zu Nk=%zu Nl=%zu S=%d sum=%.9g abs=%.9g bias=%.9g\n", n, net_c[n].size(), net_c[n][0].size(), net_c[n][0][0].size(),
               net_c[n][0][0][0].size(), scale[n], sw, aw, sb);
    }
    double so = 0;
    for (auto& d : layers.back()) for (auto& r : d) for (float v : r) so += v;
    printf("state: n_l=%d sel=%d q=%d del=%.6g alpha=%.3g fft=%d fft_l=%d sym=%d maxdiff=%d pairs=%zu out_sum=%.9g\n", n_l, sel, q, del, alpha, fft, fft_l, sym, maxdiff,
           net_c.size() / 2, so);
    if (out.is_open()) for (auto& d : layers.back()) for (auto& r : d) out.write(reinterpret_cast<const char*>(r.data()), sizeof(float) * r.size());
    return 0;
}
