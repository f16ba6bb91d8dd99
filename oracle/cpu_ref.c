/* TEST INFRASTRUCTURE ONLY -- not product code.
 *
 * Plain-C restatement of the reference's single-threaded CPU path (the `gpu=0` branch,
 * /root/reference/source/netlib.cpp): Conv (318-358), backprop (361-451), Pool (114-164),
 * Portion (292-315), on flat row-major float arrays instead of nested std::vector.
 * Loop order and float operation order follow the reference line by line so that the
 * outputs are bit-identical to the reference compiled with the same flags (checked in
 * tests/test_oracle_cpu.py against oracle/_ref/libnetlib_ref.so when that exists).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Layouts: in/out [dD][Nx][Ny], hin [dM][Nx][Ny], c [dM][dD][Nk][Nl], f [dD][dM][Nk][Nl].
 * act(x)=x, act1(x)=1 (backproplib.cu:38-51).
 */
#include <math.h>
#include <stddef.h>

static float act(float x) { return x; }
static float act1(float x) { (void)x; return 1; }

/* netlib.cpp:318-358 */
void cpu_ref_conv(const float *in, float *out, const float *c, const float *b,
                  int dD, int dM, int Nx, int Ny, int Nk, int Nl)
{
    int ak = ((Nk - 1) / 2 - 1);
    int al = ((Nl - 1) / 2 - 1);
    for (int m = 0; m < dM; m++) {
        for (int i = 0; i < Nx; i++) {
            for (int j = 0; j < Ny; j++) {
                float h = 0;
                for (int d = 0; d < dD; d++) {
                    int ik = -2 * ak - 1;
                    for (int k = 0; k < Nk; k++) {
                        int il = -2 * al - 1;
                        for (int l = 0; l < Nl; l++) {
                            /* netlib.cpp:344 -- strict '>0': row/col 0 of the input is never read */
                            if (i - ik > 0 && i - ik < Nx && j - il > 0 && j - il < Ny)
                                h += c[((m * dD + d) * Nk + k) * Nl + l] *
                                     in[((size_t)d * Nx + (i - ik)) * Ny + (j - il)];
                            il += 1;
                        }
                        ik += 1;
                    }
                }
                h += b[m];
                out[((size_t)m * Nx + i) * Ny + j] = act(h);
            }
        }
    }
}

static float clipf(float g) { return (10 < fabsf(g)) ? fabsf(g) : 10; }

/* netlib.cpp:361-451.  Returns the printed (un-normalised) squared distance (`:375-385`).
 * c, f, b, p are updated in place INSIDE the loop nest (`:437-443`), so later weight
 * elements see already-updated f -- replicated. */
float cpu_ref_backprop(const float *in, const float *out, const float *hin,
                       float *c, float *b, float *f, float *p, float del,
                       int dD, int dM, int Nx, int Ny, int Nk, int Nl)
{
    int ak = ((Nk - 1) / 2 - 1);
    int al = ((Nl - 1) / 2 - 1);
    float Norm = (float)(dD * dM * Nk * Nl * Nx * Ny);
    float dist = 0;
    for (int d1 = 0; d1 < dD; d1++)
        for (int i = 0; i < Nx; i++)
            for (int j = 0; j < Ny; j++) {
                size_t q = ((size_t)d1 * Nx + i) * Ny + j;
                /* std::pow(float,int) promotes to double (C++11), sum truncated to float */
                dist = (float)((double)dist + pow((double)(in[q] - out[q]), 2.0));
            }
    for (int m = 0; m < dM; m++) {
        for (int d = 0; d < dD; d++) {
            int ik = -2 * ak - 1;
            for (int k = 0; k < Nk; k++) {
                int il = -2 * al - 1;
                for (int l = 0; l < Nl; l++) {
                    float dDdC = 0, dDdF = 0, dDdB = 0, dDdP = 0;
                    for (int d1 = 0; d1 < dD; d1++) {
                        for (int i = 0; i < Nx; i++) {
                            for (int j = 0; j < Ny; j++) {
                                float dDdB1 = 0;
                                float dDdC1 = 0;
                                int ik1 = -2 * ak - 1;
                                for (int k1 = 0; k1 < Nk; k1++) {
                                    int il1 = -2 * al - 1;
                                    for (int l1 = 0; l1 < Nl; l1++) {
                                        if (i - ik1 > 0 && i - ik1 < Nx && j - il1 > 0 && j - il1 < Ny) {
                                            float prod = f[((d1 * dM + m) * Nk + k1) * Nl + l1] *
                                                         act1(hin[((size_t)m * Nx + (i - ik1)) * Ny + (j - il1)]);
                                            dDdB1 += prod;
                                            if (i - ik1 - ik > 0 && i - ik1 - ik < Nx && j - il1 - il > 0 && j - il1 - il < Ny)
                                                dDdC1 += prod * in[((size_t)d * Nx + (i - ik1 - ik)) * Ny + (j - il1 - il)];
                                        }
                                        il1 += 1;
                                    }
                                    ik1 += 1;
                                }
                                size_t q = ((size_t)d1 * Nx + i) * Ny + j;
                                float sum0 = (out[q] - in[q]) * act1(out[q]);
                                dDdC += sum0 * dDdC1 / Norm;
                                dDdB += sum0 * dDdB1 / Norm;
                                if (d1 == d) {
                                    if (i - ik > 0 && i - ik < Nx && j - il > 0 && j - il < Ny)
                                        dDdF += sum0 * act(hin[((size_t)m * Nx + (i - ik)) * Ny + (j - il)]) / Norm;
                                    dDdP += sum0 / Norm;
                                }
                            }
                        }
                    }
                    c[((m * dD + d) * Nk + k) * Nl + l] += -del * dDdC / clipf(dDdC);
                    f[((d * dM + m) * Nk + k) * Nl + l] += -del * dDdF / clipf(dDdF);
                    if (k == 0 && l == 0) {
                        if (d == 0) b[m] += -del * dDdB / clipf(dDdB);
                        if (m == 0) p[d] += -del * dDdP / clipf(dDdP);
                    }
                    il += 1;
                }
                ik += 1;
            }
        }
    }
    return dist;
}

/* netlib.cpp:114-164.  scale>0: max-pool with `int smax=0` (truncation to int, clamp at 0,
 * even at scale 1); scale<0: nearest-neighbour up-sample.  in [D][Nxi][Nyi], out [D][Nxo][Nyo]. */
void cpu_ref_pool(const float *in, float *out, int D, int Nxi, int Nyi, int Nxo, int Nyo, int scale)
{
    if (scale > 0) {
        int Nx = Nxi, Ny = Nyi;
        for (int d = 0; d < D; d++)
            for (int i = 0; i < Nx; i += scale)
                for (int j = 0; j < Ny; j += scale) {
                    int smax = 0;
                    for (int k = 0; k < scale; k++)
                        for (int l = 0; l < scale; l++)
                            if (i + k < Nx && j + l < Ny && in[((size_t)d * Nxi + i + k) * Nyi + j + l] > smax)
                                smax = (int)in[((size_t)d * Nxi + i + k) * Nyi + j + l];
                    out[((size_t)d * Nxo + i / scale) * Nyo + j / scale] = (float)smax;
                }
    } else {
        int Nx = Nxo, Ny = Nyo;
        scale = -scale;
        for (int d = 0; d < D; d++)
            for (int i = 0; i < Nx; i += scale)
                for (int j = 0; j < Ny; j += scale)
                    for (int k = 0; k < scale; k++)
                        for (int l = 0; l < scale; l++)
                            if (i + k < Nx && j + l < Ny)
                                out[((size_t)d * Nxo + i + k) * Nyo + j + l] =
                                    in[((size_t)d * Nxi + i / scale) * Nyi + j / scale];
    }
}

/* netlib.cpp:292-315: centred 1/q crop of one [ch][Nx][Ny] tensor into [ch][Nx/q][Ny/q]. */
void cpu_ref_portion(const float *in, float *in_s, int ch, int Nx, int Ny, int q)
{
    int dx = (Nx - Nx / q) / 2;
    int dy = (Ny - Ny / q) / 2;
    int Nxs = Nx / q, Nys = Ny / q;
    for (int i = 0; i < Nxs; i++)
        for (int j = 0; j < Nys; j++)
            for (int d = 0; d < ch; d++)
                in_s[((size_t)d * Nxs + i) * Nys + j] = in[((size_t)d * Nx + i + dx) * Ny + j + dy];
}
