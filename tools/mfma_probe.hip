// dev probe: operand/result lane layout of v_mfma_f32_4x4x1_16b_f32 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, float* d)
{
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[v * 64 + threadIdx.x] = c[v];
}
int main()
{
    float ha[64], hb[64], hd[256], *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    // a[lane] = 1 + lane, b[lane] = 1000 * (1 + lane): d = a*b identifies (lane_a, lane_b) of every result element
    for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 1000.f * (1 + l); }
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int v = 0; v < 4; ++v)
        for (int l = 0; l < 64; ++l) {
            const long p = (long)(hd[v * 64 + l] / 1000.f + 0.5f);
            // find la, lb with (1+la)*(1+lb) == p, la/4 == lb/4
            int fa = -1, fb = -1;
            for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if ((long)(1 + la) * (1 + lb) == p && la / 4 == lb / 4 && lb == l) { fa = la; fb = lb; }
            if (l < 8) printf("vgpr %d lane %2d = %8.0f  -> a lane %d, b lane %d\n", v, l, hd[v * 64 + l], fa, fb);
            // hypothesis: D[i=v][j=l%4] of block l/4 = A[lane 4*blk+v] * B[lane l]
            if (hd[v * 64 + l] != ha[4 * (l / 4) + v] * hb[l]) ok = 0;
        }
    printf("hypothesis D[vgpr v][lane l] = a[4*(l/4)+v] * b[l]: %s\n", ok ? "CONFIRMED" : "WRONG");
    return 0;
}
