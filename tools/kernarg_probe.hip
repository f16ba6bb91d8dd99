// dev probe: cost of a large by-value kernel argument read with a dynamic index vs the same data behind a device pointer
#include <hip/hip_runtime.h>
#include <cstdio>
struct Desc { const float* A; long a_r, a_k; const float* B; long b_k, b_c; float* Out; long o_r, o_c; int R, C, K; long P; float x[16]; long pad[8]; };
struct Big { Desc q[8]; int n; int start[9]; };
__global__ void by_value(const Big g, float* out)
{
    int p = 0;
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
    const Desc& q = g.q[p];
    float s = 0;
    for (int k = 0; k < q.K; ++k) s += q.x[k & 15] * (float)(q.a_r + q.b_k + q.o_r + k);
    if (s == 12345.f) out[threadIdx.x] = s + q.P + q.R + q.C;
}
__global__ void by_pointer(const Big* __restrict__ gp, float* out)
{
    const Big& g = *gp;
    int p = 0;
    for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
    const Desc& q = g.q[p];
    float s = 0;
    for (int k = 0; k < q.K; ++k) s += q.x[k & 15] * (float)(q.a_r + q.b_k + q.o_r + k);
    if (s == 12345.f) out[threadIdx.x] = s + q.P + q.R + q.C;
}
__global__ void tiny(int K, float* out)
{
    float s = 0;
    for (int k = 0; k < K; ++k) s += (float)k;
    if (s == 12345.f) out[threadIdx.x] = s;
}
int main()
{
    Big h{}; h.n = 8;
    for (int i = 0; i < 8; ++i) { h.q[i].K = 32; h.q[i].a_r = i; h.start[i] = i * 1000; }
    h.start[8] = 8000;
    Big* d; float* out;
    (void)hipMalloc(&d, sizeof(Big)); (void)hipMalloc(&out, 4096);
    (void)hipMemcpy(d, &h, sizeof(Big), hipMemcpyHostToDevice);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    printf("sizeof(Big) = %zu\n", sizeof(Big));
    for (int blocks : {64, 2048, 8192, 32768}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int w = 0; w < 5; ++w) { if (mode == 0) by_value<<<blocks, 256>>>(h, out); else if (mode == 1) by_pointer<<<blocks, 256>>>(d, out); else tiny<<<blocks, 256>>>(32, out); }
            (void)hipEventRecord(a);
            for (int w = 0; w < 200; ++w) { if (mode == 0) by_value<<<blocks, 256>>>(h, out); else if (mode == 1) by_pointer<<<blocks, 256>>>(d, out); else tiny<<<blocks, 256>>>(32, out); }
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf("blocks %6d  %-10s %7.2f us/launch\n", blocks, mode == 0 ? "by_value" : mode == 1 ? "by_pointer" : "tiny", ms * 1000 / 200);
        }
    }
    return 0;
}
