// dev tool (GPU box): what the access pattern of the per-bin contraction costs.  A workgroup of 4 waves owns 32 bins and reads, for each
// of NP planes, the 256 bytes of its bins -- (a) planar layout [plane][P]: NP pieces of 256 B one plane (P*8 bytes) apart, what
// contract_mfma_kernel does on planar kernel spectra; (b) bin-tiled layout [tile][plane][32]: one contiguous run of NP*256 bytes.
// Same bytes, same instructions (8-byte loads per lane, 4 waves sharing the planes 4 ways), a checksum so nothing is elided.
//   hipcc --offload-arch=gfx950 -O3 -o build_x/stride_probe tools/stride_probe.hip && build_x/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool TILED>
__global__ __launch_bounds__(256) void read_kernel(const float2* __restrict__ W, float* __restrict__ out, long P, int NP)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long tile = blockIdx.x;
    const int bin = lane & 31, half = lane >> 5;                  // a wave reads two planes per instruction (32 lanes x 8 B each)
    float2 acc = make_float2(0.f, 0.f);
    for (int p0 = wv * 2 + half; p0 < NP; p0 += 8 * 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = min(p0 + 8 * u, NP - 1);
            v[u] = TILED ? W[(tile * NP + p) * 32 + bin] : W[(long)p * P + tile * 32 + bin];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; }
    }
    if (acc.x == 12345.678f) out[0] = acc.y;                     // (never true: keeps the loads)
}

// (c) the contraction's own instruction shape: 16-byte loads, lane = 16*sub + blk -- one instruction covers 4 planes x 256 B;
// (d) 16-byte loads, one plane x 1 KB per instruction (a wave owns 128 bins)
template <int MODE>
__global__ __launch_bounds__(256) void read16_kernel(const float4* __restrict__ W, float* __restrict__ out, long P, int NP)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 0) {
        const long tile = blockIdx.x;                              // 32 bins = 16 float4 per plane
        const int blk = lane & 15, sub = lane >> 4;
        for (int p0 = wv * 4 + sub; p0 < NP; p0 += 16 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int p = min(p0 + 16 * u, NP - 1); v[u] = W[((long)p * P + tile * 32) / 2 + blk]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    } else {
        const long tile = blockIdx.x;                              // 128 bins = 64 float4 per plane: gridDim.x = P / 128
        for (int p0 = wv; p0 < NP; p0 += 4 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int p = min(p0 + 4 * u, NP - 1); v[u] = W[((long)p * P + tile * 128) / 2 + lane]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    if (acc.x == 12345.678f) out[0] = acc.y + acc.z + acc.w;
}

int main()
{
    const long P = 131584;                                       // bins of a 512 x 257 grid
    const int NP = 2048;                                         // planes of the 32 -> 64 pair
    const size_t n = (size_t)P * NP;
    float2* W; float* out;
    CHECK(hipMalloc(&W, n * sizeof(float2))); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(W, 0, n * sizeof(float2)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned tiles = (unsigned)(P / 32);
    float ms[2];
    for (int t = 0; t < 2; ++t) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            if (t) read_kernel<true><<<tiles, 256>>>(W, out, P, NP); else read_kernel<false><<<tiles, 256>>>(W, out, P, NP);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms[t], e0, e1));
        }
    }
    float ms16[2];
    for (int t = 0; t < 2; ++t)
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            if (t) read16_kernel<1><<<(unsigned)(P / 128), 256>>>(reinterpret_cast<const float4*>(W), out, P, NP);
            else read16_kernel<0><<<tiles, 256>>>(reinterpret_cast<const float4*>(W), out, P, NP);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms16[t], e0, e1));
        }
    const double gb = (double)n * 8 / 1e9;
    printf("{\"x16_4planes_per_instr_us\": %.1f, \"GBps\": %.0f, \"x16_1plane_1KB_per_instr_us\": %.1f, \"GBps_\": %.0f}\n", ms16[0] * 1e3, gb / ms16[0] * 1e3, ms16[1] * 1e3, gb / ms16[1] * 1e3);
    printf("{\"probe\": \"%d planes x %ld bins of complex64 (%.2f GB) read by 32-bin workgroups\", \"planar_us\": %.1f, \"planar_GBps\": %.0f, \"bin_tiled_us\": %.1f, \"bin_tiled_GBps\": %.0f}\n",
           NP, P, gb, ms[0] * 1e3, gb / ms[0] * 1e3, ms[1] * 1e3, gb / ms[1] * 1e3);
    return 0;
}
