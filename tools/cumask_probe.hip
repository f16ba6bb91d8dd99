// dev tool (GPU box): where do the bits of a hipExtStreamCreateWithCUMask mask land?  Launches many small workgroups on a masked stream,
// each records (XCC_ID, SE_ID, CU_ID) from the hardware registers; prints per-XCD counts of distinct CUs used, for a few masks.
//   hipcc --offload-arch=gfx950 -O2 -o build_x/cumask_probe tools/cumask_probe.hip && build_x/cumask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <set>
#include <vector>

__global__ void probe(unsigned* out, int spin)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the workgroup alive for a moment so that the dispatcher has to spread the grid over every CU it may use
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xF) | ((hw & 0xFFFF) << 4) | (a == 12345.f ? 1u << 31 : 0u);
}

static void run(const char* label, const std::vector<uint32_t>& mask)
{
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", label); return; }
    const int nb = 16384;
    unsigned* d; hipMalloc(&d, nb * 4);
    probe<<<nb, 64, 0, st>>>(d, 20000);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(nb);
    hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus[16];
    for (unsigned v : h) { const unsigned x = v & 0xF, hw = (v >> 4) & 0xFFFF; cus[x].insert(((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)); }
    printf("%-28s CUs in use per XCD:", label);
    int tot = 0;
    for (int x = 0; x < 8; ++x) { printf(" %2zu", cus[x].size()); tot += (int)cus[x].size(); }
    printf("  total %d\n", tot);
    hipFree(d); hipStreamDestroy(st);
}

int main()
{
    auto bits = [](int lo, int hi) { std::vector<uint32_t> m(8, 0); for (int i = lo; i < hi; ++i) m[i / 32] |= 1u << (i % 32); return m; };
    run("all 256", bits(0, 256));
    run("bits 0..63", bits(0, 64));
    run("bits 64..255", bits(64, 256));
    run("bits 0..31", bits(0, 32));
    run("bits 0..7", bits(0, 8));
    run("bits 0..127", bits(0, 128));
    run("bits 128..255", bits(128, 256));
    std::vector<uint32_t> ev(8, 0x55555555u);
    run("even bits", ev);
    return 0;
}
