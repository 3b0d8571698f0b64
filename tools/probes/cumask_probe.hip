// tools/probes/cumask_probe.hip -- which physical CUs does bit i of a hipExtStreamCreateWithCUMask mask select on an
// 8-XCD MI355X?  (experiment harness, not product).  For a few one-bit / few-bit masks, a grid of one-wave workgroups
// records HW_REG_XCC_ID and HW_REG_HW_ID; the host prints the (xcc, se, sh, cu) set that ran them.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void whereKernel(uint32_t* out, int spin)
{
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20); // HW_REG_XCC_ID
    // keep the CU busy for a moment so that the dispatcher has to spread the grid over everything the mask allows
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
}

static void run(const char* name, const std::vector<uint32_t>& mask)
{
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%-28s create failed: %s\n", name, hipGetErrorString(e)); return; }
    const int n = 4096;
    uint32_t* d;
    CK(hipMalloc((void**)&d, n * 8));
    hipLaunchKernelGGL(whereKernel, dim3(n), dim3(64), 0, s, d, 2000); // 20 us per workgroup
    CK(hipStreamSynchronize(s));
    std::vector<uint32_t> h(n * 2);
    CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    std::map<int, std::set<int>> perXcc; // xcc -> {se<<8 | sh<<4 | cu}
    for (int i = 0; i < n; ++i) {
        const uint32_t hw = h[i * 2], xcc = h[i * 2 + 1] & 15u;
        const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        perXcc[(int)xcc].insert(se << 8 | sh << 4 | cu);
    }
    int total = 0;
    printf("%-28s", name);
    for (auto& kv : perXcc) {
        total += (int)kv.second.size();
        printf(" xcc%d:%zu[", kv.first, kv.second.size());
        int k = 0;
        for (int v : kv.second) { if (k++ < 4) printf("%s%d.%d.%d", k > 1 ? "," : "", v >> 8, (v >> 4) & 1, v & 15); }
        printf("%s]", kv.second.size() > 4 ? ",.." : "");
    }
    printf("  = %d CUs\n", total);
    CK(hipFree(d));
    CK(hipStreamDestroy(s));
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
    const int words = (p.multiProcessorCount + 31) / 32;
    auto bits = [&](std::initializer_list<int> on) { std::vector<uint32_t> m(words, 0u); for (int b : on) m[b / 32] |= 1u << (b % 32); return m; };
    run("all", std::vector<uint32_t>(words, 0xffffffffu));
    run("bit 0", bits({0}));
    run("bit 1", bits({1}));
    run("bit 7", bits({7}));
    run("bit 8", bits({8}));
    run("bit 31", bits({31}));
    run("bit 32", bits({32}));
    run("bit 255", bits({255}));
    run("bits 0-7", bits({0, 1, 2, 3, 4, 5, 6, 7}));
    run("bits 0,32,64,..,224", bits({0, 32, 64, 96, 128, 160, 192, 224}));
    { std::vector<uint32_t> m(words, 0xffffffffu); m[0] &= ~0xffu; run("all but bits 0-7", m); }
    { std::vector<uint32_t> m(words, 0xffffffffu); for (int w = 0; w < words; ++w) m[w] &= ~1u; run("all but bits 0,32,..", m); }
    { std::vector<uint32_t> m(words, 0xffffffffu); m[words - 1] &= ~0xff000000u; run("all but bits 248-255", m); }
    return 0;
}
