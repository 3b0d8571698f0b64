// tools/exhaustive/exhaustive_div.hip -- experiment harness (not product), written in round 3, run in round 4 (profiles/r04/r04_div.log):
// candidate short sequences for the correctly rounded binary32 quotient a / b against hipcc's IEEE expansion
// (v_div_scale x 2, v_rcp, 5 fma, v_div_fmas, v_div_fixup: ~10 VALU instructions, 11 sites in tptTraceQueueKernel).
//
// A quotient's significand depends on the operands' significands only (scaling by powers of two is exact while every
// intermediate stays normal), so ALL pairs of significands -- 2^23 x 2^23 = 7.0e13 -- decide a candidate for every (a, b) whose
// exponents keep the intermediates normal (|exponent| <= 60 is ample: the residual a - b q is ~2^-24 a).  At ~20 instructions
// per pair that is about 40 s of an MI355X: `exhaustive_div [firstB] [countB]` walks b's significands in slices so that a run
// fits any time limit; mismatches are counted per variant and the first failing pairs are printed.
//
// Variants (y0 = v_rcp_f32(b), 1 ulp):
//   V1  q0 = a y0;                      r = fma(-b, q0, a); q = fma(r, y0, q0)                       (4 instructions)
//   V2  e = fma(-b, y0, 1); y1 = fma(e, y0, y0); q0 = a y1; r = fma(-b, q0, a); q = fma(r, y1, q0)  (6: Markstein)
//   V3  V2 + a second residual step                                                                  (8)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exhaustive/exhaustive_div.hip -o /tmp/exhaustive_div
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float divV1(float a, float b, float y0)
{
    const float q0 = a * y0;
    const float r = fmaf_(-b, q0, a);
    return fmaf_(r, y0, q0);
}
__device__ __forceinline__ float divV2(float a, float b, float y0)
{
    const float e = fmaf_(-b, y0, 1.0f);
    const float y1 = fmaf_(e, y0, y0);
    const float q0 = a * y1;
    const float r = fmaf_(-b, q0, a);
    return fmaf_(r, y1, q0);
}
__device__ __forceinline__ float divV3(float a, float b, float y0)
{
    const float e = fmaf_(-b, y0, 1.0f);
    const float y1 = fmaf_(e, y0, y0);
    float q = a * y1;
    float r = fmaf_(-b, q, a);
    q = fmaf_(r, y1, q);
    r = fmaf_(-b, q, a);
    return fmaf_(r, y1, q);
}

struct Result {
    unsigned long long bad[3];
    unsigned firstA[3], firstB[3];
    unsigned badB[3];        // how many significands of b have at least one failing a
    unsigned listB[3][16];   // the first few of them (is the failure set a handful of b's a guard could name?)
};

// one thread per significand of b; a walks all 2^23 significands of [1, 2)
__global__ void __launch_bounds__(256) divKernel(uint32_t firstB, uint32_t countB, Result* out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= countB) return;
    const uint32_t mb = firstB + t;
    const float b = u2f(0x3f800000u | mb);
    const float y0 = __builtin_amdgcn_rcpf(b);
    unsigned long long bad[3] = {0, 0, 0};
    uint32_t fa[3] = {0, 0, 0};
    for (uint32_t ma = 0; ma < (1u << 23); ++ma) {
        const float a = u2f(0x3f800000u | ma);
        const float want = a / b; // hipcc's IEEE expansion (-fno-fast-math)
        const float g1 = divV1(a, b, y0), g2 = divV2(a, b, y0), g3 = divV3(a, b, y0);
        if (f2u(g1) != f2u(want)) { if (!bad[0]) fa[0] = ma; ++bad[0]; }
        if (f2u(g2) != f2u(want)) { if (!bad[1]) fa[1] = ma; ++bad[1]; }
        if (f2u(g3) != f2u(want)) { if (!bad[2]) fa[2] = ma; ++bad[2]; }
    }
    for (int v = 0; v < 3; ++v)
        if (bad[v]) {
            if (atomicAdd(&out->bad[v], bad[v]) == 0ull) { out->firstA[v] = fa[v]; out->firstB[v] = mb; }
            const unsigned k = atomicAdd(&out->badB[v], 1u);
            if (k < 16u) out->listB[v][k] = mb;
        }
}

int main(int argc, char** argv)
{
    const uint32_t firstB = argc > 1 ? (uint32_t)strtoul(argv[1], nullptr, 0) : 0u;
    const uint32_t countB = argc > 2 ? (uint32_t)strtoul(argv[2], nullptr, 0) : (1u << 23) - firstB;
    Result* d;
    CK(hipMalloc((void**)&d, sizeof(Result)));
    CK(hipMemset(d, 0, sizeof(Result)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(divKernel, dim3((countB + 255) / 256), dim3(256), 0, 0, firstB, countB, d);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    Result h;
    CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("b significands [%u, %u) x all 2^23 significands of a: %.1f s\n", firstB, firstB + countB, ms * 1e-3);
    const char* names[3] = {"V1 (4 instr)", "V2 (6 instr, Markstein)", "V3 (8 instr)"};
    for (int v = 0; v < 3; ++v) {
        printf("%-26s mismatches %llu", names[v], h.bad[v]);
        if (h.bad[v]) {
            printf("  (first: a = 0x%08x b = 0x%08x); %u significands of b affected:", 0x3f800000u | h.firstA[v], 0x3f800000u | h.firstB[v], h.badB[v]);
            for (unsigned k = 0; k < h.badB[v] && k < 16u; ++k) printf(" %06x", h.listB[v][k]);
        }
        printf("\n");
    }
    return 0;
}
