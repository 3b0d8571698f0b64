// tools/exhaustive/exhaustive_math.hip -- experiment harness (not product): candidate fast paths for the correctly rounded
// sqrt(x) and 1/sqrt-then-reciprocal used by normalize(), compared with hipcc's correctly rounded expansions for ALL 2^32
// binary32 inputs on the device.  Prints mismatch counts per variant and the first failing inputs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float sqrtV1(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y;
    const float h = 0.5f * y;
    const float r = fmaf_(-s0, s0, x);
    return fmaf_(r, h, s0);
}
__device__ __forceinline__ float sqrtV2(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float r = fmaf_(-s0, s0, x);
    return fmaf_(r, h, s0);
}
__device__ __forceinline__ float sqrtV3(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = u2f(f2u(s) - 1u), sp = u2f(f2u(s) + 1u);
    const float rm = fmaf_(-sm, s, x), rp = fmaf_(-sp, s, x);
    float o = rm <= 0.0f ? sm : s;
    o = rp > 0.0f ? sp : o;
    return o;
}
// V4: V1 with a second correction
__device__ __forceinline__ float sqrtV4(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    float s0 = x * y;
    const float h = 0.5f * y;
    float r = fmaf_(-s0, s0, x);
    s0 = fmaf_(r, h, s0);
    r = fmaf_(-s0, s0, x);
    return fmaf_(r, h, s0);
}
// reciprocal of L = RN(sqrt(x)) starting from y ~ 1/sqrt(x)
__device__ __forceinline__ float rcpN1(float L, float y)
{
    const float e = fmaf_(-L, y, 1.0f);
    const float r1 = fmaf_(e, y, y);
    const float e1 = fmaf_(-L, r1, 1.0f);
    return fmaf_(e1, r1, r1);
}
__device__ __forceinline__ float rcpN2(float L, float y)
{
    const float e = fmaf_(-L, y, 1.0f);
    return fmaf_(e, y, y);
}

// variant id -> result for input x; ref for the same
template <int V>
__device__ __forceinline__ void eval(float x, float& got, float& ref)
{
    if (V <= 4) {
        ref = __builtin_sqrtf(x);
        got = V == 1 ? sqrtV1(x) : V == 2 ? sqrtV2(x) : V == 3 ? sqrtV3(x) : sqrtV4(x);
    } else {
        const float L = __builtin_sqrtf(x);
        ref = 1.0f / L;
        const float y = __builtin_amdgcn_rsqf(x);
        got = V == 5 ? rcpN1(L, y) : V == 6 ? rcpN2(L, y) : 0.0f;
        if (V == 7) got = rcpN1(L, __builtin_amdgcn_rcpf(L));
        if (V == 8) got = rcpN2(L, __builtin_amdgcn_rcpf(L));
    }
}

template <int V>
__global__ void sweep(uint32_t lo, uint32_t hi, unsigned long long* nBad, uint32_t* firstBad)
{
    // inputs [lo, hi] as bit patterns; 2^32 threads worth of work in a grid-stride loop
    const uint64_t n = (uint64_t)hi - lo + 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = lo + (uint32_t)i;
        float got, ref;
        eval<V>(u2f(b), got, ref);
        if (f2u(got) != f2u(ref) && !(got != got && ref != ref)) {
            const unsigned long long k = atomicAdd(nBad, 1ull);
            if (k < 8) firstBad[k] = b;
        }
    }
}

template <int V>
void run(const char* name, uint32_t lo, uint32_t hi)
{
    unsigned long long* dBad;
    uint32_t* dFirst;
    hipMalloc(&dBad, 8);
    hipMalloc(&dFirst, 32);
    hipMemset(dBad, 0, 8);
    hipMemset(dFirst, 0, 32);
    hipLaunchKernelGGL(sweep<V>, dim3(4096), dim3(256), 0, 0, lo, hi, dBad, dFirst);
    unsigned long long bad = 0;
    uint32_t first[8];
    hipMemcpy(&bad, dBad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(first, dFirst, 32, hipMemcpyDeviceToHost);
    printf("%-44s [%08x, %08x]: %llu mismatches", name, lo, hi, bad);
    for (int k = 0; k < 8 && k < (int)bad; ++k) printf(" %08x", first[k]);
    printf("\n");
    hipFree(dBad);
    hipFree(dFirst);
}

int main()
{
    // positive normal range only first (sqrt of negatives is NaN either way); then the guard candidates
    const uint32_t lo = 0x00800000u, hi = 0x7f7fffffu;
    run<1>("sqrt V1 rsq, x*y, one fma correction", lo, hi);
    run<2>("sqrt V2 v_sqrt + rsq correction", lo, hi);
    run<3>("sqrt V3 v_sqrt +-1ulp select (compiler core)", lo, hi);
    run<4>("sqrt V4 rsq, two corrections", lo, hi);
    run<5>("1/RN(sqrt) N1 from rsq, two Newton steps", lo, hi);
    run<6>("1/RN(sqrt) N2 from rsq, one Newton step", lo, hi);
    run<7>("1/RN(sqrt) N1 from v_rcp(L), two steps", lo, hi);
    run<8>("1/RN(sqrt) N2 from v_rcp(L), one step", lo, hi);
    // narrower guard range [2^-96, 2^96]
    const uint32_t glo = 0x0f800000u, ghi = 0x6f800000u;
    run<1>("sqrt V1", glo, ghi);
    run<2>("sqrt V2", glo, ghi);
    run<3>("sqrt V3", glo, ghi);
    run<4>("sqrt V4", glo, ghi);
    run<5>("N1 rsq", glo, ghi);
    run<6>("N2 rsq", glo, ghi);
    run<7>("N1 rcp", glo, ghi);
    run<8>("N2 rcp", glo, ghi);
    return 0;
}
