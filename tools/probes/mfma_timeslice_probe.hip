// Stand-alone probe for DESIGN.md 2.2: does a chain of v_mfma_f32_32x32x16_f16 whose A tiles are read from GLOBAL memory give the same
// answer every time when the process holds more hardware queues than the device runs side by side (the device then time-slices
// the queues)?  No library code: a table of binary16 A tiles, B operands made from (lane, iteration), the chain of matrixApply
// (two independent MFMAs on tile 0, two accumulating ones on tile 1), the sign bits of the 32 result registers folded into a
// per-wave checksum.  The same launch is repeated; every launch must produce the same checksums.
//
//   hipcc -O3 --offload-arch=gfx950 -o mfma_timeslice_probe mfma_timeslice_probe.hip
//   GPU_MAX_HW_QUEUES=32 ./mfma_timeslice_probe <extra streams> <launches> <mode>      mode 0: A from global, 1: A staged in LDS,
//                                                                                      2: no MFMA (the same loads, VALU fold)
//   mode 3 = mode 0 plus ingredients of the grouped kernel, argv[5] = feature mask: 1 per-lane gathers in flight around the chain,
//   2 a per-wave LDS pair list (returning ds_add, list write / read, ds_min_u64), 4 ~48 more live registers, 8 LDS request padded to
//   81 920 B per workgroup (two workgroups fill a CU), 16 a dynamically indexed private array (scratch), 32 v_permlane32_swap on the masks
//   and on B operands with gathers in flight
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);      \
            exit(2);                                                                         \
        }                                                                                    \
    } while (0)

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

constexpr int kTiles = 8;                 // 8 tables of 64 rows: what 512 groups need
constexpr int kTileVec = 4 * 64;          // uint4 per table: [sphere tile 2][k step 2][lane 64]
constexpr int kThreads = 512;

__device__ __forceinline__ v8h asH(const uint4& a)
{
    v8h h;
    __builtin_memcpy(&h, &a, 16);
    return h;
}
__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// binary16 pair with small finite values: exponent field 13..16 (0.25 .. 4), random sign and mantissa
__device__ __host__ __forceinline__ uint32_t halfPair(uint32_t r)
{
    const uint32_t lo = (r & 0x83ffu) | ((13u + ((r >> 10) & 3u)) << 10);
    const uint32_t hi = ((r >> 16) & 0x83ffu) | ((13u + ((r >> 26) & 3u)) << 10);
    return lo | (hi << 16);
}

template <int MODE>
__global__ void __launch_bounds__(kThreads, 4) probe(const uint4* __restrict__ A, int iters, unsigned long long* __restrict__ out)
{
    __shared__ uint4 ldsA[MODE == 1 ? kTiles * kTileVec : 1];
    const int lane = threadIdx.x & 63;
    if (MODE == 1) {
        for (int i = threadIdx.x; i < kTiles * kTileVec; i += kThreads) ldsA[i] = A[i];
        __syncthreads();
    }
    const uint4* T = MODE == 1 ? ldsA : A;
    unsigned long long sum = 0;
    const uint32_t seed = mix(blockIdx.x * 1315423911u + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
        // B operands: two k steps x two ray tiles, 4 registers each
        uint4 b[4];
        uint32_t r = mix(seed + (uint32_t)it * 0x9e3779b9u);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b[k].x = halfPair(r = mix(r + 1u));
            b[k].y = halfPair(r = mix(r + 2u));
            b[k].z = halfPair(r = mix(r + 3u));
            b[k].w = halfPair(r = mix(r + 4u));
        }
#pragma unroll 1
        for (int t = 0; t < kTiles; ++t) {
            const uint4* P = T + t * kTileVec;
            uint32_t W0 = 0, W1 = 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const uint4 a0 = P[(2 * half + 0) * 64 + lane], a1 = P[(2 * half + 1) * 64 + lane];
                if (MODE == 2) { // same loads, no matrix cores
                    W0 = W0 * 31u + (a0.x ^ b[0].x) + (a1.y ^ b[2].y);
                    W1 = W1 * 31u + (a0.z ^ b[1].z) + (a1.w ^ b[3].w);
                } else {
                    v16f c0 = {0}, c1 = {0};
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asH(a0), asH(b[0]), c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asH(a0), asH(b[1]), c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asH(a1), asH(b[2]), c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asH(a1), asH(b[3]), c1, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        W0 = __builtin_amdgcn_alignbit(W0, __float_as_uint(c0[q]), 31);
                        W1 = __builtin_amdgcn_alignbit(W1, __float_as_uint(c1[q]), 31);
                    }
                }
            }
            sum = sum * 0x100000001b3ull + (((unsigned long long)W0 << 32) | W1);
        }
    }
    out[(size_t)blockIdx.x * kThreads + threadIdx.x] = sum;
}

extern __shared__ __attribute__((aligned(16))) unsigned char dynLds[];
__global__ void __launch_bounds__(kThreads, 4) probeLike(const uint4* __restrict__ A, const uint4* __restrict__ members, int iters, int feat,
                                                          unsigned long long* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per-wave LDS region: pair list (192 entries), two counters, 64 keys
    volatile unsigned* list = reinterpret_cast<volatile unsigned*>(dynLds + wave * 2048);
    unsigned* count = reinterpret_cast<unsigned*>(dynLds + wave * 2048 + 768);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(dynLds + wave * 2048 + 1024);
    if (feat & 2) keys[lane] = ~0ull;
    unsigned long long sum = 0;
    float acc[48];
#pragma unroll
    for (int j = 0; j < 48; ++j) acc[j] = (float)(lane + j);
    float priv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) priv[j] = (float)j;
    const uint32_t seed = mix(blockIdx.x * 1315423911u + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
        uint4 b[4];
        uint32_t r = mix(seed + (uint32_t)it * 0x9e3779b9u);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b[k].x = halfPair(r = mix(r + 1u));
            b[k].y = halfPair(r = mix(r + 2u));
            b[k].z = halfPair(r = mix(r + 3u));
            b[k].w = halfPair(r = mix(r + 4u));
        }
#pragma unroll 1
        for (int t = 0; t < kTiles; ++t) {
            const uint4* P = A + t * kTileVec;
            uint4 g0 = {0, 0, 0, 0}, g1 = g0, g2 = g0, g3 = g0;
            if (feat & 1) { // a group's members, gathered per lane, in flight while the chain runs
                const uint4* M = members + (size_t)(mix(r + (uint32_t)t) & 4095u) * 8u;
                g0 = M[0]; g1 = M[1]; g2 = M[2]; g3 = M[3];
            }
            uint32_t W0 = 0, W1 = 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const uint4 a0 = P[(2 * half + 0) * 64 + lane], a1 = P[(2 * half + 1) * 64 + lane];
                v16f c0 = {0}, c1 = {0};
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asH(a0), asH(b[0]), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asH(a0), asH(b[1]), c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asH(a1), asH(b[2]), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asH(a1), asH(b[3]), c1, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    W0 = __builtin_amdgcn_alignbit(W0, __float_as_uint(c0[q]), 31);
                    W1 = __builtin_amdgcn_alignbit(W1, __float_as_uint(c1[q]), 31);
                }
            }
            if (feat & 2) { // the dealing's LDS protocol, one wave, no barrier
                if (lane == 0) *count = 0u;
                __builtin_amdgcn_wave_barrier();
                const unsigned n = 1u + (W0 & 1u);
                const unsigned pos = atomicAdd(count, n);
                for (unsigned k = 0; k < n; ++k)
                    if (pos + k < 192u) list[pos + k] = ((unsigned)lane << 16) | (unsigned)t;
                __builtin_amdgcn_wave_barrier();
                const unsigned total = __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned e = (unsigned)lane < total ? list[lane] : 0u;
                atomicMin(&keys[(e >> 16) & 63u], ((unsigned long long)(W1 | 1u) << 32) | (unsigned long long)(e & 0xffffu));
                __builtin_amdgcn_wave_barrier();
            }
            if (feat & 4) {
#pragma unroll
                for (int j = 0; j < 48; ++j) acc[j] = acc[j] * 1.0001f + __uint_as_float((W0 >> (j & 15)) & 0x3f800000u);
            }
            if (feat & 16) priv[(W1 + (unsigned)lane) & 15u] += 1.0f;
            if (feat & 32) { // v_permlane32_swap on the masks (as at the end of matrixApply) and on a B operand pair, with gathers in flight
                const uint4* M2 = members + (size_t)(mix(r + 77u + (uint32_t)t) & 4095u) * 8u;
                const uint4 h0 = M2[4], h1 = M2[5];
                auto sw = __builtin_amdgcn_permlane32_swap(W0, W1, false, false);
                W0 = sw[0]; W1 = sw[1];
                auto sb = __builtin_amdgcn_permlane32_swap(b[0].x, b[1].x, false, false);
                W0 += sb[0]; W1 ^= sb[1];
                W0 ^= h0.x ^ h1.w;
            }
            if (feat & 1) W0 ^= g0.x ^ g1.y ^ g2.z ^ g3.w;
            sum = sum * 0x100000001b3ull + (((unsigned long long)W0 << 32) | W1);
        }
    }
    if (feat & 2) sum ^= keys[lane];
    if (feat & 4) {
        float f = 0.0f;
#pragma unroll
        for (int j = 0; j < 48; ++j) f += acc[j];
        sum ^= __float_as_uint(f);
    }
    if (feat & 16) {
        float f = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) f += priv[j];
        sum += __float_as_uint(f);
    }
    out[(size_t)blockIdx.x * kThreads + threadIdx.x] = sum;
}

__global__ void touch(int* p) { if (p) *p = 1; }

int main(int argc, char** argv)
{
    const int extra = argc > 1 ? atoi(argv[1]) : 0, launches = argc > 2 ? atoi(argv[2]) : 40, mode = argc > 3 ? atoi(argv[3]) : 0;
    const double targetMs = argc > 4 ? atof(argv[4]) : 30.0;
    const int feat = argc > 5 ? atoi(argv[5]) : 0;
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 2;
    std::vector<uint32_t> hostA((size_t)kTiles * kTileVec * 4);
    uint32_t r = 12345u;
    for (auto& w : hostA) {
        r = r * 1664525u + 1013904223u;
        w = halfPair(r ^ (r >> 13));
    }
    uint4* dA = nullptr;
    unsigned long long* dOut = nullptr;
    const size_t nOut = (size_t)blocks * kThreads;
    CHECK(hipMalloc(reinterpret_cast<void**>(&dA), hostA.size() * 4));
    CHECK(hipMalloc(reinterpret_cast<void**>(&dOut), nOut * 8));
    CHECK(hipMemcpy(dA, hostA.data(), hostA.size() * 4, hipMemcpyHostToDevice));
    uint4* dMembers = nullptr; // 4096 groups x 8 members x 16 B
    {
        std::vector<uint32_t> hostM((size_t)4096 * 8 * 4);
        for (auto& w : hostM) {
            r = r * 1664525u + 1013904223u;
            w = r;
        }
        CHECK(hipMalloc(reinterpret_cast<void**>(&dMembers), hostM.size() * 4));
        CHECK(hipMemcpy(dMembers, hostM.data(), hostM.size() * 4, hipMemcpyHostToDevice));
    }
    const size_t likeLds = (feat & 8) ? 81920 : 8 * 2048;
    if (mode == 3) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probeLike), hipFuncAttributeMaxDynamicSharedMemorySize, (int)likeLds));
    std::vector<hipStream_t> streams(extra);
    for (auto& s : streams) { // every stream used once: it gets its hardware queue
        CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipLaunchKernelGGL(touch, dim3(1), dim3(1), 0, s, (int*)nullptr);
    }
    CHECK(hipDeviceSynchronize());
    hipStream_t main;
    CHECK(hipStreamCreateWithFlags(&main, hipStreamNonBlocking));
    auto launch = [&](int iters) {
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(kThreads), 0, main, dA, iters, dOut);
        else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(kThreads), 0, main, dA, iters, dOut);
        else if (mode == 3) hipLaunchKernelGGL(probeLike, dim3(blocks), dim3(kThreads), likeLds, main, dA, dMembers, iters, feat, dOut);
        else hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(kThreads), 0, main, dA, iters, dOut);
        CHECK(hipGetLastError());
        CHECK(hipStreamSynchronize(main));
    };
    // size the launch: ~targetMs each
    int iters = 64;
    launch(iters);
    auto t0 = std::chrono::steady_clock::now();
    launch(iters);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    iters = (int)(iters * targetMs / (ms > 0.01 ? ms : 0.01));
    if (iters < 16) iters = 16;
    std::vector<unsigned long long> ref(nOut), got(nOut);
    launch(iters);
    CHECK(hipMemcpy(ref.data(), dOut, nOut * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    size_t badWords = 0;
    t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < launches; ++k) {
        CHECK(hipMemsetAsync(dOut, 0, nOut * 8, main));
        launch(iters);
        CHECK(hipMemcpy(got.data(), dOut, nOut * 8, hipMemcpyDeviceToHost));
        size_t d = 0;
        for (size_t i = 0; i < nOut; ++i) d += got[i] != ref[i];
        if (d) {
            ++bad;
            badWords += d;
            if (bad <= 3) {
                for (size_t i = 0; i < nOut; ++i)
                    if (got[i] != ref[i]) {
                        printf("  launch %d: thread %zu (block %zu wave %zu lane %zu) %016llx instead of %016llx (%zu threads differ)\n", k, i, i / kThreads,
                               (i % kThreads) / 64, i % 64, got[i], ref[i], d);
                        break;
                    }
            }
        }
    }
    ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (mode == 3) printf("features %d: ", feat);
    printf("mode %d (%s), %d extra streams, GPU_MAX_HW_QUEUES=%s: %d of %d launches differ from the first (%zu thread sums), %d iterations per wave, %.1f ms per launch\n",
           mode, mode == 0 ? "A from global memory" : mode == 1 ? "A staged in LDS" : mode == 3 ? "A from global memory + ingredients" : "no MFMA", extra, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)",
           bad, launches, badWords, iters, ms / launches);
    return 0;
}
