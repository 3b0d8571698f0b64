// tpt_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the path tracer.
//
// One kernel family replaces the reference's DrawTest fan-out (Cpp/Source/Test.cpp:344-367:
// enkiTS task set over rows -> TraceRowJob :266-300).  Mapping to the hardware:
//
//   * work item  = one pixel (PER_PIXEL seed mode) or one row (ROW_SERIAL, the reference's
//     Test.cpp:280 RNG stream; debugging / bit-for-bit reproduction of the CPU image);
//   * lane       = runs the flattened Trace/Scatter state machine of tpt_trace.h, one ray per step;
//   * wave (64)  = 8x8 pixel tiles, so the camera rays of a wave are coherent;
//   * persistent variant: each wave pulls chunks of 256 pixels from a global atomic counter and
//     RE-FILLS idle lanes with the next pixel of its chunk (ballot + prefix count), so lanes whose
//     path ended early (sky after one ray) do not wait for the 11-bounce neighbours.  Chunks are
//     numbered bottom-up (y = 0 is the expensive, sphere-covered bottom of the image; the cheap
//     sky rows come last), which keeps the tail of the launch short.
//   * scene: phase-1 sphere pairs are read with scalar loads (wave-uniform), the {centre, r^2}
//     records, 1/r and the light list are staged into LDS once per workgroup for the per-lane
//     phase-2 gather; materials are read from global memory (L1/L2 resident, only at a hit).
//   * output: float4 accumulation buffer, RGB read-modify-write per pixel (alpha untouched),
//     ray counts reduced per wave (shuffle) -> one 64-bit atomic per wave.
//
// No MFMA: there is no dense contraction in this path (46-long select/min reduction per lane).
#include "tpt_device.h"

namespace tpt {

__device__ __forceinline__ unsigned waveReduceAdd(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// idx -> pixel.  Returns false for padding slots of partially covered tiles.
__device__ __forceinline__ bool mapItem(const KernelArgs& a, int idx, int& x, int& ly)
{
    if (a.fc.seedMode == SEED_ROW_SERIAL) {
        x = 0;
        ly = idx;
        return ly < a.nLocalRows;
    }
    int tile = idx >> 6, within = idx & 63;
    int tx = tile % a.tilesX, ty = tile / a.tilesX;
    x = tx * 8 + (within & 7);
    ly = ty * 8 + (within >> 3);
    return x < a.fc.width && ly < a.nLocalRows;
}
__device__ __forceinline__ int localRowToGlobal(const KernelArgs& a, int ly)
{
    return (ly / a.stripeRows) * a.stripeStride + a.stripeOffset + (ly % a.stripeRows);
}

__device__ __forceinline__ void storeColour(const KernelArgs& a, const Lane& L)
{
    f3 c = lanePixelColour(L, a.fc);
    f4 v;
    v.x = c.x; v.y = c.y; v.z = c.z; v.w = 0.0f;
    a.frameColour[L.pix] = v; // one 16-B store per pixel
}

// Progressive accumulation, Test.cpp:293-295: tile.rgb = tile.rgb*lerpFac + colour*(1-lerpFac); alpha kept.
// Separate from the trace kernel so that consecutive frames' trace kernels carry no dependency on each
// other and can overlap on the device (the tail of frame f runs beside the head of frame f+1).  HBM-bound:
// 48 B per pixel (read tile + colour, write tile).
__global__ void __launch_bounds__(256) tptResolveKernel(float* __restrict__ tile, const f4* __restrict__ colour, int nPixels, float lerpFac)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nPixels) return;
    f4 t = reinterpret_cast<const f4*>(tile)[i];
    f4 c = colour[i];
    f3 r = blendPixel(mk3(t.x, t.y, t.z), mk3(c.x, c.y, c.z), lerpFac);
    t.x = r.x; t.y = r.y; t.z = r.z;
    reinterpret_cast<f4*>(tile)[i] = t;
}

template <int HS, int FOLD, bool PERSIST, bool LDS_SCENE>
__global__ void __launch_bounds__(TPT_BLOCK) tptTraceKernel(const KernelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // ---- carve LDS (every offset a multiple of 16)
    const int nPad = a.scene.nPairs * 2;
    f4* ldsSph = reinterpret_cast<f4*>(smem);
    int off = LDS_SCENE ? nPad * 16 : 0;
    float* ldsInvR = reinterpret_cast<float*>(smem + off);
    off += LDS_SCENE ? ((nPad * 4 + 15) & ~15) : 0;
    f4* ldsLights = reinterpret_cast<f4*>(smem + off);
    off += a.scene.nLights * 32;
    f4* ldsStack = reinterpret_cast<f4*>(smem + off);

    SceneView sv = a.scene;
    if (LDS_SCENE) {
        for (int i = threadIdx.x; i < nPad; i += TPT_BLOCK) {
            ldsSph[i] = a.scene.sph4[i];
            ldsInvR[i] = a.scene.invR[i];
        }
        sv.sph4 = ldsSph;
        sv.invR = ldsInvR;
    }
    for (int i = threadIdx.x; i < a.scene.nLights * 2; i += TPT_BLOCK) ldsLights[i] = a.scene.lights[i];
    sv.lights = ldsLights;
    __syncthreads();

    BounceStack stack;
    stack.base = ldsStack + threadIdx.x;
    stack.stride = TPT_BLOCK;

    const FrameConsts& fc = a.fc;
    const bool rowSerial = fc.seedMode == SEED_ROW_SERIAL;
#if defined(TPT_STATS)
    const unsigned long long statT0 = wall_clock64(); // 100 MHz
#endif
    Lane L;
    L.active = false;
    L.rays = 0;

    if (!PERSIST) {
        // ---- static mapping: one work item per thread
        int idx = blockIdx.x * TPT_BLOCK + threadIdx.x;
        int x, ly;
        if (idx < a.numItems && mapItem(a, idx, x, ly)) {
            laneBeginPixel(L, fc, x, localRowToGlobal(a, ly), ly * fc.width + x, true);
        }
        while (L.active) {
            if (laneStep<HS, FOLD>(L, sv, fc, stack)) {
                storeColour(a, L);
                if (rowSerial && L.x + 1 < fc.width) {
                    laneBeginPixel(L, fc, L.x + 1, L.y, L.pix + 1, false);
                        } else {
                    L.active = false;
                }
            }
        }
    } else {
        // ---- persistent waves with per-lane refill
        const int lane = threadIdx.x & 63;
        const unsigned long long laneBelow = (1ull << lane) - 1ull;
        int chunkNext = 0, chunkEnd = 0;
        bool noMoreWork = false;
        for (;;) {
            bool need = !L.active;
            for (;;) {
                unsigned long long needMask = __ballot(need);
                if (needMask == 0ull) break;
                TPT_STAT(ST_REFILL);
                if (chunkNext >= chunkEnd) {
                    if (noMoreWork) break;
                    TPT_STAT(ST_CHUNK);
                    int c = 0;
                    if (lane == 0) c = (int)atomicAdd(&a.work[0], 1u);
                    c = __builtin_amdgcn_readfirstlane(c);
                    if (c >= a.numChunks) {
                        noMoreWork = true;
                        break;
                    }
                    chunkNext = c * a.chunkSize;
                    chunkEnd = chunkNext + a.chunkSize;
                    if (chunkEnd > a.numItems) chunkEnd = a.numItems;
                }
                int rank = __popcll(needMask & laneBelow);
                int want = __popcll(needMask);
                int avail = chunkEnd - chunkNext;
                int take = want < avail ? want : avail;
                if (need && rank < take) {
                    int x, ly;
                    if (mapItem(a, chunkNext + rank, x, ly)) {
                        laneBeginPixel(L, fc, x, localRowToGlobal(a, ly), ly * fc.width + x, true);
                                    need = false;
                    }
                }
                chunkNext += take;
            }
            if (__ballot(L.active) == 0ull) break;
            if (L.active) {
                if (laneStep<HS, FOLD>(L, sv, fc, stack)) {
                    storeColour(a, L);
                    if (rowSerial && L.x + 1 < fc.width) {
                        laneBeginPixel(L, fc, L.x + 1, L.y, L.pix + 1, false);
                                } else {
                        L.active = false;
                    }
                }
            }
        }
    }

    // ---- ray counter: one atomic per wave (Test.cpp:299 does one per task)
    unsigned waveRays = waveReduceAdd(L.rays);
#if defined(TPT_STATS)
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long statT1 = wall_clock64();
        atomicAdd(&g_tptStats[24], statT1 - statT0);  // sum of wave lifetimes (10 ns ticks)
        atomicMin(&g_tptStats[25], statT0);           // first wave start
        atomicMax(&g_tptStats[26], statT1);           // last wave end
        atomicAdd(&g_tptStats[27], 1ull);             // waves
        atomicMax(&g_tptStats[28], statT1 - statT0);  // longest wave
    }
#endif
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(a.rayCounter, (unsigned long long)waveRays);
        if (PERSIST) {
            // last wave to finish re-arms the work counter for the next launch on this stream
            unsigned done = atomicAdd(&a.work[1], 1u) + 1u;
            if (done == a.totalWaves) {
                a.work[0] = 0u;
                a.work[1] = 0u;
            }
        }
    }
}

// ---------------------------------------------------------------- unit-test kernels (GPU parity of the math layer)
// op: 0 sqrt(a) 1 a/b 2 tsinf(a) 3 tcosf(a) 4 tpow5f(a) 5 rnd01 stream (a = seed bits) 6 schlick(a,b) 7 1/sqrt-normalize.x
__global__ void tptMathTestKernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f;
    switch (op) {
    case 0: r = tsqrt(x); break;
    case 1: r = x / y; break;
    case 2: r = tsinf(x); break;
    case 3: r = tcosf(x); break;
    case 4: r = tpow5f(x); break;
    case 5: {
        uint32_t s = f2u(x) | 1u;
        for (int k = 0; k < 16; ++k) r = rnd01(s);
        break;
    }
    case 6: r = schlick(x, y); break;
    case 7: r = normalize(mk3(x, y, 1.0f)).x; break;
    }
    out[i] = r;
}

// rays: [n][6] orig,dir -> outId[n], outT[n]
template <int HS>
__global__ void tptHitTestKernel(const KernelArgs a, const float* __restrict__ rays, int* __restrict__ outId,
                                 float* __restrict__ outT, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 o = mk3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), d = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    float t;
    int id = hitSpheres<HS>(a.scene, o, d, TPT_MIN_T, TPT_MAX_T, t);
    outId[i] = id;
    outT[i] = t;
}

} // namespace tpt

// ---------------------------------------------------------------- launch glue (called from tpt_host.cpp)
using namespace tpt;

int tptReadStats(unsigned long long* out64)
{
#if defined(TPT_STATS)
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_tptStats), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -2;
#else
    (void)out64;
    return -1;
#endif
}
int tptResetStats()
{
#if defined(TPT_STATS)
    unsigned long long z[64] = {0};
    z[25] = ~0ull;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tptStats), z, sizeof(z)) == hipSuccess ? 0 : -2;
#else
    return -1;
#endif
}

size_t tptLdsBytes(const KernelArgs& a, int fold, bool ldsScene)
{
    const int nPad = a.scene.nPairs * 2;
    size_t bytes = 0;
    if (ldsScene) bytes += (size_t)nPad * 16 + (((size_t)nPad * 4 + 15) & ~(size_t)15);
    bytes += (size_t)a.scene.nLights * 32;
    if (fold == FOLD_RECURSIVE) bytes += (size_t)TPT_MAX_DEPTH * TPT_BLOCK * 16;
    return bytes;
}

template <int HS, int FOLD, bool PERSIST, bool LDS_SCENE>
static hipError_t launchOne(const KernelArgs& a, int blocks, size_t lds, hipStream_t stream)
{
    auto k = tptTraceKernel<HS, FOLD, PERSIST, LDS_SCENE>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(blocks), dim3(TPT_BLOCK), lds, stream, a);
    return hipGetLastError();
}

template <int HS, int FOLD, bool PERSIST, bool LDS_SCENE>
static int occupancyOne(size_t lds)
{
    int nb = 0;
    auto k = tptTraceKernel<HS, FOLD, PERSIST, LDS_SCENE>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), TPT_BLOCK, lds) != hipSuccess) nb = 1;
    return nb < 1 ? 1 : nb;
}

#define TPT_DISPATCH(FN, ...)                                                                                      \
    do {                                                                                                           \
        const int key = (hs ? 8 : 0) | (fold ? 4 : 0) | (persist ? 2 : 0) | (ldsScene ? 1 : 0);                    \
        switch (key) {                                                                                             \
        case 0: return FN<HS_TWO_PHASE, FOLD_RECURSIVE, false, false>(__VA_ARGS__);                                \
        case 1: return FN<HS_TWO_PHASE, FOLD_RECURSIVE, false, true>(__VA_ARGS__);                                 \
        case 2: return FN<HS_TWO_PHASE, FOLD_RECURSIVE, true, false>(__VA_ARGS__);                                 \
        case 3: return FN<HS_TWO_PHASE, FOLD_RECURSIVE, true, true>(__VA_ARGS__);                                  \
        case 4: return FN<HS_TWO_PHASE, FOLD_FORWARD, false, false>(__VA_ARGS__);                                  \
        case 5: return FN<HS_TWO_PHASE, FOLD_FORWARD, false, true>(__VA_ARGS__);                                   \
        case 6: return FN<HS_TWO_PHASE, FOLD_FORWARD, true, false>(__VA_ARGS__);                                   \
        case 7: return FN<HS_TWO_PHASE, FOLD_FORWARD, true, true>(__VA_ARGS__);                                    \
        case 8: return FN<HS_SIMPLE, FOLD_RECURSIVE, false, false>(__VA_ARGS__);                                   \
        case 9: return FN<HS_SIMPLE, FOLD_RECURSIVE, false, true>(__VA_ARGS__);                                    \
        case 10: return FN<HS_SIMPLE, FOLD_RECURSIVE, true, false>(__VA_ARGS__);                                   \
        case 11: return FN<HS_SIMPLE, FOLD_RECURSIVE, true, true>(__VA_ARGS__);                                    \
        case 12: return FN<HS_SIMPLE, FOLD_FORWARD, false, false>(__VA_ARGS__);                                    \
        case 13: return FN<HS_SIMPLE, FOLD_FORWARD, false, true>(__VA_ARGS__);                                     \
        case 14: return FN<HS_SIMPLE, FOLD_FORWARD, true, false>(__VA_ARGS__);                                     \
        default: return FN<HS_SIMPLE, FOLD_FORWARD, true, true>(__VA_ARGS__);                                      \
        }                                                                                                          \
    } while (0)

hipError_t tptLaunchTrace(const KernelArgs& a, int hs, int fold, bool persist, bool ldsScene, int blocks, size_t lds, hipStream_t stream)
{
    TPT_DISPATCH(launchOne, a, blocks, lds, stream);
}
int tptTraceOccupancy(int hs, int fold, bool persist, bool ldsScene, size_t lds)
{
    TPT_DISPATCH(occupancyOne, lds);
}

hipError_t tptLaunchResolve(float* tile, const f4* frameColour, int nPixels, float lerpFac, hipStream_t stream)
{
    hipLaunchKernelGGL(tptResolveKernel, dim3((nPixels + 255) / 256), dim3(256), 0, stream, tile, frameColour, nPixels, lerpFac);
    return hipGetLastError();
}

hipError_t tptLaunchMathTest(int op, const float* a, const float* b, float* out, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(tptMathTestKernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, a, b, out, n);
    return hipGetLastError();
}
hipError_t tptLaunchHitTest(const KernelArgs& a, int hs, const float* rays, int* outId, float* outT, int n, hipStream_t stream)
{
    if (hs == HS_SIMPLE)
        hipLaunchKernelGGL(tptHitTestKernel<HS_SIMPLE>, dim3((n + 255) / 256), dim3(256), 0, stream, a, rays, outId, outT, n);
    else
        hipLaunchKernelGGL(tptHitTestKernel<HS_TWO_PHASE>, dim3((n + 255) / 256), dim3(256), 0, stream, a, rays, outId, outT, n);
    return hipGetLastError();
}
