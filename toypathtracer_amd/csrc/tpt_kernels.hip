// tpt_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the path tracer.
//
// They replace the reference's DrawTest fan-out (Cpp/Source/Test.cpp:344-367: enkiTS task set over rows ->
// TraceRowJob :266-300).  All of them run the same per-lane code (tpt_trace.h: hitSpheres + lanePost, one ray per
// step) and produce the same bits; they differ in how rays are mapped onto lanes:
//
//   * tptTraceQueueKernel   (default)  workgroups of 8 waves own 952 paths whose state lives in LDS; waves pop
//                                      batches of paths that need the SAME code from per-class rings, so Scatter and the
//                                      intersections run at (nearly) full lane utilisation;
//   * tptTraceKernel                   one-wave workgroups pull 8x8-pixel chunks from a global counter and re-fill idle
//                                      lanes with the next pixel (fallback for row-serial seeds / forward fold / simple
//                                      HitSpheres; cost-ordered chunks).
//
//   * work item  = one pixel (PER_PIXEL seed mode) or one row (ROW_SERIAL, the reference's Test.cpp:280 RNG stream;
//     bit-for-bit reproduction of the CPU image);
//   * scene: phase-1 sphere pairs are read with scalar loads (wave-uniform); the {centre, r^2} records, 1/r, the light
//     list and the materials are staged into LDS once per workgroup when they fit (large scenes: global memory + the
//     grouped traversal of tpt_trace.h);
//   * output: the frame's colour per pixel (one 16-B store), blended into the float4 accumulation tile by
//     tptResolveKernel (RGB read-modify-write, alpha untouched); ray counts reduced per wave -> one 64-bit atomic.
//
// Matrix cores: HitSpheres itself has no dense contraction (a 46-long select / min reduction per lane, Maths.cpp:165-202), but
// its conservative FILTER does -- the discriminant is bilinear in (sphere, ray), so the path-queue kernel evaluates
// [64 spheres x 32 slots] x [32 slots x 64 rays] with v_mfma_f32_32x32x16_f16 on f16-split operands (tpt_trace.h phase1MatrixH)
// and runs the reference's exact arithmetic only on what passes.  Everything a branch can depend on stays on the VALU.
#include "tpt_device.h"
#include "tpt_shard.h"

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
    const int tilesX = uniformHere(a.tilesX);
    int tx = tile % tilesX, ty = tile / tilesX;
    x = tx * 8 + (within & 7);
    ly = ty * 8 + (within >> 3);
    return x < a.fc.width && ly < a.nLocalRows;
}
__device__ __forceinline__ int localRowToGlobal(const KernelArgs& a, int ly) { return shardKernelLocalToGlobal(ly, uniformHere(a.stripeRows), uniformHere(a.stripeStride), a.stripeOffset); }
__device__ __forceinline__ int globalRowToLocal(const KernelArgs& a, int gy) { return shardKernelGlobalToLocal(gy, uniformHere(a.stripeRows), uniformHere(a.stripeStride), a.stripeOffset); } // rows of this rank

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
// frameRays (host-pointer path): the frame's own ray count, which this kernel folds into the context's running total.
__global__ void __launch_bounds__(256) tptResolveKernel(float* __restrict__ tile, const f4* __restrict__ colour, int nPixels, float lerpFac,
                                                        const unsigned long long* frameRays, unsigned long long* totalRays)
{
    // These few waves land on CUs saturated with persistent trace waves, and VALU issue is arbitrated by priority, then
    // AGE: the newcomer gets the leftover slots (a 6-us kernel took 40-1000 us; the ordered resolve chain is what bounds
    // small frames).  Raise the wave's priority for its short life.
    __builtin_amdgcn_s_setprio(3);
    if (blockIdx.x == 0 && threadIdx.x == 0 && frameRays) atomicAdd(totalRays, *frameRays);
    // grid-stride: a few hundred workgroups however large the tile is (tptLaunchResolve) -- every workgroup is one more
    // dispatch that has to find a slot on a machine full of persistent trace workgroups
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nPixels; i += gridDim.x * 256) {
        f4 t = reinterpret_cast<const f4*>(tile)[i];
        f4 c = colour[i];
        f3 r = blendPixel(mk3(t.x, t.y, t.z), mk3(c.x, c.y, c.z), lerpFac);
        t.x = r.x; t.y = r.y; t.z = r.z;
        reinterpret_cast<f4*>(tile)[i] = t;
    }
}
// Same, and the blended pixel also goes to `mirror` (the snapshot a sharded host hands to its collective while the
// next frames keep accumulating into the tile) and the current value of the ray counter to `counterOut`: one kernel
// in the frame's dependency chain instead of three.
__global__ void __launch_bounds__(256) tptResolveMirrorKernel(float* __restrict__ tile, const f4* __restrict__ colour, int nPixels, float lerpFac,
                                                              f4* __restrict__ mirror, unsigned long long* rayCounter,
                                                              unsigned long long* counterOut, const unsigned long long* frameRays)
{
    __builtin_amdgcn_s_setprio(3); // see tptResolveKernel
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // frameRays: the frame's own ray count (it was traced as part of a batch), folded into the running total here, in
        // frame order -- the snapshot of the counter that travels with the mirrored tile then is exact for "frames up to this one"
        unsigned long long total = frameRays ? atomicAdd(rayCounter, *frameRays) + *frameRays
                                             : __hip_atomic_load(rayCounter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (counterOut) *counterOut = total;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nPixels; i += gridDim.x * 256) {
        f4 t = reinterpret_cast<const f4*>(tile)[i];
        f4 c = colour[i];
        f3 r = blendPixel(mk3(t.x, t.y, t.z), mk3(c.x, c.y, c.z), lerpFac);
        t.x = r.x; t.y = r.y; t.z = r.z;
        reinterpret_cast<f4*>(tile)[i] = t;
        mirror[i] = t;
    }
}

// The blends of a batch of frames (tptDrawDeviceBatch), applied in frame order to each pixel by one launch: exactly the
// arithmetic of nFrames tptResolveKernel launches (same blendPixel, same order), 1 / nFrames of the launches.
__global__ void __launch_bounds__(256) tptResolveBatchKernel(float* __restrict__ tile, const f4* __restrict__ colour, int nPixels, int planeStride,
                                                             int nFrames, tptLerpTable lerp, f4* __restrict__ mirror,
                                                             const unsigned long long* rayCounter, unsigned long long* counterOut)
{
    __builtin_amdgcn_s_setprio(3); // see tptResolveKernel
    if (blockIdx.x == 0 && threadIdx.x == 0 && counterOut) *counterOut = __hip_atomic_load(rayCounter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nPixels; i += gridDim.x * 256) {
        f4 t = reinterpret_cast<const f4*>(tile)[i];
        f3 r = mk3(t.x, t.y, t.z);
        for (int j = 0; j < nFrames; ++j) {
            const f4 c = colour[(size_t)j * planeStride + i];
            r = blendPixel(r, mk3(c.x, c.y, c.z), lerp.v[j]);
        }
        t.x = r.x; t.y = r.y; t.z = r.z;
        reinterpret_cast<f4*>(tile)[i] = t;
        if (mirror) mirror[i] = t;
    }
}

// Rank 0 of a sharded frame: the gathered tiles [rank][padRows + 1][width] f4 (row stripes dealt round-robin, one extra row
// per rank whose first 8 bytes carry that rank's ray counter) -> the image [height][width] f4.  HBM-bound copy, one f4 per
// lane, coalesced on both sides; replaces the per-row joins of DrawTest's task set (Test.cpp:357-361).
__global__ void __launch_bounds__(256) tptAssembleKernel(const f4* __restrict__ gathered, f4* __restrict__ image, int width, int height, int stripeRows,
                                                         int nRanks, int padRows)
{
    __builtin_amdgcn_s_setprio(3);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= width * height) return;
    const int gy = i / width, x = i - gy * width;
    image[i] = gathered[shardGatheredPixel(x, gy, width, stripeRows, nRanks, padRows)]; // (tpt_shard.h)
}

// One wave that spins for `ticks` of the 100 MHz wall clock: tptInitialize launches one per trace stream to measure how
// many of them the runtime really runs side by side (hardware queues granted to this process).
__global__ void tptQueueProbeKernel(unsigned long long ticks, unsigned* sink)
{
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks) ++n;
    if (sink && n == 0xffffffffu) *sink = n;
}

#ifndef TPT_MIN_WAVES_PER_SIMD
#define TPT_MIN_WAVES_PER_SIMD 4 // caps the allocation at 128 VGPRs: 16 waves per CU
#endif
// Work-distribution helper: order[] = chunk indices sorted by accumulated cost, expensive first (counting sort into
// 256 cost buckets; order inside a bucket is arbitrary -- it only affects scheduling, never results).  Long pixels
// then start early and the cheap sky tiles fill the end of the launch: -11 % wave-steps at configs[1]
// (tools/sim_sched.py).  One workgroup; ~10 us for 14 400 chunks.
__global__ void __launch_bounds__(1024) tptChunkOrderKernel(const unsigned* __restrict__ cost, unsigned* __restrict__ snap,
                                                            unsigned* __restrict__ order, int n)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned maxCost;
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0u;
    if (tid == 0) maxCost = 1u;
    __syncthreads();
    // trace kernels of other frames may be adding to cost[] right now: read every element ONCE into a snapshot so that
    // the histogram and the scatter below see the same buckets (otherwise the table would not be a permutation)
    unsigned m = 0u;
    for (int i = tid; i < n; i += 1024) {
        const unsigned c = __hip_atomic_load(&cost[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        snap[i] = c;
        m = c > m ? c : m;
    }
    atomicMax(&maxCost, m);
    __syncthreads();
    const float scale = 255.0f / (float)maxCost;
    for (int i = tid; i < n; i += 1024) { // each thread re-reads only what it wrote itself
        unsigned b = 255u - (unsigned)((float)snap[i] * scale); // bucket 0 = most expensive
        atomicAdd(&hist[b > 255u ? 255u : b], 1u);
    }
    __syncthreads();
    if (tid == 0) { // exclusive prefix sum over 256 buckets
        unsigned run = 0u;
        for (int b = 0; b < 256; ++b) {
            unsigned c = hist[b];
            hist[b] = run;
            run += c;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        unsigned b = 255u - (unsigned)((float)snap[i] * scale);
        unsigned pos = atomicAdd(&hist[b > 255u ? 255u : b], 1u);
        order[pos] = (unsigned)i;
    }
}

// Display conversion of the linear float accumulation buffer, the way the reference's C++ path shows it in the
// browser (Cpp/Emscripten/main.cpp:63-79): rows flipped (row 0 of the buffer is the bottom of the image), cheap sRGB
// approximation c8 = (uint8) min(sqrtf(c) * 255, 255), alpha 255.  HBM-bound: 16 B read + 4 B written per pixel,
// one uchar4 per lane, fully coalesced.  Negative / NaN inputs (which the reference would convert with UB) give 0.
__global__ void __launch_bounds__(256) tptDisplayKernel(const f4* __restrict__ tile, uint32_t* __restrict__ rgba, int width, int height)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= width * height) return;
    const int y = i / width, x = i - y * width;
    const f4 c = tile[(height - 1 - y) * width + x];
    auto to8 = [](float v) -> uint32_t {
        float s = tsqrt(v) * 255.0f;
        s = s < 255.0f ? s : 255.0f; // std::min(s, 255.0f)
        return s > 0.0f ? (uint32_t)s : 0u;
    };
    rgba[i] = to8(c.x) | (to8(c.y) << 8) | (to8(c.z) << 16) | 0xff000000u;
}

template <int HS, int FOLD, bool LDS_SCENE>
// 112 VGPRs x 4 waves/SIMD leaves 64 registers per SIMD lane for the resolve kernel's waves (see tptTraceQueueKernel;
// amdgpu_num_vgpr counts half of the unified file on gfx90a+, so 56 means 112)
__global__ void __launch_bounds__(TPT_BLOCK, TPT_MIN_WAVES_PER_SIMD) __attribute__((amdgpu_num_vgpr(56)))
tptTraceKernel(const KernelArgs a)
{
    constexpr int HSX = (HS == HS_TWO_PHASE && !LDS_SCENE) ? HS_TWO_PHASE_GROUPS : HS; // grouped scenes are never LDS-staged
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // ---- carve LDS (every offset a multiple of 16)
    const int nPad = a.scene.nPairs * 2;
    f4* ldsSph = reinterpret_cast<f4*>(smem);
    int off = LDS_SCENE ? nPad * 16 : 0;
    float* ldsInvR = reinterpret_cast<float*>(smem + off);
    off += LDS_SCENE ? ((nPad * 4 + 15) & ~15) : 0;
    f4* ldsLights = reinterpret_cast<f4*>(smem + off);
    off += a.scene.nLights * 32;
    f4* ldsMats = reinterpret_cast<f4*>(smem + off);
    off += LDS_SCENE ? a.scene.nSpheres * 48 : 0;
    f4* ldsStack = reinterpret_cast<f4*>(smem + off);

    SceneView sv = a.scene;
    if (LDS_SCENE) {
        for (int i = threadIdx.x; i < nPad; i += TPT_BLOCK) {
            ldsSph[i] = a.scene.sph4[i];
            ldsInvR[i] = a.scene.invR[i];
        }
        for (int i = threadIdx.x; i < a.scene.nSpheres * 3; i += TPT_BLOCK) ldsMats[i] = a.scene.mats[i];
        sv.sph4 = ldsSph;
        sv.invR = ldsInvR;
        sv.mats = ldsMats; // materials are read on every hit and every fold level: LDS latency instead of L2
    }
    for (int i = threadIdx.x; i < a.scene.nLights * 2; i += TPT_BLOCK) ldsLights[i] = a.scene.lights[i];
    sv.lights = ldsLights;
    __syncthreads();

    BounceStack stack;
    stack.base = ldsStack + threadIdx.x;
    stack.stride = TPT_BLOCK;
    stack.fastLevels = a.ldsStackLevels;
    stack.spill = a.stackBuf + (blockIdx.x * TPT_BLOCK + threadIdx.x);
    stack.spillStride = a.stackStride;

    const FrameConsts& fc = a.fc;
    const bool rowSerial = fc.seedMode == SEED_ROW_SERIAL;
#if defined(TPT_STATS)
    const unsigned long long statT0 = wall_clock64(); // 100 MHz
#endif
    Lane L;
    L.active = false;
    L.rays = 0;

    {
        // ---- persistent waves with per-lane refill
        const int lane = threadIdx.x & 63;
        const unsigned long long laneBelow = (1ull << lane) - 1ull;
        int chunkNext = 0, chunkEnd = 0, chunkFrame = 0;
        bool noMoreWork = false;
        for (;;) {
            bool need = !L.active && lane < a.laneCap; // (laneCap < 64: few, long work items -- spread them over more waves)
            // one thread per pixel (persistent 0, the north_star's shape): nothing is handed out while any lane of the wave still works
            // on the pixel it was given -- a lane whose path ended after one ray waits for its 11-bounce neighbour
            if (a.noRefill && __ballot(L.active) != 0ull) need = false;
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
                    if (a.chunkOrder) c = (int)a.chunkOrder[c]; // expensive chunks first (statistics of previous frames)
                    chunkFrame = 0;
                    if (a.batchFrames > 1) { // batched launch (row-serial seeds): chunk c belongs to frame c / chunksPerFrame of the batch
                        chunkFrame = c / a.chunksPerFrame;
                        c -= chunkFrame * a.chunksPerFrame;
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
                        laneBeginPixel(L, fc, x, localRowToGlobal(a, ly), chunkFrame * a.framePlane + ly * fc.width + x, true);
                        if (a.batchFrames > 1) L.rng = pixelSeed(fc.seedMode, L.x, L.y, fc.frame + chunkFrame);
                        L.frameIdx = chunkFrame;
                        L.item = chunkNext + rank;
                        L.rays0 = L.rays;
                        need = false;
                    }
                }
                chunkNext += take;
            }
            if (__ballot(L.active) == 0ull) break;
            if (L.active) {
                if (laneStep<HSX, FOLD>(L, sv, fc, stack)) {
                    storeColour(a, L);
                    if (rowSerial && L.x + 1 < fc.width) {
                        laneBeginPixel(L, fc, L.x + 1, L.y, L.pix + 1, false);
                    } else {
                        // statistics for the cost-ordered work distribution: only the long pixels matter for the order
                        // (and 90 % fewer atomics than counting every pixel)
                        if (a.chunkCost && L.rays - L.rays0 > (uint32_t)(8 * fc.spp))
                            atomicAdd(&a.chunkCost[L.item >> a.chunkShift], L.rays - L.rays0);
                        if (rowSerial && a.batchFrames > 1) {
                            // batched row-serial launch: the row's rays go to its frame's own counter (a host that is served
                            // the frames one by one returns each frame's count, Test.cpp:366) -- one atomic per row
                            atomicAdd(a.rayCounter + (size_t)L.frameIdx * a.rayCounterStride, (unsigned long long)(L.rays - L.rays0));
                            L.rays = L.rays0; // (not again in the wave's total below)
                        }
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
        {
            // last wave to finish re-arms the work counter for the next launch on this stream
            unsigned done = atomicAdd(&a.work[1], 1u) + 1u;
            if (done == a.totalWaves) {
                a.work[0] = 0u;
                a.work[1] = 0u;
            }
        }
    }
}

__device__ __forceinline__ f4 mk4(float x, float y, float z, float w)
{
    f4 v;
    v.x = x; v.y = y; v.z = z; v.w = w;
    return v;
}

// ---------------------------------------------------------------- path-queue variant
// A workgroup of 8 waves owns a pool of TPT_Q_PATHS = 952 paths (~2 per lane, so the queues below stay deep enough to hand
// out full batches; the rings hold TPT_Q_P = 1024 entries); two such workgroups fit on a CU.  Waves are interchangeable workers that pop batches of up to 64
// path ids from per-operation queues in LDS:
//   FREE    -> [assign a pixel, camera ray]                       \
//   END     -> [sky / emission, fold, next sample's camera ray]    |  then, in the SAME iteration, HitWorld for the
//   DIEL    -> [Scatter: dielectric]                               |  rays the batch produced, classify the hits and
//   METAL   -> [Scatter: metal]                                    |  push every path to the queue of its next class
//   LAMBERT -> [Scatter: lambert + the whole light loop: every    /   (or FREE when its pixel is done)
//               shadow ray is intersected and shaded in place]
//   INT     -> overflow only: rays of batches that kept fewer than TPT_Q_FUSE_MIN lanes are re-batched here first
// A batch holds paths that all need the same code, so Scatter and the intersections that follow run at (nearly)
// full lane utilisation instead of ~40 %, a lane never idles because "its" pixel ended (any wave picks up any path),
// and a path crosses a queue once per bounce (not once per ray: the shadow rays of a Lambert hit stay in registers).
// No barriers: the queues are multi-producer / multi-consumer rings (reserve with an LDS atomic, publish by
// overwriting a 0xFFFF sentinel).  Path state: a 64-B record in LDS (ray, rng, flags, hit; the pixel's colour sum; level 0 of
// the bounce stack -- see "The path record" below); levels 1-9 of the stack live in global memory (L2-resident).
// FOLD_RECURSIVE only.
#ifndef TPT_Q_WAVES
#define TPT_Q_WAVES 8
#endif
#define TPT_Q_T (64 * TPT_Q_WAVES)
#ifndef TPT_Q_P
#define TPT_Q_P 1024 // capacity of every ring (power of two)
#endif
#ifndef TPT_MATRIX_FILTER
#define TPT_MATRIX_FILTER 1 // phase 1 of HitSpheres on the matrix cores (v_mfma_f32_32x32x16_f16, f16-split operands) for scenes with a table; 0: packed VALU filter only
#endif
#ifndef TPT_GROUP_DEAL
// Grouped traversal of large scenes (the kernel instantiated without LDS scene staging): 1 = the (ray, group) pairs of a wave
// are dealt out evenly over its lanes through a pair list in LDS (hitSpheresGroupedDeal); 0 = every lane walks the groups its
// own ray touches (tpt_trace.h hitSpheresGrouped: 10.5 trips per wave at 20 busy lanes for 4.0 groups per ray on the
// 4096-sphere scene, tools/stats_c5.py, profiles/r04/r04_run4.log)
#define TPT_GROUP_DEAL 1
#endif
#ifndef TPT_GROUP_MATRIX_BOUNDS
// The groups' bounds on the matrix cores (hitSpheres variant 4) are compiled into the HOOKS build only: a wave that has executed that
// path is not safe in a time-sliced process, it is no faster than the two-level VALU filter any more, and without it
// the product's grouped instantiation executes no MFMA at all -- and needs fewer registers.
#if defined(TPT_TEST_HOOKS)
#define TPT_GROUP_MATRIX_BOUNDS 1
#else
#define TPT_GROUP_MATRIX_BOUNDS 0
#endif
#endif
#ifndef TPT_GROUP_DEAL_EXACT
#define TPT_GROUP_DEAL_EXACT 1 // the members that pass the member filter are dealt out again for their exact tests (see hitSpheresGroupedDeal)
#endif
#ifndef TPT_DEAL_HALF_LINE
#define TPT_DEAL_HALF_LINE 1 // the three-stage dealing drops bounds that lie wholly behind the ray's origin (tpt_trace.h phase1PairT<true>); 0: line test only
#endif
#ifndef TPT_MEMBER_UNROLL
#define TPT_MEMBER_UNROLL 8 // member records requested together in the member filter of a (ray, group) pair: all eight (a latency-bound gather from L2; 4: -3 %, profiles/r06/r06_run14.log)
#endif
// (_Pragma with a stringised macro: `#pragma unroll MACRO` is not expanded when the source is preprocessed separately --
//  -save-temps, ccache, distcc -- and the build broke there)
#define TPT_PRAGMA_STR(x) _Pragma(#x)
#define TPT_PRAGMA_UNROLL(n) TPT_PRAGMA_STR(unroll n)
#ifndef TPT_GROUP_DEAL_CAP
// pair-list entries per wave and round of the FLAT variants (hitSpheres 3 / 4: one (path, group) list filled by the owners, a multiple
// of 64).  The sweeps that chose 448 (128 ... 640 entries against the path pool they leave: profiles/r06/r06_run14-16.log, r06_run26.log)
// were made while the default traversal used this list too; it deals in three stages now (TPT_DEAL_CA / CB / CS below).
#define TPT_GROUP_DEAL_CAP 448
#endif
// The three-stage dealing (dealThreeStage) cuts the wave's list area into (path, super-group) entries of a round, the stack of (path,
// group) entries waiting for a member pass and the stack of survivors waiting for an exact pass; the flat / matrix-core variants use the
// first TPT_GROUP_DEAL_CAP entries as one pair list.  Four counters behind the entries.
#ifndef TPT_DEAL_CA
#define TPT_DEAL_CA (TPT_SUPER == 8 ? 256 : 192)
#endif
#ifndef TPT_DEAL_CB
#define TPT_DEAL_CB (TPT_SUPER == 8 ? 256 : 320) // (a sub-round of 64 super-group entries leaves ~90 group entries on average -- more with super-groups of 16 --, 64 x TPT_SUPER at most; fewer than 64 wait when it starts)
#endif
#ifndef TPT_DEAL_CS
#define TPT_DEAL_CS 128 // (a member pass leaves 17 survivors on average, 512 at most; fewer than 64 wait when it starts)
#endif
// The hooks build can SHRINK the three areas at run time (tptTestSetDealCapacities): the suite renders a grouped scene with 64-entry
// areas, where super-group entries spill into further rounds and group entries / survivors that find their stack full are served in
// place -- paths a frame at the shipped sizes almost never takes.  The product build uses the constants.
#if defined(TPT_TEST_HOOKS)
__device__ unsigned g_dealCaps[3] = {TPT_DEAL_CA, TPT_DEAL_CB, TPT_DEAL_CS};
#define TPT_DEAL_CA_RT (g_dealCaps[0])
#define TPT_DEAL_CB_RT (g_dealCaps[1])
#define TPT_DEAL_CS_RT (g_dealCaps[2])
#else
#define TPT_DEAL_CA_RT ((unsigned)TPT_DEAL_CA)
#define TPT_DEAL_CB_RT ((unsigned)TPT_DEAL_CB)
#define TPT_DEAL_CS_RT ((unsigned)TPT_DEAL_CS)
#endif
#define TPT_GROUP_DEAL_ENTRIES (TPT_DEAL_CA + TPT_DEAL_CB + TPT_DEAL_CS)
static_assert(TPT_GROUP_DEAL_ENTRIES >= TPT_GROUP_DEAL_CAP, "the flat variants' pair list lives in the same area");
#define TPT_GROUP_DEAL_WAVE_BYTES (TPT_GROUP_DEAL_ENTRIES * 4 + 16)
#define TPT_Q_SPH_FIXED 1024 /* bytes at LDS offset 0 for {centre, r^2} of scenes of <= 64 spheres: DS offsets fold into the instructions */
#ifndef TPT_Q_PATHS
// paths per workgroup (<= TPT_Q_P): what the path records in LDS are sized for.  960 with the matrix filter: its 4-KB operand
// table and the fixed 1-KB sphere area have to fit beside them for two workgroups per CU (2 x 80 KB minus the launch code's
// 256-B margin per workgroup: chooseKernel); 960 measured no slower than 1024 (profiles/r03/r03_run10.log)
#define TPT_Q_PATHS (TPT_MATRIX_FILTER ? 952 : TPT_Q_P)
#endif
#ifndef TPT_Q_PATHS_GROUPED
// ... and of the instantiation for GROUPED scenes (no scene staging, no matrix-filter table): 608.  The LDS the smaller pool frees holds
// the entry areas of the three-stage dealing (640 entries per wave) and the groups' bounding spheres (pair records, 144 B per super-group
// of 8 groups, for up to TPT_Q_GROUP_LDS_BYTES: stage B reads them per lane, and from L2 that stage would be latency-bound).  624 ... 752
// paths measured within 1 % of each other (profiles/r06/r06_run26.log).
#define TPT_Q_PATHS_GROUPED 608
#endif
// The groups' pair records in LDS: a super-group's four records (128 B) are read per lane by lanes that hold DIFFERENT super-groups. At a
// stride of 128 B every lane's read of "record q, half h" lands on one of two 16-byte bank groups of the 16 -- an 8-way conflict on
// every read (68 % of the LDS's active cycles were conflict cycles, profiles/r06/r06_run30.log).  Nine bank groups per super-group
// (144 B: 16 B of padding) spread consecutive super-groups over all sixteen.
#define TPT_GPAIR_FLOATS ((TPT_SUPER / 2) * 8) /* floats of a super-group's pair records: 32 (64 for super-groups of 16) */
#define TPT_GPAIR_LDS_STRIDE (TPT_GPAIR_FLOATS + 4) /* floats per super-group in LDS: 9 (17) bank groups of 16 bytes */
#define TPT_Q_GROUP_LDS_BYTES 9808 /* group pair records in LDS at most: 68 super-groups x 144 B + 16 (a launch that would lose its second workgroup per CU to them reads them from global memory instead: chooseKernel) */
#ifndef TPT_Q_FUSE_MIN
#define TPT_Q_FUSE_MIN 48 // a batch intersects its own rays when at least this many lanes still hold one
#endif
#define TPT_Q_NF4 4
enum { Q_FREE = 0, Q_INT = 1, Q_END = 2, Q_DIEL = 3, Q_METAL = 4, Q_LAMBERT = 5, Q_COUNT = 6 };
struct QueueCtl {
    unsigned head[8];
    unsigned tail[8];
    unsigned poolTotal;       // pixels sitting in the private chunk pools of this workgroup's waves (+ fetches in flight)
    unsigned globalExhausted; // some wave saw the global chunk counter run out
    unsigned frameRays[32];   // batched launch: rays traced for each frame of the batch by this workgroup (flushed to the global counter every 2^31: see the push)
};

// The rings (and the pair lists of the grouped traversal) are polled with volatile accesses, and LLVM's address-space inference
// leaves volatile accesses alone: through a generic pointer they compile to FLAT loads / stores, which reach LDS through the
// vector-memory path (seen in the ISA of rounds 2-3: flat_load_ushort in the pop and push spins).  Typed as LDS they are ds_ ops.
#if defined(__HIP_DEVICE_COMPILE__)
typedef volatile unsigned short __attribute__((address_space(3))) * LdsRing;
typedef volatile unsigned __attribute__((address_space(3))) * LdsList;
#else
typedef volatile unsigned short* LdsRing;
typedef volatile unsigned* LdsList;
#endif
// 0, but only once `v` has arrived: orders a second atomic behind the RETURN of a first one without a fence (tpt_device.h)
__device__ __forceinline__ unsigned dependentZero(unsigned v)
{
    unsigned z;
    asm volatile("v_and_b32_e32 %0, 0, %1" : "=v"(z) : "v"(v));
    return z;
}
// Push every lane's path id to the queue of its class `cls` (Q_FREE..Q_LAMBERT, or -1 for none): one returning LDS atomic
// per lane reserves the slot (the LDS unit serialises the lanes that hit the same tail word -- its time, not the VALU's:
// the ballot / popcount / readlane version of this cost ~35 VALU instructions per batch).
__device__ __forceinline__ void qPushByClass(LdsRing q, QueueCtl* ctl, int cls, int pathId, int lane)
{
    (void)lane;
    if (cls >= 0) {
        const unsigned pos = atomicAdd(&ctl->tail[cls], 1u);
        LdsRing slot = q + cls * TPT_Q_P + (pos & (TPT_Q_P - 1));
        // A consumer advances the head BEFORE it reads its slots, and ids can cycle through a ring any number of times
        // while one consumer stalls between those two steps (FREE: pop, no pixel left, push again): the tail may lap a
        // reserved-but-unread slot.  Publish only into a slot whose previous entry has been taken (sentinel restored).
        while (*slot != 0xFFFFu) {
        }
        *slot = (unsigned short)pathId;
    }
}
// Pops up to 64 ids (uniform count returned); lanes < count receive a path id.
// h0 / t0: the head and tail the wave read when it chose this queue -- the first reservation is attempted with them (one LDS
// round trip less per iteration than reading both again first: -DTPT_Q_POP_SNAPSHOT=0); a stale pair just fails the CAS.
#ifndef TPT_Q_POP_SNAPSHOT
#define TPT_Q_POP_SNAPSHOT 1
#endif
__device__ __forceinline__ int qPop(LdsRing q, unsigned* head, unsigned* tail, int lane, int& pathId, unsigned h0, unsigned t0)
{
    unsigned h = 0, n = 0;
    if (lane == 0) {
        bool first = TPT_Q_POP_SNAPSHOT != 0;
        for (;;) {
            unsigned t;
            if (first) {
                h = h0;
                t = t0;
            } else {
                h = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                t = __hip_atomic_load(tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // A (head, tail) pair read by two loads may be torn: a tail older than the head it is paired with -- consumers and
            // producers both moved in between -- makes tail - head wrap to 4 billion, and a CAS that still finds that head would
            // reserve 64 slots nobody will ever fill (every wave of the workgroup ends up spinning on an unpublished slot: seen once
            // the snapshot pair was reused, profiles/r04/r04_run11.log).  A pair is only ever used when tail - head is positive as
            // a SIGNED number: then tail - head <= (tail now) - head, whatever the order of the two loads.
            const int avail = (int)(t - h);
            n = avail <= 0 ? 0u : (avail < 64 ? (unsigned)avail : 64u);
            if (n == 0u) {
                if (first) { // the snapshot is stale or torn: look again before giving up
                    first = false;
                    continue;
                }
                break;
            }
            first = false;
            if (atomicCAS(head, h, h + n) == h) break;
        }
    }
    h = (unsigned)__builtin_amdgcn_readfirstlane((int)h);
    n = (unsigned)__builtin_amdgcn_readfirstlane((int)n);
    pathId = 0;
    if ((unsigned)lane < n) {
        const unsigned slot = (h + (unsigned)lane) & (TPT_Q_P - 1);
        unsigned v;
        do {
            v = q[slot]; // the producer reserved this slot; wait until it has written the id
        } while (v == 0xFFFFu);
        q[slot] = 0xFFFFu;
        pathId = (int)v;
    }
    return (int)n;
}


#if TPT_GROUP_DEAL
// HitWorld over a GROUPED scene for a whole wave (every lane calls it; lanes without a ray pass go = false): the same tests
// as tpt_trace.h's hitSpheresGrouped -- big spheres exactly, the super-groups' and groups' bounding spheres through the packed
// conservative filter, the members of every touched group through the per-member filter, the reference's exact test
// (Maths.cpp:171-190) for what passes, lowest original index among equal t -- but nothing below the first level is done by "the lane
// that owns the ray": (ray, super-group), (ray, group) and (ray, member) pairs are DEALT OUT evenly over the lanes through entry lists
// in LDS (dealThreeStage; on the 4096-sphere scene a ray touches 3.5 super-groups, 4.9 groups and 1.3 members, the busiest lane of a
// wave 8.4, 15.5 and 10.5).  A lane's ray {o, d} and its best hit so far -- the 64-bit key (t bits << 32 | sphere id) -- are parked in
// planes 0 / 1 of its own path record, which are dead while this wave holds the path; whoever finds a hit merges it into the owner's
// key with ds_min_u64: smaller t wins, equal t: the lower ORIGINAL sphere index -- the reference's first-strictly-less rule made
// explicit (t > tMin > 0: the bit patterns order like the values).  The flat variants (hitSpheres 3: one packed filter over all
// groups; 4, hooks build: the groups' bounds on the matrix cores) deal the (ray, group) pairs only: owners reserve list entries with one
// LDS atomic, lane j takes entry j, pairs that do not fit the list (TPT_GROUP_DEAL_CAP per round) stay in their lane's mask.
// One wave, no barrier: a wave's LDS operations execute in order; wave_barrier only pins the compiler.
// the packed filter of phase1Pair for one pair record read PER LANE (from LDS): two more sign bits shifted into m
// (HALF: the half-line form of tpt_trace.h's phase1PairT -- bounds whose centre is behind the ray's origin, the origin outside them)
template <bool HALF>
__device__ __forceinline__ void phase1PairLaneT(const float* rec, v2f ox, v2f oy, v2f oz, v2f dx, v2f dy, v2f dz, uint32_t& m)
{
    const f4 r0 = *reinterpret_cast<const f4*>(rec), r1 = *reinterpret_cast<const f4*>(rec + 4);
    const v2f cx = {r0.x, r0.y}, cy = {r0.z, r0.w}, cz = {r1.x, r1.y}, nsq = {r1.z, r1.w};
    const v2f coX = cx - ox, coY = cy - oy, coZ = cz - oz;
    const v2f nb = fma2(coZ, dz, fma2(coY, dy, coX * dx));
    const v2f e = fma2(coZ, coZ, fma2(coY, coY, fma2(coX, coX, nsq)));
    const v2f discr = fma2(nb, nb, -e);
    if (HALF) {
        const v2f hc = {-TPT_HALF_C, -TPT_HALF_C};
        const v2f wn = fma2(nsq, hc, -e);
        m = alignbit(m, (f2u(nb[0]) & f2u(wn[0])) | f2u(discr[0]), 31);
        m = alignbit(m, (f2u(nb[1]) & f2u(wn[1])) | f2u(discr[1]), 31);
    } else {
        m = alignbit(m, f2u(discr[0]), 31);
        m = alignbit(m, f2u(discr[1]), 31);
    }
}
__device__ __forceinline__ void phase1PairLane(const float* rec, v2f ox, v2f oy, v2f oz, v2f dx, v2f dy, v2f dz, uint32_t& m)
{
    phase1PairLaneT<false>(rec, ox, oy, oz, dx, dy, dz, m);
}
// profiling build (-DTPT_STATS=2): s_memtime ticks of the dealing's stages, summed by lane 0 of every wave into g_tptStats[90..99]
// ([90] big spheres [91] super-groups' bounds, wave-wide [92] groups' bounds per lane + list entries [93] member filter (list, parked
// ray, gathers) [94] survivors dealt + exact tests; [95] sub-rounds of 64 pairs [96] pairs [97] rounds [98] survivors [99] calls [100] exact passes [101] wave trips of the per-lane super-group loop [102] of the entry-writing loop [103] of the survivors' push loop)
#if defined(__HIP_DEVICE_COMPILE__) && defined(TPT_STATS) && TPT_STATS >= 2
#define TPT_DEAL_T(v) TPT_HS_STAMP(v)
__shared__ unsigned long long g_dealLds[18]; // per-workgroup sums (LDS atomics: global ones made the build 60 x slower), flushed when the workgroup ends
#define TPT_DEAL_ADD(slot, a, b) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_dealLds[(slot) - 90], (unsigned long long)((b) - (a))); } while (0)
#define TPT_DEAL_COUNT(slot, n) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_dealLds[(slot) - 90], (unsigned long long)(n)); } while (0)
#define TPT_DEAL_TRIP(slot) do { if (TPT_HS_FIRST()) atomicAdd(&g_dealLds[(slot) - 90], 1ull); } while (0) /* inside divergent loops: the first active lane counts the wave's trip */
#else
#define TPT_DEAL_TRIP(slot) do { } while (0)
#define TPT_DEAL_T(v) do { } while (0)
#define TPT_DEAL_ADD(slot, a, b) do { } while (0)
#define TPT_DEAL_COUNT(slot, n) do { } while (0)
#endif
// Exclusive prefix sum of v over the wave's 64 lanes (all of them must be executing), and the total: six DPP additions -- row shifts
// by 1, 2, 4, 8 inside the rows of 16, then the rows' totals broadcast onward -- instead of a returning LDS atomic on one counter, which
// the LDS serialises lane by lane and every wave of the CU waits behind (half of the remaining LDS conflict cycles, r06_run31.log).
__device__ __forceinline__ unsigned wavePrefix(unsigned v, unsigned& total)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    total = (unsigned)__builtin_amdgcn_readlane(x, 63);
    return (unsigned)x - v;
#else
    total = v;
    return 0u;
#endif
}
// One exact pass of the dealing: lane j < n takes survivor entry list[first + j] = (owner's path id << 20 | member slot), reads the
// owner's parked ray and the member, runs the reference's test (Maths.cpp:171-190) and merges a hit into the owner's key.
template <int PATHS>
__device__ __forceinline__ void dealExactPass(const SceneView& sv, LdsList list, unsigned first, unsigned n, f4* st, int lane)
{
    TPT_DEAL_COUNT(100, 1);
    if ((unsigned)lane < n) {
        const unsigned e2 = list[first + (unsigned)lane];
        const int po2 = (int)(e2 >> 20), slot = (int)(e2 & 0xfffffu);
        const f4 q0 = st[po2], q1 = st[PATHS + po2];
        float ht2 = TPT_MAX_T;
        int hid2 = -1;
        TPT_STAT(ST_PHASE2);
        // (both gathers requested together: left to itself the compiler asks for the original index only once the discriminant is known to
        //  be positive -- a second L2 round trip behind the first)
        const f4 sph = sv.gsph[slot];
        int sid = sv.gid[slot];
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(sid));
#endif
        testSphereTie(sph, sid, mk3(q0.z, q0.w, q1.x), mk3(q1.y, q1.z, q1.w), TPT_MIN_T, ht2, hid2);
        if (hid2 >= 0) atomicMin(reinterpret_cast<unsigned long long*>(&st[po2]), ((unsigned long long)f2u(ht2) << 32) | (unsigned long long)(uint32_t)hid2);
    }
}
// One round of the dealing, consumer side: the wave's pair list holds (owner's path id << 16 | group) entries; lane j takes entry j
// (sub-rounds of 64), reads the owner's parked ray, filters the group's members, deals the survivors out once more for their exact
// tests and merges hits into the owners' keys (see hitSpheresGroupedDeal).
template <int PATHS>
__device__ __forceinline__ void dealProcessList(const SceneView& sv, LdsList list, unsigned* listCount, f4* st, int lane)
{
    unsigned total = __hip_atomic_load(listCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    total = total < (unsigned)TPT_GROUP_DEAL_CAP ? total : (unsigned)TPT_GROUP_DEAL_CAP;
    unsigned pend = 0u; // survivors waiting for their exact test in list[0 .. pend) (wave-uniform; listCount[1] is the same number, for the atomics)
    if (lane == 0) listCount[1] = 0u;
    for (unsigned base = 0; base < total; base += 64u) {
        TPT_DEAL_T(tA_);
        TPT_DEAL_COUNT(95, 1);
        TPT_DEAL_COUNT(96, total - base < 64u ? total - base : 64u);
        const bool have = base + (unsigned)lane < total;
        unsigned e = 0;
        if (have) e = list[base + (unsigned)lane];
        const int po = (int)(e >> 16), g = (int)(e & 0xffffu);
        uint32_t mm = 0;
        f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0);
        const f4* mem = sv.gsph + (size_t)g * TPT_GROUP;
        if (have) {
            const f4 r0 = st[po], r1 = st[PATHS + po];
            ro = mk3(r0.z, r0.w, r1.x);
            rd = mk3(r1.y, r1.z, r1.w);
            const f3 dk = mk3(rd.x * TPT_P1_K, rd.y * TPT_P1_K, rd.z * TPT_P1_K);
            TPT_STAT(ST_SPHERELOOP); // profiling build: group visits
            TPT_PRAGMA_UNROLL(TPT_MEMBER_UNROLL)
            for (int j = 0; j < TPT_GROUP; ++j) mm |= (memberFilter(mem[j], ro, dk) ? 1u : 0u) << j;
        }
        TPT_DEAL_T(tB_);
        TPT_DEAL_ADD(93, tA_, tB_);
#if TPT_GROUP_DEAL_EXACT
        // The members that passed -- 0.27 per pair, ~15 per sub-round -- are dealt out once more for their exact tests: (owner's path id
        // << 20 | member slot) entries, pushed on a stack of survivors that grows in the list positions the sub-rounds have consumed so far
        // ([0, base + 64)).  An exact pass (45 instructions + two dependent LDS / L2 round trips) runs only when 64 survivors are
        // waiting, and once at the end for the rest: 1.8 passes per call instead of one per sub-round (5.2 at 15 busy lanes).
        // (A survivor that finds the stack full is tested in place by the lane that found it.)
        const unsigned cap = base + 64u;
        __builtin_amdgcn_wave_barrier();
        float ht = TPT_MAX_T;
        int hid = -1;
        if (mm) {
            unsigned pos = atomicAdd(&listCount[1], (unsigned)__popc(mm));
            while (mm) {
                TPT_DEAL_TRIP(103);
                const int j = __builtin_ctz(mm);
                mm &= mm - 1u;
                if (pos < cap) {
                    list[pos] = ((unsigned)po << 20) | (unsigned)(g * TPT_GROUP + j);
                } else {
                    TPT_STAT(ST_PHASE2);
                    testSphereTie(mem[j], sv.gid[g * TPT_GROUP + j], ro, rd, TPT_MIN_T, ht, hid);
                }
                ++pos;
            }
            if (hid >= 0) atomicMin(reinterpret_cast<unsigned long long*>(&st[po]), ((unsigned long long)f2u(ht) << 32) | (unsigned long long)(uint32_t)hid);
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned counted = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&listCount[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        TPT_DEAL_COUNT(98, counted - pend);
        pend = counted < cap ? counted : cap;
        if (pend >= 64u || counted > cap) {
            while (pend >= 64u) {
                pend -= 64u;
                dealExactPass<PATHS>(sv, list, pend, 64u, st, lane);
            }
            if (lane == 0) listCount[1] = pend;
        }
        __builtin_amdgcn_wave_barrier();
#else
        if (have) {
            float ht = TPT_MAX_T;
            int hid = -1;
            while (mm) {
                const int j = __builtin_ctz(mm);
                mm &= mm - 1u;
                TPT_STAT(ST_PHASE2);
                testSphereTie(mem[j], sv.gid[g * TPT_GROUP + j], ro, rd, TPT_MIN_T, ht, hid);
            }
            if (hid >= 0) atomicMin(reinterpret_cast<unsigned long long*>(&st[po]), ((unsigned long long)f2u(ht) << 32) | (unsigned long long)(uint32_t)hid);
        }
#endif
        TPT_DEAL_T(tC_);
        TPT_DEAL_ADD(94, tB_, tC_);
    }
#if TPT_GROUP_DEAL_EXACT
    if (pend != 0u) { // the survivors still waiting: one last pass (the stack is empty again for the next round)
        TPT_DEAL_T(tD_);
        dealExactPass<PATHS>(sv, list, 0u, pend, st, lane);
        if (lane == 0) listCount[1] = 0u;
        __builtin_amdgcn_wave_barrier();
        TPT_DEAL_T(tE_);
        TPT_DEAL_ADD(94, tD_, tE_);
    }
#endif
}
// A (ray, group) pair that found the list of waiting pairs full: its lane filters and tests the group's members itself (rare; slow).
template <int PATHS>
__device__ __forceinline__ void dealPairInPlace(const SceneView& sv, int po, int g, f4* st)
{
    const f4 r0 = st[po], r1 = st[PATHS + po]; // (read again here: the rare path must not keep the ray's six registers alive in the common one)
    const f3 ro = mk3(r0.z, r0.w, r1.x), rd = mk3(r1.y, r1.z, r1.w);
    const f3 dk = mk3(rd.x * TPT_P1_K, rd.y * TPT_P1_K, rd.z * TPT_P1_K);
    const f4* mem = sv.gsph + (size_t)g * TPT_GROUP;
    float ht = TPT_MAX_T;
    int hid = -1;
    for (int j = 0; j < TPT_GROUP; ++j) {
        const f4 s = mem[j];
        if (memberFilter(s, ro, dk)) {
            TPT_STAT(ST_PHASE2);
            testSphereTie(s, sv.gid[g * TPT_GROUP + j], ro, rd, TPT_MIN_T, ht, hid);
        }
    }
    if (hid >= 0) atomicMin(reinterpret_cast<unsigned long long*>(&st[po]), ((unsigned long long)f2u(ht) << 32) | (unsigned long long)(uint32_t)hid);
}
// One member pass of the three-stage dealing: lane j < n takes the (owner's path id << 16 | group) entry B[first + j], reads the owner's
// parked ray, filters the group's eight members and pushes the survivors -- (owner << 20 | member slot) -- on the stack S; an exact pass
// runs whenever 64 survivors are waiting.  pendS: survivors on the stack (wave-uniform; positions come from a prefix sum over the lanes).
template <int PATHS>
__device__ __forceinline__ void dealMemberPass(const SceneView& sv, LdsList B, unsigned first, unsigned n, LdsList S, unsigned& pendS, f4* st, int lane)
{
    TPT_DEAL_T(tA_);
    TPT_DEAL_COUNT(95, 1);
    TPT_DEAL_COUNT(96, n);
    const unsigned capS = TPT_DEAL_CS_RT;
    const bool have = (unsigned)lane < n;
    unsigned e = 0;
    if (have) e = B[first + (unsigned)lane];
    const int po = (int)(e >> 16), g = (int)(e & 0xffffu);
    uint32_t mm = 0;
    const f4* mem = sv.gsph + (size_t)g * TPT_GROUP;
    if (have) {
        const f4 r0 = st[po], r1 = st[PATHS + po];
        const f3 ro = mk3(r0.z, r0.w, r1.x), rd = mk3(r1.y, r1.z, r1.w);
        const f3 dk = mk3(rd.x * TPT_P1_K, rd.y * TPT_P1_K, rd.z * TPT_P1_K);
        TPT_STAT(ST_SPHERELOOP); // profiling build: group visits
        // (the filter's sign bits shifted in one v_alignbit each, like phase1PairLane: member j ends at bit 7 - j, set = rejected)
        uint32_t rej = 0;
        TPT_PRAGMA_UNROLL(TPT_MEMBER_UNROLL)
        for (int j = 0; j < TPT_GROUP; ++j) rej = alignbit(rej, f2u(memberFilterValue(mem[j], ro, dk)), 31);
        mm = ~rej & ((1u << TPT_GROUP) - 1u);
    }
    TPT_DEAL_T(tB_);
    TPT_DEAL_ADD(93, tA_, tB_);
    __builtin_amdgcn_wave_barrier();
    unsigned nSurv;
    unsigned pos = pendS + wavePrefix((unsigned)__popc(mm), nSurv);
    TPT_DEAL_COUNT(98, nSurv);
    if (mm) {
        float ht = TPT_MAX_T;
        int hid = -1;
        while (mm) {
            TPT_DEAL_TRIP(103);
            const int j = TPT_GROUP - 1 - __builtin_ctz(mm);
            mm &= mm - 1u;
            if (pos < capS) {
                S[pos] = ((unsigned)po << 20) | (unsigned)(g * TPT_GROUP + j);
            } else { // (the stack is full: this survivor is tested where it was found; the ray is read again -- see dealPairInPlace)
                TPT_STAT(ST_PHASE2);
                const f4 q0 = st[po], q1 = st[PATHS + po];
                testSphereTie(mem[j], sv.gid[g * TPT_GROUP + j], mk3(q0.z, q0.w, q1.x), mk3(q1.y, q1.z, q1.w), TPT_MIN_T, ht, hid);
            }
            ++pos;
        }
        if (hid >= 0) atomicMin(reinterpret_cast<unsigned long long*>(&st[po]), ((unsigned long long)f2u(ht) << 32) | (unsigned long long)(uint32_t)hid);
    }
    __builtin_amdgcn_wave_barrier();
    pendS += nSurv;
    pendS = pendS < capS ? pendS : capS;
    while (pendS >= 64u) {
        pendS -= 64u;
        dealExactPass<PATHS>(sv, S, pendS, 64u, st, lane);
    }
    __builtin_amdgcn_wave_barrier();
    TPT_DEAL_T(tC_);
    TPT_DEAL_ADD(94, tB_, tC_);
}
// The grouped traversal as THREE dealt stages (round 6, final form).  Per chunk of 64 super-groups (512 groups) the super-groups' bounds
// go through the wave-uniform packed filter (32 pair records, scalar loads); from there on nothing is done by "the lane that owns the
// ray" any more -- a ray touches 3.5 super-groups, 4.9 groups and 1.3 members on the 4096-sphere scene, the busiest lane of a wave 8.4,
// 15.5 and 10.5 (profiles/r06/r06_run28.log), so every per-owner loop ran at a third of the lanes:
//   A  every lane writes one (path, super-group) entry per super-group its ray touches (and parks its ray);
//   B  lane j takes A entry j (sub-rounds of 64): the owner's ray against the 8 groups of the super-group (4 pair records at gpairsLane:
//      LDS when they fit), one (path, group) entry per group touched, pushed on the stack B; whenever 64 entries are waiting there:
//   C  a member pass (dealMemberPass): 8 member filters per entry, survivors pushed on the stack S; whenever 64 are waiting: an exact pass.
// What is left on B and S when the last chunk is through is drained by one partial pass each.  Order is irrelevant: hits merge into the
// owners' keys with ds_min_u64 on (t bits, original index).  An entry that finds its stack full is served in place by the lane holding it.
// Both bounds levels test the HALF-line (TPT_DEAL_HALF_LINE; tpt_trace.h phase1PairT<true>): a bound whose centre lies behind the ray's
// origin, the origin outside it by a margin, holds nothing the reference could accept -- a quarter of what the line test kept.
template <int PATHS>
__device__ __forceinline__ void dealThreeStage(const SceneView& sv, const float* gpairsLane, int recStride, bool go, f3 o, f3 d, v2f ox, v2f oy, v2f oz, v2f dx, v2f dy, v2f dz,
                                               LdsList list, unsigned* cnt, f4* st, int p, int lane, float hitT, int id, bool& parked)
{
    LdsList A = list, B = list + TPT_DEAL_CA, S = list + TPT_DEAL_CA + TPT_DEAL_CB;
    const unsigned capA = TPT_DEAL_CA_RT, capB = TPT_DEAL_CB_RT;
    (void)cnt; // (the counters behind the entries serve the flat variants; here every position comes from a prefix sum over the lanes)
    unsigned nB = 0u, pendS = 0u; // entries waiting on B / S (wave-uniform)
    for (int sc0 = 0; sc0 < sv.nSuperPairs; sc0 += 32) {
        const int leftS = sv.nSuperPairs - sc0;
        TPT_DEAL_T(t0_);
        uint64_t sm = phase1ChunkT<TPT_DEAL_HALF_LINE != 0>(pairPtr(sv.spairs + (size_t)sc0 * 8), leftS < 32 ? leftS : 32, ox, oy, oz, dx, dy, dz); // super-group k of the chunk: bit 63 - k
        if (!go) sm = 0ull;
        TPT_DEAL_T(t1_);
        TPT_DEAL_ADD(91, t0_, t1_);
        while (__ballot(sm != 0ull) != 0ull) { // rounds: until every touched super-group of every lane has been entered (TPT_DEAL_CA per round)
            TPT_DEAL_T(t2_);
            TPT_DEAL_COUNT(97, 1);
            // ---- A: (path, super-group) entries
            unsigned nA;
            {
                unsigned pos = wavePrefix((unsigned)__popcll(sm), nA);
                if (sm != 0ull) {
                    while (pos < capA && sm != 0ull) { // the entries that fit; the rest stay in the mask for the next round
                        TPT_DEAL_TRIP(101);
                        const int k = __builtin_clzll(sm);
                        sm &= ~(0x8000000000000000ull >> k);
                        A[pos] = ((unsigned)p << 16) | (unsigned)(sc0 * 2 + k);
                        ++pos;
                    }
                    if (!parked) { // (o and d do not change during the call; the key is kept current by the atomics)
                        st[p] = mk4(u2f((uint32_t)id), hitT, o.x, o.y); // {key lo = sphere id, key hi = t bits, o.x, o.y}
                        st[PATHS + p] = mk4(o.z, d.x, d.y, d.z);
                        parked = true;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            TPT_DEAL_T(tR_);
            TPT_DEAL_ADD(104, t2_, tR_);
            nA = nA < capA ? nA : capA;
            // ---- B: lane j takes (path, super-group) entry j
            for (unsigned a0 = 0; a0 < nA; a0 += 64u) {
                TPT_DEAL_T(tS_);
                const bool have = a0 + (unsigned)lane < nA;
                unsigned e = 0;
                if (have) e = A[a0 + (unsigned)lane];
                const int po = (int)(e >> 16), sg = (int)(e & 0xffffu);
                uint32_t c8 = 0;
#if defined(TPT_STATS) && TPT_STATS >= 2
                unsigned statSgBehind = 0u, statGBehind = 0u;
#endif
                if (have) {
                    const f4 r0 = st[po], r1 = st[PATHS + po];
                    const f3 ro = mk3(r0.z, r0.w, r1.x), rd = mk3(r1.y, r1.z, r1.w);
                    // the owner's operands, made the way the owner makes them (hitSpheresGroupedDeal): the same bits, whoever computes
                    const v2f qx = {ro.x, ro.x}, qy = {ro.y, ro.y}, qz = {ro.z, ro.z};
                    const float gx = rd.x * TPT_PG_K, gy = rd.y * TPT_PG_K, gz = rd.z * TPT_PG_K;
                    const v2f ex = {gx, gx}, ey = {gy, gy}, ez = {gz, gz};
                    const float* rec = gpairsLane + sg * recStride; // (recStride: floats per super-group -- 32, or TPT_GPAIR_LDS_STRIDE in LDS)
                    uint32_t m = 0;
#pragma unroll
                    for (int q = 0; q < TPT_SUPER / 2; ++q) phase1PairLaneT<TPT_DEAL_HALF_LINE != 0>(rec + q * 8, qx, qy, qz, ex, ey, ez, m);
                    c8 = ~m & ((1u << TPT_SUPER) - 1u); // bit TPT_SUPER - 1 = the super-group's first group
#if defined(TPT_STATS) && TPT_STATS >= 2
                    // profiling build: how many of these candidates lie wholly BEHIND the ray's origin (centre behind: nb < 0; origin
                    // outside the bound by a margin: e' > 2^-11 R'^2) -- what a half-line test on top of the line test would take away
                    {
                        auto behind = [&](float cx, float cy, float cz, float nsq) {
                            const float ax = cx - ro.x, ay = cy - ro.y, az = cz - ro.z;
                            const float nb = fma1(az, gz, fma1(ay, gy, ax * gx));
                            const float ee = fma1(az, az, fma1(ay, ay, fma1(ax, ax, nsq)));
                            const float w = fma1(nsq, 0.00049316406f, ee);
                            return nb < 0.0f && w > 0.0f;
                        };
                        const float* sp = sv.spairs + (size_t)(sg >> 1) * 8 + (sg & 1);
                        statSgBehind = behind(sp[0], sp[2], sp[4], sp[6]) ? 1u : 0u;
                        for (int q = 0; q < TPT_SUPER / 2; ++q)
                            for (int h = 0; h < 2; ++h)
                                if (((c8 >> (TPT_SUPER - 1 - (2 * q + h))) & 1u) && behind(rec[q * 8 + h], rec[q * 8 + 2 + h], rec[q * 8 + 4 + h], rec[q * 8 + 6 + h])) statGBehind++;
                    }
#endif
                }
#if defined(TPT_STATS) && TPT_STATS >= 2
                {
                    unsigned t0_, t1_;
                    (void)wavePrefix(statSgBehind, t0_);
                    (void)wavePrefix(statGBehind, t1_);
                    TPT_DEAL_COUNT(105, t0_);
                    TPT_DEAL_COUNT(106, t1_);
                    TPT_DEAL_COUNT(107, nA - a0 < 64u ? nA - a0 : 64u);
                }
#endif
                __builtin_amdgcn_wave_barrier();
                unsigned nNew;
                unsigned pos = nB + wavePrefix((unsigned)__popc(c8), nNew);
                while (c8) {
                    TPT_DEAL_TRIP(102);
                    const int b = __builtin_ctz(c8);
                    c8 &= c8 - 1u;
                    const int g = sg * TPT_SUPER + (TPT_SUPER - 1) - b;
                    if (pos < capB)
                        B[pos] = ((unsigned)po << 16) | (unsigned)g;
                    else
                        dealPairInPlace<PATHS>(sv, po, g, st);
                    ++pos;
                }
                __builtin_amdgcn_wave_barrier();
                nB += nNew;
                nB = nB < capB ? nB : capB;
                TPT_DEAL_T(t3_);
                TPT_DEAL_ADD(92, tS_, t3_);
                // ---- C: member passes while 64 (path, group) entries are waiting
                while (nB >= 64u) {
                    nB -= 64u;
                    dealMemberPass<PATHS>(sv, B, nB, 64u, S, pendS, st, lane);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    // what is still waiting: one partial member pass, one partial exact pass (both stacks are empty again for the next call)
    if (nB != 0u) dealMemberPass<PATHS>(sv, B, 0u, nB, S, pendS, st, lane);
    if (pendS != 0u) {
        TPT_DEAL_T(tD_);
        dealExactPass<PATHS>(sv, S, 0u, pendS, st, lane);
        TPT_DEAL_T(tE_);
        TPT_DEAL_ADD(94, tD_, tE_);
    }
    __builtin_amdgcn_wave_barrier();
}
template <int PATHS>
__device__ __forceinline__ int hitSpheresGroupedDeal(const SceneView& sv, bool go, f3 o, f3 d, float& outT, LdsList list, unsigned* listCount,
                                                     f4* st, int p, int lane, const float* ldsGpairs, int boundsMode)
{
    float hitT = TPT_MAX_T;
    int id = -1;
    TPT_DEAL_T(tb0_);
    TPT_DEAL_COUNT(99, 1);
    {
        // The big spheres (ground, lights, dissolved groups: at most 64), wave-uniform in b: {centre, r^2} and the original index come
        // through SCALAR loads (constant address space, like the pair records) and feed the VALU as scalar operands; the per-sphere
        // conservative filter first (memberFilter = phase 1's arithmetic, 12 instructions), the exact test (45) under the lanes' mask only
        // where it passes (skipped when no lane is left).  Until round 6 this was a vector load per sphere with a full wait behind each --
        // five L2 round trips in a row per call, 10.7 % of the wave time at configs[4] (profiles/r06/r06_run22.log).
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const f4 __attribute__((address_space(4))) * BigPtr;
        typedef const int __attribute__((address_space(4))) * BigIdPtr;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        const BigPtr bs = (BigPtr)(sv.bsph);
        const BigIdPtr bi = (BigIdPtr)(sv.bid);
#pragma clang diagnostic pop
#else
        const f4* bs = sv.bsph;
        const int* bi = sv.bid;
#endif
        const f3 dk1 = mk3(d.x * TPT_P1_K, d.y * TPT_P1_K, d.z * TPT_P1_K);
#pragma unroll 4
        for (int b = 0; b < sv.nBig; ++b) {
            const f4 s = bs[b];
            const int sid = bi[b];
            if (go && memberFilter(s, o, dk1)) testSphereTie(s, sid, o, d, TPT_MIN_T, hitT, id);
        }
    }
    TPT_DEAL_T(tb1_);
    TPT_DEAL_ADD(90, tb0_, tb1_);
    const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
    const float gx = d.x * TPT_PG_K, gy = d.y * TPT_PG_K, gz = d.z * TPT_PG_K;
    const v2f dx = {gx, gx}, dy = {gy, gy}, dz = {gz, gz};
    bool parked = false;
    // The bounding spheres of the groups, three ways (the host decides: tptSetKernelVariant / tptGetSceneInfo):
    //   * two levels on the VALU (the default): the super-groups' bounds (TPT_SUPER = 8 consecutive groups each) through the
    //     wave-uniform packed filter -- 16 pair records per 256 groups instead of 128 --, then every lane tests the 8 groups of
    //     each super-group its own ray touches, their pair records read from LDS (from global memory when the scene has more groups
    //     than the LDS area holds);
    //   * on the matrix cores (one 4-KB tile pair per 64 groups through the vector cache) -- opt-in: in a time-sliced process
    //     (more hardware queues than the device runs side by side) a wave that has executed this path now and then sees a
    //     member gather of the dealing below deliver wrong data (profiles/r06);
    //   * the flat packed filter over all groups (hitSpheres variant 3: the A/B).
    const bool boundsOnMatrix = TPT_MATRIX_FILTER && TPT_GROUP_MATRIX_BOUNDS && sv.gmxTiles > 0;
    const bool twoLevel = !boundsOnMatrix && boundsMode >= 0 && sv.nSuperPairs > 0; // (boundsMode = KernelArgs::ldsGroupPairs)
    if (twoLevel) {
        // (two call sites: the LDS pointer must stay an LDS pointer -- merged with the global one it becomes a generic pointer and
        //  the per-lane reads FLAT loads, which reach LDS through the vector-memory path)
        if (boundsMode > 0)
            dealThreeStage<PATHS>(sv, ldsGpairs, TPT_GPAIR_LDS_STRIDE, go, o, d, ox, oy, oz, dx, dy, dz, list, listCount, st, p, lane, hitT, id, parked);
        else
            dealThreeStage<PATHS>(sv, sv.gpairs, (TPT_SUPER / 2) * 8, go, o, d, ox, oy, oz, dx, dy, dz, list, listCount, st, p, lane, hitT, id, parked);
    }
    MatrixRayOps mops;
    if (boundsOnMatrix) matrixRayOperands(o, d, 1, mops);
    for (int pb0 = 0; !twoLevel && pb0 < sv.nGroupPairs; pb0 += 128) {
        // wave-uniform filter on the bounding spheres of up to 256 groups: four 64-bit candidate masks per lane
        uint64_t cm0 = 0, cm1 = 0, cm2 = 0, cm3 = 0;
        if (boundsOnMatrix) {
            const int g0 = pb0 * 2, left = sv.nGroups - g0; // groups from here on
            const uint4* A = reinterpret_cast<const uint4*>(sv.gmatH) + (size_t)(g0 / 64) * (TPT_MXH_TABLE_DWORDS / 4);
            cm0 = matrixApply(A, 16, left < 64 ? left : 64, mops);
            if (left > 64) cm1 = matrixApply(A + TPT_MXH_TABLE_DWORDS / 4, 16, left - 64 < 64 ? left - 64 : 64, mops);
            if (left > 128) cm2 = matrixApply(A + 2 * (TPT_MXH_TABLE_DWORDS / 4), 16, left - 128 < 64 ? left - 128 : 64, mops);
            if (left > 192) cm3 = matrixApply(A + 3 * (TPT_MXH_TABLE_DWORDS / 4), 16, left - 192 < 64 ? left - 192 : 64, mops);
        } else {
            const int left = sv.nGroupPairs - pb0;
            cm0 = phase1Chunk(pairPtr(sv.gpairs + (size_t)pb0 * 8), left < 32 ? left : 32, ox, oy, oz, dx, dy, dz);
            if (left > 32) cm1 = phase1Chunk(pairPtr(sv.gpairs + (size_t)(pb0 + 32) * 8), left - 32 < 32 ? left - 32 : 32, ox, oy, oz, dx, dy, dz);
            if (left > 64) cm2 = phase1Chunk(pairPtr(sv.gpairs + (size_t)(pb0 + 64) * 8), left - 64 < 32 ? left - 64 : 32, ox, oy, oz, dx, dy, dz);
            if (left > 96) cm3 = phase1Chunk(pairPtr(sv.gpairs + (size_t)(pb0 + 96) * 8), left - 96 < 32 ? left - 96 : 32, ox, oy, oz, dx, dy, dz);
        }
        if (!go) cm0 = cm1 = cm2 = cm3 = 0ull;
        while (__ballot((cm0 | cm1 | cm2 | cm3) != 0ull) != 0ull) { // rounds: until every touched group of every lane has been dealt
            const unsigned n = (unsigned)(__popcll(cm0) + __popcll(cm1) + __popcll(cm2) + __popcll(cm3));
            if (lane == 0) *listCount = 0u;
            __builtin_amdgcn_wave_barrier();
            if (n != 0u) {
                unsigned pos = atomicAdd(listCount, n);
                while (pos < (unsigned)TPT_GROUP_DEAL_CAP) { // write the entries that fit; the rest stay in the masks
                    const int sel = cm0 ? 0 : cm1 ? 1 : cm2 ? 2 : 3;
                    const uint64_t w = cm0 ? cm0 : cm1 ? cm1 : cm2 ? cm2 : cm3;
                    if (!w) break;
                    const int k = __builtin_clzll(w);
                    const uint64_t keep = ~(0x8000000000000000ull >> k);
                    cm0 &= sel == 0 ? keep : ~0ull;
                    cm1 &= sel == 1 ? keep : ~0ull;
                    cm2 &= sel == 2 ? keep : ~0ull;
                    cm3 &= sel == 3 ? keep : ~0ull;
                    list[pos] = ((unsigned)p << 16) | (unsigned)((pb0 + sel * 32) * 2 + k);
                    ++pos;
                }
                if (!parked) { // (o and d do not change between rounds; the key is kept current by the atomics)
                    st[p] = mk4(u2f((uint32_t)id), hitT, o.x, o.y); // {key lo = sphere id, key hi = t bits, o.x, o.y}
                    st[PATHS + p] = mk4(o.z, d.x, d.y, d.z);
                    parked = true;
                }
            }
            __builtin_amdgcn_wave_barrier();
            dealProcessList<PATHS>(sv, list, listCount, st, lane);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (parked) {
        const f4 k = st[p];
        id = (int)f2u(k.x);
        hitT = k.y;
    }
    outT = hitT;
    return id;
}
#endif

// The path record, 64 B in LDS, four f4 planes [plane][path]:
//   [0] ray origin -- for a path waiting in a class queue: the HIT POSITION orig + dir * t (Maths.cpp:195) -- .xyz, rng
//   [1] ray direction .xyz, {sample : 11, depth : 4, doMatE : 1, hit id : 16}
//   [2] colour sum of the pixel's finished samples .xyz (Test.cpp:289), pixel x | y << 16
//   [3] bounce-stack level 0 {matE + lightE, attenuation id}: 56 % of all pushes; deeper levels live in global memory
// Everything else of a Lane is constant while a path sits in a queue (a main-chain ray, alive, no camera ray pending;
// sp == depth with the recursive fold) or is recomputed by the class code.
#ifndef TPT_Q_MIN_WAVES_PER_SIMD
#define TPT_Q_MIN_WAVES_PER_SIMD 4
#endif
// 4 waves/SIMD x 120 VGPRs leaves 32 registers per SIMD lane free: the resolve kernel's waves (10 VGPRs) can then
// start beside a machine full of persistent trace workgroups instead of waiting for one of them to retire (measured:
// resolve took 40-300 us instead of 4-10, and the ordered resolve chain is what bounds small / sharded frames).
// (On gfx90a+ LLVM doubles the requested number -- it assumes an equal AGPR half of the unified file -- so 60 means 120.)
#ifndef TPT_Q_MAX_VGPR
#define TPT_Q_MAX_VGPR 60
#endif
// BATCH: the launch traces a.batchFrames consecutive frames (tptDrawDeviceBatch).  A compile-time switch: the frame index
// a path carries costs the single-frame kernel two more spilled registers if it is a run-time one.
template <bool LDS_SCENE, bool BATCH>
__device__ __forceinline__ void traceQueueBody(const KernelArgs& a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS layout: everything of fixed size first, at compile-time offsets (immediates in the DS instructions instead of base
    // registers): path records, rings, control block, frame constants; then the scene arrays, whose sizes the launch decides
    // (kernels that stage the scene keep {centre, r^2} of up to 64 spheres -- every scene the matrix filter serves -- at offset
    //  0: phase 2 then addresses a sphere with sphere index x 16 and an immediate, like the path records)
    constexpr int kPaths = LDS_SCENE ? TPT_Q_PATHS : TPT_Q_PATHS_GROUPED; // paths this workgroup owns
    constexpr int kOffSt = LDS_SCENE ? TPT_Q_SPH_FIXED : 0;
    constexpr int kOffQ = kOffSt + TPT_Q_NF4 * kPaths * 16;
    constexpr int kOffCtl = kOffQ + Q_COUNT * TPT_Q_P * 2;
    constexpr int kOffDeal = kOffCtl + (((int)sizeof(QueueCtl) + 63) & ~63);
    constexpr int kDealBytes = (!LDS_SCENE && TPT_GROUP_DEAL) ? TPT_Q_WAVES * TPT_GROUP_DEAL_WAVE_BYTES : 0; // pair lists of the grouped traversal
    constexpr int kOffFc = kOffDeal + kDealBytes;
    constexpr int kOffScene = kOffFc + (((int)sizeof(FrameConsts) + 15) & ~15);
    f4* st = reinterpret_cast<f4*>(smem + kOffSt);
    LdsRing q = (LdsRing)(smem + kOffQ);
    QueueCtl* ctl = reinterpret_cast<QueueCtl*>(smem + kOffCtl);
    // the frame constants the camera code reads (22 camera floats, 1/w, 1/h): in LDS, read where a sample starts, instead of
    // ~30 SGPRs held (and spilled) across the whole loop
    FrameConsts* ldsFc = reinterpret_cast<FrameConsts*>(smem + kOffFc);
    const int nPad = a.scene.nPairs * 2;
    const bool sphFixed = LDS_SCENE && nPad * 16 <= TPT_Q_SPH_FIXED;
    f4* ldsSphFixed = reinterpret_cast<f4*>(smem);
    f4* ldsSph = sphFixed ? ldsSphFixed : reinterpret_cast<f4*>(smem + kOffScene);
    int off = kOffScene + ((LDS_SCENE && !sphFixed) ? nPad * 16 : 0);
    float* ldsInvR = reinterpret_cast<float*>(smem + off);
    off += LDS_SCENE ? ((nPad * 4 + 15) & ~15) : 0;
    f4* ldsLights = reinterpret_cast<f4*>(smem + off);
    off += a.scene.nLights * 32;
    f4* ldsMats = reinterpret_cast<f4*>(smem + off);
    off += LDS_SCENE ? a.scene.nSpheres * 48 : 0;
#if TPT_MATRIX_FILTER
    // phase 1 on the matrix cores (scenes with a table: <= 64 spheres in binary16 range): the A-operand table, 4 KB
    const bool useMatrix = LDS_SCENE && a.scene.mxR1 >= 0;
    uint32_t* ldsA = reinterpret_cast<uint32_t*>(smem + off);
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const bool helper = a.helperBase > 0;
    if (helper) { // (workgroup-uniform: one thread registers and looks, the barrier shares what it saw)
        unsigned* seen = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) {
            // register, THEN look (the launch's last wave closes, THEN looks for registrations).  Both sides use returning atomics --
            // performed at the device's coherence point -- and feed the first one's result into the second (a dependency the hardware
            // has to honour): no cache write-back / invalidate as a seq_cst fence at agent scope would cost every helper workgroup
            const unsigned was = __hip_atomic_fetch_add(&a.work[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned closed = __hip_atomic_fetch_or(&a.work[3], dependentZero(was), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned taken = __hip_atomic_load(&a.work[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long left = (long long)a.numChunks - (long long)taken;
            // closed >= gen (serials only grow; compared as a signed difference): the launch has finished, or the block serves a later one
            const bool join = (int)(closed - a.gen) < 0 && left > 0 && left * 100 >= (long long)a.numChunks * a.helperPct;
            if (!join) __hip_atomic_fetch_sub(&a.work[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *seen = join ? 1u : 0u;
        }
        __syncthreads();
        const unsigned join = *seen;
        __syncthreads();
        if (!join) return;
    }
    SceneView sv = a.scene;
    if (LDS_SCENE) {
        for (int i = tid; i < nPad; i += TPT_Q_T) {
            ldsSph[i] = a.scene.sph4[i];
            ldsInvR[i] = a.scene.invR[i];
        }
        for (int i = tid; i < a.scene.nSpheres * 3; i += TPT_Q_T) ldsMats[i] = a.scene.mats[i];
        sv.sph4 = ldsSph;
        sv.invR = ldsInvR;
        sv.mats = ldsMats;
    }
    for (int i = tid; i < a.scene.nLights * 2; i += TPT_Q_T) ldsLights[i] = a.scene.lights[i];
    sv.lights = ldsLights;
    // grouped scenes: the groups' bounding spheres (pair records) for the per-lane second level of the bounds filter, padded with
    // never-a-candidate records to whole super-groups (hitSpheresGroupedDeal); not when they do not fit (the launch says: ldsGroupPairs)
    // (a.ldsGroupPairs: > 0 records staged in LDS, 0 read from global memory, < 0 flat filter)
    float* ldsGpairs = reinterpret_cast<float*>(smem + ((off + 15) & ~15));
    if (!LDS_SCENE && a.ldsGroupPairs > 0) // (the host pads to whole super-groups; in LDS each super-group's 32 floats sit at a stride of TPT_GPAIR_LDS_STRIDE)
        for (int i = tid; i < a.ldsGroupPairs * 8; i += TPT_Q_T) ldsGpairs[(i / TPT_GPAIR_FLOATS) * TPT_GPAIR_LDS_STRIDE + (i % TPT_GPAIR_FLOATS)] = a.scene.gpairs[i];
#if TPT_MATRIX_FILTER
    if (useMatrix)
        for (int i = tid; i < TPT_MXH_TABLE_DWORDS; i += TPT_Q_T) ldsA[i] = a.scene.amatH[i];
    const int mxR1 = a.scene.mxR1;
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(TPT_STATS) && TPT_STATS >= 2
    if (tid < 4) g_hsLds[tid] = 0ull;
#if TPT_GROUP_DEAL
    if (tid < 18) g_dealLds[tid] = 0ull;
#endif
#endif
    for (int i = tid; i < (int)(sizeof(FrameConsts) / 4); i += TPT_Q_T) reinterpret_cast<uint32_t*>(ldsFc)[i] = reinterpret_cast<const uint32_t*>(&a.fc)[i];
    // every path starts in the FREE queue; all other queues empty (sentinel everywhere)
    for (int i = tid; i < Q_COUNT * TPT_Q_P; i += TPT_Q_T) q[i] = (unsigned short)(i < kPaths ? i : 0xFFFF);
    if (tid < 8) {
        ctl->head[tid] = 0u;
        ctl->tail[tid] = tid == Q_FREE ? (unsigned)kPaths : 0u;
    }
    if (tid == 0) {
        ctl->poolTotal = 0u;
        ctl->globalExhausted = 0u;
    }
    if (BATCH && tid < 32) ctl->frameRays[tid] = 0u;
    __syncthreads();

    const FrameConsts& fc = a.fc;
    const unsigned long long laneBelow = (1ull << lane) - 1ull;
#if TPT_MATRIX_FILTER
    SceneView svM = sv; // what phase 2 behind the matrix filter reads: {centre, r^2} at their compile-time LDS address
    svM.sph4 = ldsSphFixed;
#endif
#if TPT_GROUP_DEAL
    LdsList dealList = (LdsList)(smem + kOffDeal + (tid >> 6) * TPT_GROUP_DEAL_WAVE_BYTES);
    unsigned* dealCount = reinterpret_cast<unsigned*>(smem + kOffDeal + (tid >> 6) * TPT_GROUP_DEAL_WAVE_BYTES + TPT_GROUP_DEAL_ENTRIES * 4);
    const bool groupDeal = !LDS_SCENE && sv.nGroups > 0 && sv.nGroups <= 65536; // (16 bits of group index, 20 of member slot, in a list entry)
#endif
    f4* colSum = st + 2 * kPaths;                        // plane 2: per-path colour sums + pixel coordinates
    int chunkNext = 0, chunkEnd = 0; // this wave's private pixel pool
    int chunkFrame = 0;              // batched launch: the frame of the batch that pool belongs to
    bool noMoreChunks = false;
    unsigned myRays = 0;
#if defined(TPT_STATS)
    const unsigned long long qT0 = wall_clock64();
    unsigned qSteps = 0, qBatches = 0, qIdle = 0;
    unsigned long long qSec[5] = {0, 0, 0, 0, 0}; // s_memtime ticks: pick+pop, class code, intersect, push, idle
#endif

#if defined(TPT_STATS)
#define TPT_TSTAMP(v)                      \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_waitcnt(0);         \
    const unsigned long long v = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0)
#if TPT_STATS < 2
#define TPT_TADD(slot, a, b) do { if (lane == 0) atomicAdd(&g_tptStats[slot], (b) - (a)); } while (0)
#else // level 2: per-wave sums by KIND of section (slot & 3; idle = 4), flushed when the wave ends
#define TPT_TADD(slot, a, b) do { qSec[(slot) == 64 + 24 ? 4 : ((slot) & 3)] += (b) - (a); } while (0)
#endif
#else
#define TPT_TSTAMP(v) do { } while (0)
#define TPT_TADD(slot, a, b) do { } while (0)
#endif
    for (;;) {
        TPT_TSTAMP(tsTop);
        // ---- what is waiting?  lanes 0..5 read one queue each, broadcast through readlane
        unsigned myAvail = 0, myHead = 0, myTail = 0;
        if (lane < Q_COUNT) {
            myTail = __hip_atomic_load(&ctl->tail[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            myHead = __hip_atomic_load(&ctl->head[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            myAvail = myTail - myHead;
        }
        unsigned avail[Q_COUNT];
#pragma unroll
        for (int c = 0; c < Q_COUNT; ++c) avail[c] = (unsigned)__builtin_amdgcn_readlane((int)myAvail, c);
        const bool canStart = (chunkEnd > chunkNext) || !noMoreChunks;
        int pick = -1;
        unsigned best = 0;
        // full batches first (the fullest queue), FREE preferred so the pool of live paths stays full
        if (canStart && avail[Q_FREE] >= 64u) pick = Q_FREE;
        if (pick < 0) {
#pragma unroll
            for (int c = 1; c < Q_COUNT; ++c)
                if (avail[c] >= 64u && avail[c] > best) {
                    best = avail[c];
                    pick = c;
                }
        }
        if (pick < 0) { // no full batch anywhere: take the largest partial one
#pragma unroll
            for (int c = 0; c < Q_COUNT; ++c)
                if ((c != Q_FREE || canStart) && avail[c] > best) {
                    best = avail[c];
                    pick = c;
                }
        }
        if (pick < 0) {
            // nothing to do for this wave right now: done if every path is free and no pixel is left anywhere
            const unsigned pool = __hip_atomic_load(&ctl->poolTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned exhausted = __hip_atomic_load(&ctl->globalExhausted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (avail[Q_FREE] == (unsigned)kPaths && pool == 0u && exhausted != 0u && !canStart) break;
            if (exhausted != 0u && pool == 0u) noMoreChunks = true;
            TPT_STAT(ST_REFILL); // idle polls
#if defined(TPT_STATS)
            qIdle++;
#endif
            __builtin_amdgcn_s_sleep(4);
            TPT_TSTAMP(tsIdle);
            TPT_TADD(64 + 24, tsTop, tsIdle);
            continue;
        }
        int p = 0;
        const int n = qPop(q + pick * TPT_Q_P, &ctl->head[pick], &ctl->tail[pick], lane, p, (unsigned)__builtin_amdgcn_readlane((int)myHead, pick),
                           (unsigned)__builtin_amdgcn_readlane((int)myTail, pick));
        if (n == 0) continue;
#if defined(TPT_STATS)
        qBatches++;
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        TPT_TSTAMP(tsPop);
        TPT_TADD(64 + pick * 4 + 0, tsTop, tsPop);
        const bool mine = lane < n;
#if defined(TPT_STATS)
        if (mine) { TPT_STAT(16 + pick); } // [16+pick] batches popped per queue, [48+pick] paths in them
#endif

        // ---- this iteration's path state, in registers: the ray the batch produces (or, from a class queue, the hit
        //      position and the incoming direction), the rng, the flags of the record's word
        f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0);
        uint32_t rng = 0;
        int sample = 0, depth = 0, recId = -1;
        bool doMatE = true;
        bool ray = false;    // this lane holds a ray that still has to be intersected
        bool toFree = false; // this lane's path goes back to the FREE queue
        bool toEnd = false;  // Metal whose scattered ray points into the surface: the path ends (END class), nothing to intersect
        QStack stack;
        stack.l0 = (LdsF4Ptr)(st + 3 * kPaths + p); // level 0 in the path record
        stack.spill = a.stackBuf + ((size_t)(blockIdx.x + (unsigned)a.helperBase) * TPT_Q_PATHS + p);
        stack.stride = a.stackStride;
        QLambert lam;
        lam.sdir = lam.nl = lam.albedo = lam.lightE = mk3(0, 0, 0);
        lam.cosAMax = 0.0f;
        if (pick != Q_FREE && mine) {
            const f4 r0 = st[0 * kPaths + p], r1 = st[1 * kPaths + p];
            ro = mk3(r0.x, r0.y, r0.z);
            rng = f2u(r0.w);
            rd = mk3(r1.x, r1.y, r1.z);
            const uint32_t w = f2u(r1.w);
            sample = (int)(w & 0x7ffu);
            depth = (int)((w >> 11) & 15u);
            doMatE = ((w >> 15) & 1u) != 0;
            recId = (w >> 16) == 0xffffu ? -1 : (int)(w >> 16);
        }

        if (pick == Q_FREE) {
            // ---- start pixels on free paths (this wave's chunk pool, refilled from the global counter)
            bool need = mine, got = false;
            int px = 0, py = 0, laneFrame = 0;
            for (;;) {
                const unsigned long long needMask = __ballot(need);
                if (needMask == 0ull) break;
                if (chunkNext >= chunkEnd) {
                    if (noMoreChunks) break;
                    int c = 0;
                    if (lane == 0) {
                        atomicAdd(&ctl->poolTotal, 64u); // optimistic: keeps "pixels left" non-zero while the fetch is in flight
                        c = (int)atomicAdd(&a.work[0], 1u);
                    }
                    c = __builtin_amdgcn_readfirstlane(c);
                    if (c >= a.numChunks) {
                        if (lane == 0) {
                            atomicSub(&ctl->poolTotal, 64u);
                            __hip_atomic_store(&ctl->globalExhausted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        noMoreChunks = true;
                        break;
                    }
                    chunkFrame = 0;
                    if (BATCH) { // batched launch: chunk c belongs to frame c / chunksPerFrame of the batch
                        chunkFrame = c / a.chunksPerFrame;
                        c -= chunkFrame * a.chunksPerFrame;
                    }
                    chunkNext = c * a.chunkSize;
                    chunkEnd = chunkNext + a.chunkSize;
                    if (chunkEnd > a.numItems) chunkEnd = a.numItems;
                    if (lane == 0 && chunkEnd - chunkNext != 64) atomicSub(&ctl->poolTotal, (unsigned)(64 - (chunkEnd - chunkNext)));
                }
                const int rank = __popcll(needMask & laneBelow);
                const int want = __popcll(needMask);
                const int availPx = chunkEnd - chunkNext;
                const int take = want < availPx ? want : availPx;
                if (need && rank < take) {
                    int x, ly;
                    if (mapItem(a, chunkNext + rank, x, ly)) {
                        px = x;
                        py = localRowToGlobal(a, ly);
                        laneFrame = chunkFrame;
                        got = true;
                        need = false;
                    }
                }
                chunkNext += take;
                if (lane == 0) atomicSub(&ctl->poolTotal, (unsigned)take);
            }
            if (got) {
                rng = pixelSeed(fc.seedMode, px, py, fc.frame + (BATCH ? laneFrame : 0));
                // colour sum = 0; the pixel: x | y << 16, or in a batched launch x | y << 13 | frame << 26
                const uint32_t where = BATCH ? ((uint32_t)px | ((uint32_t)py << 13) | ((uint32_t)laneFrame << 26))
                                             : ((uint32_t)px | ((uint32_t)py << 16));
                colSum[p] = mk4(0.0f, 0.0f, 0.0f, u2f(where));
                qCamera(*ldsFc, px, py, rng, ro, rd);
                ray = true;
            } else if (mine) {
                toFree = true; // no pixel left for this path
            }
        } else if (pick == Q_INT) {
            ray = mine; // overflow: rays of sparse batches, re-batched
        } else if (pick == Q_END) {
            // ---- a path ended (sky / emission): fold, add the sample to the pixel's sum, next sample or pixel done
            if (mine) {
                const f3 c = qFold(sv, qEndTerm(sv, fc, rd, recId), depth, stack);
                const f4 c3 = colSum[p];
                const f3 col = mk3(c3.x, c3.y, c3.z) + c; // same order of additions as Test.cpp:289
                int px, py, plane = 0;
                if (BATCH) {
                    px = (int)(f2u(c3.w) & 0x1fffu);
                    py = (int)((f2u(c3.w) >> 13) & 0x1fffu);
                    plane = (int)(f2u(c3.w) >> 26) * a.framePlane;
                } else {
                    px = (int)(f2u(c3.w) & 0xffffu);
                    py = (int)(f2u(c3.w) >> 16);
                }
                sample++;
                if (sample < fc.spp) {
                    colSum[p] = mk4(col.x, col.y, col.z, c3.w);
                    qCamera(*ldsFc, px, py, rng, ro, rd);
                    depth = 0;
                    doMatE = true;
                    ray = true;
                } else {
                    const f3 out = col * fc.invSpp; // Test.cpp:291
                    a.frameColour[plane + globalRowToLocal(a, py) * fc.width + px] = mk4(out.x, out.y, out.z, 0.0f); // one 16-B store per pixel
                    toFree = true;
                }
            }
        } else if (pick == Q_DIEL) {
            if (mine) {
                f3 e;
                rd = qDielectric(sv, fc, ro, rd, recId, doMatE, rng, e);
                qStackPush(stack, depth, e, -1);
                depth++;
                doMatE = true; // Test.cpp:214
                ray = true;
            }
        } else if (pick == Q_METAL) {
            if (mine) {
                f3 e, nd;
                if (qMetal(sv, fc, ro, rd, recId, doMatE, rng, e, nd)) {
                    qStackPush(stack, depth, e, recId);
                    depth++;
                    doMatE = true;
                    rd = nd;
                    ray = true;
                } else {
                    toEnd = true; // Test.cpp:218-221: return matE -- the END class does that from the record as it stands
                }
            }
        } else { // Q_LAMBERT
            if (mine) {
                qLambertBegin(sv, ro, rd, recId, rng, lam);
                ray = true;
            }
        }

        TPT_TSTAMP(tsClass);
        TPT_TADD(64 + pick * 4 + 1, tsPop, tsClass);
        // ---- HitWorld for the rays this batch produced.  A Lambert batch first runs its light loop (Test.cpp:96-133), wave-
        //      uniform in j: shadow ray, intersection, shading, all in registers; its last trip intersects the bounce ray.
        int cls = -1;
        unsigned iterRays = 0; // (batched launch: this lane's rays of this iteration, counted for the frame its path belongs to)
        const int nRay = __popcll(__ballot(ray));
        if (pick == Q_FREE || pick == Q_INT || pick == Q_LAMBERT || nRay >= TPT_Q_FUSE_MIN) {
            const int nShadow = (pick == Q_LAMBERT && (fc.config & CFG_LIGHT_SAMPLING)) ? sv.nLights : 0;
            int hitId = -1;
            float hitT = 0.0f;
            for (int j = 0; j <= nShadow; ++j) {
#if defined(TPT_STATS)
                qSteps++;
#endif
                const bool shadow = j < nShadow;
                if (pick == Q_LAMBERT && !shadow && ray) {
                    // the light loop is over: what this level adds to the fold, then the bounce ray (Test.cpp:91, 210-216)
                    qStackPush(stack, depth, qLambertE(sv, recId, doMatE, lam), recId);
                    depth++;
                    doMatE = !(fc.config & CFG_LIGHT_SAMPLING); // Test.cpp:209-214: only with light sampling
                    rd = lam.sdir;
                }
                f3 d2 = rd;
                bool go = ray;
                int lightId = -2;
                f4 l1 = mk4(0.0f, 0.0f, 0.0f, 0.0f);
                if (shadow) {
                    l1 = sv.lights[j * 2 + 1];
                    lightId = (int)f2u(l1.w);
                    go = ray && lightId != recId; // Test.cpp:100: not the sphere itself
                    if (go) d2 = qLightRay(sv.lights[j * 2], ro, rng, lam.cosAMax, (sv.flags & SCENE_LIGHT_R2_DIV_SAFE) != 0);
                }
#if TPT_MATRIX_FILTER
                // phase 1 of HitSpheres for the whole wave on the matrix cores: every lane takes part (this loop is wave-uniform);
                // lanes without a ray feed the finite values they hold and ignore their mask
                uint64_t cand = 0ull;
                if (LDS_SCENE && useMatrix) cand = phase1MatrixH(ldsA, mxR1, sv.nSpheres, ro, d2);
#endif
#if TPT_GROUP_DEAL
                int dealId = -1;
                float dealT = TPT_MAX_T;
                if (!LDS_SCENE && groupDeal) // (wave-uniform: every lane takes part, lanes without a ray as helpers only)
                    dealId = hitSpheresGroupedDeal<kPaths>(sv, go, ro, d2, dealT, dealList, dealCount, st, p, lane, ldsGpairs, a.ldsGroupPairs);
#endif
                if (go) {
                    TPT_STAT(ST_STEP);
                    float t;
                    int id;
#if TPT_GROUP_DEAL
                    if (!LDS_SCENE && groupDeal) {
                        id = dealId;
                        t = dealT;
                    } else
#endif
#if TPT_MATRIX_FILTER
                    if (LDS_SCENE && useMatrix)
                        id = hitSpheresCandidates(svM, cand, ro, d2, TPT_MIN_T, TPT_MAX_T, t);
                    else
#endif
                        id = hitSpheres<LDS_SCENE ? HS_TWO_PHASE : HS_TWO_PHASE_GROUPS>(sv, ro, d2, TPT_MIN_T, TPT_MAX_T, t);
                    if (BATCH) iterRays++; else myRays++;
                    if (shadow) {
                        if (id == lightId) qLightShade(l1, d2, lam);
                    } else {
                        hitId = id;
                        hitT = t;
                    }
                }
            }
            if (ray) {
                if (hitId < 0 || depth >= TPT_MAX_DEPTH)
                    cls = Q_END;
                else {
                    const int type = (int)f2u(sv.mats[hitId * 3].w);
                    cls = type == MAT_LAMBERT ? Q_LAMBERT : type == MAT_METAL ? Q_METAL : type == MAT_DIELECTRIC ? Q_DIEL : Q_END;
                }
                if (hitId >= 0) ro = ro + rd * hitT; // the hit position (Maths.cpp:195), all the class code needs of {orig, t}
                recId = hitId;
            }
        } else if (ray) {
            cls = Q_INT;
        }
        if (toEnd) cls = Q_END;
        if (ray || toEnd) {
            const uint32_t w = ((uint32_t)sample & 0x7ffu) | (((uint32_t)depth & 15u) << 11) | ((uint32_t)doMatE << 15) | (((uint32_t)recId & 0xffffu) << 16);
            st[0 * kPaths + p] = mk4(ro.x, ro.y, ro.z, u2f(rng));
            st[1 * kPaths + p] = mk4(rd.x, rd.y, rd.z, u2f(w));
        }
        if (toFree) cls = Q_FREE;
        if (BATCH && iterRays != 0u) { // (one LDS atomic per lane and iteration)
            const unsigned fr = f2u(colSum[p].w) >> 26;
            const unsigned old = atomicAdd(&ctl->frameRays[fr], iterRays);
            // a 32-bit counter per workgroup and frame can wrap on a very large frame at high spp on a small grid: the lane whose
            // addition crosses 2^31 moves 2^31 rays to the frame's global counter (one lane per crossing)
            if (__builtin_expect(old < 0x80000000u && old + iterRays >= 0x80000000u, 0)) {
                atomicSub(&ctl->frameRays[fr], 0x80000000u);
                atomicAdd(a.rayCounter + (size_t)fr * a.rayCounterStride, 0x80000000ull);
            }
        }
        TPT_TSTAMP(tsInt);
        TPT_TADD(64 + pick * 4 + 2, tsClass, tsInt);
        // (the cold state is global memory, but every wave that can pop this path runs on this CU and shares its L1:
        //  the workgroup-scope release orders the stores before the queue entry becomes visible)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        qPushByClass(q, ctl, mine ? cls : -1, p, lane);
        TPT_TSTAMP(tsPush);
        TPT_TADD(64 + pick * 4 + 3, tsInt, tsPush);
    }

    const unsigned waveRays = waveReduceAdd(myRays);
    if (BATCH) {
        // every frame of the batch has its own counter (rayCounterStride 1: a caller that is served the frames one by one
        // gets each frame's own count; 0: they all add to the context's running total)
        __syncthreads();
        if (tid < a.batchFrames && ctl->frameRays[tid] != 0u) atomicAdd(a.rayCounter + (size_t)tid * a.rayCounterStride, (unsigned long long)ctl->frameRays[tid]);
    }
    if (helper) {
        // this workgroup's pixels are stored and its rays counted: leave the launch (release: the stores reach memory first)
        if (lane == 0 && !BATCH) atomicAdd(a.rayCounter, (unsigned long long)waveRays);
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_sub(&a.work[2], 1u, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (lane == 0) {
        if (!BATCH) atomicAdd(a.rayCounter, (unsigned long long)waveRays);
        unsigned done = atomicAdd(&a.work[1], 1u) + 1u;
        if (done == a.totalWaves) {
            if (a.gen != 0u) {
                // close, THEN look for registered helpers (they register, then look for "closed": one side always sees the other).
                // They are resident workgroups finishing the chunks they took: bounded work, so the wait has NO cap -- a cap that
                // expired (long chunks of the grouped kernel, a time-sliced or debugged process) re-armed the pool under helpers
                // that were still running, and their late stores landed in the slot's next frame without anyone noticing.
                const unsigned prev = __hip_atomic_exchange(&a.work[3], a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned busy = __hip_atomic_fetch_or(&a.work[2], dependentZero(prev), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (busy != 0u) {
                    __builtin_amdgcn_s_sleep(127);
                    busy = __hip_atomic_fetch_or(&a.work[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            a.work[0] = 0u;
            a.work[1] = 0u;
        }
    }
#if defined(__HIP_DEVICE_COMPILE__) && defined(TPT_STATS) && TPT_STATS >= 2
    __syncthreads();
    if (tid < 4) TPT_COUNT(120 + tid, g_hsLds[tid]);
#if TPT_GROUP_DEAL
    if (tid < 15) TPT_COUNT(90 + tid, g_dealLds[tid]);
    if (tid >= 15 && tid < 18) TPT_COUNT(124 + tid - 15, g_dealLds[tid]); // (slots 105-110 belong to the wave statistics)
#endif
#endif
#if defined(TPT_STATS)
    if (lane == 0) {
        const unsigned long long qT1 = wall_clock64();
        TPT_COUNT(106, qSteps);       // intersection iterations of this wave
        TPT_COUNT(107, waveRays);     // rays in them
        TPT_COUNT(108, qT1 - qT0);    // wave lifetime, 10 ns ticks
        TPT_COUNT(109, 1);            // waves
        TPT_COUNT(110, qBatches);     // batches popped
        TPT_COUNT(105, qIdle);        // idle polls
#if TPT_STATS >= 2
        for (int k = 0; k < 5; ++k) TPT_COUNT(112 + k, qSec[k]);
#endif
    }
#endif
}

// The kernels proper.  Scenes staged in LDS (<= 64 spheres with the matrix filter: the headline) keep the 120-register cap above.
// Grouped scenes (no LDS scene: the 4096-sphere stress scene) hold four candidate masks, the dealing state and the group's member
// gathers on top: at 120 registers that instantiation spilled 32 of them (128 B of scratch per lane, 11x the algorithmic HBM traffic);
// it gets the full 128 of a 4-wave SIMD -- its launches last ~200 ms, the blend behind them can wait for a workgroup to retire.
#ifndef TPT_Q_MAX_VGPR_GROUPED
#define TPT_Q_MAX_VGPR_GROUPED 64
#endif
template <bool LDS_SCENE, bool BATCH = false>
__global__ void __launch_bounds__(TPT_Q_T, TPT_Q_MIN_WAVES_PER_SIMD) __attribute__((amdgpu_num_vgpr(TPT_Q_MAX_VGPR)))
tptTraceQueueKernel(const KernelArgs a)
{
    traceQueueBody<LDS_SCENE, BATCH>(a);
}
template <>
__global__ void __launch_bounds__(TPT_Q_T, TPT_Q_MIN_WAVES_PER_SIMD) __attribute__((amdgpu_num_vgpr(TPT_Q_MAX_VGPR_GROUPED)))
tptTraceQueueKernel<false, false>(const KernelArgs a)
{
    traceQueueBody<false, false>(a);
}
template <>
__global__ void __launch_bounds__(TPT_Q_T, TPT_Q_MIN_WAVES_PER_SIMD) __attribute__((amdgpu_num_vgpr(TPT_Q_MAX_VGPR_GROUPED)))
tptTraceQueueKernel<false, true>(const KernelArgs a)
{
    traceQueueBody<false, true>(a);
}

#if defined(TPT_TEST_HOOKS)
// ---------------------------------------------------------------- unit-test kernels (GPU parity of the math layer)
// op: 0 sqrt(a) 1 a/b 2 tsinf(a) 3 tcosf(a) 4 tpow5f(a) 5 rnd01 stream (a = seed bits) 6 schlick(a,b) 7 1/sqrt-normalize.x 8 / 9 sin / cos of tsincosf(a) 10 tdivSafeNum(a, b) 11 tdivByPi(a)
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
    case 8: { float sn, cs; tsincosf(x, sn, cs); r = sn; break; } // the pair the path uses, against ops 2 / 3
    case 9: { float sn, cs; tsincosf(x, sn, cs); r = cs; break; }
    case 10: r = tdivSafeNum(x, y); break; // Scatter's r^2 / d^2 (numerator range-checked on the host, divisor guarded in the function)
    case 11: r = tdivByPi(x); break;       // Scatter's x / kPI
    }
    out[i] = r;
}

// Exhaustive self-check of the fast correctly-rounded paths of tpt_math.h against the compiler's own correctly rounded
// expansions, for every binary32 bit pattern in [lo, hi]: op 0 tsqrt(x) vs sqrtf(x), op 1 trsqrt2(x) vs 1.0f / sqrtf(x).
// (NaN results compare equal to NaN.)  out[0] = mismatches, firstBad[0..7] = the first few offending inputs.
__global__ void __launch_bounds__(256) tptMathExhaustiveKernel(int op, uint32_t lo, uint32_t hi, unsigned long long* nBad, uint32_t* firstBad)
{
    const uint64_t n = (uint64_t)hi - lo + 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = lo + (uint32_t)i;
        const float x = u2f(b);
        const float want = op == 0 ? __builtin_sqrtf(x) : 1.0f / __builtin_sqrtf(x);
        const float got = op == 0 ? tsqrt(x) : trsqrt2(x);
        if (f2u(got) != f2u(want) && !(got != got && want != want)) {
            const unsigned long long k = atomicAdd(nBad, 1ull);
            if (k < 8) firstBad[k] = b;
        }
    }
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

// matrix-core filter alone: n rays (n a multiple of 64; one wave per 64), candidate masks out; outId / outT (optional): the
// nearest hit through the filter + the exact test for its candidates (what the path-queue kernel runs)
__global__ void __launch_bounds__(64) tptMatrixFilterTestKernel(const KernelArgs a, const float* __restrict__ rays, unsigned long long* __restrict__ outMask,
                                                                int* __restrict__ outId, float* __restrict__ outT, int n)
{
    __shared__ __attribute__((aligned(16))) uint32_t ldsA[TPT_MXH_TABLE_DWORDS];
    for (int i = threadIdx.x; i < TPT_MXH_TABLE_DWORDS; i += 64) ldsA[i] = a.scene.amatH[i];
    __syncthreads();
    const int i = blockIdx.x * 64 + threadIdx.x; // n is padded to whole waves by the caller
    f3 o = mk3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), d = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    const uint64_t cand = phase1MatrixH(ldsA, a.scene.mxR1, a.scene.nSpheres, o, d);
    if (outMask) outMask[i] = cand;
    if (outId) {
        float t;
        outId[i] = hitSpheresCandidates(a.scene, cand, o, d, TPT_MIN_T, TPT_MAX_T, t);
        outT[i] = t;
    }
}

// the matrix-core filter over the group bounds against the exact test of every member: one wave per 64 rays (rays beyond n are
// padding: a valid far-away ray whose results are ignored); out[0] = violations, out[1] = groups kept, out[2] = exact line hits
__global__ void __launch_bounds__(64) tptGroupFilterTestKernel(const KernelArgs a, const float* __restrict__ rays, int n, unsigned long long* __restrict__ out)
{
    const SceneView& sv = a.scene;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const bool real = i < n;
    f3 o = mk3(0.0f, 1.0e4f, 0.0f), d = mk3(0.0f, 1.0f, 0.0f);
    if (real) {
        o = mk3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]);
        d = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    }
    unsigned long long bad = 0, kept = 0, exact = 0, keptHalf = 0;
    if (sv.gmxTiles == 0) {
        // no matrix-core table in the scene set (the default): the two-level packed VALU filter of the path-queue kernel, the groups'
        // pair records read per lane from global memory here (the kernel stages them in LDS when they fit)
        const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
        const float gx = d.x * TPT_PG_K, gy = d.y * TPT_PG_K, gz = d.z * TPT_PG_K;
        const v2f dx = {gx, gx}, dy = {gy, gy}, dz = {gz, gz};
        // per lane, super-group by super-group: the super-group's own bound, then its groups' -- in the line form and in the half-line
        // form the three-stage dealing uses, the latter held against the reference's WHOLE acceptance (Maths.cpp:171-190: a positive
        // discriminant and a root beyond tMin); it must be a subset of the line form
        const int nSupers = (sv.nGroups + TPT_SUPER - 1) / TPT_SUPER;
        for (int sg = 0; sg < nSupers && real; ++sg) {
            uint32_t sl = 0, sh = 0;
            phase1PairLaneT<false>(sv.spairs + (size_t)(sg >> 1) * 8, ox, oy, oz, dx, dy, dz, sl);
            phase1PairLaneT<true>(sv.spairs + (size_t)(sg >> 1) * 8, ox, oy, oz, dx, dy, dz, sh);
            const bool sgLine = !((sl >> (1 - (sg & 1))) & 1u), sgHalf = !((sh >> (1 - (sg & 1))) & 1u);
            for (int q = 0; q < TPT_SUPER / 2; ++q) {
                const float* rec = sv.gpairs + ((size_t)sg * (TPT_SUPER / 2) + q) * 8;
                uint32_t gl = 0, gh = 0;
                phase1PairLaneT<false>(rec, ox, oy, oz, dx, dy, dz, gl);
                phase1PairLaneT<true>(rec, ox, oy, oz, dx, dy, dz, gh);
                for (int hIdx = 0; hIdx < 2; ++hIdx) {
                    const int grp = sg * TPT_SUPER + q * 2 + hIdx;
                    if (grp >= sv.nGroups) continue;
                    const bool keptBit = sgLine && !((gl >> (1 - hIdx)) & 1u), keptHalfBit = sgHalf && !((gh >> (1 - hIdx)) & 1u);
                    kept += keptBit;
                    keptHalf += keptHalfBit;
                    if (keptHalfBit && !keptBit) ++bad;
                    const f4* mem = sv.gsph + (size_t)grp * TPT_GROUP;
                    for (int j = 0; j < TPT_GROUP; ++j) {
                        const f4 s = mem[j];
                        const float coX = s.x - o.x, coY = s.y - o.y, coZ = s.z - o.z;
                        const float nb = coX * d.x + coY * d.y + coZ * d.z;
                        const float c = coX * coX + coY * coY + coZ * coZ - s.w;
                        const float discr = nb * nb - c;
                        if (discr > 0) {
                            ++exact;
                            if (!keptBit) ++bad;
                            const float sq = tsqrt(discr);
                            float t = nb - sq;
                            if (t <= TPT_MIN_T) t = nb + sq;
                            if (t > TPT_MIN_T && !keptHalfBit) ++bad; // a hit the reference takes (whatever hitT it holds: tMax = 1e7) in a dropped group
                        }
                    }
                }
            }
        }
        if (bad) atomicAdd(&out[0], bad);
        atomicAdd(&out[1], kept);
        atomicAdd(&out[2], exact);
        atomicAdd(&out[3], keptHalf);
        return;
    }
    MatrixRayOps mops;
    matrixRayOperands(o, d, 1, mops);
    for (int t = 0; t < sv.gmxTiles; ++t) {
        const int left = sv.nGroups - t * 64;
        const uint64_t m = matrixApply(reinterpret_cast<const uint4*>(sv.gmatH) + (size_t)t * (TPT_MXH_TABLE_DWORDS / 4), 16, left < 64 ? left : 64, mops);
        if (!real) continue;
        kept += (unsigned long long)__popcll(m);
        for (int q = 0; q < 64 && q < left; ++q) {
            const bool keptBit = (m >> (63 - q)) & 1ull;
            const f4* mem = sv.gsph + (size_t)(t * 64 + q) * TPT_GROUP;
            for (int j = 0; j < TPT_GROUP; ++j) {
                const f4 s = mem[j];
                // the reference's discriminant, Maths.cpp:171-178 (padding members carry r^2 = -inf: never positive)
                const float coX = s.x - o.x, coY = s.y - o.y, coZ = s.z - o.z;
                const float nb = coX * d.x + coY * d.y + coZ * d.z;
                const float c = coX * coX + coY * coY + coZ * coZ - s.w;
                const float discr = nb * nb - c;
                if (discr > 0) {
                    ++exact;
                    if (!keptBit) ++bad;
                }
            }
        }
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], kept);
    atomicAdd(&out[2], exact);
}
#endif // TPT_TEST_HOOKS

} // namespace tpt

// ---------------------------------------------------------------- launch glue (called from tpt_host.cpp)
using namespace tpt;

#if defined(TPT_TEST_HOOKS)
int tptReadStats(unsigned long long* out64)
{
#if defined(TPT_STATS)
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_tptStats), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -2;
#else
    (void)out64;
    return -1;
#endif
}
int tptResetStats()
{
#if defined(TPT_STATS)
    unsigned long long z[128] = {0};
    z[25] = ~0ull;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tptStats), z, sizeof(z)) == hipSuccess ? 0 : -2;
#else
    return -1;
#endif
}
#endif // TPT_TEST_HOOKS

size_t tptLdsBytes(const KernelArgs& a, int fold, bool ldsScene)
{
    const int nPad = a.scene.nPairs * 2;
    size_t bytes = 0;
    if (ldsScene) bytes += (size_t)nPad * 16 + (((size_t)nPad * 4 + 15) & ~(size_t)15);
    bytes += (size_t)a.scene.nLights * 32;
    if (ldsScene) bytes += (size_t)a.scene.nSpheres * 48;
    if (fold == FOLD_RECURSIVE) bytes += (size_t)a.ldsStackLevels * TPT_BLOCK * 16;
    return bytes;
}

template <int HS, int FOLD, bool LDS_SCENE>
static hipError_t launchOne(const KernelArgs& a, int blocks, size_t lds, hipStream_t stream)
{
    auto k = tptTraceKernel<HS, FOLD, LDS_SCENE>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(blocks), dim3(TPT_BLOCK), lds, stream, a);
    return hipGetLastError();
}

template <int HS, int FOLD, bool LDS_SCENE>
static int occupancyOne(size_t lds)
{
    int nb = 0;
    auto k = tptTraceKernel<HS, FOLD, LDS_SCENE>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), TPT_BLOCK, lds) != hipSuccess) nb = 1;
    return nb < 1 ? 1 : nb;
}

#define TPT_DISPATCH(FN, ...)                                                                  \
    do {                                                                                       \
        const int key = (hs ? 4 : 0) | (fold ? 2 : 0) | (ldsScene ? 1 : 0);                    \
        switch (key) {                                                                         \
        case 0: return FN<HS_TWO_PHASE, FOLD_RECURSIVE, false>(__VA_ARGS__);                   \
        case 1: return FN<HS_TWO_PHASE, FOLD_RECURSIVE, true>(__VA_ARGS__);                    \
        case 2: return FN<HS_TWO_PHASE, FOLD_FORWARD, false>(__VA_ARGS__);                     \
        case 3: return FN<HS_TWO_PHASE, FOLD_FORWARD, true>(__VA_ARGS__);                      \
        case 4: return FN<HS_SIMPLE, FOLD_RECURSIVE, false>(__VA_ARGS__);                      \
        case 5: return FN<HS_SIMPLE, FOLD_RECURSIVE, true>(__VA_ARGS__);                       \
        case 6: return FN<HS_SIMPLE, FOLD_FORWARD, false>(__VA_ARGS__);                        \
        default: return FN<HS_SIMPLE, FOLD_FORWARD, true>(__VA_ARGS__);                        \
        }                                                                                      \
    } while (0)

hipError_t tptLaunchTrace(const KernelArgs& a, int hs, int fold, bool ldsScene, int blocks, size_t lds, hipStream_t stream)
{
    TPT_DISPATCH(launchOne, a, blocks, lds, stream);
}
int tptTraceOccupancy(int hs, int fold, bool ldsScene, size_t lds)
{
    TPT_DISPATCH(occupancyOne, lds);
}

// Two workgroups per CU is what the path-queue kernel is tuned for; the launch code (chooseKernel) drops the LDS scene --
// and with it the matrix-core filter: 58 -> 41 Gray/s -- as soon as 2 x (LDS + 256-B margin) exceeds 160 KB.  The built-in
// 46-sphere scene with 2 lights must fit: checked at compile time, because a few hundred bytes too many are silent at run time.
namespace {
constexpr size_t kQueueLdsFixedPart = (size_t)TPT_Q_NF4 * TPT_Q_PATHS * 16 + (size_t)Q_COUNT * TPT_Q_P * 2 + ((sizeof(tpt::QueueCtl) + 63) & ~(size_t)63) +
                                      ((sizeof(tpt::FrameConsts) + 15) & ~(size_t)15);
constexpr size_t kDefaultSceneLds = TPT_Q_SPH_FIXED + ((46 * 4 + 15) & ~15) + 46 * 48 + 2 * 32 + (TPT_MATRIX_FILTER ? TPT_MXH_TABLE_DWORDS * 4 + 64 : 0);
static_assert(2 * (kQueueLdsFixedPart + kDefaultSceneLds + 256) <= 160 * 1024, "the default scene no longer fits two path-queue workgroups per CU: shrink TPT_Q_PATHS");
}
size_t tptQueueLdsBytes(const KernelArgs& a, bool ldsScene)
{
    const int nPad = a.scene.nPairs * 2;
    size_t bytes = 0;
    if (ldsScene) bytes += TPT_Q_SPH_FIXED + ((size_t)nPad * 16 <= TPT_Q_SPH_FIXED ? 0 : (size_t)nPad * 16) + (((size_t)nPad * 4 + 15) & ~(size_t)15) + (size_t)a.scene.nSpheres * 48;
    bytes += (size_t)a.scene.nLights * 32;
    bytes += (size_t)TPT_Q_NF4 * (ldsScene ? TPT_Q_PATHS : TPT_Q_PATHS_GROUPED) * 16 + (size_t)Q_COUNT * TPT_Q_P * 2 + ((sizeof(QueueCtl) + 63) & ~(size_t)63) + ((sizeof(FrameConsts) + 15) & ~(size_t)15);
    if (!ldsScene && TPT_GROUP_DEAL) bytes += (size_t)TPT_Q_WAVES * TPT_GROUP_DEAL_WAVE_BYTES;
    if (!ldsScene && a.ldsGroupPairs > 0) bytes += 16 + (size_t)(a.ldsGroupPairs / (TPT_SUPER / 2)) * TPT_GPAIR_LDS_STRIDE * 4; // the groups' bounds for the second filter level (tptQueueGroupPairsInLds), padded stride
#if TPT_MATRIX_FILTER
    if (ldsScene && a.scene.mxR1 >= 0) bytes += TPT_MXH_TABLE_DWORDS * sizeof(uint32_t) + 64;
#endif
    return bytes;
}
template <bool LDS_SCENE, bool BATCH>
static hipError_t launchTraceQueue(const KernelArgs& a, int blocks, size_t lds, hipStream_t stream)
{
    auto k = tptTraceQueueKernel<LDS_SCENE, BATCH>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(TPT_Q_T), lds, stream, a);
    return hipSuccess;
}
hipError_t tptLaunchTraceQueue(const KernelArgs& a, bool ldsScene, int blocks, size_t lds, hipStream_t stream)
{
    const bool batch = a.batchFrames > 1;
    hipError_t e = ldsScene ? (batch ? launchTraceQueue<true, true>(a, blocks, lds, stream) : launchTraceQueue<true, false>(a, blocks, lds, stream))
                            : (batch ? launchTraceQueue<false, true>(a, blocks, lds, stream) : launchTraceQueue<false, false>(a, blocks, lds, stream));
    if (e != hipSuccess) return e;
    return hipGetLastError();
}
#if defined(TPT_TEST_HOOKS)
// hooks build: run-time sizes of the three-stage dealing's entry areas (0, 0, 0: the compiled ones); each between 64 and its compiled size
hipError_t tptSetDealCapacitiesForTest(int ca, int cb, int cs)
{
    unsigned v[3] = {TPT_DEAL_CA, TPT_DEAL_CB, TPT_DEAL_CS};
    if (ca || cb || cs) {
        if (ca < 64 || ca > TPT_DEAL_CA || cb < 64 || cb > TPT_DEAL_CB || cs < 64 || cs > TPT_DEAL_CS) return hipErrorInvalidValue;
        v[0] = (unsigned)ca; v[1] = (unsigned)cb; v[2] = (unsigned)cs;
    }
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dealCaps), v, sizeof(v));
}
#endif
int tptQueuePathsPerBlock() { return TPT_Q_PATHS; } // (the larger of the two pools: what per-workgroup buffers are sized for)
// Pair records of the groups' bounds the grouped instantiation keeps in LDS for a scene of nGroups groups: whole super-groups
// (padded), or 0 when they do not fit the area the smaller path pool leaves (the flat filter runs over all groups then)
int tptQueueGroupPairsInLds(int nGroups, int nSuperPairs)
{
    if (nGroups <= 0 || nSuperPairs <= 0) return 0;
    const int pairs = ((nGroups + TPT_SUPER - 1) / TPT_SUPER) * (TPT_SUPER / 2);
    return (size_t)(pairs / (TPT_SUPER / 2)) * TPT_GPAIR_LDS_STRIDE * 4 + 16 <= (size_t)TPT_Q_GROUP_LDS_BYTES ? pairs : 0;
}
int tptQueueMatrixFilter() { return TPT_MATRIX_FILTER; }
int tptQueueGroupMatrixBounds() { return TPT_MATRIX_FILTER && TPT_GROUP_MATRIX_BOUNDS; }
int tptQueueThreadsPerBlock() { return TPT_Q_T; }

hipError_t tptLaunchDisplay(const float* tile, unsigned char* rgba, int width, int height, hipStream_t stream)
{
    hipLaunchKernelGGL(tptDisplayKernel, dim3((width * height + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const f4*>(tile),
                       reinterpret_cast<uint32_t*>(rgba), width, height);
    return hipGetLastError();
}

hipError_t tptLaunchAssemble(const float* gathered, float* image, int width, int height, int stripeRows, int nRanks, int padRows, hipStream_t stream)
{
    hipLaunchKernelGGL(tptAssembleKernel, dim3((width * height + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const f4*>(gathered),
                       reinterpret_cast<f4*>(image), width, height, stripeRows, nRanks, padRows);
    return hipGetLastError();
}
hipError_t tptLaunchQueueProbe(unsigned long long ticks, hipStream_t stream)
{
    hipLaunchKernelGGL(tptQueueProbeKernel, dim3(1), dim3(64), 0, stream, ticks, static_cast<unsigned*>(nullptr));
    return hipGetLastError();
}
hipError_t tptLaunchChunkOrder(const unsigned* cost, unsigned* snap, unsigned* order, int numChunks, hipStream_t stream)
{
    hipLaunchKernelGGL(tptChunkOrderKernel, dim3(1), dim3(1024), 0, stream, cost, snap, order, numChunks);
    return hipGetLastError();
}

// Workgroups of a blend launch: one per 256 pixels up to 512, grid-stride beyond (every workgroup is one more dispatch that has
// to find a slot on a machine full of persistent trace workgroups; 128-512 measured alike, profiles/r03).
static int tptResolveBlocks(int nPixels)
{
    const int need = (nPixels + 255) / 256;
    return need < 512 ? need : 512;
}
hipError_t tptLaunchResolve(float* tile, const f4* frameColour, int nPixels, float lerpFac, float* mirror,
                            unsigned long long* rayCounter, unsigned long long* counterOut, const unsigned long long* frameRays, hipStream_t stream)
{
    if (nPixels <= 0) return hipSuccess;
    const int blocks = tptResolveBlocks(nPixels);
    if (mirror)
        hipLaunchKernelGGL(tptResolveMirrorKernel, dim3(blocks), dim3(256), 0, stream, tile, frameColour, nPixels, lerpFac,
                           reinterpret_cast<f4*>(mirror), rayCounter, counterOut, frameRays);
    else
        hipLaunchKernelGGL(tptResolveKernel, dim3(blocks), dim3(256), 0, stream, tile, frameColour, nPixels, lerpFac, frameRays, rayCounter);
    return hipGetLastError();
}

hipError_t tptLaunchResolveBatch(float* tile, const f4* frameColour, int nPixels, int planeStride, int nFrames, const tptLerpTable& lerp,
                                 float* mirror, unsigned long long* rayCounter, unsigned long long* counterOut, hipStream_t stream)
{
    if (nPixels <= 0) return hipSuccess;
    hipLaunchKernelGGL(tptResolveBatchKernel, dim3(tptResolveBlocks(nPixels)), dim3(256), 0, stream, tile, frameColour, nPixels, planeStride, nFrames, lerp,
                       reinterpret_cast<f4*>(mirror), rayCounter, counterOut);
    return hipGetLastError();
}

#if defined(TPT_TEST_HOOKS)
hipError_t tptLaunchMathTest(int op, const float* a, const float* b, float* out, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(tptMathTestKernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, a, b, out, n);
    return hipGetLastError();
}
hipError_t tptLaunchMathExhaustive(int op, unsigned lo, unsigned hi, unsigned long long* nBad, unsigned* firstBad, hipStream_t stream)
{
    hipLaunchKernelGGL(tptMathExhaustiveKernel, dim3(8192), dim3(256), 0, stream, op, lo, hi, nBad, firstBad);
    return hipGetLastError();
}
hipError_t tptLaunchMatrixFilterTest(const KernelArgs& a, const float* rays, unsigned long long* outMask, int* outId, float* outT, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(tptMatrixFilterTestKernel, dim3(n / 64), dim3(64), 0, stream, a, rays, outMask, outId, outT, n);
    return hipGetLastError();
}
hipError_t tptLaunchGroupFilterTest(const KernelArgs& a, const float* rays, int n, int nPad, unsigned long long* out4, hipStream_t stream)
{
    hipLaunchKernelGGL(tptGroupFilterTestKernel, dim3(nPad / 64), dim3(64), 0, stream, a, rays, n, out4);
    return hipGetLastError();
}
hipError_t tptLaunchHitTest(const KernelArgs& a, int hs, const float* rays, int* outId, float* outT, int n, hipStream_t stream)
{
    if (hs == HS_SIMPLE)
        hipLaunchKernelGGL(tptHitTestKernel<HS_SIMPLE>, dim3((n + 255) / 256), dim3(256), 0, stream, a, rays, outId, outT, n);
    else
        hipLaunchKernelGGL(tptHitTestKernel<HS_TWO_PHASE_GROUPS>, dim3((n + 255) / 256), dim3(256), 0, stream, a, rays, outId, outT, n);
    return hipGetLastError();
}
#endif // TPT_TEST_HOOKS
