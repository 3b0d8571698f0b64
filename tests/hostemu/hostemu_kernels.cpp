// TEST INFRASTRUCTURE (see hip/hip_runtime.h in this directory): the launch functions of csrc/tpt_device.h restated for the host --
// each one enqueues a closure on the emulated stream that does, pixel by pixel with the product's own lane headers (tpt_trace.h,
// tpt_math.h, tpt_shard.h), what the kernel of csrc/tpt_kernels.hip does with its argument block: which pixel an item is, which
// frame of a batch a chunk belongs to, where the colour goes, which counter the rays go to, how the work counters are re-armed.
// The per-ray arithmetic is the lane state machine / the path-queue class code that tests/lane_emu.cpp already holds against the
// oracle; what THIS file exists for is everything csrc/tpt_host.cpp builds around the kernels.
#include "tpt_device.h"
#include "tpt_shard.h"
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

using namespace tpt;

namespace {
// (mirrors of constants that live in tpt_kernels.hip: path records, rings, control block of the path-queue kernel)
const int kQPaths = 952, kQPathsGrouped = 608, kQRing = 1024, kQClasses = 6, kQThreads = 512, kQWaves = 8;
const size_t kQCtlBytes = 256, kQDealWaveBytes = (256 + 256 + 128) * 4 + 16; // (entry areas of the three-stage dealing)
const size_t kQGroupPairStride = ((TPT_SUPER / 2) * 8 + 4) * 4, kQGroupLdsBytes = 9808; // (144 B per super-group of 8 groups in LDS: TPT_GPAIR_LDS_STRIDE)

bool mapItem(const KernelArgs& a, int idx, int& x, int& ly) // tpt_kernels.hip: mapItem
{
    if (a.fc.seedMode == SEED_ROW_SERIAL) {
        x = 0;
        ly = idx;
        return ly < a.nLocalRows;
    }
    const int tile = idx >> 6, within = idx & 63;
    const int tx = tile % a.tilesX, ty = tile / a.tilesX;
    x = tx * 8 + (within & 7);
    ly = ty * 8 + (within >> 3);
    return x < a.fc.width && ly < a.nLocalRows;
}
int localRowToGlobal(const KernelArgs& a, int ly) { return shardKernelLocalToGlobal(ly, a.stripeRows, a.stripeStride, a.stripeOffset); }

long long g_helperGrids[3] = {0, 0, 0}; // helper grids that found their launch closed / the pool dry / took chunks

struct TraceJob {
    KernelArgs a;
    int hs, fold;
    bool ldsScene, queue;
};

void checkCounters(const KernelArgs& a)
{
    // the kernels rely on the previous launch on this slot having re-armed the counters (and nobody else touching them)
    if (a.work[0] != 0u || a.work[1] != 0u) {
        fprintf(stderr, "hostemu: a trace launch found its work counters at %u / %u: two launches share a counter block, or one started before the previous had finished\n", a.work[0], a.work[1]);
        abort();
    }
}

template <int HS, int FOLD>
void traceLaneRefill(const KernelArgs& a)
{
    const FrameConsts& fc = a.fc;
    const SceneView& sv = a.scene;
    const bool rowSerial = fc.seedMode == SEED_ROW_SERIAL;
    f4 stackMem[TPT_MAX_DEPTH];
    BounceStack stack;
    stack.base = stackMem;
    stack.stride = 1;
    stack.fastLevels = a.ldsStackLevels > TPT_MAX_DEPTH ? TPT_MAX_DEPTH : a.ldsStackLevels;
    stack.spill = stackMem + stack.fastLevels;
    stack.spillStride = 1;
    unsigned long long total = 0;
    for (int c0 = 0; c0 < a.numChunks; ++c0) {
        int c = a.chunkOrder ? (int)a.chunkOrder[c0] : c0;
        int chunkFrame = 0;
        if (a.batchFrames > 1) {
            chunkFrame = c / a.chunksPerFrame;
            c -= chunkFrame * a.chunksPerFrame;
        }
        const int first = c * a.chunkSize, last = std::min(first + a.chunkSize, a.numItems);
        for (int item = first; item < last; ++item) {
            int x, ly;
            if (!mapItem(a, item, x, ly)) continue;
            Lane L;
            L.rays = 0;
            L.active = false;
            laneBeginPixel(L, fc, x, localRowToGlobal(a, ly), chunkFrame * a.framePlane + ly * fc.width + x, true);
            if (a.batchFrames > 1) L.rng = pixelSeed(fc.seedMode, L.x, L.y, fc.frame + chunkFrame);
            for (;;) {
                while (!laneStep<HS, FOLD>(L, sv, fc, stack)) {
                }
                const f3 col = lanePixelColour(L, fc);
                f4 v; v.x = col.x; v.y = col.y; v.z = col.z; v.w = 0.0f;
                a.frameColour[L.pix] = v;
                if (rowSerial && L.x + 1 < fc.width) laneBeginPixel(L, fc, L.x + 1, L.y, L.pix + 1, false);
                else break;
            }
            if (a.chunkCost && L.rays > (uint32_t)(8 * fc.spp)) a.chunkCost[item >> a.chunkShift] += L.rays;
            if (rowSerial && a.batchFrames > 1) a.rayCounter[(size_t)chunkFrame * a.rayCounterStride] += L.rays;
            else total += L.rays;
        }
    }
    a.rayCounter[0] += total;
}

void traceQueueClasses(const KernelArgs& a, bool ldsScene, int chunk0, int chunk1) // chunks [chunk0, chunk1) of the launch's pool
{
    const FrameConsts& fc = a.fc;
    const SceneView& sv = a.scene;
    const bool matrix = ldsScene && sv.mxR1 >= 0;
    auto hit = [&](f3 o, f3 d, float& t) {
        return matrix ? hitSpheres<HS_MATRIX>(sv, o, d, TPT_MIN_T, TPT_MAX_T, t) : hitSpheres<HS_TWO_PHASE_GROUPS>(sv, o, d, TPT_MIN_T, TPT_MAX_T, t);
    };
    const bool fastDiv = (sv.flags & SCENE_LIGHT_R2_DIV_SAFE) != 0;
    f4 level0, spill[TPT_MAX_DEPTH];
    QStack stack;
    stack.l0 = &level0;
    stack.spill = spill;
    stack.stride = 1;
    std::vector<unsigned long long> frameRays((size_t)std::max(a.batchFrames, 1), 0ull);
    for (int c = chunk0; c < chunk1; ++c) {
        const int chunkFrame = a.batchFrames > 1 ? c / a.chunksPerFrame : 0;
        const int cf = c - chunkFrame * a.chunksPerFrame;
        const int first = cf * a.chunkSize, last = std::min(first + a.chunkSize, a.numItems);
        for (int item = first; item < last; ++item) {
            int px, ly;
            if (!mapItem(a, item, px, ly)) continue;
            const int py = localRowToGlobal(a, ly);
            uint32_t rng = pixelSeed(fc.seedMode, px, py, fc.frame + chunkFrame);
            f3 col = mk3(0, 0, 0), ro, rd;
            unsigned long long rays = 0;
            for (int sample = 0; sample < fc.spp; ++sample) {
                qCamera(fc, px, py, rng, ro, rd);
                int depth = 0, recId = -1;
                bool doMatE = true;
                for (;;) {
                    float t;
                    const int hitId = hit(ro, rd, t);
                    ++rays;
                    int cls = -1;
                    if (hitId >= 0 && depth < TPT_MAX_DEPTH) cls = (int)f2u(sv.mats[hitId * 3].w);
                    if (hitId >= 0) ro = ro + rd * t;
                    recId = hitId;
                    bool ended = cls != MAT_LAMBERT && cls != MAT_METAL && cls != MAT_DIELECTRIC;
                    if (cls == MAT_DIELECTRIC) {
                        f3 e;
                        rd = qDielectric(sv, fc, ro, rd, recId, doMatE, rng, e);
                        qStackPush(stack, depth, e, -1);
                        depth++;
                        doMatE = true;
                    } else if (cls == MAT_METAL) {
                        f3 e, nd;
                        if (qMetal(sv, fc, ro, rd, recId, doMatE, rng, e, nd)) {
                            qStackPush(stack, depth, e, recId);
                            depth++;
                            doMatE = true;
                            rd = nd;
                        } else {
                            ended = true;
                        }
                    } else if (cls == MAT_LAMBERT) {
                        QLambert lam;
                        qLambertBegin(sv, ro, rd, recId, rng, lam);
                        const int nShadow = (fc.config & CFG_LIGHT_SAMPLING) ? sv.nLights : 0;
                        for (int j = 0; j < nShadow; ++j) {
                            const f4 l1 = sv.lights[j * 2 + 1];
                            const int lightId = (int)f2u(l1.w);
                            if (lightId == recId) continue;
                            const f3 d2 = qLightRay(sv.lights[j * 2], ro, rng, lam.cosAMax, fastDiv);
                            float ts;
                            const int id = hit(ro, d2, ts);
                            ++rays;
                            if (id == lightId) qLightShade(l1, d2, lam);
                        }
                        qStackPush(stack, depth, qLambertE(sv, recId, doMatE, lam), recId);
                        depth++;
                        doMatE = !(fc.config & CFG_LIGHT_SAMPLING);
                        rd = lam.sdir;
                    }
                    if (ended) {
                        col = col + qFold(sv, qEndTerm(sv, fc, rd, recId), depth, stack);
                        break;
                    }
                }
            }
            const f3 out = col * fc.invSpp;
            f4 v; v.x = out.x; v.y = out.y; v.z = out.z; v.w = 0.0f;
            a.frameColour[(size_t)chunkFrame * a.framePlane + (size_t)ly * fc.width + px] = v;
            frameRays[(size_t)chunkFrame] += rays;
        }
    }
    if (a.batchFrames > 1)
        for (int f = 0; f < a.batchFrames; ++f) a.rayCounter[(size_t)f * a.rayCounterStride] += frameRays[(size_t)f];
    else
        a.rayCounter[0] += frameRays[0];
}

void runTrace(void* p)
{
    const TraceJob& J = *static_cast<const TraceJob*>(p);
    const KernelArgs& a = J.a;
    int firstChunk = 0;
    // The tail helpers (csrc/tpt_device.h).  A launch is one indivisible step here, so a helper grid either finds its
    // launch closed (it ran after it: nothing to do), or runs BEFORE it and takes the first half of what is left of the pool;
    // the launch itself then starts where the counter stands.  The counter block is checked hard: a helper that meets a block
    // in any other state than "re-armed, or part-consumed by helpers of the same launch" would have corrupted a frame.
    if (a.helperBase > 0) {
        if (!J.queue || a.batchFrames > 1) { fprintf(stderr, "hostemu: a helper grid for a launch that takes none\n"); abort(); }
        if ((int)(a.work[3] - a.gen) >= 0) { g_helperGrids[0]++; return; } // closed: its launch has finished, or the block serves a later one
        if (a.work[2] != 0u || a.work[1] != 0u) { fprintf(stderr, "hostemu: helper grid of launch %u met a counter block in use (busy %u, done %u)\n", a.gen, a.work[2], a.work[1]); abort(); }
        const int taken = (int)a.work[0];
        if (taken >= a.numChunks) { g_helperGrids[1]++; return; }
        g_helperGrids[2]++;
        const int upto = taken + (a.numChunks - taken + 1) / 2;
        traceQueueClasses(a, J.ldsScene, taken, upto);
        a.work[0] = (unsigned)upto;
        return;
    }
    if (J.queue && a.gen != 0u) {
        if (a.work[1] != 0u || a.work[2] != 0u || (int)a.work[0] > a.numChunks) { fprintf(stderr, "hostemu: launch %u met its counter block at %u / %u / busy %u\n", a.gen, a.work[0], a.work[1], a.work[2]); abort(); }
        firstChunk = (int)a.work[0]; // (what helper grids of this launch have taken already)
    } else
    checkCounters(a);
    a.work[0] = (unsigned)a.numChunks + 1u; // (what the counters look like while the launch runs)
    a.work[1] = 1u;
    if (J.queue) {
        traceQueueClasses(a, J.ldsScene, firstChunk, a.numChunks);
        if (a.gen != 0u) a.work[3] = a.gen; // the last wave closes the block before it re-arms the counters
    } else {
        const bool groups = !J.ldsScene;
        if (J.hs == HS_SIMPLE) {
            if (J.fold == FOLD_FORWARD) traceLaneRefill<HS_SIMPLE, FOLD_FORWARD>(a);
            else traceLaneRefill<HS_SIMPLE, FOLD_RECURSIVE>(a);
        } else if (groups) {
            if (J.fold == FOLD_FORWARD) traceLaneRefill<HS_TWO_PHASE_GROUPS, FOLD_FORWARD>(a);
            else traceLaneRefill<HS_TWO_PHASE_GROUPS, FOLD_RECURSIVE>(a);
        } else {
            if (J.fold == FOLD_FORWARD) traceLaneRefill<HS_TWO_PHASE, FOLD_FORWARD>(a);
            else traceLaneRefill<HS_TWO_PHASE, FOLD_RECURSIVE>(a);
        }
    }
    a.work[0] = 0u; // the last wave re-arms the counters
    a.work[1] = 0u;
}

struct ResolveJob {
    float* tile;
    const f4* colour;
    int nPixels, planeStride, nFrames;
    float lerpFac;
    tptLerpTable lerp;
    f4* mirror;
    unsigned long long *rayCounter, *counterOut;
    const unsigned long long* frameRays;
    bool batch;
};
void runResolve(void* p)
{
    const ResolveJob& J = *static_cast<const ResolveJob*>(p);
    if (!J.batch) { // tptResolveKernel / tptResolveMirrorKernel
        if (J.mirror) {
            const unsigned long long total = J.frameRays ? (*J.rayCounter += *J.frameRays) : *J.rayCounter;
            if (J.counterOut) *J.counterOut = total;
        } else if (J.frameRays) {
            *J.rayCounter += *J.frameRays;
        }
    } else if (J.counterOut) {
        *J.counterOut = *J.rayCounter;
    }
    for (int i = 0; i < J.nPixels; ++i) {
        f4 t = reinterpret_cast<f4*>(J.tile)[i];
        f3 r = mk3(t.x, t.y, t.z);
        if (J.batch)
            for (int j = 0; j < J.nFrames; ++j) {
                const f4 c = J.colour[(size_t)j * J.planeStride + i];
                r = blendPixel(r, mk3(c.x, c.y, c.z), J.lerp.v[j]);
            }
        else {
            const f4 c = J.colour[i];
            r = blendPixel(r, mk3(c.x, c.y, c.z), J.lerpFac);
        }
        t.x = r.x; t.y = r.y; t.z = r.z;
        reinterpret_cast<f4*>(J.tile)[i] = t;
        if (J.mirror) J.mirror[i] = t;
    }
}

struct AssembleJob { const f4* gathered; f4* image; int width, height, stripeRows, nRanks, padRows; };
void runAssemble(void* p)
{
    const AssembleJob& J = *static_cast<const AssembleJob*>(p);
    for (int i = 0; i < J.width * J.height; ++i) {
        const int gy = i / J.width, x = i - gy * J.width;
        J.image[i] = J.gathered[shardGatheredPixel(x, gy, J.width, J.stripeRows, J.nRanks, J.padRows)];
    }
}
struct DisplayJob { const f4* tile; uint32_t* rgba; int width, height; };
void runDisplay(void* p)
{
    const DisplayJob& J = *static_cast<const DisplayJob*>(p);
    auto to8 = [](float v) -> uint32_t {
        float s = tsqrt(v) * 255.0f;
        s = s < 255.0f ? s : 255.0f;
        return s > 0.0f ? (uint32_t)s : 0u;
    };
    for (int i = 0; i < J.width * J.height; ++i) {
        const int y = i / J.width, x = i - y * J.width;
        const f4 c = J.tile[(J.height - 1 - y) * J.width + x];
        J.rgba[i] = to8(c.x) | (to8(c.y) << 8) | (to8(c.z) << 16) | 0xff000000u;
    }
}
struct OrderJob { const unsigned* cost; unsigned *snap, *order; int n; };
void runOrder(void* p)
{
    const OrderJob& J = *static_cast<const OrderJob*>(p);
    for (int i = 0; i < J.n; ++i) { J.snap[i] = J.cost[i]; J.order[i] = (unsigned)i; }
    std::stable_sort(J.order, J.order + J.n, [&](unsigned x, unsigned y) { return J.snap[x] > J.snap[y]; }); // expensive chunks first
}
void runNothing(void*) {}
} // namespace

size_t tptLdsBytes(const KernelArgs& a, int fold, bool ldsScene) // = tpt_kernels.hip
{
    const int nPad = a.scene.nPairs * 2;
    size_t bytes = 0;
    if (ldsScene) bytes += (size_t)nPad * 16 + (((size_t)nPad * 4 + 15) & ~(size_t)15);
    bytes += (size_t)a.scene.nLights * 32;
    if (ldsScene) bytes += (size_t)a.scene.nSpheres * 48;
    if (fold == FOLD_RECURSIVE) bytes += (size_t)a.ldsStackLevels * TPT_BLOCK * 16;
    return bytes;
}
int tptTraceOccupancy(int, int, bool, size_t lds) { const int byLds = (int)(160 * 1024 / (lds + 256)); return byLds < 1 ? 1 : (byLds > 16 ? 16 : byLds); }
size_t tptQueueLdsBytes(const KernelArgs& a, bool ldsScene) // = tpt_kernels.hip (constants mirrored above)
{
    const int nPad = a.scene.nPairs * 2;
    size_t bytes = 0;
    if (ldsScene) bytes += 1024 + ((size_t)nPad * 16 <= 1024 ? 0 : (size_t)nPad * 16) + (((size_t)nPad * 4 + 15) & ~(size_t)15) + (size_t)a.scene.nSpheres * 48;
    bytes += (size_t)a.scene.nLights * 32;
    bytes += (size_t)4 * (ldsScene ? kQPaths : kQPathsGrouped) * 16 + (size_t)kQClasses * kQRing * 2 + kQCtlBytes + ((sizeof(FrameConsts) + 15) & ~(size_t)15);
    if (!ldsScene) bytes += (size_t)kQWaves * kQDealWaveBytes + (a.ldsGroupPairs > 0 ? 16 + (size_t)(a.ldsGroupPairs / (TPT_SUPER / 2)) * kQGroupPairStride : 0);
    if (ldsScene && a.scene.mxR1 >= 0) bytes += TPT_MXH_TABLE_DWORDS * sizeof(uint32_t) + 64;
    return bytes;
}
int tptQueuePathsPerBlock() { return kQPaths; }
int tptQueueGroupPairsInLds(int nGroups, int nSuperPairs) // = tpt_kernels.hip
{
    if (nGroups <= 0 || nSuperPairs <= 0) return 0;
    const int pairs = ((nGroups + TPT_SUPER - 1) / TPT_SUPER) * (TPT_SUPER / 2);
    return (size_t)(pairs / (TPT_SUPER / 2)) * kQGroupPairStride + 16 <= kQGroupLdsBytes ? pairs : 0;
}
int tptQueueMatrixFilter() { return 1; }
int tptQueueGroupMatrixBounds() { return 1; }
int tptQueueThreadsPerBlock() { return kQThreads; }

hipError_t tptLaunchTrace(const KernelArgs& a, int hs, int fold, bool ldsScene, int blocks, size_t, hipStream_t stream)
{
    if (blocks < 1 || (unsigned)(blocks * (TPT_BLOCK / 64)) != a.totalWaves) return hipErrorInvalidValue;
    TraceJob J; J.a = a; J.hs = hs; J.fold = fold; J.ldsScene = ldsScene; J.queue = false;
    hostemuEnqueue(stream, runTrace, &J, sizeof(J));
    return hipSuccess;
}
hipError_t tptLaunchTraceQueue(const KernelArgs& a, bool ldsScene, int blocks, size_t lds, hipStream_t stream)
{
    if (blocks < 1 || lds > 160 * 1024) return hipErrorInvalidValue;
    TraceJob J; J.a = a; J.hs = 0; J.fold = FOLD_RECURSIVE; J.ldsScene = ldsScene; J.queue = true;
    hostemuEnqueue(stream, runTrace, &J, sizeof(J));
    return hipSuccess;
}
hipError_t tptLaunchDisplay(const float* tile, unsigned char* rgba, int width, int height, hipStream_t stream)
{
    DisplayJob J = {reinterpret_cast<const f4*>(tile), reinterpret_cast<uint32_t*>(rgba), width, height};
    hostemuEnqueue(stream, runDisplay, &J, sizeof(J));
    return hipSuccess;
}
hipError_t tptLaunchAssemble(const float* gathered, float* image, int width, int height, int stripeRows, int nRanks, int padRows, hipStream_t stream)
{
    AssembleJob J = {reinterpret_cast<const f4*>(gathered), reinterpret_cast<f4*>(image), width, height, stripeRows, nRanks, padRows};
    hostemuEnqueue(stream, runAssemble, &J, sizeof(J));
    return hipSuccess;
}
hipError_t tptLaunchQueueProbe(unsigned long long, hipStream_t stream)
{
    int dummy = 0;
    hostemuEnqueue(stream, runNothing, &dummy, sizeof(dummy));
    return hipSuccess;
}
hipError_t tptLaunchChunkOrder(const unsigned* cost, unsigned* snap, unsigned* order, int numChunks, hipStream_t stream)
{
    OrderJob J = {cost, snap, order, numChunks};
    hostemuEnqueue(stream, runOrder, &J, sizeof(J));
    return hipSuccess;
}
hipError_t tptLaunchResolve(float* tile, const f4* frameColour, int nPixels, float lerpFac, float* mirror, unsigned long long* rayCounter,
                            unsigned long long* counterOut, const unsigned long long* frameRays, hipStream_t stream)
{
    if (nPixels <= 0) return hipSuccess;
    ResolveJob J;
    memset(&J, 0, sizeof(J));
    J.tile = tile; J.colour = frameColour; J.nPixels = nPixels; J.lerpFac = lerpFac; J.mirror = reinterpret_cast<f4*>(mirror);
    J.rayCounter = rayCounter; J.counterOut = counterOut; J.frameRays = frameRays; J.batch = false;
    hostemuEnqueue(stream, runResolve, &J, sizeof(J));
    return hipSuccess;
}
hipError_t tptLaunchResolveBatch(float* tile, const f4* frameColour, int nPixels, int planeStride, int nFrames, const tptLerpTable& lerp,
                                 float* mirror, unsigned long long* rayCounter, unsigned long long* counterOut, hipStream_t stream)
{
    if (nPixels <= 0) return hipSuccess;
    ResolveJob J;
    memset(&J, 0, sizeof(J));
    J.tile = tile; J.colour = frameColour; J.nPixels = nPixels; J.planeStride = planeStride; J.nFrames = nFrames; J.lerp = lerp;
    J.mirror = reinterpret_cast<f4*>(mirror); J.rayCounter = rayCounter; J.counterOut = counterOut; J.batch = true;
    hostemuEnqueue(stream, runResolve, &J, sizeof(J));
    return hipSuccess;
}
int tptReadStats(unsigned long long*) { return -1; }
int tptResetStats() { return -1; }
extern "C" void hostemu_helper_stats(long long* out3)
{
    for (int i = 0; i < 3; ++i) out3[i] = g_helperGrids[i];
}
