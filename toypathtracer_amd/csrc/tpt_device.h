// tpt_device.h -- kernel argument block shared by tpt_kernels.hip (device) and tpt_host.cpp (host).
#pragma once
#include "tpt_trace.h"

#ifndef TPT_BLOCK
#define TPT_BLOCK 64         // threads per workgroup: one wave, so a finished wave frees its LDS/VGPRs at once
#endif
#define TPT_CHUNK_PIXELS 256 // pixels a persistent wave pulls per atomic (4 tiles of 8x8)
// Tail helpers: when the caller BLOCKS (tptSynchronize, tptTimerEnd, tptShardedFinish) while path-queue launches are still in flight,
// the newest of them get a second grid of workgroups that takes chunks from the same pool -- the last launches of a burst otherwise
// finish on a half-empty machine (the anatomy of the driver's command: 20 frames 47.5 -> 50 k Mray/s).  The launch's counter
// block (KernelArgs::work) carries the hand-shake: [2] helper workgroups registered, [3] serial of the last launch that CLOSED on this
// block.  A helper registers, then looks: closed (or a later launch's block) -> it leaves without touching anything; the launch's
// last wave closes, then waits for the registered helpers before it re-arms the counters and lets the kernel end -- so "launch
// complete" still means "frame complete" for the ordered blend behind it.  Model-checked (tests/test_helper_handshake_model.py), run
// under the emulated runtime's schedules (tests/test_host_logic.py); env TPT_TAIL_HELPERS=0 switches the second grids off.

namespace tpt {

struct KernelArgs {
    SceneView scene;
    FrameConsts fc;
    f4* frameColour;   // device, [nLocalRows][width] f4: THIS frame's colour per pixel (xyz; w unused), written once
                       // per pixel by the trace kernel and blended into the accumulation tile by tptResolveKernel
    // row sharding: local row ly <-> image row (ly / stripeRows) * stripeStride + stripeOffset + ly % stripeRows
    int nLocalRows, stripeRows, stripeStride, stripeOffset;
    int tilesX;     // 8x8 tiles per row of tiles
    int numItems;   // pixels incl. tile padding (PER_PIXEL) or rows (ROW_SERIAL)
    int numChunks, chunkSize;
    // Batched launch (path-queue kernel, tptDrawDeviceBatch): the launch traces batchFrames consecutive frames of the same
    // scene and camera -- chunk c belongs to frame fc.frame + c / chunksPerFrame -- and writes their colour planes one
    // after the other into frameColour (framePlane pixels apart).  1 = a single frame (chunksPerFrame == numChunks).
    int batchFrames, chunksPerFrame, framePlane;
    unsigned totalWaves;
    int laneCap;    // lane-refill kernel: lanes of a wave that take work items (64; fewer for row-serial seeds, whose items are whole rows)
    int noRefill;   // lane-refill kernel: 1 = a wave takes 64 pixels and every lane keeps ITS pixel until all 64 are done (the north_star's one-thread-per-pixel shape, tptSetKernelVariant persistent 0: an A/B instantiation); 0 = idle lanes are re-filled at once
    // FOLD_RECURSIVE bounce stack: the first ldsStackLevels levels live in LDS (per thread), deeper ones in
    // stackBuf [TPT_MAX_DEPTH - ldsStackLevels][stackStride] (global, one column per thread of the launch).
    f4* stackBuf;
    int stackStride;
    int ldsStackLevels;
    f4* pathBuf;                     // path-queue kernel: cold path state [workgroups][paths][4] f4 (global, L2-resident)
    // cost-ordered work distribution (persistent kernel): chunk c of the queue is chunkOrder[c]; every finished pixel
    // adds its ray count to chunkCost[chunk] (running sum over frames); tptChunkOrderKernel re-sorts, expensive first
    const unsigned* chunkOrder;      // may be null: image order
    unsigned* chunkCost;             // may be null
    int chunkShift;                  // log2(chunkSize)
    unsigned* work;                  // [0] next chunk, [1] finished waves (persistent variants)
    unsigned long long* rayCounter;  // monotonic total of rays traced by this context
    int rayCounterStride;            // batched row-serial launch: frame j of the batch counts into rayCounter[j * stride] (0: one counter)
    unsigned gen;                    // tail helpers: serial of this launch (1, 2, ...; 0: takes no helpers)
    int helperBase;                  // 0: the launch itself; > 0: its helper grid, whose workgroup b plays workgroup helperBase + b (stack columns)
    int helperPct;                   // a helper workgroup joins only while at least this % of the pool is unclaimed
    int ldsGroupPairs;               // grouped scenes, path-queue kernel, the two-level bounds filter: > 0 pair records of the groups' bounds staged in
                                     // LDS (whole super-groups), 0 = read them from global memory (too many for the LDS area), < 0 = flat filter over all groups
};

} // namespace tpt

size_t tptLdsBytes(const tpt::KernelArgs& a, int fold, bool ldsScene); // uses a.ldsStackLevels
hipError_t tptLaunchTrace(const tpt::KernelArgs& a, int hs, int fold, bool ldsScene, int blocks, size_t lds, hipStream_t stream);
int tptTraceOccupancy(int hs, int fold, bool ldsScene, size_t lds);
size_t tptQueueLdsBytes(const tpt::KernelArgs& a, bool ldsScene);
hipError_t tptLaunchTraceQueue(const tpt::KernelArgs& a, bool ldsScene, int blocks, size_t lds, hipStream_t stream);
int tptQueuePathsPerBlock();
int tptQueueGroupPairsInLds(int nGroups, int nSuperPairs);
int tptQueueMatrixFilter();
int tptQueueGroupMatrixBounds(); // 1: this build carries the groups' bounds on the matrix cores (hooks build only)
int tptQueueThreadsPerBlock();
hipError_t tptLaunchDisplay(const float* tile, unsigned char* rgba, int width, int height, hipStream_t stream);
hipError_t tptLaunchAssemble(const float* gathered, float* image, int width, int height, int stripeRows, int nRanks, int padRows, hipStream_t stream);
hipError_t tptLaunchQueueProbe(unsigned long long ticks, hipStream_t stream);
hipError_t tptLaunchChunkOrder(const unsigned* cost, unsigned* snap, unsigned* order, int numChunks, hipStream_t stream);
struct tptLerpTable { float v[32]; }; // lerp factor of each frame of a batch (Test.cpp:272-276), by value in the kernel arguments
hipError_t tptLaunchResolveBatch(float* tile, const tpt::f4* frameColour, int nPixels, int planeStride, int nFrames, const tptLerpTable& lerp,
                                 float* mirror, unsigned long long* rayCounter, unsigned long long* counterOut, hipStream_t stream);
hipError_t tptLaunchResolve(float* tile, const tpt::f4* frameColour, int nPixels, float lerpFac, float* mirror,
                            unsigned long long* rayCounter, unsigned long long* counterOut, const unsigned long long* frameRays, hipStream_t stream);
hipError_t tptLaunchMathTest(int op, const float* a, const float* b, float* out, int n, hipStream_t stream);
hipError_t tptLaunchMathExhaustive(int op, unsigned lo, unsigned hi, unsigned long long* nBad, unsigned* firstBad, hipStream_t stream);
hipError_t tptLaunchMatrixFilterTest(const tpt::KernelArgs& a, const float* rays, unsigned long long* outMask, int* outId, float* outT, int n, hipStream_t stream);
hipError_t tptSetDealCapacitiesForTest(int ca, int cb, int cs); // (hooks build)
hipError_t tptLaunchGroupFilterTest(const tpt::KernelArgs& a, const float* rays, int n, int nPad, unsigned long long* out4, hipStream_t stream);
hipError_t tptLaunchHitTest(const tpt::KernelArgs& a, int hs, const float* rays, int* outId, float* outT, int n, hipStream_t stream);
int tptReadStats(unsigned long long* out64);  // profiling build (-DTPT_STATS) only, else -1
int tptResetStats();
