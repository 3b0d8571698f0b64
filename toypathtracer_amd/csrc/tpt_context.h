// tpt_context.h -- the host runtime's state (one context per process, like the reference's statics Test.cpp:13-69, 237) and the
// functions its translation units share.  Not an interface of the library: include/tpt_hip.h is.
//
//   tpt_host.cpp           context, initialisation, scene staging, the setters, UpdateTest, the reference's C++ symbols
//   tpt_host_pipeline.cpp  one frame: plan, buffers, trace launch, ordered blend; tail helpers; tptDrawDevice / tptDrawDeviceBatch
//   tpt_host_draw.cpp      DrawTest on a host backbuffer: look-ahead, row-serial batches, banded copies; display conversion
//   tpt_host_shard.cpp     multi-GPU inside the library: RCCL (dlopen), tptDrawSharded, tptShardedFinish
//   tpt_host_hooks.cpp     unit-test / profiling entry points (include/tpt_test_hooks.h; the second build only)
#pragma once
#include "../../include/tpt_hip.h"
#if defined(TPT_TEST_HOOKS)
#include "../../include/tpt_test_hooks.h" // unit-test / profiling entry points: the second build only (csrc/build.sh)
#endif
#include "../../include/tpt_test_api.h"
#include "tpt_device.h"
#include "tpt_scene.h"
#include "tpt_shard.h"
#include <hip/hip_runtime.h>
#include <thread>
#include <rccl/rccl.h> // types and prototypes only: the library is dlopen()ed when tptCommInit is called
#include <chrono>
#include <dlfcn.h>
#include <map>
#include <stdio.h>
#include <stdlib.h>
#include <string>


#define TPT_DEFAULT_HW_QUEUES "20" /* what tptInitialize exports as GPU_MAX_HW_QUEUES when the host has not (tpt_host.cpp) */
namespace tpth {
using namespace tpt;

// What a trace launch leaves behind for the blend that follows it (now, or -- host path with look-ahead -- later).
const int kMaxBatch = 32; // frames per batched launch (tptDrawDeviceBatch): 6 bits in the path record, 32 lerp factors by value
struct TraceTicket {
    int slot = 0, nPixels = 0;
    bool pipelined = false, valid = false;
    float lerpFac = 0;
    const f4* colour = nullptr;
    int batch = 1;           // frames traced by the launch; their colour planes lie nPixels apart
    tptLerpTable lerp = {};  // batch > 1: each frame's lerp factor
};

struct Context {
    static const int kMaxOverlap = 16;              // frames in flight (trace streams, colour buffers, ...): one hardware queue each
    static const int kMaxSlots = 2 * kMaxOverlap;   // frame slots (colour buffers, events): a frame holds its slot from trace to blend
    static const int kOrderTables = kMaxSlots + 2;  // rotating chunk-order tables: more than frames in flight
    bool inited = false;
    int device = 0, numCUs = 0;
    int traceCUs = 0;      // what a trace launch can occupy (= numCUs)
    std::string deviceName, err;
    hipStream_t ownStream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // host scene state (the reference's statics, Test.cpp:13-69)
    std::vector<SpherePOD> spheres;
    std::vector<MaterialPOD> mats;
    CameraSetup camSetup = defaultCameraSetup();
    CameraPOD cam;
    PackedScene packed;
    bool sceneDirty = true; // host arrays changed since last pack
    bool updated = false;   // tptUpdate ran at least once

    // device scene: a ring of scene sets, so that an animated scene (kFlagAnimate re-packs every frame,
    // Test.cpp:304-308,321-339) is uploaded asynchronously while earlier frames still read the older sets.
    // One device blob + one pinned host staging blob per set, laid out pairs | sph4 | invR | mats | lights.
    // A frame uploads at most one set, a set is reused after kSceneSets uploads, at most kMaxSlots frames
    // are in flight and the upload is stream-ordered behind the resolve of frame f - overlap: no kernel still reads the
    // set that is being overwritten.
    static const int kSceneSets = 2 * kMaxSlots;
    struct SceneSet {
        char* dev = nullptr;
        char* stage = nullptr; // pinned
        size_t cap = 0, bytes = 0;
        size_t offSph4 = 0, offInvR = 0, offMats = 0, offLights = 0;
        size_t offGPairs = 0, offGSph = 0, offGId = 0, offBSph = 0, offBId = 0; // grouped representation (large scenes)
        size_t offSPairs = 0;                                                    // ... and the super-group bounds over it
        int nSuperPairs = 0;
        size_t offAmat = 0; // matrix-core filter table (small scenes)
        int mxR1 = -1;
        size_t offGmat = 0; // the same for the group bounds of a grouped scene
        int gmxTiles = 0;
        int flags = 0;
        int nSpheres = 0, nPairs = 0, nLights = 0;
        int nGroups = 0, nGroupPairs = 0, nBig = 0;
        hipEvent_t evUploaded = nullptr;
        hipStream_t uploadStream = nullptr;
        bool copyEnqueued = false, copyDone = false;
    } sets[kSceneSets];
    int curSet = -1, pendingSet = -1;
    unsigned uploadSeq = 0;

    // run-time versions of the reference's compile-time switches
    int spp = 4;                     // DO_SAMPLES_PER_PIXEL, Config.h:22
    int config = CFG_LIGHT_SAMPLING; // DO_LIGHT_SAMPLING 1, DO_MITSUBA_COMPARE 0, Config.h:24-25
    float animateSmoothing = 0.9f;   // DO_ANIMATE_SMOOTHING, Config.h:23
    int seedMode = SEED_PER_PIXEL;
    int foldMode = FOLD_RECURSIVE;
    int allowGroups = 1; // hitSpheres variant 2 = two-phase, brute force even for large scenes
    int useMatrix = 1;   // phase 1 of HitSpheres on the matrix cores where it applies (hitSpheres variant 3 = VALU filter everywhere)
    int groupMatrix = 0; // grouped scenes: the groups' bounds on the matrix cores (hitSpheres variant 4; opt-in: DESIGN.md 2.2) instead of the two-level VALU filter
    int hs = HS_TWO_PHASE, persist = 3, ldsScene = -1; // persist 3 = path queues (falls back to 1 where they do not apply)
    int stripeRows = 0, numParts = 1, part = 0;
    int gridFill = 0;                           // env TPT_GRID_FILL: % of the resident slots all in-flight launches ask for
    int gridDiv = 0;                            // env TPT_GRID_DIV: launch resident/gridDiv workgroups per frame; 0 = adaptive
    unsigned long long oldestPending = 0;       // adaptive grid: oldest frame whose trace kernel may still be running
    int streamDepth = 1, prevInFlight = -1;     // adaptive grid: deepest pipeline the caller has built / in flight at the previous enqueue
    int framesSinceIdle = 0;                    // adaptive grid: frames enqueued since one found the pipeline empty
    int depthOverride = 0;                      // > 0: frames that share the machine, known to the caller of enqueueTrace (tptDraw)
    int ldsStackLevels = 6;                     // recursive fold, lane-refill kernel: bounce-stack levels kept in LDS

    float* mirror = nullptr;                    // tptSetTileMirror: second destination of the resolve kernel
    unsigned long long* mirrorCounter = nullptr;
    unsigned* dWork = nullptr;
    unsigned long long* dRays = nullptr;    // the counter kernels add to (own or caller-provided)
    unsigned long long* dRaysOwn = nullptr;
    long long lastTotal = 0;

    f4* dStack[kMaxSlots] = {};         // recursive fold: global bounce stacks / spill levels (one per trace stream: the first kMaxOverlap entries)
    size_t stackCap = 0, colourCap = 0, pathCap = 0; // bytes per slot; all reserved slots have the same capacities
    int slotsReserved = 0;              // slots [0, slotsReserved) hold buffers of those capacities
    int smallStreak = 0;                // consecutive launches that needed a quarter of the reserved colour slot or less (reserveSlotBuffers)
    int slotReservations = 0;           // how often the slot buffers were (re-)allocated (tptGetPipelineInfo)
    // cost-ordered chunk distribution (persistent kernel)
    unsigned* dChunkCost = nullptr;
    unsigned* dChunkOrder[kOrderTables] = {};
    unsigned* dChunkSnap[kMaxOverlap] = {}; // per trace stream: cost snapshot of the sort kernel
    int chunkCap = 0, chunkCount = 0; // chunkCount: numChunks the statistics belong to
    int costOrder = 1;                // expensive tiles first (lane-refill kernel)
    hipEvent_t evOrder = nullptr;     // the last sort of an order table (recorded on the stream that ran it)
    hipStream_t orderStream = nullptr;
    bool orderDone = true;
    unsigned long long orderSeq = 0;
    int lastOrderTable = 0;
    f4* dPath[kMaxSlots] = {};          // (unused since the path record moved into LDS; kept for the size bookkeeping)
    float* dFrame = nullptr; // device tile behind the host-pointer DrawTest
    // ---- host-pointer path (tptDraw / DrawTest)
    hipStream_t hostStream2 = nullptr;  // second stream of the banded upload / blend / download (full-duplex PCIe)
    hipEvent_t evBand = nullptr, evBandEnd = nullptr;
    int hostTrust = 0;                  // tptSetHostBufferMode(1): only DrawTest writes the backbuffer -> never re-upload it
    const float* tileSrc = nullptr;     // which host buffer (and size) the device tile g.dFrame currently mirrors
    int tileW = 0, tileH = 0;
    int lookahead = 2;                  // tptSetHostLookahead: frames traced ahead of the caller's next DrawTest
    struct Ahead {                      // a frame traced ahead: what it was traced for, where its result sits
        int frameCount, w, h;
        unsigned flags;
        unsigned long long configKey;   // everything else a trace depends on (see hostConfigKey)
        int raySlot;
        bool used;
    };
    Ahead ahead[4];
    TraceTicket aheadTicket[4];
    // The same in the reference's own seed mode (one RNG stream per row, Test.cpp:280): a frame alone offers `rows` lanes of
    // work, so the frames ahead are traced as ONE batched launch (rows x frames lanes, tptDrawDeviceBatch's kernel path) with a
    // ray counter per frame, and served one by one; [0] is being served, [1] is the batch after it, launched when [0] starts.
    struct RowSerialBatch {
        bool used = false;
        int firstFrame = 0, n = 0, next = 0, w = 0, h = 0;
        unsigned flags = 0;
        unsigned long long key = 0;
        TraceTicket T;
        int counterBase = 0;
    } rsb[2];
    struct HostCaller { // tptDraw: are the calls consecutive frames of one configuration?  (gates the row-serial batches)
        int frame = 0, w = 0, h = 0, streak = 0;
        unsigned flags = 0;
        unsigned long long key = 0;
        // a configuration whose batched launch was refused (frame too large for a batch, not enough device memory): served frame
        // by frame from then on instead of failing (or retrying the reservation) on every call
        int refusedW = 0, refusedH = 0;
        unsigned long long refusedKey = 0;
    } hostCaller;
    unsigned long long* dRaysBatch = nullptr; // [2][kMaxBatch] per-frame ray counters of those two batches
    // Streaming callers of tptDrawDevice / tptDrawSharded with SMALL frames (tiles of a sharded frame, 640x360): a launch cannot
    // be shorter than its longest pixel's sequential samples, so frame by frame such callers are bound by launch latency, not
    // by arithmetic.  When the calls are consecutive frames of one static configuration, the next call's frames are traced in
    // the SAME launch (2-8 frames, tptDrawDeviceBatch's kernel path, a ray counter per frame) and each later call only blends
    // its own plane -- every frame is still delivered, in order, with its own ray count.  A wrong guess costs GPU time only.
    struct StreamBatch {
        bool used = false;
        int firstFrame = 0, n = 0, next = 0, w = 0, h = 0;
        unsigned flags = 0;
        unsigned long long key = 0;
        TraceTicket T;
        int counterBase = 0;
    } sbatch;
    static const int kStreamBatchMax = 8, kStreamRing = 64;
    unsigned long long* dRaysStream = nullptr; // [kStreamRing][kStreamBatchMax]
    unsigned long long streamBatches = 0;       // batches launched (ring index)
    int streamBatch = 1;                        // on by default since round 4; tptSetStreamBatching(0) / env TPT_STREAM_BATCH=0 turn it off
    // tptDrawDevice: is the caller synchronous (the previous frame's blend has completed by the time the next call arrives)
    // and are its calls consecutive frames of one configuration?  Then the next frames are traced ahead for it too.
    struct DeviceCaller {
        int lastSlot = -1, frame = 0, w = 0, h = 0;
        unsigned flags = 0;
        unsigned long long key = 0;
        int syncStreak = 0, seqStreak = 0;
    } devCaller;
    long long aheadHits = 0;            // frames that were found traced ahead when their call arrived (tptGetLookaheadHits)
    unsigned long long* dRaysAhead = nullptr; // [kMaxSlots] per-slot ray counters of frames traced ahead of their call (both synchronous paths)
    unsigned long long configEpoch = 1;       // bumped by every call that changes what a frame looks like

    // ---- multi-GPU inside the library (one process per GPU, RCCL): tptCommInit .. tptDrawSharded
    struct Shard {
        static const int kRing = 4;     // send snapshots: a gather may trail the renderer by this many frames
        void* lib = nullptr;            // librccl, loaded on first use (no link-time dependency: a single-GPU host never needs it)
        ncclComm_t comm = nullptr;
        bool active = false, loopback = false; // loopback: rank 0 of nRanks with a device copy in place of the gather (tptCommInitLoopback)
        int nRanks = 0, rank = 0, stripeRows = 8;
        int w = 0, h = 0, padRows = 0;
        hipStream_t commStream = nullptr;
        float* tile = nullptr;          // this rank's resident accumulation tile [localRows][w] f4
        float* send[kRing] = {};        // snapshots [padRows + 1][w] f4: blended tile + the row carrying the ray counter
        float* gathered = nullptr;      // rank 0: [nRanks][padRows + 1][w] f4
        hipEvent_t evSnap[kRing] = {}, evSent[kRing] = {};
        bool sentRecorded[kRing] = {};
        unsigned long long frames = 0;  // exchanges enqueued (index into the snapshot ring)
        // Exchange interval (tptSetShardExchangeInterval): with small tiles the chain behind a frame -- trace launch, blend + snapshot,
        // gather, de-interleave: four dispatches, each a ~40 us quantum beside a machine full of trace workgroups -- bounds the frame
        // rate, not the arithmetic.  tptDrawSharded then DEFERS such frames: k consecutive frames of one configuration are
        // issued as one tptDrawShardedBatch (one trace launch, one blend, one exchange) when the k-th arrives, when anything about the
        // configuration is about to change, or when the caller waits (tptShardedFinish, tptSynchronize, tptRayCounterRead).
        int exchangeEvery = 0;          // 0 = automatic (1 for tiles of >= 2.4 M samples per frame, 2 / 4 / 8 below), else the host's choice
        int pendCount = 0, pendFirst = 0, pendW = 0, pendH = 0; // frames accepted but not issued yet: [pendFirst, pendFirst + pendCount)
        unsigned pendFlags = 0;
        float pendTime = 0.0f;
        float* lastImage = nullptr;     // the root's image pointer of the most recent call (a deferred batch writes there)
        decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
        decltype(&ncclCommInitRank) CommInitRank = nullptr;
        decltype(&ncclCommDestroy) CommDestroy = nullptr;
        decltype(&ncclGather) Gather = nullptr;
        decltype(&ncclCommCount) CommCount = nullptr;
        decltype(&ncclCommUserRank) CommUserRank = nullptr;
        decltype(&ncclGetErrorString) GetErrorString = nullptr;
    } shard;
    size_t frameCap = 0;

    // frame pipelining: trace kernels of consecutive frames run on alternating internal streams and write
    // their own per-frame colour buffer; the (ordered) resolve kernels run on g.stream
    int overlap = 16;
    hipStream_t traceStream[kMaxOverlap] = {};
    hipEvent_t evTrace[kMaxSlots] = {}, evResolve[kMaxSlots] = {};
    bool resolveRecorded[kMaxSlots] = {};
    f4* dColour[kMaxSlots] = {};
    int hwQueues = 0, overlapCap = kMaxOverlap; // measured at tptInitialize (probeHardwareQueues)
    int slotFactor = 2;                         // colour slots per trace stream (enqueueTrace)
    // tail helpers (tpt_device.h): second grids for the launches still in flight when the caller blocks
    hipEvent_t evPre[kMaxSlots] = {};           // recorded on the slot's stream right before its trace launch: what a helper grid has to wait for
    struct HelperRec {
        KernelArgs a;
        bool ldsScene = false, valid = false, helped = false;
        hipStream_t ts = nullptr;
        int blocks = 0, maxBlocks = 0;
        size_t lds = 0;
    } hrec[kMaxSlots];
    unsigned launchGen = 0;
    int helpersOn = 1;                          // env TPT_TAIL_HELPERS=0: no second grids (the launches still close their counter blocks)
    static const int kHelperPct = 3;            // a helper workgroup joins only while this % of its launch's pool is unclaimed
    static const int kHelperMax = 8;            // launches helped per wait (the newest half of those in flight; sweep: profiles/r05/r05_run2.log)
    long long helperLaunches = 0;
    int hostPace = 1;                           // env TPT_HOST_PACE=0: let the host run ahead of the pipeline (enqueueTrace)
    int shardOverlapCap = kMaxOverlap;          // 8 while the frame is sharded over more than two parts (tptSetRowShard)
    unsigned long long frameSeq = 0;

    // per-launch timing of the trace kernel: hipEvent pairs on the stream each launch goes to
    bool kernelTiming = false;
    std::vector<hipEvent_t> ktStart, ktStop;
    size_t ktUsed = 0;

    std::map<int, int> occCache;
    int lastBlocksPerCU = 0, lastLds = 0, lastGrid = 0;
};

extern Context g;

// Events that order work between the streams of this context (trace -> resolve -> next use of a colour buffer, scene
// upload -> trace, order-table sort -> trace).  Plain events: hipEventDisableSystemFence was measured (no gain: the
// fences are not what bounds small frames) and dropped again -- a dependency between kernels on different streams is
// exactly where the release/acquire of an event matters, and one unexplained mismatch in a full test run was not worth it.
const unsigned kOrderingEvent = hipEventDisableTiming;
const unsigned kTimingEvent = hipEventDefault;

inline int fail(const std::string& what)
{
    g.err = what;
    return -1;
}
inline int hipFail(hipError_t e, const char* what)
{
    g.err = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError(); // clear the runtime's sticky error: the next launch's hipGetLastError() must not report this one again
    return -2;
}
// A request the pipeline declines -- too large for a batch, not enough device memory, frame slots still held by frames traced
// ahead -- as opposed to something that went wrong: callers that can retry with less (the row-serial batches of tptDraw) do so
// on this code only and pass every other error on.
const int kRefused = -4;
inline int refuse(const std::string& what)
{
    g.err = what;
    return kRefused;
}
#define HIPCHK(x)                                   \
    do {                                            \
        hipError_t _e = (x);                        \
        if (_e != hipSuccess) return hipFail(_e, #x); \
    } while (0)

// tpt_host.cpp
int localRows(int h);
int localToGlobal(int ly);
int stageScene();
Context::SceneSet* activeSet();
SceneView deviceView();
int enqueueSceneUpload(hipStream_t ts);
int framesInFlight(int nOverlap);
int uploadBackbuffer(const float* backbuffer, int w, int h);
int requireInit();
int drainPipeline();
int effectiveOverlap();
// tpt_host_pipeline.cpp
int enqueueTrace(int frameCount, int w, int h, unsigned testFlags, unsigned long long* frameRays, TraceTicket& T, int batch = 1, int rayStride = 0);
int enqueueResolve(const TraceTicket& T, float* deviceTile, const unsigned long long* frameRays);
int syncAllStreams();
int launchTailHelpers();
// tpt_host_draw.cpp
int discardLookahead();
int flushShardDeferred(); // tpt_host_shard.cpp: issue the sharded frames tptDrawSharded has accepted but deferred (none: nothing happens)
int takeAhead(TraceTicket& T, int& raySlot);
int traceAhead(int frameCount, int w, int h, unsigned testFlags, unsigned long long key, int want);

} // namespace tpth
