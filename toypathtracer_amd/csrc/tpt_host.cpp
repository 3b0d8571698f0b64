// tpt_host.cpp -- host runtime + C ABI (include/tpt_hip.h) + the reference's C++ Test API
// (include/tpt_test_api.h == Cpp/Source/Test.h:10-17) for the MI355X path tracer.
//
// Replaces, on the host side: InitializeTest/ShutdownTest (Test.cpp:240-253; the enkiTS scheduler is
// replaced by a HIP stream), UpdateTest (Test.cpp:302-342; scene prep stays on the host, results are
// uploaded to HBM when dirty), DrawTest (Test.cpp:344-367; the row fan-out becomes one kernel
// launch), GetObjectCount/GetSceneDesc (Test.cpp:369-384).
//
// There is deliberately NO CPU rendering path in this library: if HIP is unavailable every entry
// point fails loudly.
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

using namespace tpt;

namespace {

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
        unsigned long long frames = 0;
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

Context g;

// Events that order work between the streams of this context (trace -> resolve -> next use of a colour buffer, scene
// upload -> trace, order-table sort -> trace).  Plain events: hipEventDisableSystemFence was measured (no gain: the
// fences are not what bounds small frames) and dropped again -- a dependency between kernels on different streams is
// exactly where the release/acquire of an event matters, and one unexplained mismatch in a full test run was not worth it.
const unsigned kOrderingEvent = hipEventDisableTiming;
const unsigned kTimingEvent = hipEventDefault;

int fail(const std::string& what)
{
    g.err = what;
    return -1;
}
int hipFail(hipError_t e, const char* what)
{
    g.err = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError(); // clear the runtime's sticky error: the next launch's hipGetLastError() must not report this one again
    return -2;
}
// A request the pipeline declines -- too large for a batch, not enough device memory, frame slots still held by frames traced
// ahead -- as opposed to something that went wrong: callers that can retry with less (the row-serial batches of tptDraw) do so
// on this code only and pass every other error on.
const int kRefused = -4;
int refuse(const std::string& what)
{
    g.err = what;
    return kRefused;
}
#define HIPCHK(x)                                   \
    do {                                            \
        hipError_t _e = (x);                        \
        if (_e != hipSuccess) return hipFail(_e, #x); \
    } while (0)

int localRows(int h) { return shardLocalRows(h, g.stripeRows, g.numParts, g.part); }          // (tpt_shard.h)
int localToGlobal(int ly) { return shardLocalToGlobal(ly, g.stripeRows, g.numParts, g.part); }

template <class T>
int ensureDev(T*& p, int& cap, int need)
{
    if (need <= cap && p) return 0;
    if (p) HIPCHK(hipFree(p));
    p = nullptr;
    int n = need < 64 ? 64 : need;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * (size_t)n));
    cap = n;
    return 0;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// UpdateTest's scene half: pack the host scene into the pinned staging blob of a fresh scene set.  Nothing is
// enqueued here; the draw that follows copies the blob on its own trace stream (enqueueSceneUpload).
int stageScene()
{
    packScene(g.spheres, g.mats, g.packed);
    const PackedScene& P = g.packed;
    if (g.pendingSet < 0) g.pendingSet = (int)(g.uploadSeq++ % Context::kSceneSets);
    Context::SceneSet& S = g.sets[g.pendingSet];
    const size_t bPairs = P.pairs.size() * sizeof(float), bSph4 = P.sph4.size() * sizeof(f4), bInvR = P.invR.size() * sizeof(float),
                 bMats = P.mats.size() * sizeof(f4), bLights = P.lights.size() * sizeof(f4);
    const bool grouped = g.allowGroups && P.nGroups > 0;
    const size_t bGPairs = grouped ? P.gpairs.size() * sizeof(float) : 0, bGSph = grouped ? P.gsph.size() * sizeof(f4) : 0,
                 bGId = grouped ? P.gid.size() * sizeof(int) : 0, bBSph = grouped ? P.bsph.size() * sizeof(f4) : 0,
                 bBId = grouped ? P.bid.size() * sizeof(int) : 0;
    const size_t offSph4 = align256(bPairs), offInvR = offSph4 + align256(bSph4), offMats = offInvR + align256(bInvR),
                 offLights = offMats + align256(bMats), offGPairs = offLights + align256(bLights + 32),
                 offGSph = offGPairs + align256(bGPairs), offGId = offGSph + align256(bGSph), offBSph = offGId + align256(bGId),
                 offBId = offBSph + align256(bBSph), offAmat = offBId + align256(bBId + 32);
    const size_t bAmat = (g.useMatrix && tptQueueMatrixFilter() && P.mxR1 >= 0) ? P.amatH.size() * sizeof(uint32_t) : 0;
    const size_t offGmat = offAmat + align256(bAmat + 32);
    const size_t bGmat = (grouped && g.useMatrix && tptQueueMatrixFilter() && P.gmxTiles > 0) ? P.gmatH.size() * sizeof(uint32_t) : 0;
    const size_t total = offGmat + align256(bGmat + 32);
    if (!S.evUploaded) HIPCHK(hipEventCreateWithFlags(&S.evUploaded, kOrderingEvent));
    // the previous copy out of this staging blob (kSceneSets uploads ago) must have left the host before we overwrite it:
    // only ever waits when the host has run more than 16 animated frames ahead of the GPU
    if (S.copyEnqueued && !S.copyDone) HIPCHK(hipEventSynchronize(S.evUploaded));
    if (total > S.cap) {
        // Grow EVERY set of the ring at once (one drain now instead of one per first use of a set: an animated scene would
        // otherwise synchronise the device on each of its first kSceneSets frames).  The set in use keeps its contents.
        HIPCHK(hipDeviceSynchronize());
        const size_t cap = total < 4096 ? 4096 : total + total / 4;
        const bool growAll = cap <= (8u << 20); // (a huge scene is rarely animated: its sets grow as they are first used)
        for (int k = 0; k < Context::kSceneSets; ++k) {
            Context::SceneSet& Q = g.sets[k];
            if (Q.cap >= cap || (!growAll && &Q != &S)) continue;
            char *dev = nullptr, *stage = nullptr;
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&dev), cap));
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&stage), cap, hipHostMallocDefault));
            if (k == g.curSet && Q.dev && Q.bytes) { // frames still to be enqueued may read it without a new upload
                HIPCHK(hipMemcpy(dev, Q.dev, Q.bytes, hipMemcpyDeviceToDevice));
                memcpy(stage, Q.stage, Q.bytes);
            }
            if (Q.dev) HIPCHK(hipFree(Q.dev));
            if (Q.stage) HIPCHK(hipHostFree(Q.stage));
            Q.dev = dev; Q.stage = stage; Q.cap = cap;
            Q.copyEnqueued = false; Q.copyDone = true; // (everything has completed: synchronised above)
        }
    }
    memcpy(S.stage, P.pairs.data(), bPairs);
    memcpy(S.stage + offSph4, P.sph4.data(), bSph4);
    memcpy(S.stage + offInvR, P.invR.data(), bInvR);
    memcpy(S.stage + offMats, P.mats.data(), bMats);
    if (bLights) memcpy(S.stage + offLights, P.lights.data(), bLights);
    if (grouped) {
        memcpy(S.stage + offGPairs, P.gpairs.data(), bGPairs);
        memcpy(S.stage + offGSph, P.gsph.data(), bGSph);
        memcpy(S.stage + offGId, P.gid.data(), bGId);
        if (bBSph) memcpy(S.stage + offBSph, P.bsph.data(), bBSph);
        if (bBId) memcpy(S.stage + offBId, P.bid.data(), bBId);
    }
    if (bAmat) memcpy(S.stage + offAmat, P.amatH.data(), bAmat);
    if (bGmat) memcpy(S.stage + offGmat, P.gmatH.data(), bGmat);
    S.bytes = offGmat + bGmat;
    S.offAmat = offAmat;
    S.mxR1 = bAmat ? P.mxR1 : -1;
    S.offGmat = offGmat;
    S.gmxTiles = bGmat ? P.gmxTiles : 0;
    S.flags = P.flags;
    S.offSph4 = offSph4; S.offInvR = offInvR; S.offMats = offMats; S.offLights = offLights;
    S.offGPairs = offGPairs; S.offGSph = offGSph; S.offGId = offGId; S.offBSph = offBSph; S.offBId = offBId;
    S.nSpheres = P.nSpheres; S.nPairs = P.nPairs; S.nLights = P.nLights;
    S.nGroups = grouped ? P.nGroups : 0; S.nGroupPairs = grouped ? P.nGroupPairs : 0; S.nBig = grouped ? P.nBig : 0;
    S.copyEnqueued = false; S.copyDone = false; S.uploadStream = nullptr;
    g.sceneDirty = false;
    return 0;
}

// the set the next launch reads (the staged one if an upload is pending)
Context::SceneSet* activeSet()
{
    const int k = g.pendingSet >= 0 ? g.pendingSet : g.curSet;
    return k >= 0 ? &g.sets[k] : nullptr;
}

SceneView deviceView()
{
    SceneView sv;
    memset(&sv, 0, sizeof(sv));
    Context::SceneSet* S = activeSet();
    if (!S) return sv;
    sv.pairs = reinterpret_cast<const float*>(S->dev);
    sv.sph4 = reinterpret_cast<const f4*>(S->dev + S->offSph4);
    sv.invR = reinterpret_cast<const float*>(S->dev + S->offInvR);
    sv.mats = reinterpret_cast<const f4*>(S->dev + S->offMats);
    sv.lights = reinterpret_cast<const f4*>(S->dev + S->offLights);
    sv.nSpheres = S->nSpheres;
    sv.nPairs = S->nPairs;
    sv.nLights = S->nLights;
    sv.gpairs = reinterpret_cast<const float*>(S->dev + S->offGPairs);
    sv.gsph = reinterpret_cast<const f4*>(S->dev + S->offGSph);
    sv.gid = reinterpret_cast<const int*>(S->dev + S->offGId);
    sv.bsph = reinterpret_cast<const f4*>(S->dev + S->offBSph);
    sv.bid = reinterpret_cast<const int*>(S->dev + S->offBId);
    sv.nGroups = S->nGroups;
    sv.nGroupPairs = S->nGroupPairs;
    sv.nBig = S->nBig;
    sv.amatH = reinterpret_cast<const uint32_t*>(S->dev + S->offAmat);
    sv.mxR1 = S->mxR1;
    sv.gmatH = reinterpret_cast<const uint32_t*>(S->dev + S->offGmat);
    sv.gmxTiles = S->gmxTiles;
    sv.flags = S->flags;
    return sv;
}

// Make the active scene set visible to work enqueued on `ts` from here on: copy a pending set on `ts` itself, or
// make `ts` wait for the copy another stream carries.
int enqueueSceneUpload(hipStream_t ts)
{
    if (g.pendingSet >= 0) {
        Context::SceneSet& S = g.sets[g.pendingSet];
        HIPCHK(hipMemcpyAsync(S.dev, S.stage, S.bytes, hipMemcpyHostToDevice, ts));
        HIPCHK(hipEventRecord(S.evUploaded, ts));
        S.copyEnqueued = true; S.copyDone = false; S.uploadStream = ts;
        g.curSet = g.pendingSet;
        g.pendingSet = -1;
        return 0;
    }
    if (g.curSet < 0) return fail("tpt: no scene uploaded (call tptUpdate first)");
    Context::SceneSet& S = g.sets[g.curSet];
    // (always the stream wait, never an event QUERY as a shortcut: a query on a re-recorded event has been seen to answer
    //  "done" before the new record's work was -- tptDraw's banded path caught it red-handed, and a trace kernel that reads a
    //  scene set before its upload has landed is the kind of once-in-a-thousand mismatch round 1 could not explain)
    if (S.uploadStream != ts) HIPCHK(hipStreamWaitEvent(ts, S.evUploaded, 0));
    return 0;
}

// Number of earlier frames whose trace kernel has not finished yet (their events complete in order: amortised one
// hipEventQuery per frame).
int framesInFlight(int nOverlap) // nOverlap: frame slots in use
{
    if (nOverlap <= 1) return 0;
    if (g.oldestPending + (unsigned long long)nOverlap < g.frameSeq) g.oldestPending = g.frameSeq - (unsigned long long)nOverlap;
    // (called after frameSeq was advanced for the frame being enqueued: that frame itself does not count)
    const unsigned long long cur = g.frameSeq - 1;
    while (g.oldestPending < cur) {
        const int s = (int)(g.oldestPending % (unsigned long long)nOverlap);
        if (hipEventQuery(g.evTrace[s]) != hipSuccess) break;
        g.oldestPending++;
    }
    (void)hipGetLastError(); // hipErrorNotReady is not an error
    return (int)(cur - g.oldestPending);
}

// tptDraw's upload of the caller's backbuffer (previous frame's RGB, caller-owned alpha) into g.dFrame, this rank's rows
int uploadBackbuffer(const float* backbuffer, int w, int h)
{
    const int rows = localRows(h);
    const size_t rowBytes = (size_t)w * 4 * sizeof(float);
    const bool sharded = g.numParts > 1 && g.stripeRows > 0;
    if (!sharded) {
        HIPCHK(hipMemcpyAsync(g.dFrame, backbuffer, rowBytes * rows, hipMemcpyHostToDevice, g.stream));
        return 0;
    }
    for (int ly = 0; ly < rows; ly += g.stripeRows) {
        int n = rows - ly < g.stripeRows ? rows - ly : g.stripeRows;
        HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(g.dFrame) + rowBytes * ly,
                              reinterpret_cast<const char*>(backbuffer) + rowBytes * localToGlobal(ly), rowBytes * n,
                              hipMemcpyHostToDevice, g.stream));
    }
    return 0;
}

int discardLookahead();
struct TraceTicket;
int takeAhead(TraceTicket& T, int& raySlot);
int traceAhead(int frameCount, int w, int h, unsigned testFlags, unsigned long long key, int want);

int requireInit()
{
    if (!g.inited) return fail("tpt: not initialised (call tptInitialize / InitializeTest first)");
    return 0;
}

// How many of the trace streams does the runtime really run side by side?  ROCm maps streams onto GPU_MAX_HW_QUEUES
// hardware queues (default 4) and reads that variable when the runtime starts -- a host that touched HIP before loading
// this library keeps its 4, whatever tptInitialize puts into the environment, and streams that share a queue serialise
// (3 frames in flight on 4 queues were SLOWER than 2).  So measure instead of assuming: one wave spinning 2 ms on each
// trace stream; 16 concurrent ones take ~2 ms, 4 queues take 4 rounds.  The frame pipeline is then clamped to what the
// queues can carry (tptGetPipelineInfo reports both numbers).
int probeHardwareQueues()
{
    auto clampCap = [] {
        // the ordered resolve chain, the scene uploads and the caller's own streams need queues too: with fewer than
        // ~3 queues per 2 trace streams to spare, two frames in flight is the best there is
        g.overlapCap = g.hwQueues >= Context::kMaxOverlap ? Context::kMaxOverlap : (g.hwQueues >= 8 ? g.hwQueues - 3 : 2);
        if (const char* e = getenv("TPT_OVERLAP_CAP")) g.overlapCap = atoi(e) < 1 ? 1 : (atoi(e) > Context::kMaxOverlap ? Context::kMaxOverlap : atoi(e));
    };
    // env TPT_HW_QUEUES=n: the host knows how many hardware queues this process has (GPU_MAX_HW_QUEUES as the runtime read it):
    // no probe (it costs 2-4 ms at start-up and measures a busy, shared GPU pessimistically)
    if (const char* eq = getenv("TPT_HW_QUEUES")) {
        const int q = atoi(eq);
        g.hwQueues = q < 1 ? 1 : (q > Context::kMaxOverlap ? Context::kMaxOverlap : q);
        clampCap();
        return 0;
    }
    // Long enough that enqueueing the 16 probes (~20 us each) does not matter: 2 ms of the 100 MHz wall clock.
    const double spinUs = 2000.0;
    const unsigned long long ticks = (unsigned long long)(spinUs * 100.0);
    for (int rep = 0; rep < 2; ++rep) { // first round: warm-up (code object load, queue creation)
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < Context::kMaxOverlap; ++k) HIPCHK(tptLaunchQueueProbe(rep ? ticks : 100ull, g.traceStream[k]));
        for (int k = 0; k < Context::kMaxOverlap; ++k) HIPCHK(hipStreamSynchronize(g.traceStream[k]));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (rep) {
            const double rounds = us / spinUs; // 1.0x with a queue per stream, 4 with the default 4 queues
            int q = (int)((double)Context::kMaxOverlap / (rounds > 1.0 ? rounds : 1.0) + 0.5);
            if (rounds < 1.5) q = Context::kMaxOverlap; // all side by side
            g.hwQueues = q < 1 ? 1 : q;
        }
    }
    clampCap();
    return 0;
}

// The trace streams: non-blocking (their kernels write only the library's own buffers; nothing the caller does on the default
// stream may serialise them).  (Round 4 tried CU-masked trace streams that leave a few CUs to the blend chain,
// hipExtStreamCreateWithCUMask: the blend did not get faster and the trace rate fell by 5-9 % -- profiles/r04/r04_run1.log,
// tools/probes/cumask_probe.hip; removed in round 5.)
int createTraceStreams()
{
    g.traceCUs = g.numCUs;
    for (int k = 0; k < Context::kMaxOverlap; ++k) HIPCHK(hipStreamCreateWithFlags(&g.traceStream[k], hipStreamNonBlocking));
    return 0;
}

} // namespace

extern "C" {

const char* tptGetLastError(void) { return g.err.c_str(); }
const char* tptGetDeviceName(void) { return g.deviceName.c_str(); }

int tptInitialize(void)
{
    if (g.inited) return 0;
    // Frame pipelining wants one hardware queue per in-flight trace kernel; the ROCm runtime exposes 4 by default and
    // maps further streams onto them round-robin (3 streams then run slower than 2).  Only effective if the HIP
    // runtime has not been initialised yet by the host application; harmless otherwise.
    setenv("GPU_MAX_HW_QUEUES", "32", 0);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail("tptInitialize: no HIP device visible (this library has no CPU fallback)");
    int dev = 0;
    const char* env = getenv("TPT_DEVICE");
    if (!env) env = getenv("LOCAL_RANK");
    if (env) dev = atoi(env);
    if (dev < 0) return fail("tptInitialize: negative device index in TPT_DEVICE / LOCAL_RANK");
    if (dev >= count) dev = dev % count;
    HIPCHK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    g.device = dev;
    g.numCUs = prop.multiProcessorCount;
    g.deviceName = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    // The context's own stream -- the ordered chain that touches the CALLER's device buffers (blend into the tile, mirror and
    // counter snapshot, display conversion) -- is a BLOCKING stream: HIP then orders it against the legacy default stream in
    // both directions, like any library that "works on the default stream".  A host that fills its tile with hipMemset / a
    // torch op on the default stream and calls tptDrawDevice straight away is ordered (the blend waits for the fill), and so is
    // a host that reads the tile from the default stream after the call.  (Round 3 had it non-blocking: a fill still queued
    // behind earlier GPU work landed AFTER the library's writes.)  The trace streams stay non-blocking: their kernels write
    // only the library's own colour slots, and nothing the caller does on the default stream may serialise them.  A caller
    // that hands over its own stream (tptSetStream) gets everything enqueued there instead.
    HIPCHK(hipStreamCreateWithFlags(&g.ownStream, hipStreamDefault));
    g.stream = g.ownStream;
    HIPCHK(hipEventCreateWithFlags(&g.evOrder, kOrderingEvent));
    g.orderDone = true;
    HIPCHK(hipEventCreate(&g.ev0));
    HIPCHK(hipEventCreate(&g.ev1));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.dWork), 64 * Context::kMaxSlots));
    // (memsets go to the context's own stream, synchronised below before anything is launched)
    HIPCHK(hipMemsetAsync(g.dWork, 0, 64 * Context::kMaxSlots, g.stream));
    {
        int rc = createTraceStreams();
        if (rc) return rc;
    }
    for (int k = 0; k < Context::kMaxSlots; ++k) {
        HIPCHK(hipEventCreateWithFlags(&g.evTrace[k], kOrderingEvent));
        HIPCHK(hipEventCreateWithFlags(&g.evResolve[k], kOrderingEvent));
        g.resolveRecorded[k] = false;
    }

    g.frameSeq = 0;
    g.oldestPending = 0;
    HIPCHK(hipStreamCreateWithFlags(&g.hostStream2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&g.evBand, kOrderingEvent));
    HIPCHK(hipEventCreateWithFlags(&g.evBandEnd, kOrderingEvent));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.dRaysAhead), sizeof(unsigned long long) * Context::kMaxSlots));
    HIPCHK(hipMemsetAsync(g.dRaysAhead, 0, sizeof(unsigned long long) * Context::kMaxSlots, g.stream));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.dRaysStream), sizeof(unsigned long long) * Context::kStreamRing * Context::kStreamBatchMax));
    HIPCHK(hipMemsetAsync(g.dRaysStream, 0, sizeof(unsigned long long) * Context::kStreamRing * Context::kStreamBatchMax, g.stream));
    g.sbatch.used = false;
    if (const char* esb = getenv("TPT_STREAM_BATCH")) g.streamBatch = atoi(esb) != 0; // (default on)
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.dRaysBatch), sizeof(unsigned long long) * 2 * kMaxBatch));
    HIPCHK(hipMemsetAsync(g.dRaysBatch, 0, sizeof(unsigned long long) * 2 * kMaxBatch, g.stream));
    g.rsb[0].used = g.rsb[1].used = false;
    for (int k = 0; k < 4; ++k) g.ahead[k].used = false;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.dRaysOwn), 64));
    HIPCHK(hipMemsetAsync(g.dRaysOwn, 0, 64, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    g.dRays = g.dRaysOwn;
    g.lastTotal = 0;
    if (g.spheres.empty()) defaultScene(g.spheres, g.mats);
    if (const char* e6 = getenv("TPT_GRID_FILL")) g.gridFill = atoi(e6);
    if (const char* e7 = getenv("TPT_HOST_PACE")) g.hostPace = atoi(e7);
    if (const char* eh = getenv("TPT_TAIL_HELPERS")) g.helpersOn = atoi(eh) != 0;
    for (int k = 0; k < Context::kMaxSlots; ++k) {
        HIPCHK(hipEventCreateWithFlags(&g.evPre[k], kOrderingEvent));
        g.hrec[k].valid = false;
    }
    g.launchGen = 0;
    if (const char* e5 = getenv("TPT_GRID_DIV")) g.gridDiv = atoi(e5) > 0 ? atoi(e5) : 0;
    g.sceneDirty = true;
    {
        int rc = probeHardwareQueues();
        if (rc) return rc;
    }
    g.inited = true;
    return 0;
}

int tptShutdown(void)
{
    if (!g.inited) return 0;
    (void)discardLookahead();
    (void)tptCommDestroy();
    (void)hipStreamSynchronize(g.stream);
    (void)hipDeviceSynchronize();
    for (int k = 0; k < Context::kMaxSlots; ++k) {
        if (g.evPre[k]) (void)hipEventDestroy(g.evPre[k]);
        g.evPre[k] = nullptr;
        g.hrec[k].valid = false;
    }
    for (int k = 0; k < Context::kSceneSets; ++k) {
        Context::SceneSet& S = g.sets[k];
        (void)hipFree(S.dev);
        if (S.stage) (void)hipHostFree(S.stage);
        if (S.evUploaded) (void)hipEventDestroy(S.evUploaded);
        S = Context::SceneSet();
    }
    g.curSet = -1; g.pendingSet = -1; g.uploadSeq = 0;
    (void)hipFree(g.dWork); (void)hipFree(g.dRaysOwn); (void)hipFree(g.dFrame);
    (void)hipFree(g.dChunkCost); g.dChunkCost = nullptr; g.chunkCap = 0; g.chunkCount = 0; g.orderSeq = 0;
    for (int k = 0; k < Context::kOrderTables; ++k) { (void)hipFree(g.dChunkOrder[k]); g.dChunkOrder[k] = nullptr; }
    for (int k = 0; k < Context::kMaxOverlap; ++k) { (void)hipFree(g.dChunkSnap[k]); g.dChunkSnap[k] = nullptr; }
    g.dWork = nullptr; g.dRays = nullptr; g.dRaysOwn = nullptr; g.dFrame = nullptr;
    g.frameCap = 0;
    for (size_t i = 0; i < g.ktStart.size(); ++i) { (void)hipEventDestroy(g.ktStart[i]); (void)hipEventDestroy(g.ktStop[i]); }
    g.ktStart.clear(); g.ktStop.clear(); g.ktUsed = 0; g.kernelTiming = false;
    for (int k = 0; k < Context::kMaxOverlap; ++k) {
        if (g.traceStream[k]) { (void)hipStreamSynchronize(g.traceStream[k]); (void)hipStreamDestroy(g.traceStream[k]); }
        g.traceStream[k] = nullptr;
    }
    for (int k = 0; k < Context::kMaxSlots; ++k) {
        if (g.evTrace[k]) (void)hipEventDestroy(g.evTrace[k]);
        if (g.evResolve[k]) (void)hipEventDestroy(g.evResolve[k]);
        (void)hipFree(g.dColour[k]);
        (void)hipFree(g.dStack[k]); g.dStack[k] = nullptr;
        (void)hipFree(g.dPath[k]); g.dPath[k] = nullptr;
        g.evTrace[k] = nullptr; g.evResolve[k] = nullptr; g.dColour[k] = nullptr;
    }
    g.stackCap = g.colourCap = g.pathCap = 0; g.slotsReserved = 0; g.slotReservations = 0;
    (void)hipEventDestroy(g.ev0); (void)hipEventDestroy(g.ev1);
    if (g.evOrder) { (void)hipEventDestroy(g.evOrder); g.evOrder = nullptr; }
    (void)hipStreamDestroy(g.ownStream);
    g.ownStream = g.stream = nullptr;
    g.inited = false;
    g.updated = false;
    g.occCache.clear();
    g.mirror = nullptr; g.mirrorCounter = nullptr;
    if (g.hostStream2) { (void)hipStreamSynchronize(g.hostStream2); (void)hipStreamDestroy(g.hostStream2); g.hostStream2 = nullptr; }
    if (g.evBand) { (void)hipEventDestroy(g.evBand); g.evBand = nullptr; }
    if (g.evBandEnd) { (void)hipEventDestroy(g.evBandEnd); g.evBandEnd = nullptr; }
    g.tileSrc = nullptr; g.tileW = g.tileH = 0;
    for (int k = 0; k < 4; ++k) g.ahead[k].used = false;
    (void)hipFree(g.dRaysAhead); g.dRaysAhead = nullptr;
    (void)hipFree(g.dRaysBatch); g.dRaysBatch = nullptr;
    (void)hipFree(g.dRaysStream); g.dRaysStream = nullptr;
    g.sbatch.used = false;
    g.rsb[0].used = g.rsb[1].used = false;
    g.orderDone = true; g.orderStream = nullptr; g.oldestPending = 0; g.frameSeq = 0;
    g.streamDepth = 1; g.prevInFlight = -1;
    g.hostCaller = Context::HostCaller(); g.devCaller = Context::DeviceCaller(); // (a refusal or a streak remembered for a configuration
    g.smallStreak = 0; g.framesSinceIdle = 0; g.configEpoch = 1;                 //  does not survive re-initialisation)
    return 0;
}

int tptSetStream(void* hipStream)
{
    if (requireInit()) return -1;
    if (discardLookahead()) return -2;
    HIPCHK(hipStreamSynchronize(g.stream));
    g.stream = hipStream ? reinterpret_cast<hipStream_t>(hipStream) : g.ownStream;
    return 0;
}

int tptSetSamplesPerPixel(int spp)
{
    if (spp < 1 || spp > 65536) return fail("tptSetSamplesPerPixel: spp out of range");
    if (spp == g.spp) return 0; // (a setter that changes nothing must not invalidate frames traced ahead)
    g.spp = spp;
    g.configEpoch++;
    return 0;
}
int tptSetConfig(int lightSampling, float animateSmoothing, int mitsubaCompare)
{
    const int config = (lightSampling ? CFG_LIGHT_SAMPLING : 0) | (mitsubaCompare ? CFG_MITSUBA_COMPARE : 0);
    if (config == g.config && animateSmoothing == g.animateSmoothing) return 0;
    g.config = config;
    g.animateSmoothing = animateSmoothing;
    g.configEpoch++;
    return 0;
}
int tptSetSeedMode(int mode)
{
    if (mode != SEED_ROW_SERIAL && mode != SEED_PER_PIXEL) return fail("tptSetSeedMode: 0 (ROW_SERIAL) or 1 (PER_PIXEL)");
    if (mode == g.seedMode) return 0;
    g.seedMode = mode;
    g.configEpoch++;
    return 0;
}
int tptSetFoldMode(int mode)
{
    if (mode != FOLD_RECURSIVE && mode != FOLD_FORWARD) return fail("tptSetFoldMode: 0 (RECURSIVE) or 1 (FORWARD)");
    if (mode == g.foldMode) return 0;
    g.foldMode = mode;
    g.configEpoch++;
    return 0;
}
int tptKernelTimingBegin(int maxLaunches)
{
    if (requireInit()) return -1;
    if (maxLaunches < 1) maxLaunches = 1;
    while ((int)g.ktStart.size() < maxLaunches) {
        hipEvent_t a = nullptr, b = nullptr;
        HIPCHK(hipEventCreateWithFlags(&a, kTimingEvent));
        HIPCHK(hipEventCreateWithFlags(&b, kTimingEvent));
        g.ktStart.push_back(a);
        g.ktStop.push_back(b);
    }
    g.ktUsed = 0;
    g.kernelTiming = true;
    return 0;
}

int tptKernelTimingEnd(float* outSumMs, int* outLaunches)
{
    if (requireInit()) return -1;
    g.kernelTiming = false;
    HIPCHK(hipStreamSynchronize(g.stream));
    double sum = 0;
    for (size_t i = 0; i < g.ktUsed; ++i) {
        HIPCHK(hipEventSynchronize(g.ktStop[i]));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, g.ktStart[i], g.ktStop[i]));
        sum += ms;
    }
    if (outSumMs) *outSumMs = (float)sum;
    if (outLaunches) *outLaunches = (int)g.ktUsed;
    g.ktUsed = 0;
    return 0;
}

namespace {
// Everything enqueued so far completes; the slot bookkeeping starts afresh (the number of slots is about to change).
int drainPipeline()
{
    if (g.inited) {
        if (discardLookahead()) return -2;
        HIPCHK(hipStreamSynchronize(g.stream));
        for (int k = 0; k < Context::kMaxOverlap; ++k)
            if (g.traceStream[k]) HIPCHK(hipStreamSynchronize(g.traceStream[k]));
    }
    g.oldestPending = g.frameSeq;
    g.streamDepth = 1; g.prevInFlight = -1;
    for (int k = 0; k < Context::kMaxSlots; ++k) g.resolveRecorded[k] = false;
    return 0;
}
// frames in flight: what the caller asked for, what the hardware queues carry, and what the tile size rewards
int effectiveOverlap()
{
    int n = g.overlap < 1 ? 1 : (g.overlap > Context::kMaxOverlap ? Context::kMaxOverlap : g.overlap);
    if (n > g.overlapCap) n = g.overlapCap;
    if (n > g.shardOverlapCap) n = g.shardOverlapCap;
    return n;
}
} // namespace

int tptSetFrameOverlap(int frames)
{
    if (frames < 1 || frames > Context::kMaxOverlap) return fail("tptSetFrameOverlap: 1..16");
    int rc = drainPipeline();
    if (rc) return rc;
    g.overlap = frames;
    return 0;
}

int tptSetKernelVariant(int hitSpheres, int persistent, int ldsScene)
{
    if (hitSpheres < 0 || hitSpheres > 3) return fail("tptSetKernelVariant: hitSpheres 0 (two-phase) 1 (simple) 2 (two-phase, no groups) 3 (two-phase, VALU filter)");
    if (persistent != 1 && persistent != 3)
        return fail("tptSetKernelVariant: persistent 3 (path queues, the default) or 1 (lane refill); the thread-per-pixel (0) and lane-sorting (2) kernels were removed in round 3");
    g.hs = hitSpheres == 1 ? HS_SIMPLE : HS_TWO_PHASE;
    const int allow = hitSpheres == 2 ? 0 : 1, matrix = hitSpheres == 3 ? 0 : 1; // 3: the packed VALU filter everywhere (no matrix-core table)
    if (allow != g.allowGroups || matrix != g.useMatrix) {
        g.allowGroups = allow;
        g.useMatrix = matrix;
        g.sceneDirty = true; // the staged scene set carries (or not) the grouped arrays / the matrix table
    }
    g.persist = persistent; // 3 = path queues (default), 1 = lane-refill kernel
    g.configEpoch++;
    g.ldsScene = ldsScene < 0 ? -1 : (ldsScene ? 1 : 0);
    return 0;
}

int tptSetScene(const void* spheres, const void* materials, int count)
{
    if (!spheres || !materials || count <= 0) {
        defaultScene(g.spheres, g.mats);
    } else {
        if (count > (1 << 20)) return fail("tptSetScene: too many spheres");
        const SpherePOD* s = static_cast<const SpherePOD*>(spheres);
        const MaterialPOD* m = static_cast<const MaterialPOD*>(materials);
        g.spheres.assign(s, s + count);
        g.mats.assign(m, m + count);
    }
    g.sceneDirty = true;
    g.configEpoch++;
    return 0;
}

int tptSetCamera(const float* lookFrom, const float* lookAt, float vfov, float aperture, float focusDist)
{
    g.configEpoch++;
    if (!lookFrom || !lookAt) {
        g.camSetup = defaultCameraSetup();
        return 0;
    }
    for (int i = 0; i < 3; ++i) {
        g.camSetup.lookFrom[i] = lookFrom[i];
        g.camSetup.lookAt[i] = lookAt[i];
    }
    g.camSetup.vfov = vfov;
    g.camSetup.aperture = aperture;
    g.camSetup.focusDist = focusDist;
    return 0;
}

int tptSetRowShard(int stripeRows, int numParts, int part)
{
    g.configEpoch++;
    const bool sharded = numParts > 1 && stripeRows > 0;
    if (sharded && (part < 0 || part >= numParts)) return fail("tptSetRowShard: part out of range");
    // Tiles of a quarter frame and less are small enough that every further launch in flight costs more queue latency
    // than its overlap buys (one-GPU emulation of rank 0, C2: 16 in flight 29 / 39 Gray/s aggregate at 4 / 8 ranks, 8 in
    // flight 92 / 99; profiles/r02/r02_run36.log).
    const int cap = sharded && numParts > 2 ? 8 : Context::kMaxOverlap;
    if (cap != g.shardOverlapCap) {
        int rc = drainPipeline();
        if (rc) return rc;
        g.shardOverlapCap = cap;
    }
    if (!sharded) {
        g.stripeRows = 0; g.numParts = 1; g.part = 0;
        return 0;
    }
    g.stripeRows = stripeRows; g.numParts = numParts; g.part = part;
    return 0;
}
int tptLocalRowCount(int screenHeight) { return localRows(screenHeight); }
int tptLocalRowToGlobal(int localRow) { return localToGlobal(localRow); }

// UpdateTest, Test.cpp:302-342
int tptUpdate(float time, int frameCount, int screenWidth, int screenHeight, unsigned testFlags)
{
    (void)frameCount;
    if (requireInit()) return -1;
    if (screenWidth <= 0 || screenHeight <= 0) return fail("tptUpdate: bad size");
    if ((testFlags & TPT_FLAG_ANIMATE) && g.spheres.size() > 8) { // Test.cpp:304-308
        g.spheres[1].cy = cosf(time) + 1.0f;
        g.spheres[8].cz = sinf(time) * 0.3f;
        g.sceneDirty = true;
    }
    if (g.sceneDirty || (g.curSet < 0 && g.pendingSet < 0)) {
        int rc = stageScene();
        if (rc) return rc;
    }
    CameraSetup cs = g.camSetup;
    if (g.config & CFG_MITSUBA_COMPARE) cs.aperture = 0.0f; // Test.cpp:312-313
    g.cam = makeCamera(cs, float(screenWidth) / float(screenHeight)); // Test.cpp:341
    g.updated = true;
    return 0;
}

// ---------------------------------------------------------------- one frame: plan, buffers, enqueue
} // extern "C"

namespace {

// Everything decided about a frame before anything is enqueued.
struct FramePlan {
    KernelArgs a;
    bool rowSerial = false, queued = false, ldsScene = false, useOrder = false;
    size_t lds = 0;
    int occ = 0, threadsPerBlock = 0, blocks = 0;
    int nOverlap = 1;           // launches that may run side by side (trace streams in use)
    int nSlots = 1, slot = 0;   // frames that may be enqueued ahead / this frame's slot (colour, stack, path buffers, events)
    int batch = 1;              // frames traced by this launch (tptDrawDeviceBatch)
};

// Per-slot device buffers (frame colour, bounce stacks, path colour sums) are allocated for ALL slots of the pipeline at
// once, sized for the largest grid this kernel can ever be launched with at this frame shape -- never on the per-frame
// path: a lazily grown slot drained the whole pipeline (two stream syncs + hipFree/hipMalloc) on every first use, and with
// fewer warm-up frames than slots those drains landed inside the caller's timed region (round-1 driver bench: 17 instead
// of 35 Gray/s).  A re-allocation happens only when the frame shape / kernel variant / overlap asks for MORE than any
// earlier frame did; it synchronises everything once.
int syncAllStreams();
int reserveSlotBuffers(int nSlots, size_t colourBytes, size_t stackBytes, size_t pathBytes)
{ // (stack / path buffers are used while the kernel runs only: indexed by stream, allocated for the first kMaxOverlap slots)
    // Memory that a large batched frame pinned is given back when the caller returns to frames a quarter of that size and more
    // than 1 GiB of colour slots is held (one drain, like a growth); anything smaller stays (no churn between similar shapes).
    // ... and only after 8 launches in a row were that small: a caller that alternates large batches with a small tail chunk
    // (33..40 frames through tptDrawDeviceBatch: 32 + 1..8) must not free and re-allocate gigabytes on every call.
    const bool small = colourBytes * 4 <= g.colourCap && g.colourCap * (size_t)g.slotsReserved > (1ull << 30);
    g.smallStreak = small ? g.smallStreak + 1 : 0;
    // ... and never while a frame that was traced ahead (look-ahead, a row-serial or stream batch being served) still waits
    // for its blend: its ticket points into the very buffers a shrink frees.
    bool ticketsOut = g.rsb[0].used || g.rsb[1].used || g.sbatch.used;
    for (int k = 0; k < 4; ++k) ticketsOut = ticketsOut || g.ahead[k].used;
    const bool shrink = small && g.smallStreak >= 8 && !ticketsOut;
    if (shrink) g.smallStreak = 0;
    if (!shrink && nSlots <= g.slotsReserved && colourBytes <= g.colourCap && stackBytes <= g.stackCap && pathBytes <= g.pathCap) return 0;
    // ... nor may the slots GROW under such a frame: growth frees and re-allocates every colour slot (found by the round-4 advisor:
    // the second row-serial batch asking for more than the first had got).  The caller retries with less or drops its look-ahead.
    if (ticketsOut && colourBytes > g.colourCap)
        return refuse("frame buffers: the colour slots are held by frames traced ahead of their call; a larger launch has to wait for them");
    int rc = syncAllStreams();
    if (rc) return rc;
    if (shrink) {
        for (int k = 0; k < g.slotsReserved; ++k) {
            if (g.dColour[k]) HIPCHK(hipFree(g.dColour[k]));
            g.dColour[k] = nullptr;
        }
        g.colourCap = 0;
    }
    const size_t cb = colourBytes > g.colourCap ? colourBytes : g.colourCap, sb = stackBytes > g.stackCap ? stackBytes : g.stackCap,
                 pb = pathBytes > g.pathCap ? pathBytes : g.pathCap;
    const int n = nSlots > g.slotsReserved ? nSlots : g.slotsReserved;
    {
        // refuse BEFORE anything is freed when the device cannot hold the request (a failed hipMalloc half-way would leave the
        // context without its buffers)
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
            size_t need = 0;
            for (int k = 0; k < n; ++k) {
                const bool fresh = k >= g.slotsReserved;
                if (fresh || cb > g.colourCap) need += cb;
                if ((fresh || sb > g.stackCap) && k < Context::kMaxOverlap) need += sb;
            }
            const size_t held = (cb > g.colourCap ? g.colourCap * (size_t)g.slotsReserved : 0);
            if (need > freeB + held)
                return refuse("frame buffers: " + std::to_string(need >> 20) + " MiB needed for " + std::to_string(n) + " frame slots, " +
                            std::to_string((freeB + held) >> 20) + " MiB available on the device (smaller batch / frame, or fewer frames in flight: tptSetFrameOverlap)");
        }
    }
    auto grow = [&]() -> int {
        for (int k = 0; k < n; ++k) {
            const bool fresh = k >= g.slotsReserved;
            if (fresh || cb > g.colourCap) {
                if (g.dColour[k]) HIPCHK(hipFree(g.dColour[k]));
                g.dColour[k] = nullptr;
                if (cb) HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.dColour[k]), cb));
            }
            if (fresh || sb > g.stackCap) {
                if (g.dStack[k]) HIPCHK(hipFree(g.dStack[k]));
                g.dStack[k] = nullptr;
                if (sb && k < Context::kMaxOverlap) HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.dStack[k]), sb));
            }
        }
        return 0;
    };
    if ((rc = grow())) {
        // an allocation failed half-way (the memory check above is advisory: another process may have taken the memory): no slot may
        // keep a capacity its buffer does not have -- give everything back, the next frame reserves afresh
        for (int k = 0; k < Context::kMaxSlots; ++k) {
            (void)hipFree(g.dColour[k]); g.dColour[k] = nullptr;
            (void)hipFree(g.dStack[k]); g.dStack[k] = nullptr;
        }
        (void)hipGetLastError();
        g.colourCap = g.stackCap = g.pathCap = 0; g.slotsReserved = 0;
        return rc;
    }
    g.colourCap = cb; g.stackCap = sb; g.pathCap = pb; g.slotsReserved = n;
    g.slotReservations++;
    return 0;
}

// Which kernel runs this frame, how much LDS it takes, how many workgroups fit on a CU.
int chooseKernel(FramePlan& P)
{
    KernelArgs& a = P.a;
    P.rowSerial = g.seedMode == SEED_ROW_SERIAL;
    // LDS scene staging: default when {centre, r^2} + 1/r (20 B per padded sphere) + 48 B of material per sphere fit in
    // 40 KB (46 spheres: 3.2 KB; up to ~600 spheres)
    const int nPad = a.scene.nPairs * 2;
    P.ldsScene = g.ldsScene < 0 ? ((size_t)nPad * 20 + (size_t)a.scene.nSpheres * 48 <= 40960) : (g.ldsScene != 0);
    if (a.scene.nGroups > 0) P.ldsScene = false; // the LDS-staging kernels are built without the grouped traversal
    // bounce stack: the lane-refill kernel keeps the first levels in LDS and spills the rare deep ones to global memory
    a.ldsStackLevels = g.foldMode == FOLD_RECURSIVE ? g.ldsStackLevels : TPT_MAX_DEPTH;
    const size_t ldsV1 = tptLdsBytes(a, g.foldMode, P.ldsScene);
    // path-queue kernel: PER_PIXEL seeds, recursive fold, two-phase HitSpheres
    // (it packs a pixel as x | y << 16 and a path id as 16 bits: larger frames take the lane-refill kernel)
    // (so does its 64-B path record: 11 bits of sample index, 16 of sphere id)
    P.queued = g.persist == 3 && !P.rowSerial && g.hs == HS_TWO_PHASE && g.foldMode == FOLD_RECURSIVE && a.fc.width <= 65535 &&
               a.fc.height <= 65535 && g.spp <= 2047 && a.scene.nSpheres <= 65534;
    P.lds = P.queued ? tptQueueLdsBytes(a, P.ldsScene) : ldsV1;
    if ((size_t)a.scene.nLights * 32 > 96 * 1024)
        return fail("tptDrawDevice: too many emissive spheres for the LDS light table (3072 at most)");
    if (P.lds > 160 * 1024) return fail("tptDrawDevice: scene too large for LDS staging; use tptSetKernelVariant(.., .., 0)");
    if (P.queued) {
        a.ldsStackLevels = 1; // level 0 of the bounce stack sits in the path record (LDS), levels 1-9 in global memory
        // two workgroups per CU are worth more than the scene in LDS: a scene that costs the second workgroup its place
        // is read from global memory (L2) instead
        if (g.ldsScene < 0 && P.ldsScene && 160 * 1024 / (P.lds + 256) < 2 && 160 * 1024 / (tptQueueLdsBytes(a, false) + 256) >= 2) {
            P.ldsScene = false;
            P.lds = tptQueueLdsBytes(a, false);
        }
    }
    const int key = (P.queued ? (1 << 30) : 0) | (g.hs ? 8 : 0) | (g.foldMode ? 4 : 0) | (P.ldsScene ? 1 : 0) | ((int)(P.lds / 256) << 5);
    auto it = g.occCache.find(key);
    if (it == g.occCache.end()) {
        P.occ = P.queued ? (int)(160 * 1024 / (P.lds + 256)) : tptTraceOccupancy(g.hs, g.foldMode, P.ldsScene, P.lds);
        g.occCache[key] = P.occ;
    } else {
        P.occ = it->second;
    }
    P.threadsPerBlock = P.queued ? tptQueueThreadsPerBlock() : TPT_BLOCK;
    return 0;
}

// Work items, chunk size and the number of workgroups of this launch.
void sizeGrid(FramePlan& P)
{
    KernelArgs& a = P.a;
    const int resident = g.traceCUs * P.occ; // workgroups that can be co-resident (on the CUs the trace streams may use)
    const int wavesPerBlock = P.threadsPerBlock / 64;
    int chunk = P.rowSerial ? 1 : TPT_CHUNK_PIXELS;
    // small frames: hand out single 8x8 tiles so every resident wave gets several chunks
    if (!P.rowSerial && a.numItems / TPT_CHUNK_PIXELS < 8 * resident * wavesPerBlock) chunk = 64;
    if (P.queued) chunk = 64; // the path-queue kernel accounts its pixel pools in 64-pixel chunks
    a.chunkSize = chunk;
    a.numChunks = (a.numItems + chunk - 1) / chunk;
    a.chunksPerFrame = a.numChunks;
    a.numChunks *= P.batch; // a batched launch hands out the chunks of all its frames, frame after frame
    int blocks = (a.numChunks + wavesPerBlock - 1) / wavesPerBlock;
    a.laneCap = 64;
    if (P.rowSerial && !P.queued) {
        // Row-serial seeds: a work item is a whole image row (thousands of sequential rays), and there are few of them -- rows x
        // frames of the batch.  A wave that fills all 64 lanes leaves most SIMDs idle; a SIMD runs one wave's instructions at
        // the same rate whether 8 or 64 of its lanes are alive, so the items are dealt out over as many waves as there are
        // SIMDs (4 per CU), at least 4 lanes each.
        // (k launches in flight -- the deepest pipeline this caller has built so far -- share the SIMDs: k times the lanes)
        const int simds = g.traceCUs * 4, k = g.depthOverride > 0 ? g.depthOverride : (g.streamDepth > 1 ? g.streamDepth : 1);
        int cap = (int)(((long long)a.numChunks * k + simds - 1) / simds);
        cap = cap < 4 ? 4 : (cap > 64 ? 64 : cap);
        a.laneCap = cap;
        blocks = (a.numChunks + cap - 1) / cap;
    }
    // Frames in flight share the machine: with k trace kernels side by side each one gets fill / k of the resident
    // workgroups -- its pools then stay in steady state longer before they drain, and the launches behind it fill the
    // gaps.  fill = 200 % on a single GPU (measured best), 100 % when the frame is sharded over ranks (oversubscription
    // buys nothing on small tiles).  A caller that synchronises every frame has nothing in flight and gets the full grid.
    int cap;
    if (g.gridDiv > 0) {
        cap = resident / g.gridDiv;
    } else {
        // fill = how many times the machine the launches in flight ask for together: 200 % on a single GPU (64 workgroups per
        // launch at 16 in flight: long steady states; 100 %: 54.6 vs 55.9 Gray/s), 100 % when the frame is sharded over ranks
        // (oversubscription buys nothing on small tiles).  Rounds 2-3 gave the first 24 frames after an idle pipeline 400 %:
        // worth +4.5 % on a burst of exactly 20 frames (whose last launches then fill the machine as it empties), but -8 % on 30
        // frames and -1.5 % on 100 (profiles/r04/r04_run9.log) and 1.8x the memory traffic per launch -- a constant fitted to
        // one command line; removed in round 4.
        const int fill = g.gridFill > 0 ? g.gridFill : (g.numParts > 1 ? 100 : 200);
        // k = how many launches share the machine.  Not just what is in flight right now: a caller that streams frames
        // (enqueue, enqueue, ..., synchronise once) starts every burst with an empty pipeline, and whole-machine grids
        // for the first frames of a burst serialise them (each with its own tail) -- a 20-frame burst ran at 24 instead
        // of 33 Gray/s.  So the deepest pipeline this caller has built is remembered (streamDepth) and only forgotten
        // when two consecutive frames find the pipeline empty: that is a caller who synchronises every frame
        // (the reference's DrawTest contract) and gets the whole machine.
        const int inFlight = framesInFlight(P.nSlots);
        g.framesSinceIdle = inFlight == 0 ? 0 : g.framesSinceIdle + 1;
        if (inFlight == 0 && g.prevInFlight == 0) g.streamDepth = 1;
        if (inFlight + 1 > g.streamDepth) g.streamDepth = inFlight + 1;
        g.prevInFlight = inFlight;
        int k = g.streamDepth;
        if (g.depthOverride > 0) k = g.depthOverride; // the host-pointer path knows exactly how deep its pipeline is
        if (k > P.nOverlap) k = P.nOverlap;
        cap = (int)((long long)resident * fill / (100ll * k));
        if (cap > resident) cap = resident;
        const int floorBlocks = resident / (2 * (P.nOverlap > 1 ? P.nOverlap : 1));
        if (cap < floorBlocks) cap = floorBlocks;
    }
    if (P.rowSerial && !P.queued) cap = resident; // (row-serial launches are latency-bound: one short wave per SIMD, whatever else is in flight)
    if (cap < 1) cap = 1;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    P.blocks = blocks;
    a.totalWaves = (unsigned)(blocks * wavesPerBlock);
}

// Largest number of workgroups sizeGrid can ever pick for this kernel at this frame shape.
int maxGridBlocks(const FramePlan& P)
{
    const KernelArgs& a = P.a;
    const int wavesPerBlock = P.threadsPerBlock / 64;
    const int resident = g.traceCUs * P.occ;
    const int minChunk = P.rowSerial ? 1 : 64;
    const int byWork = (((a.numItems + minChunk - 1) / minChunk) * P.batch + wavesPerBlock - 1) / wavesPerBlock;
    int m = resident < byWork ? resident : byWork;
    return m < 1 ? 1 : m;
}

// Per-slot buffers of this frame: colour, bounce-stack spill / per-path stacks, path colour sums.
int ensureFrameBuffers(FramePlan& P, int w)
{
    KernelArgs& a = P.a;
    const int slot = P.slot;
    const int maxBlocks = maxGridBlocks(P);
    const bool needStack = g.foldMode == FOLD_RECURSIVE && a.ldsStackLevels < TPT_MAX_DEPTH;
    const size_t maxColumns = (size_t)maxBlocks * (size_t)(P.queued ? tptQueuePathsPerBlock() : P.threadsPerBlock);
    const size_t stackBytes = needStack ? maxColumns * (size_t)(TPT_MAX_DEPTH - a.ldsStackLevels) * sizeof(f4) : 0;
    const size_t pathBytes = 0; // (the path-queue kernel's per-path colour sums moved into LDS)
    int rc = reserveSlotBuffers(P.nSlots, (size_t)a.nLocalRows * w * sizeof(f4) * (size_t)P.batch, stackBytes, pathBytes);
    if (rc) return rc;
    a.frameColour = g.dColour[slot];
    a.work = g.dWork + 16 * slot;
    a.rayCounter = g.dRays;
    a.stackBuf = nullptr;
    a.stackStride = 0;
    if (needStack) {
        a.stackBuf = g.dStack[slot % P.nOverlap];
        a.stackStride = P.queued ? P.blocks * tptQueuePathsPerBlock() : P.blocks * P.threadsPerBlock;
        // the columns of a helper grid (workgroups blocks .. 2 * blocks - 1 at most) lie behind the launch's own: one stride for both
        if (P.queued) a.stackStride = (2 * P.blocks < maxBlocks ? 2 * P.blocks : maxBlocks) * tptQueuePathsPerBlock();
    }
    a.pathBuf = nullptr;
    return 0;
}

int syncAllStreams()
{
    for (int k = 0; k < Context::kMaxOverlap; ++k) HIPCHK(hipStreamSynchronize(g.traceStream[k]));
    for (int k = 0; k < Context::kMaxSlots; ++k) g.hrec[k].valid = false; // (nothing is in flight any more)
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

// The caller is about to block: give the newest launches that have not finished a second grid each (tpt_device.h).  The newest launches
// first -- they have the most left -- and at most helperMax of them; a launch is helped once.  hipEventQuery is a hint only: a launch
// that finishes a microsecond later closes its counter block and the helpers leave at once.
int launchTailHelpers()
{
    if (!g.helpersOn) return 0;
    int order[Context::kMaxSlots], n = 0;
    for (int s = 0; s < Context::kMaxSlots; ++s) {
        Context::HelperRec& R = g.hrec[s];
        if (!R.valid) continue;
        if (hipEventQuery(g.evTrace[s]) == hipSuccess) { R.valid = false; continue; }
        (void)hipGetLastError();
        order[n++] = s; // every launch still in flight, helped already or not
    }
    if (n < 2) return 0; // (a caller that waits for every frame has nothing to rebalance)
    for (int i = 1; i < n; ++i) // newest first
        for (int j = i; j > 0 && (int)(g.hrec[order[j]].a.gen - g.hrec[order[j - 1]].a.gen) > 0; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    // The k-th newest launch is helped from the stream of the k-th OLDEST launch that has nothing queued behind it: that stream is
    // the next to fall idle for good -- a burst longer than the 16 streams has its last launches queued behind its first ones, and a
    // helper placed there would start when everything is over (profiles/r04/r04_run23.log).  The helper may well start before the
    // launch it helps (it only waits for what that launch waits for): the pool is simply part-consumed when the launch arrives.
    // (Streams of their own were tried first: four more streams in the process cost the whole pipeline a factor 2.4 -- the
    // runtime's hardware queues are a small shared pool; r04_run21.log, r04_run22.log.)
    hipStream_t freeSoon[Context::kMaxSlots];
    int nFree = 0;
    for (int i = n - 1; i >= 0; --i) { // oldest first
        hipStream_t ts = g.hrec[order[i]].ts;
        int queued = 0;
        for (int j = 0; j < n; ++j) queued += g.hrec[order[j]].ts == ts ? 1 : 0;
        if (queued == 1) freeSoon[nFree++] = ts;
    }
    for (int i = 0; i < n / 2 && i < Context::kHelperMax && i < nFree; ++i) {
        Context::HelperRec& R = g.hrec[order[i]];
        if (R.helped) continue;
        R.helped = true;
        int extra = R.maxBlocks - R.blocks;
        if (extra > R.blocks) extra = R.blocks; // (two and three times the launch's own grid measured no better, profiles/r05/r05_run2.log)
        if (extra < 1) continue;
        KernelArgs h = R.a;
        h.helperBase = R.blocks;
        h.helperPct = Context::kHelperPct;
        hipStream_t hs = freeSoon[i];
        if (hs == R.ts) continue; // (its own stream: it would run after the launch it is meant to help)
        HIPCHK(hipStreamWaitEvent(hs, g.evPre[order[i]], 0));
        HIPCHK(tptLaunchTraceQueue(h, R.ldsScene, extra, R.lds, hs));
        g.helperLaunches++;
    }
    return 0;
}

// Cost-ordered work distribution of the lane-refill kernel: statistics and order tables for this chunk count.
int prepareChunkOrder(FramePlan& P)
{
    KernelArgs& a = P.a;
    a.chunkOrder = nullptr;
    a.chunkCost = nullptr;
    a.chunkShift = 6;
    P.useOrder = g.costOrder && !P.rowSerial && !P.queued && a.numChunks > 1 &&
                 (a.chunkSize & (a.chunkSize - 1)) == 0;
    if (!P.useOrder) return 0;
    int sh = 0;
    while ((1 << sh) < a.chunkSize) ++sh;
    a.chunkShift = sh;
    const size_t bytes = sizeof(unsigned) * (size_t)a.numChunks;
    if (a.numChunks > g.chunkCap) {
        int rc = syncAllStreams();
        if (rc) return rc;
        if (g.dChunkCost) HIPCHK(hipFree(g.dChunkCost));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.dChunkCost), bytes));
        for (int k = 0; k < Context::kOrderTables; ++k) {
            if (g.dChunkOrder[k]) HIPCHK(hipFree(g.dChunkOrder[k]));
            g.dChunkOrder[k] = nullptr;
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.dChunkOrder[k]), bytes));
        }
        for (int k = 0; k < Context::kMaxOverlap; ++k) {
            if (g.dChunkSnap[k]) HIPCHK(hipFree(g.dChunkSnap[k]));
            g.dChunkSnap[k] = nullptr;
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.dChunkSnap[k]), bytes));
        }
        g.chunkCap = a.numChunks;
        g.chunkCount = 0;
    }
    if (g.chunkCount != a.numChunks) { // new resolution / sharding: statistics start over
        int rc = syncAllStreams();
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(g.dChunkCost, 0, bytes, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
        g.chunkCount = a.numChunks;
        g.orderSeq = 0;
        g.orderDone = true;
    }
    a.chunkCost = g.dChunkCost;
    return 0;
}

// Give the launch on `ts` an order table: re-sorted from the statistics gathered so far (every frame until the first
// frames' statistics have certainly arrived -- the sort runs beside up to nOverlap unfinished frames -- then every
// 32nd), or the most recent one.  Tables rotate over kOrderTables buffers (> frames in flight): a trace kernel still in
// flight keeps reading the one it was given.
int enqueueChunkOrder(FramePlan& P, hipStream_t ts)
{
    if (!P.useOrder) return 0;
    if (g.orderSeq > 0) {
        const int fresh = (int)(g.orderSeq % Context::kOrderTables);
        if (g.orderSeq <= (unsigned long long)(2 * P.nSlots + 2) || (g.orderSeq & 31ull) == 0ull) {
            HIPCHK(tptLaunchChunkOrder(g.dChunkCost, g.dChunkSnap[P.slot % P.nOverlap], g.dChunkOrder[fresh], P.a.numChunks, ts));
            HIPCHK(hipEventRecord(g.evOrder, ts));
            g.orderStream = ts;
            g.orderDone = false;
            g.lastOrderTable = fresh;
        } else if (g.orderStream && g.orderStream != ts) {
            // the most recent table may still be being written by another stream's sort kernel
            HIPCHK(hipStreamWaitEvent(ts, g.evOrder, 0));
        }
        P.a.chunkOrder = g.dChunkOrder[g.lastOrderTable];
    }
    g.orderSeq++;
    return 0;
}

} // namespace

extern "C" {

} // extern "C"

namespace {

// First half of a frame: plan, buffers, trace kernel on the slot's stream.  `frameRays`: where the kernel adds its ray
// count (the context's counter, or a per-slot one for frames that are traced ahead of their DrawTest call).
int enqueueTrace(int frameCount, int w, int h, unsigned testFlags, unsigned long long* frameRays, TraceTicket& T, int batch = 1, int rayStride = 0)
{
    if (g.sceneDirty || (g.curSet < 0 && g.pendingSet < 0)) { // tptSetScene after the last tptUpdate
        int rc = stageScene();
        if (rc) return rc;
    }
    FramePlan P;
    KernelArgs& a = P.a;
    a.scene = deviceView(); // pointers of the set this frame reads; its upload is enqueued below, on the frame's stream
    a.fc = makeFrameConsts(g.cam, w, h, g.spp, frameCount, testFlags, g.seedMode, g.config, g.animateSmoothing);
    a.nLocalRows = localRows(h);
    if (g.numParts > 1 && g.stripeRows > 0) {
        a.stripeRows = g.stripeRows;
        a.stripeStride = g.stripeRows * g.numParts;
        a.stripeOffset = g.stripeRows * g.part;
    } else {
        a.stripeRows = h > 0 ? h : 1;
        a.stripeStride = a.stripeRows;
        a.stripeOffset = 0;
    }
    T.valid = false;
    if (a.nLocalRows <= 0) return 0; // nothing to do on this rank
    a.tilesX = (w + 7) / 8;
    const int tilesY = (a.nLocalRows + 7) / 8;
    a.numItems = g.seedMode == SEED_ROW_SERIAL ? a.nLocalRows : a.tilesX * tilesY * 64;
    P.batch = batch;
    a.batchFrames = batch;
    a.framePlane = a.nLocalRows * w;
    a.chunksPerFrame = 0; // (sizeGrid)
    P.nOverlap = effectiveOverlap();
    // Twice as many colour slots as trace streams for frames up to 32 MB of colour (2 M pixels): the blends are ordered
    // (frame f after f - 1) but the trace kernels finish out of order, so with one slot per stream a stream whose kernel
    // finished early sits idle until every earlier frame has been blended.  With a spare slot its next kernel starts at
    // once.  Worth +3-8 % on tiles of a sharded C2 frame (rank 0 of 2 / 4 / 8), nothing at C2 on one GPU (the machine is
    // full either way), and -4 % at C3, where 16 launches of 190 ms running at once only crowd the caches: large frames
    // keep one slot per stream (profiles/r02/r02_run42.log, r02_evidence2.log).
    P.nSlots = P.nOverlap;
    const size_t colourBytesPerSlot = (size_t)a.nLocalRows * (size_t)w * sizeof(f4) * (size_t)batch;
    if (P.nOverlap > 1 && g.slotFactor > 1 && colourBytesPerSlot <= (32ull << 20)) P.nSlots = 2 * P.nOverlap;
    // Every slot is sized for the largest frame seen, so the number of slots bounds the memory a large (batched) frame pins:
    // all colour slots together stay under 8 GiB (1280x720 x 32 frames per launch: 16 slots x 472 MB = 7.5 GB; a 4K x 8-frame
    // batch: 4 slots instead of 16), never fewer than 2 (one being traced, one being blended); a single slot above 4 GiB is
    // refused here, before anything is drained or freed.
    if (colourBytesPerSlot > (4ull << 30))
        return refuse("tptDrawDeviceBatch: " + std::to_string(colourBytesPerSlot >> 20) + " MiB of frame colour per launch (rows x width x 16 B x frames): over the 4096 MiB limit, use a smaller batch");
    while (P.nSlots > 2 && colourBytesPerSlot * (size_t)P.nSlots > (8ull << 30)) P.nSlots /= 2;
    if (rayStride > 0 && g.seedMode == SEED_ROW_SERIAL && P.nSlots > 4) P.nSlots = 4; // (the host path's row-serial batches: two alive at a time)
    if (P.nOverlap > P.nSlots) P.nOverlap = P.nSlots;
    P.slot = (int)(g.frameSeq % (unsigned long long)P.nSlots);
    g.frameSeq++;
    struct SeqGuard { // an enqueue that fails before its launch does not consume a slot of the pipeline
        bool launched = false;
        ~SeqGuard() { if (!launched) g.frameSeq--; }
    } seqGuard;

    int rc = chooseKernel(P);
    if (rc) return rc;
    // a batch is traced by the path-queue kernel (per-pixel seeds) or, in the reference's own seed mode, by the lane-refill
    // kernel: one lane per (frame, row) -- rows AND frames are independent RNG streams there (Test.cpp:280)
    if (batch > 1 && (!(P.queued || P.rowSerial) || w > 8192 || h > 8192 || (long long)a.nLocalRows * w * batch > (1ll << 30)))
        return refuse("tptDrawDeviceBatch: needs the path-queue kernel (per-pixel seeds, recursive fold, two-phase HitSpheres) or row-serial seeds, and a frame of at most 8192 x 8192 (2^30 pixels per batch)");
    sizeGrid(P);
    if ((rc = ensureFrameBuffers(P, w))) return rc;
    if (frameRays) a.rayCounter = frameRays;
    a.rayCounterStride = rayStride; // (batched row-serial launch for the host path: one counter per frame of the batch)
    if ((rc = prepareChunkOrder(P))) return rc;
    g.lastBlocksPerCU = P.occ;
    g.lastLds = (int)P.lds;
    g.lastGrid = P.blocks;

    // trace(f) on its own stream (no dependency on the previous frame); the ordered blend follows on g.stream
    const int slot = P.slot;
    const bool pipelined = P.nOverlap > 1;
    hipStream_t ts = pipelined ? g.traceStream[slot % P.nOverlap] : g.stream;
    if (pipelined && g.resolveRecorded[slot]) {
        // Host pacing: the caller's thread waits here until the slot's previous frame has been blended, so it never runs more
        // than nSlots frames ahead and the queue's barrier below is already satisfied when the command processor reaches it.
        // A host that runs far ahead leaves every queue with an unsatisfied barrier at its head, and the command processor
        // polls them all: small frames retire at half the rate (C1, 400 frames: 6.7 -> 13.6 Gray/s; rank 0 of 8: 99 -> 128
        // aggregate; C2 unchanged; profiles/r02/r02_run40.log).  Pacing only: the stream wait below is what orders the work
        // (an event query may report "done" early on a re-recorded event).
        if (g.hostPace) {
            // (bounded: a caller whose stream is blocked behind something it will only enqueue later must not hang here)
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (hipEventQuery(g.evResolve[slot]) == hipErrorNotReady) {
                std::this_thread::yield();
                if ((++spins & 255u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
            }
            (void)hipGetLastError();
        }
        HIPCHK(hipStreamWaitEvent(ts, g.evResolve[slot], 0)); // colour buffer free again
    }
    if ((rc = enqueueSceneUpload(ts))) return rc; // behind the wait above: nobody reads the set being replaced any more
    if ((rc = enqueueChunkOrder(P, ts))) return rc;
    if (frameRays && frameRays != g.dRays) HIPCHK(hipMemsetAsync(frameRays, 0, sizeof(unsigned long long) * (size_t)(rayStride > 0 ? batch : 1), ts));
    const bool helpable = P.queued && pipelined && batch == 1 && !P.rowSerial; // (single frames of the path-queue kernel)
    a.helperBase = 0;
    a.helperPct = 0;
    a.gen = 0u;
    g.hrec[slot].valid = false;
    if (helpable) {
        if (++g.launchGen == 0u) g.launchGen = 1u;
        a.gen = g.launchGen;
        HIPCHK(hipEventRecord(g.evPre[slot], ts)); // the set upload and the slot's previous users are behind this point
    }
    const bool timeIt = g.kernelTiming && g.ktUsed < g.ktStart.size();
    if (timeIt) HIPCHK(hipEventRecord(g.ktStart[g.ktUsed], ts));
    if (P.queued)
        HIPCHK(tptLaunchTraceQueue(a, P.ldsScene, P.blocks, P.lds, ts));
    else
        HIPCHK(tptLaunchTrace(a, g.hs, g.foldMode, P.ldsScene, P.blocks, P.lds, ts));
    seqGuard.launched = true;
    if (timeIt) {
        HIPCHK(hipEventRecord(g.ktStop[g.ktUsed], ts));
        g.ktUsed++;
    }
    if (pipelined) HIPCHK(hipEventRecord(g.evTrace[slot], ts));
    if (helpable) {
        Context::HelperRec& R = g.hrec[slot];
        R.a = a; R.ldsScene = P.ldsScene; R.blocks = P.blocks; R.maxBlocks = maxGridBlocks(P); R.lds = P.lds;
        R.helped = false; R.valid = true; R.ts = ts;
    }
    T.slot = slot;
    T.nPixels = a.nLocalRows * w;
    T.pipelined = pipelined;
    T.lerpFac = a.fc.lerpFac;
    T.colour = a.frameColour;
    T.batch = batch;
    for (int j = 0; j < batch && batch > 1; ++j)
        T.lerp.v[j] = makeFrameConsts(g.cam, w, h, g.spp, frameCount + j, testFlags, g.seedMode, g.config, g.animateSmoothing).lerpFac;
    T.valid = true;
    return 0;
}

// Second half: the progressive blend of the frame's colour into the accumulation tile (Test.cpp:293-295), in frame order
// on g.stream.  `frameRays` (host path): a per-slot ray count the kernel also adds to the context's running total.
int enqueueResolve(const TraceTicket& T, float* deviceTile, const unsigned long long* frameRays)
{
    if (!T.valid) return 0;
    if (T.pipelined) HIPCHK(hipStreamWaitEvent(g.stream, g.evTrace[T.slot], 0));
    if (T.batch > 1)
        HIPCHK(tptLaunchResolveBatch(deviceTile, T.colour, T.nPixels, T.nPixels, T.batch, T.lerp, g.mirror, g.dRays, g.mirrorCounter, g.stream));
    else
        HIPCHK(tptLaunchResolve(deviceTile, T.colour, T.nPixels, T.lerpFac, g.mirror, g.dRays, g.mirrorCounter, frameRays, g.stream));
    if (T.pipelined) {
        HIPCHK(hipEventRecord(g.evResolve[T.slot], g.stream));
        g.resolveRecorded[T.slot] = true;
    }
    return 0;
}

} // namespace

extern "C" {

int tptDrawDevice(float time, int frameCount, int w, int h, float* deviceTile, unsigned testFlags)
{
    (void)time; // stored but never read by the reference either (Test.cpp:257,347)
    if (requireInit()) return -1;
    if (!g.updated) return fail("tptDrawDevice: call tptUpdate (UpdateTest) first");
    if (!deviceTile || w <= 0 || h <= 0) return fail("tptDrawDevice: bad arguments");
    // A caller that waits for every frame before it asks for the next (the reference's DrawTest contract, on a device tile)
    // would leave each frame alone on the GPU, bound by its longest paths: 0.98 ms per C2 frame against 0.45 in a stream.
    // Such a caller shows: when its call arrives, the previous frame's blend has already completed.  After two such calls
    // for consecutive frames of one configuration the next frames are traced ahead of it, exactly as tptDraw does for the
    // host-pointer path (same bookkeeping, same per-slot ray counters; a wrong guess costs GPU time only).  A caller that
    // streams frames never meets the condition and takes the plain path below.
    Context::DeviceCaller& D = g.devCaller;
    const unsigned long long key = g.configEpoch;
    const bool pipelined = effectiveOverlap() > 1;
    const bool stable = !g.sceneDirty && g.pendingSet < 0 && !(testFlags & TPT_FLAG_ANIMATE);
    bool prevDone = false;
    if (D.lastSlot >= 0 && g.resolveRecorded[D.lastSlot]) {
        prevDone = hipEventQuery(g.evResolve[D.lastSlot]) == hipSuccess;
        (void)hipGetLastError();
    }
    D.syncStreak = prevDone ? D.syncStreak + 1 : 0;
    D.seqStreak = (frameCount == D.frame + 1 && w == D.w && h == D.h && testFlags == D.flags && key == D.key) ? D.seqStreak + 1 : 0;
    D.frame = frameCount; D.w = w; D.h = h; D.flags = testFlags; D.key = key;
    const bool lookAhead = pipelined && stable && !g.mirror && g.lookahead > 0 && D.syncStreak >= 2 && D.seqStreak >= 2;

    TraceTicket T;
    int rc;
    const Context::Ahead& front = g.ahead[0];
    const bool hit = front.used && front.frameCount == frameCount && front.w == w && front.h == h && front.flags == testFlags &&
                     front.configKey == key && stable && !g.mirror;
    if (!hit && !lookAhead) {
        // ---- a streaming caller with small frames: served from / starting a stream batch (see Context::StreamBatch)
        Context::StreamBatch& SB = g.sbatch;
        if (SB.used && SB.w == w && SB.h == h && SB.flags == testFlags && SB.key == key && stable && frameCount == SB.firstFrame + SB.next) {
            const int j = SB.next++;
            T = SB.T;
            T.colour = SB.T.colour + (size_t)j * (size_t)SB.T.nPixels;
            T.lerpFac = SB.T.lerp.v[j];
            T.batch = 1;
            if (SB.next == SB.n) SB.used = false;
            rc = enqueueResolve(T, deviceTile, g.dRaysStream + SB.counterBase + j);
            D.lastSlot = T.slot;
            return rc;
        }
        if ((rc = discardLookahead())) return rc; // (also closes a stream batch that did not continue as guessed)
        int nBatch = 1;
        if (g.streamBatch && pipelined && stable && D.seqStreak >= 2 && g.persist == 3 && g.seedMode == SEED_PER_PIXEL && g.foldMode == FOLD_RECURSIVE &&
            g.hs == HS_TWO_PHASE && w <= 8192 && h <= 8192 && g.spp <= 2047) {
            // how many frames make a launch long enough to amortise its fixed cost: 1 at 1280x720x4 (3.7 M samples), 2 / 4 / 8 for
            // halves / quarters / eighths of that (profiles/r03/r03_run19.log: where several frames per launch pay)
            const long long samples = (long long)localRows(h) * w * g.spp;
            nBatch = samples >= 2400000 ? 1 : samples >= 1200000 ? 2 : samples >= 600000 ? 4 : Context::kStreamBatchMax;
            if (samples <= 0) nBatch = 1;
        }
        if (nBatch > 1) {
            SB.firstFrame = frameCount; SB.n = nBatch; SB.next = 1; SB.w = w; SB.h = h; SB.flags = testFlags; SB.key = key;
            SB.counterBase = (int)(g.streamBatches++ % (unsigned long long)Context::kStreamRing) * Context::kStreamBatchMax;
            if ((rc = enqueueTrace(frameCount, w, h, testFlags, g.dRaysStream + SB.counterBase, SB.T, nBatch, 1))) return rc;
            SB.used = SB.T.valid;
            T = SB.T;
            T.lerpFac = SB.T.lerp.v[0];
            T.batch = 1;
            rc = enqueueResolve(T, deviceTile, T.valid ? g.dRaysStream + SB.counterBase : nullptr);
            if (T.valid) D.lastSlot = T.slot;
            return rc;
        }
        // the plain path: trace + blend, the kernel adds its rays to the running total itself
        if ((rc = enqueueTrace(frameCount, w, h, testFlags, nullptr, T))) return rc;
        rc = enqueueResolve(T, deviceTile, nullptr);
        if (T.valid) D.lastSlot = T.slot;
        return rc;
    }
    // one frame more than the host-pointer path looks ahead: there the PCIe copies fill the caller's time (2 ahead: 0.88 ms
    // per frame, 3: 0.92), here nothing does (2: 0.598 ms, 3: 0.561; profiles/r02/r02_run50.log)
    const int devAhead = g.lookahead + 1 < 3 ? g.lookahead + 1 : 3;
    struct DepthScope { // launches made from here share the machine with the frames traced ahead, not with a deep pipeline
        explicit DepthScope(int d) { g.depthOverride = d; }
        ~DepthScope() { g.depthOverride = 0; }
    } depthScope(1 + devAhead);
    int raySlot = -1;
    if (hit) {
        if ((rc = takeAhead(T, raySlot))) return rc;
    } else {
        if ((rc = discardLookahead())) return rc;
        raySlot = (int)(g.frameSeq % (unsigned long long)Context::kMaxSlots);
        if ((rc = enqueueTrace(frameCount, w, h, testFlags, g.dRaysAhead + raySlot, T))) return rc;
    }
    if (lookAhead && T.valid && (rc = traceAhead(frameCount, w, h, testFlags, key, devAhead))) return rc;
    rc = enqueueResolve(T, deviceTile, T.valid ? g.dRaysAhead + raySlot : nullptr);
    if (T.valid) D.lastSlot = T.slot;
    return rc;
}

// nFrames consecutive frames (frameCount = firstFrame ... firstFrame + nFrames - 1) of the scene and camera as of the last
// tptUpdate, traced by ONE launch and blended in frame order by one: the same bits as nFrames tptDrawDevice calls.
int tptDrawDeviceBatch(float time, int firstFrame, int nFrames, int w, int h, float* deviceTile, unsigned testFlags)
{
    (void)time;
    if (requireInit()) return -1;
    if (!g.updated) return fail("tptDrawDeviceBatch: call tptUpdate (UpdateTest) first");
    if (!deviceTile || w <= 0 || h <= 0 || nFrames < 1) return fail("tptDrawDeviceBatch: bad arguments");
    if (nFrames > 1 && (testFlags & TPT_FLAG_ANIMATE))
        return fail("tptDrawDeviceBatch: an animated scene changes every frame (Test.cpp:304-308): one tptUpdate + tptDrawDevice per frame");
    int rc = discardLookahead();
    if (rc) return rc;
    for (int f = 0; f < nFrames; f += kMaxBatch) {
        const int n = nFrames - f < kMaxBatch ? nFrames - f : kMaxBatch;
        TraceTicket T;
        if ((rc = enqueueTrace(firstFrame + f, w, h, testFlags, nullptr, T, n))) return rc;
        if ((rc = enqueueResolve(T, deviceTile, nullptr))) return rc;
    }
    return 0;
}

int tptRayCounterRead(int64_t* outTotalRays)
{
    if (requireInit()) return -1;
    unsigned long long v = 0;
    HIPCHK(hipMemcpyAsync(&v, g.dRays, sizeof(v), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    if (outTotalRays) *outTotalRays = (int64_t)v;
    return 0;
}

int tptSetTileMirror(float* deviceMirror, void* deviceCounterOut)
{
    g.mirror = deviceMirror;
    g.mirrorCounter = deviceMirror ? static_cast<unsigned long long*>(deviceCounterOut) : nullptr;
    return 0;
}

int tptSetRayCounter(void* deviceU64)
{
    if (requireInit()) return -1;
    if (discardLookahead()) return -2;
    HIPCHK(hipStreamSynchronize(g.stream));
    g.dRays = deviceU64 ? static_cast<unsigned long long*>(deviceU64) : g.dRaysOwn;
    int64_t total = 0;
    int rc = tptRayCounterRead(&total);
    if (rc) return rc;
    g.lastTotal = total; // DrawTest reports per-frame differences of the active counter
    return 0;
}

int tptSynchronize(void)
{
    if (requireInit()) return -1;
    if (int rc = launchTailHelpers()) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int tptTimerBegin(void)
{
    if (requireInit()) return -1;
    HIPCHK(hipEventRecord(g.ev0, g.stream));
    return 0;
}
int tptTimerEnd(float* outMs)
{
    if (requireInit()) return -1;
    HIPCHK(hipEventRecord(g.ev1, g.stream));
    if (int rc = launchTailHelpers()) return rc;
    HIPCHK(hipEventSynchronize(g.ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, g.ev0, g.ev1));
    if (outMs) *outMs = ms;
    return 0;
}

// ---------------------------------------------------------------- DrawTest, Test.cpp:344-367 (host backbuffer, synchronous)
} // extern "C"

namespace {

// The frames traced ahead belong to a DrawTest sequence that did not continue as predicted (or the device path is about
// to be used): let them finish and forget them.  Their colour buffers were never blended into anything.
int discardLookahead()
{
    g.sbatch.used = false; // (an open stream batch needs no wait: its unserved planes are simply never blended)
    bool any = g.rsb[0].used || g.rsb[1].used;
    for (int k = 0; k < 4; ++k) any = any || g.ahead[k].used;
    if (!any) return 0;
    for (int k = 0; k < Context::kMaxOverlap; ++k) HIPCHK(hipStreamSynchronize(g.traceStream[k]));
    for (int k = 0; k < 4; ++k) g.ahead[k].used = false;
    g.rsb[0].used = g.rsb[1].used = false;
    g.sbatch.used = false;
    return 0;
}

// The frame at the head of the look-ahead queue becomes the caller's frame.
int takeAhead(TraceTicket& T, int& raySlot)
{
    g.aheadHits++;
    T = g.aheadTicket[0];
    raySlot = g.ahead[0].raySlot;
    for (int k = 0; k + 1 < 4; ++k) { g.ahead[k] = g.ahead[k + 1]; g.aheadTicket[k] = g.aheadTicket[k + 1]; }
    g.ahead[3].used = false;
    return 0;
}

// Trace the frames after `frameCount` ahead of the caller, up to tptSetHostLookahead of them: the reference's hosts call
// DrawTest(f), DrawTest(f + 1), ... with nothing else changing (TestWin.cpp:313-316, Renderer.mm:225, main.cpp:59-60); a
// frame alone on the GPU is bound by its longest paths (one frame in flight: 1.0 ms, three: 0.55 ms per frame).
int traceAhead(int frameCount, int w, int h, unsigned testFlags, unsigned long long key, int want)
{
    int have = 0;
    while (have < 4 && g.ahead[have].used) ++have;
    int nextFrame = have ? g.ahead[have - 1].frameCount + 1 : frameCount + 1;
    // every frame traced but not yet blended holds a slot (its colour buffer): this one plus the ones ahead must leave one
    // slot spare, whatever the hardware-queue probe clamped the pipeline to
    const int nSlots = effectiveOverlap();
    const int maxAhead = want < nSlots - 2 ? want : nSlots - 2;
    while (have < maxAhead) {
        Context::Ahead& A = g.ahead[have];
        A.frameCount = nextFrame; A.w = w; A.h = h; A.flags = testFlags; A.configKey = key;
        A.raySlot = (int)(g.frameSeq % (unsigned long long)Context::kMaxSlots);
        int rc = enqueueTrace(nextFrame, w, h, testFlags, g.dRaysAhead + A.raySlot, g.aheadTicket[have]);
        if (rc) return rc;
        A.used = g.aheadTicket[have].valid;
        if (!A.used) break;
        ++have;
        ++nextFrame;
    }
    return 0;
}

} // namespace

extern "C" {

int tptSetHostBufferMode(int hostBufferOnlyWrittenByDrawTest)
{
    g.hostTrust = hostBufferOnlyWrittenByDrawTest ? 1 : 0;
    g.tileSrc = nullptr; // next DrawTest uploads once
    return 0;
}

int tptSetStreamBatching(int enable)
{
    if (requireInit()) return -1;
    int rc = discardLookahead();
    if (rc) return rc;
    g.streamBatch = enable ? 1 : 0;
    return 0;
}

int tptSetHostLookahead(int frames)
{
    if (frames < 0 || frames > 3) return fail("tptSetHostLookahead: 0..3");
    if (g.inited) {
        int rc = discardLookahead();
        if (rc) return rc;
    }
    g.lookahead = frames;
    return 0;
}

int tptDraw(float time, int frameCount, int w, int h, float* backbuffer, int* outRayCount, unsigned testFlags)
{
    (void)time;
    if (requireInit()) return -1;
    if (!g.updated) return fail("tptDraw: call tptUpdate (UpdateTest) first");
    if (!backbuffer || w <= 0 || h <= 0) return fail("tptDraw: bad arguments");
    const int rows = localRows(h);
    const size_t rowBytes = (size_t)w * 4 * sizeof(float);
    const size_t need = rowBytes * (size_t)(rows > 0 ? rows : 1);
    if (need > g.frameCap) {
        int rc = discardLookahead();
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(g.stream));
        if (g.dFrame) HIPCHK(hipFree(g.dFrame));
        g.dFrame = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.dFrame), need));
        g.frameCap = need;
        g.tileSrc = nullptr;
    }
    const bool sharded = g.numParts > 1 && g.stripeRows > 0;
    const bool pipelined = effectiveOverlap() > 1;
    // What a traced frame depends on besides (frameCount, w, h, flags): scene, camera, spp, seed / fold mode, kernel variant,
    // sharding.  Every call that changes one of them bumps configEpoch; a pending scene change (tptSetScene, kFlagAnimate)
    // shows as sceneDirty / a pending scene set.
    const unsigned long long key = g.configEpoch;
    const bool stable = !g.sceneDirty && g.pendingSet < 0 && !(testFlags & TPT_FLAG_ANIMATE);

    // ---- 1. this frame's trace: traced ahead by an earlier call, or now
    struct DepthScope { // launches made from here share the machine with the frames traced ahead, not with a deep device-path pipeline
        explicit DepthScope(int d) { g.depthOverride = d; }
        ~DepthScope() { g.depthOverride = 0; }
    } depthScope(pipelined && stable ? 1 + (g.lookahead < 3 ? g.lookahead : 3) : 1);
    TraceTicket T;
    int raySlot = -1;
    const unsigned long long* rayPtr = nullptr;
    bool servedFromBatch = false;
    Context::HostCaller& HC = g.hostCaller;
    HC.streak = (frameCount == HC.frame + 1 && w == HC.w && h == HC.h && testFlags == HC.flags && key == HC.key) ? HC.streak + 1 : 0;
    HC.frame = frameCount; HC.w = w; HC.h = h; HC.flags = testFlags; HC.key = key;
    const bool batchRefused = HC.refusedKey == key && HC.refusedW == w && HC.refusedH == h;
    if (g.seedMode == SEED_ROW_SERIAL && stable && pipelined && !sharded && g.lookahead > 0 && rows > 0 && !g.mirror && !batchRefused) {
        // ---- 1r. the reference's own seed mode: a frame alone is `rows` lanes of work, so the frames AHEAD are traced as one
        //          launch (rows x frames lanes) and served one by one.  A batch is 32 frames of GPU work for one delivered
        //          frame, so it is only launched for a caller that has shown its pattern -- the third consecutive frame of one
        //          configuration (a one-shot DrawTest, or a host that jumps about, takes the plain path below).  A batch the pipeline
        //          refuses (frame wider than 8192, over 4 GiB of colour planes, not enough device memory) is retried at half the size, down to 2 frames; if nothing fits
        //          the configuration is served frame by frame: DrawTest never fails because of the look-ahead.
        auto matches = [&](const Context::RowSerialBatch& B) {
            return B.used && B.w == w && B.h == h && B.flags == testFlags && B.key == key && frameCount == B.firstFrame + B.next;
        };
        auto launch = [&](int which, int firstFrame) -> int {
            Context::RowSerialBatch& B = g.rsb[which];
            B.used = false;
            // (two banks of per-frame counters; the batch being served keeps its bank when it moves from [1] to [0])
            const int bank = (which == 1 && g.rsb[0].used && g.rsb[0].counterBase == 0) ? kMaxBatch : 0;
            // the batch behind one that is being served starts at THAT batch's size: a larger one would have to grow the colour
            // slots the first still reads (refused now) after draining the pipeline to find that out
            const int nMax = (which == 1 && g.rsb[0].used) ? g.rsb[0].n : kMaxBatch;
            for (int n = nMax; n >= 2; n /= 2) {
                if (w > 8192 || h > 8192 || (long long)rows * w * n > (1ll << 30) || (unsigned long long)rows * w * 16ull * n > (4ull << 30)) continue;
                B.firstFrame = firstFrame; B.n = n; B.next = 0; B.w = w; B.h = h; B.flags = testFlags; B.key = key;
                B.counterBase = bank;
                const int rc = enqueueTrace(firstFrame, w, h, testFlags, g.dRaysBatch + B.counterBase, B.T, B.n, 1);
                if (rc == 0) {
                    B.used = B.T.valid;
                    return 0;
                }
                if (rc != kRefused) return rc; // a real failure (HIP error, no scene): not something a smaller batch cures
            }
            if (which == 0) { HC.refusedKey = key; HC.refusedW = w; HC.refusedH = h; } // nothing fits: frame by frame from here on
            return 0;
        };
        if (matches(g.rsb[0])) {
            g.aheadHits++;
        } else {
            if (g.rsb[0].used || g.rsb[1].used) {
                int rc = discardLookahead();
                if (rc) return rc;
            }
            if (HC.streak >= 2) {
                int rc = discardLookahead();
                if (rc) return rc;
                if ((rc = launch(0, frameCount))) return rc;
            }
        }
        Context::RowSerialBatch& B = g.rsb[0];
        if (B.used) {
            // (the batch after this one is launched at once: holding it back until the first hit -- the batch above only completes
            //  when its slowest row has, 60-90 ms -- serialises the batches and costs the sequential caller 2.7x: 1.6 instead of
            //  4.3 Gray/s, profiles/r04/r04_evidence.log; the caller has shown three consecutive frames by now)
            if (!g.rsb[1].used && !(HC.refusedKey == key && HC.refusedW == w && HC.refusedH == h)) {
                int rc = launch(1, B.firstFrame + B.n);
                if (rc) return rc;
            }
            const int j = B.next;
            T = B.T;
            T.colour = B.T.colour + (size_t)j * (size_t)B.T.nPixels;
            T.lerpFac = B.T.lerp.v[j];
            T.batch = 1;
            rayPtr = g.dRaysBatch + B.counterBase + j;
            servedFromBatch = true;
            if (++B.next == B.n) { // the batch is used up with this frame: the one after it becomes current
                g.rsb[0] = g.rsb[1];
                g.rsb[0].counterBase = g.rsb[1].counterBase;
                g.rsb[1].used = false;
            }
        }
    }
    Context::Ahead& front = g.ahead[0];
    if (servedFromBatch) {
        // (nothing more to trace)
    } else if (front.used && front.frameCount == frameCount && front.w == w && front.h == h && front.flags == testFlags && front.configKey == key && stable) {
        int rc = takeAhead(T, raySlot);
        if (rc) return rc;
    } else {
        int rc = discardLookahead();
        if (rc) return rc;
        raySlot = (int)(g.frameSeq % (unsigned long long)Context::kMaxSlots);
        if ((rc = enqueueTrace(frameCount, w, h, testFlags, g.dRaysAhead + raySlot, T))) return rc;
    }
    // ---- 2. trace the next frames ahead (a wrong guess costs GPU time only)
    // (in the reference's own seed mode the batches above ARE the look-ahead: single frames traced ahead would be 60-90 ms of
    //  GPU work each, dropped again when the batch is launched -- only a configuration whose batch was refused gets them)
    const bool rowSerialBatches = g.seedMode == SEED_ROW_SERIAL && !batchRefused && !sharded && !g.mirror && rows > 0;
    if (pipelined && stable && T.valid && !servedFromBatch && !rowSerialBatches) {
        int rc = traceAhead(frameCount, w, h, testFlags, key, g.lookahead);
        if (rc) return rc;
    }
    if (!servedFromBatch) rayPtr = T.valid ? g.dRaysAhead + raySlot : nullptr;
    // ---- 3. the previous image: the host buffer is the source of truth (previous frame's RGB, caller-owned alpha) unless
    //         the caller has promised that only DrawTest writes it (tptSetHostBufferMode): then the device tile is, and the
    //         upload happens once per buffer.  Then blend and download.
    const bool upload = rows > 0 && !(g.hostTrust && g.tileSrc == backbuffer && g.tileW == w && g.tileH == h && frameCount != 0);
    if (upload) { g.tileSrc = backbuffer; g.tileW = w; g.tileH = h; }
    if (upload && !sharded && T.valid && T.pipelined && rows >= 64 && !g.mirror) {
        // Banded: rows in four bands, alternating between two streams, each band upload -> blend -> download, so that a
        // band's blend and download do not wait for the whole upload.  The caller's buffer is pageable (page-locking the
        // CALLER's memory is not ours to do -- it may be freed between calls), and a copy on pageable memory does not return
        // before it is done: the two directions do NOT overlap on the link (profiles/r03/r03_h2d_probe.log: 0.27-0.30 ms each
        // way at 50-55 GB/s, 0.28 ms for half up + half down "at once").  Going through a pinned staging buffer filled and
        // emptied by helper threads does overlap them and was tried in round 3: 0.74-0.78 instead of 0.80 ms per frame in a
        // plain process, 0.97-1.07 instead of 0.81 in one whose HIP context torch had initialised -- dropped.
        const int kBands = 4;
        HIPCHK(hipEventRecord(g.evBand, g.stream)); // (orders stream 2 behind everything earlier on g.stream)
        HIPCHK(hipStreamWaitEvent(g.hostStream2, g.evBand, 0));
        // Trace still running (nothing was traced ahead)?  Then all uploads go first, beside it; otherwise they are interleaved
        // with the downloads.  The query only picks the ORDER of the copies: the blends wait for the trace event either way
        // (an event query that said "done" too early made a blend read the colour buffer before its frame was in it).
        const bool traceDone = hipEventQuery(g.evTrace[T.slot]) == hipSuccess;
        (void)hipGetLastError();
        if (traceDone) {
            HIPCHK(hipStreamWaitEvent(g.stream, g.evTrace[T.slot], 0));
            HIPCHK(hipStreamWaitEvent(g.hostStream2, g.evTrace[T.slot], 0));
        }
        for (int pass = 0; pass < 2; ++pass) {
            for (int b = 0; b < kBands; ++b) {
                const int r0 = (int)((long long)rows * b / kBands), r1 = (int)((long long)rows * (b + 1) / kBands);
                hipStream_t st = (b & 1) ? g.hostStream2 : g.stream;
                char* hb = reinterpret_cast<char*>(backbuffer) + rowBytes * r0;
                float* db = g.dFrame + (size_t)r0 * w * 4;
                if (pass == 0) HIPCHK(hipMemcpyAsync(db, hb, rowBytes * (size_t)(r1 - r0), hipMemcpyHostToDevice, st));
                if (pass == 0 && !traceDone) continue;
                HIPCHK(tptLaunchResolve(db, T.colour + (size_t)r0 * w, (r1 - r0) * w, T.lerpFac, nullptr, g.dRays, nullptr, b == 0 ? rayPtr : nullptr, st));
                HIPCHK(hipMemcpyAsync(hb, db, rowBytes * (size_t)(r1 - r0), hipMemcpyDeviceToHost, st));
            }
            if (traceDone) break;
            if (pass == 0) { // uploads are on their way: now the blends wait for the trace
                HIPCHK(hipStreamWaitEvent(g.stream, g.evTrace[T.slot], 0));
                HIPCHK(hipStreamWaitEvent(g.hostStream2, g.evTrace[T.slot], 0));
            }
        }
        HIPCHK(hipEventRecord(g.evBandEnd, g.hostStream2));
        HIPCHK(hipStreamWaitEvent(g.stream, g.evBandEnd, 0));
        HIPCHK(hipEventRecord(g.evResolve[T.slot], g.stream));
        g.resolveRecorded[T.slot] = true;
    } else {
        if (upload) {
            int rc = uploadBackbuffer(backbuffer, w, h);
            if (rc) return rc;
        }
        int rc = enqueueResolve(T, g.dFrame, rayPtr);
        if (rc) return rc;
        if (rows > 0) {
            if (!sharded) {
                HIPCHK(hipMemcpyAsync(backbuffer, g.dFrame, rowBytes * rows, hipMemcpyDeviceToHost, g.stream));
            } else {
                for (int ly = 0; ly < rows; ly += g.stripeRows) {
                    int n = rows - ly < g.stripeRows ? rows - ly : g.stripeRows;
                    HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(backbuffer) + rowBytes * localToGlobal(ly),
                                          reinterpret_cast<const char*>(g.dFrame) + rowBytes * ly, rowBytes * n,
                                          hipMemcpyDeviceToHost, g.stream));
                }
            }
        }
    }
    unsigned long long frameRays = 0;
    if (T.valid) HIPCHK(hipMemcpyAsync(&frameRays, rayPtr, sizeof(frameRays), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    if (outRayCount) *outRayCount = (int)frameRays;
    return 0;
}

// Display conversion (Cpp/Emscripten/main.cpp:63-79): linear float tile -> RGBA8, top row first.
int tptDisplayRGBA8(const float* deviceTile, int w, int h, unsigned char* deviceRGBA)
{
    if (requireInit()) return -1;
    if (!deviceTile || !deviceRGBA || w <= 0 || h <= 0) return fail("tptDisplayRGBA8: bad arguments");
    HIPCHK(tptLaunchDisplay(deviceTile, deviceRGBA, w, h, g.stream));
    return 0;
}

int tptGetObjectCount(int* outCount, int* outObjectSize, int* outMaterialSize, int* outCamSize) // Test.cpp:369-375
{
    if (g.spheres.empty()) defaultScene(g.spheres, g.mats);
    if (outCount) *outCount = (int)g.spheres.size();
    if (outObjectSize) *outObjectSize = (int)sizeof(SpherePOD);
    if (outMaterialSize) *outMaterialSize = (int)sizeof(MaterialPOD);
    if (outCamSize) *outCamSize = (int)sizeof(CameraPOD);
    return 0;
}

int tptGetSceneDesc(void* outObjects, void* outMaterials, void* outCam, void* outEmissives, int* outEmissiveCount) // Test.cpp:377-384
{
    if (g.spheres.empty()) defaultScene(g.spheres, g.mats);
    if (g.sceneDirty || g.packed.nSpheres != (int)g.spheres.size()) packScene(g.spheres, g.mats, g.packed); // fills invRadius + emissive ids
    if (outObjects) memcpy(outObjects, g.spheres.data(), g.spheres.size() * sizeof(SpherePOD));
    if (outMaterials) memcpy(outMaterials, g.mats.data(), g.mats.size() * sizeof(MaterialPOD));
    if (outCam) memcpy(outCam, &g.cam, sizeof(CameraPOD));
    if (outEmissives && !g.packed.emissive.empty())
        memcpy(outEmissives, g.packed.emissive.data(), g.packed.emissive.size() * sizeof(int));
    if (outEmissiveCount) *outEmissiveCount = (int)g.packed.emissive.size();
    return 0;
}

// ---------------------------------------------------------------- multi-GPU inside the library
// One process per GPU.  The image's rows are dealt out in stripes round-robin (tptSetRowShard); every rank renders its
// stripes into its own resident tile; per frame ONE collective: ncclGather (rccl.h:745) of the blended tile + one extra
// row whose first 8 bytes are the rank's 64-bit ray counter, to rank 0, on a communication stream, from a ring of
// snapshots the resolve kernel itself writes (tptSetTileMirror) -- so the gather of frame f overlaps the tracing of the
// following frames.  Rank 0 de-interleaves the gathered tiles into the caller's image.  Replaces the row fan-out / join of
// DrawTest (Test.cpp:357-361) across GPUs; no Python, no torch: a C++ host that links this library shards by itself
// (examples/multi_gpu_host.cpp).
namespace {
int loadRccl()
{
    Context::Shard& S = g.shard;
    if (S.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
        if ((S.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!S.lib) return fail(std::string("tptComm: cannot load librccl: ") + dlerror());
    S.GetUniqueId = reinterpret_cast<decltype(S.GetUniqueId)>(dlsym(S.lib, "ncclGetUniqueId"));
    S.CommInitRank = reinterpret_cast<decltype(S.CommInitRank)>(dlsym(S.lib, "ncclCommInitRank"));
    S.CommDestroy = reinterpret_cast<decltype(S.CommDestroy)>(dlsym(S.lib, "ncclCommDestroy"));
    S.Gather = reinterpret_cast<decltype(S.Gather)>(dlsym(S.lib, "ncclGather"));
    S.GetErrorString = reinterpret_cast<decltype(S.GetErrorString)>(dlsym(S.lib, "ncclGetErrorString"));
    S.CommCount = reinterpret_cast<decltype(S.CommCount)>(dlsym(S.lib, "ncclCommCount"));
    S.CommUserRank = reinterpret_cast<decltype(S.CommUserRank)>(dlsym(S.lib, "ncclCommUserRank"));
    if (!S.GetUniqueId || !S.CommInitRank || !S.CommDestroy || !S.Gather || !S.GetErrorString || !S.CommCount || !S.CommUserRank)
        return fail("tptComm: librccl lacks a needed symbol");
    return 0;
}
int ncclFail(ncclResult_t r, const char* what)
{
    g.err = std::string(what) + ": " + (g.shard.GetErrorString ? g.shard.GetErrorString(r) : "RCCL error");
    return -3;
}
#define NCCLCHK(x)                                        \
    do {                                                  \
        ncclResult_t _r = (x);                            \
        if (_r != ncclSuccess) return ncclFail(_r, #x);   \
    } while (0)

int releaseShardBuffers()
{
    Context::Shard& S = g.shard;
    if (S.commStream) HIPCHK(hipStreamSynchronize(S.commStream));
    (void)hipFree(S.tile); S.tile = nullptr;
    (void)hipFree(S.gathered); S.gathered = nullptr;
    for (int k = 0; k < Context::Shard::kRing; ++k) { (void)hipFree(S.send[k]); S.send[k] = nullptr; S.sentRecorded[k] = false; }
    S.w = S.h = S.padRows = 0;
    return 0;
}
} // namespace

int tptCommGetUniqueId(void* out128)
{
    if (!out128) return fail("tptCommGetUniqueId: NULL");
    if (loadRccl()) return -1;
    ncclUniqueId id;
    NCCLCHK(g.shard.GetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return 0;
}

namespace {
int startShard(int nRanks, int rank, int stripeRows)
{
    Context::Shard& S = g.shard;
    S.nRanks = nRanks; S.rank = rank; S.stripeRows = stripeRows; S.frames = 0;
    HIPCHK(hipStreamCreateWithFlags(&S.commStream, hipStreamDefault)); // blocking, like the context's own stream: the assemble kernel writes the caller's image
    for (int k = 0; k < Context::Shard::kRing; ++k) {
        HIPCHK(hipEventCreateWithFlags(&S.evSnap[k], kOrderingEvent));
        HIPCHK(hipEventCreateWithFlags(&S.evSent[k], kOrderingEvent));
        S.sentRecorded[k] = false;
    }
    S.active = true;
    return tptSetRowShard(stripeRows, nRanks, rank);
}
} // namespace

int tptCommInit(const void* id128, int nRanks, int rank, int stripeRows)
{
    if (requireInit()) return -1;
    if (!id128 || nRanks < 1 || rank < 0 || rank >= nRanks || stripeRows < 1) return fail("tptCommInit: bad arguments");
    if (g.shard.active) return fail("tptCommInit: already initialised (tptCommDestroy first)");
    if (loadRccl()) return -1;
    Context::Shard& S = g.shard;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    NCCLCHK(S.CommInitRank(&S.comm, nRanks, id, rank));
    S.loopback = false;
    return startShard(nRanks, rank, stripeRows);
}

// Measurement aid: this process plays rank 0 of an nRanks-way sharded run on its own -- same tile, snapshot ring, events and
// assemble kernel as tptCommInit, a device copy of rank 0's slice standing in for the gather (the other ranks' rows stay
// zero).  What one GPU sustains as rank 0, RCCL apart; tools/shard_exchange_emu.py, bench.py --emulate-ranks.
int tptCommInitLoopback(int nRanks, int stripeRows)
{
    if (requireInit()) return -1;
    if (nRanks < 1 || stripeRows < 1) return fail("tptCommInitLoopback: bad arguments");
    if (g.shard.active) return fail("tptCommInitLoopback: already initialised (tptCommDestroy first)");
    g.shard.loopback = true;
    return startShard(nRanks, 0, stripeRows);
}

// What the communicator itself says about its size and this process's rank (ncclCommCount / ncclCommUserRank -- not the
// arguments tptCommInit was given), and whether it is the loopback stand-in.
int tptCommInfo(int* outRanks, int* outRank, int* outLoopback)
{
    Context::Shard& S = g.shard;
    if (!S.active) return fail("tptCommInfo: call tptCommInit first");
    int n = S.nRanks, r = S.rank;
    if (!S.loopback) {
        NCCLCHK(S.CommCount(S.comm, &n));
        NCCLCHK(S.CommUserRank(S.comm, &r));
    }
    if (outRanks) *outRanks = n;
    if (outRank) *outRank = r;
    if (outLoopback) *outLoopback = S.loopback ? 1 : 0;
    return 0;
}

int tptCommDestroy(void)
{
    Context::Shard& S = g.shard;
    if (!S.active) return 0;
    (void)discardLookahead();
    if (g.stream) (void)hipStreamSynchronize(g.stream);
    (void)releaseShardBuffers();
    (void)tptSetTileMirror(nullptr, nullptr);
    if (S.comm) NCCLCHK(S.CommDestroy(S.comm));
    S.comm = nullptr;
    S.active = S.loopback = false;
    for (int k = 0; k < Context::Shard::kRing; ++k) {
        if (S.evSnap[k]) (void)hipEventDestroy(S.evSnap[k]);
        if (S.evSent[k]) (void)hipEventDestroy(S.evSent[k]);
        S.evSnap[k] = S.evSent[k] = nullptr;
    }
    if (S.commStream) (void)hipStreamDestroy(S.commStream);
    S.commStream = nullptr;
    S.nRanks = 0;
    return tptSetRowShard(0, 1, 0);
}

// DrawTest for a frame sharded over the ranks of the communicator: asynchronous; `deviceImageOnRoot` (rank 0: w*h*4 floats in
// device memory, may be NULL elsewhere) holds frame f once tptShardedFinish (or a later call's gather) has completed.
int tptDrawSharded(float time, int frameCount, int w, int h, float* deviceImageOnRoot, unsigned testFlags)
{
    return tptDrawShardedBatch(time, frameCount, 1, w, h, deviceImageOnRoot, testFlags);
}

// nFrames consecutive frames per rank in one launch (tptDrawDeviceBatch), then ONE exchange: the image on rank 0 is that of
// the batch's last frame.  Same bits as nFrames tptDrawSharded calls; 1 / nFrames of the launches and gathers.
int tptDrawShardedBatch(float time, int frameCount, int nFrames, int w, int h, float* deviceImageOnRoot, unsigned testFlags)
{
    if (requireInit()) return -1;
    Context::Shard& S = g.shard;
    if (!S.active) return fail("tptDrawSharded: call tptCommInit first");
    if (w <= 0 || h <= 0 || nFrames < 1 || nFrames > kMaxBatch) return fail("tptDrawSharded: bad size / batch (1..32 frames)");
    if (S.rank == 0 && !deviceImageOnRoot) return fail("tptDrawSharded: rank 0 needs the image buffer");
    if (w != S.w || h != S.h) { // (re)allocate for this frame size: every rank the same padded tile height
        int rc = releaseShardBuffers();
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(g.stream));
        S.padRows = shardPadRows(h, S.stripeRows, S.nRanks); // rank 0 owns the most stripes; whole stripes
        const size_t rowBytes = (size_t)w * 4 * sizeof(float);
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&S.tile), rowBytes * (size_t)S.padRows));
        HIPCHK(hipMemsetAsync(S.tile, 0, rowBytes * (size_t)S.padRows, g.stream));
        for (int k = 0; k < Context::Shard::kRing; ++k) {
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&S.send[k]), shardSnapshotPixels(S.padRows, w) * sizeof(f4)));
            HIPCHK(hipMemsetAsync(S.send[k], 0, shardSnapshotPixels(S.padRows, w) * sizeof(f4), g.stream));
        }
        if (S.rank == 0) {
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&S.gathered), shardSnapshotPixels(S.padRows, w) * sizeof(f4) * (size_t)S.nRanks));
            HIPCHK(hipMemsetAsync(S.gathered, 0, shardSnapshotPixels(S.padRows, w) * sizeof(f4) * (size_t)S.nRanks, g.stream));
            HIPCHK(hipStreamSynchronize(g.stream)); // the communication stream writes it next
        }
        S.w = w; S.h = h;
    }
    const int k = shardRingSlot(S.frames, Context::Shard::kRing);
    S.frames++;
    // the snapshot this frame's resolve kernel writes must have left the GPU (gather of the frame that used it last)
    if (S.sentRecorded[k]) HIPCHK(hipStreamWaitEvent(g.stream, S.evSent[k], 0));
    const size_t tileFloats = shardCounterPixel(S.padRows, w) * 4;
    int rc = tptSetTileMirror(S.send[k], S.send[k] + tileFloats); // blended tile -> snapshot, ray counter -> first 8 bytes of the extra row
    if (rc) return rc;
    if ((rc = nFrames > 1 ? tptDrawDeviceBatch(time, frameCount, nFrames, w, h, S.tile, testFlags) : tptDrawDevice(time, frameCount, w, h, S.tile, testFlags))) return rc;
    HIPCHK(hipEventRecord(S.evSnap[k], g.stream));
    HIPCHK(hipStreamWaitEvent(S.commStream, S.evSnap[k], 0));
    const size_t count = shardSnapshotPixels(S.padRows, w) * 4;
    if (S.loopback) HIPCHK(hipMemcpyAsync(S.gathered, S.send[k], count * sizeof(float), hipMemcpyDeviceToDevice, S.commStream));
    else NCCLCHK(S.Gather(S.send[k], S.gathered, count, ncclFloat32, 0, S.comm, S.commStream));
    if (S.rank == 0) HIPCHK(tptLaunchAssemble(S.gathered, deviceImageOnRoot, w, h, S.stripeRows, S.nRanks, S.padRows, S.commStream));
    HIPCHK(hipEventRecord(S.evSent[k], S.commStream));
    S.sentRecorded[k] = true;
    return 0;
}

// Waits for every exchange enqueued so far; on rank 0 *outTotalRays = sum over the ranks of their ray counters as of the last
// gathered frame (exact 64-bit integers: they travel bit-cast in the float payload), elsewhere this rank's own.
int tptShardedFinish(int64_t* outTotalRays)
{
    if (requireInit()) return -1;
    Context::Shard& S = g.shard;
    if (!S.active) return fail("tptShardedFinish: call tptCommInit first");
    if (int rc = launchTailHelpers()) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipStreamSynchronize(S.commStream));
    long long total = 0;
    if (S.frames == 0 || !S.w) { // nothing gathered yet: this rank's own running total (the other ranks' are not known here)
        int64_t own = 0;
        int rc = tptRayCounterRead(&own);
        if (rc) return rc;
        if (outTotalRays) *outTotalRays = own;
        return 0;
    }
    if (S.rank == 0) {
        for (int r = 0; r < S.nRanks; ++r) {
            unsigned long long v = 0;
            const char* src = reinterpret_cast<const char*>(S.gathered) + sizeof(f4) * ((size_t)r * shardSnapshotPixels(S.padRows, S.w) + shardCounterPixel(S.padRows, S.w));
            HIPCHK(hipMemcpy(&v, src, sizeof(v), hipMemcpyDeviceToHost));
            total += (long long)v;
        }
    } else {
        const int k = shardRingSlot(S.frames - 1, Context::Shard::kRing);
        unsigned long long v = 0;
        HIPCHK(hipMemcpy(&v, reinterpret_cast<const char*>(S.send[k]) + sizeof(f4) * shardCounterPixel(S.padRows, S.w), sizeof(v), hipMemcpyDeviceToHost));
        total = (long long)v;
    }
    if (outTotalRays) *outTotalRays = total;
    return 0;
}

int tptGetLookaheadHits(long long* outHits)
{
    if (outHits) *outHits = g.aheadHits;
    return 0;
}

int tptGetPipelineInfo(int* outHwQueues, int* outOverlapEffective, int* outStreamDepth, int* outSlotReservations)
{
    if (outHwQueues) *outHwQueues = g.hwQueues;
    if (outOverlapEffective) *outOverlapEffective = effectiveOverlap();
    if (outStreamDepth) *outStreamDepth = g.streamDepth;
    if (outSlotReservations) *outSlotReservations = g.slotReservations;
    return 0;
}

int tptGetLaunchInfo(int* outBlocksPerCU, int* outLdsBytes, int* outGridBlocks, int* outNumCUs)
{
    if (outBlocksPerCU) *outBlocksPerCU = g.lastBlocksPerCU;
    if (outLdsBytes) *outLdsBytes = g.lastLds;
    if (outGridBlocks) *outGridBlocks = g.lastGrid;
    if (outNumCUs) *outNumCUs = g.numCUs;
    return 0;
}

#if defined(TPT_TEST_HOOKS) // ---- unit-test / profiling entry points (include/tpt_test_hooks.h): not in the product library
// debugging aid for the cost-ordered work distribution: copies the accumulated per-chunk ray counts and the order
// table given to the most recent launch (either pointer may be NULL); returns the number of chunks
int tptDebugChunkOrder(unsigned* outCost, unsigned* outOrder, int capacity)
{
    if (requireInit()) return -1;
    HIPCHK(hipStreamSynchronize(g.stream));
    for (int k = 0; k < Context::kMaxOverlap; ++k) HIPCHK(hipStreamSynchronize(g.traceStream[k]));
    int n = g.chunkCount < capacity ? g.chunkCount : capacity;
    if (n <= 0 || !g.dChunkCost) return 0;
    if (outCost) HIPCHK(hipMemcpy(outCost, g.dChunkCost, sizeof(unsigned) * n, hipMemcpyDeviceToHost));
    if (outOrder) HIPCHK(hipMemcpy(outOrder, g.dChunkOrder[g.lastOrderTable], sizeof(unsigned) * n, hipMemcpyDeviceToHost));
    return n;
}

int tptDebugStats(unsigned long long* out64, int reset)
{
    if (requireInit()) return -1;
    HIPCHK(hipStreamSynchronize(g.stream));
    int rc = out64 ? tptReadStats(out64) : 0;
    if (rc == -1) return fail("tptDebugStats: library not built with -DTPT_STATS (profiling build, tools/build_stats.sh)");
    if (rc) return fail("tptDebugStats: hipMemcpyFromSymbol failed");
    if (reset && tptResetStats()) return fail("tptDebugStats: reset failed");
    return 0;
}

int tptTestMath(int op, const float* a, const float* b, float* out, int n)
{
    if (requireInit()) return -1;
    if (!a || !out || n <= 0) return fail("tptTestMath: bad arguments");
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&da), sizeof(float) * n));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dout), sizeof(float) * n));
    HIPCHK(hipMemcpy(da, a, sizeof(float) * n, hipMemcpyHostToDevice));
    if (b) {
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&db), sizeof(float) * n));
        HIPCHK(hipMemcpy(db, b, sizeof(float) * n, hipMemcpyHostToDevice));
    }
    HIPCHK(tptLaunchMathTest(op, da, db, dout, n, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(out, dout, sizeof(float) * n, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return 0;
}

// The fast correctly-rounded sqrt / normalize scale of tpt_math.h against the compiler's expansions for EVERY bit pattern in
// [lo, hi] (op 0: tsqrt, op 1: trsqrt2); returns the mismatch count and the first offending inputs.
int tptTestMathExhaustive(int op, unsigned lo, unsigned hi, unsigned long long* outMismatches, unsigned* outFirst8)
{
    if (requireInit()) return -1;
    if (!outMismatches || !outFirst8 || hi < lo || op < 0 || op > 1) return fail("tptTestMathExhaustive: bad arguments");
    unsigned long long* dBad = nullptr;
    unsigned* dFirst = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dBad), sizeof(unsigned long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dFirst), sizeof(unsigned) * 8));
    HIPCHK(hipMemsetAsync(dBad, 0, sizeof(unsigned long long), g.stream));
    HIPCHK(hipMemsetAsync(dFirst, 0, sizeof(unsigned) * 8, g.stream));
    HIPCHK(tptLaunchMathExhaustive(op, lo, hi, dBad, dFirst, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(outMismatches, dBad, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(outFirst8, dFirst, sizeof(unsigned) * 8, hipMemcpyDeviceToHost));
    (void)hipFree(dBad); (void)hipFree(dFirst);
    return 0;
}

// Phase 1 on the matrix cores (phase1MatrixH) for n host rays against the current scene: candidate masks (sphere p at bit
// 63 - p) and / or the nearest hit through the filter + the exact test of its candidates, as the path-queue kernel runs it.
int tptTestMatrixFilter(const float* rays, unsigned long long* outMask, int* outId, float* outT, int n)
{
    if (requireInit()) return -1;
    if (!rays || (!outMask && !outId) || (outId && !outT) || n <= 0) return fail("tptTestMatrixFilter: bad arguments");
    if (g.sceneDirty || (g.curSet < 0 && g.pendingSet < 0)) {
        int rc = stageScene();
        if (rc) return rc;
    }
    {
        int rc = enqueueSceneUpload(g.stream);
        if (rc) return rc;
    }
    KernelArgs a;
    memset(&a, 0, sizeof(a));
    a.scene = deviceView();
    if (a.scene.mxR1 < 0)
        return fail("tptTestMatrixFilter: the current scene has no matrix table (more than 64 spheres, a sphere outside binary16 range, hit-spheres variant 3, or a build without the filter)");
    const int nPad = (n + 63) / 64 * 64;
    std::vector<float> padded((size_t)nPad * 6, 0.0f);
    memcpy(padded.data(), rays, sizeof(float) * 6 * (size_t)n);
    float *dr = nullptr, *dt = nullptr;
    int* di = nullptr;
    unsigned long long* dm = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dr), sizeof(float) * 6 * nPad));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dm), sizeof(unsigned long long) * nPad));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&di), sizeof(int) * nPad));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dt), sizeof(float) * nPad));
    HIPCHK(hipMemcpy(dr, padded.data(), sizeof(float) * 6 * nPad, hipMemcpyHostToDevice));
    HIPCHK(tptLaunchMatrixFilterTest(a, dr, dm, outId ? di : nullptr, dt, nPad, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    if (outMask) HIPCHK(hipMemcpy(outMask, dm, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    if (outId) {
        HIPCHK(hipMemcpy(outId, di, sizeof(int) * n, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(outT, dt, sizeof(float) * n, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dr); (void)hipFree(dm); (void)hipFree(di); (void)hipFree(dt);
    return 0;
}

// The matrix-core filter over the group bounds of the current (grouped) scene against the exact test of every member:
// outViolations = (ray, member) pairs the reference's discriminant accepts (discr > 0, Maths.cpp:176-178) whose group the
// filter dropped -- must be 0; outTouched = groups kept per ray (summed), outExact = exact line hits (summed).
int tptTestGroupFilter(const float* rays, int n, unsigned long long* outViolations, unsigned long long* outTouched, unsigned long long* outExact)
{
    if (requireInit()) return -1;
    if (!rays || n <= 0 || !outViolations) return fail("tptTestGroupFilter: bad arguments");
    if (g.sceneDirty || (g.curSet < 0 && g.pendingSet < 0)) {
        int rc = stageScene();
        if (rc) return rc;
    }
    int rc = enqueueSceneUpload(g.stream);
    if (rc) return rc;
    KernelArgs a;
    memset(&a, 0, sizeof(a));
    a.scene = deviceView();
    if (a.scene.nGroups <= 0 || a.scene.gmxTiles <= 0) return fail("tptTestGroupFilter: the current scene has no group-bound table (not grouped, a group too loose, or hit-spheres variant 2 / 3)");
    const int nPad = (n + 63) / 64 * 64;
    float* dr = nullptr;
    unsigned long long* dout = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dr), sizeof(float) * 6 * (size_t)nPad));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dout), sizeof(unsigned long long) * 4));
    HIPCHK(hipMemsetAsync(dr, 0, sizeof(float) * 6 * (size_t)nPad, g.stream));
    HIPCHK(hipMemsetAsync(dout, 0, sizeof(unsigned long long) * 4, g.stream));
    HIPCHK(hipMemcpyAsync(dr, rays, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice, g.stream));
    HIPCHK(tptLaunchGroupFilterTest(a, dr, n, nPad, dout, g.stream));
    unsigned long long h[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(h, dout, sizeof(h), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    *outViolations = h[0];
    if (outTouched) *outTouched = h[1];
    if (outExact) *outExact = h[2];
    (void)hipFree(dr); (void)hipFree(dout);
    return 0;
}

int tptTestHitSpheres(int hitSpheres, const float* rays, int* outId, float* outT, int n)
{
    if (requireInit()) return -1;
    if (!rays || !outId || !outT || n <= 0) return fail("tptTestHitSpheres: bad arguments");
    if (g.sceneDirty || (g.curSet < 0 && g.pendingSet < 0)) {
        int rc = stageScene();
        if (rc) return rc;
    }
    {
        int rc = enqueueSceneUpload(g.stream);
        if (rc) return rc;
    }
    float *dr = nullptr, *dt = nullptr;
    int* di = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dr), sizeof(float) * 6 * n));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dt), sizeof(float) * n));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&di), sizeof(int) * n));
    HIPCHK(hipMemcpy(dr, rays, sizeof(float) * 6 * n, hipMemcpyHostToDevice));
    KernelArgs a;
    memset(&a, 0, sizeof(a));
    a.scene = deviceView();
    HIPCHK(tptLaunchHitTest(a, hitSpheres == 1 ? HS_SIMPLE : HS_TWO_PHASE, dr, di, dt, n, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(outId, di, sizeof(int) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(outT, dt, sizeof(float) * n, hipMemcpyDeviceToHost));
    (void)hipFree(dr); (void)hipFree(dt); (void)hipFree(di);
    return 0;
}

#endif // TPT_TEST_HOOKS

} // extern "C"

// ---------------------------------------------------------------- the reference's C++ Test API (Test.h:10-17)
// Same names, signatures and C++ linkage, so `nm` shows the very symbols Test.cpp exports
// (_Z14InitializeTestv, _Z12ShutdownTestv, _Z10UpdateTestfiiij, _Z8DrawTestfiiiPfRij,
// _Z14GetObjectCountRiS_S_S_, _Z12GetSceneDescPvS_S_S_Pi).  The reference functions return void and
// have no error channel: a HIP failure is reported on stderr and the process aborts.
static void dieOn(int rc, const char* where)
{
    if (rc) {
        fprintf(stderr, "toypathtracer_hip: %s failed: %s\n", where, tptGetLastError());
        abort();
    }
}
void InitializeTest() { dieOn(tptInitialize(), "InitializeTest"); }
void ShutdownTest() { dieOn(tptShutdown(), "ShutdownTest"); }
void UpdateTest(float time, int frameCount, int screenWidth, int screenHeight, unsigned testFlags)
{
    dieOn(tptUpdate(time, frameCount, screenWidth, screenHeight, testFlags), "UpdateTest");
}
void DrawTest(float time, int frameCount, int screenWidth, int screenHeight, float* backbuffer, int& outRayCount, unsigned testFlags)
{
    dieOn(tptDraw(time, frameCount, screenWidth, screenHeight, backbuffer, &outRayCount, testFlags), "DrawTest");
}
void GetObjectCount(int& outCount, int& outObjectSize, int& outMaterialSize, int& outCamSize)
{
    dieOn(tptGetObjectCount(&outCount, &outObjectSize, &outMaterialSize, &outCamSize), "GetObjectCount");
}
void GetSceneDesc(void* outObjects, void* outMaterials, void* outCam, void* outEmissives, int* outEmissiveCount)
{
    dieOn(tptGetSceneDesc(outObjects, outMaterials, outCam, outEmissives, outEmissiveCount), "GetSceneDesc");
}
