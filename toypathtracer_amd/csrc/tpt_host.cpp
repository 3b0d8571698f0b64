// tpt_host.cpp -- host runtime (context, initialisation, scene, setters) + C ABI (include/tpt_hip.h) + the reference's C++ Test API
// (include/tpt_test_api.h == Cpp/Source/Test.h:10-17) for the MI355X path tracer.
//
// Replaces, on the host side: InitializeTest/ShutdownTest (Test.cpp:240-253; the enkiTS scheduler is
// replaced by a HIP stream), UpdateTest (Test.cpp:302-342; scene prep stays on the host, results are
// uploaded to HBM when dirty), DrawTest (Test.cpp:344-367; the row fan-out becomes one kernel
// launch), GetObjectCount/GetSceneDesc (Test.cpp:369-384).
//
// There is deliberately NO CPU rendering path in this library: if HIP is unavailable every entry
// point fails loudly.
#include "tpt_context.h"

using namespace tpt;
using namespace tpth;

namespace tpth {

Context g;

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
    const size_t bSPairs = grouped ? P.spairs.size() * sizeof(float) : 0; // super-group bounds (second level over the groups)
    const size_t offSPairs = offAmat + align256(bAmat + 32);
    const size_t offGmat = offSPairs + align256(bSPairs + 32);
    // (only for a host that asked for it, tptSetKernelVariant(4, ..): DESIGN.md 2.2)
    const size_t bGmat = (grouped && g.useMatrix && g.groupMatrix && tptQueueGroupMatrixBounds() && P.gmxTiles > 0) ? P.gmatH.size() * sizeof(uint32_t) : 0;
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
    if (bSPairs) memcpy(S.stage + offSPairs, P.spairs.data(), bSPairs);
    S.offSPairs = offSPairs;
    S.nSuperPairs = bSPairs ? P.nSuperPairs : 0;
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
    sv.spairs = reinterpret_cast<const float*>(S->dev + S->offSPairs);
    sv.nSuperPairs = S->nSuperPairs;
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

} // namespace tpth

extern "C" {

const char* tptGetLastError(void) { return g.err.c_str(); }
const char* tptGetDeviceName(void) { return g.deviceName.c_str(); }

int tptInitialize(void)
{
    if (g.inited) return 0;
    // Frame pipelining wants one hardware queue per in-flight trace kernel; the ROCm runtime exposes 4 by default and
    // maps further streams onto them round-robin (3 streams then run slower than 2).  Only effective if the HIP
    // runtime has not been initialised yet by the host application; harmless otherwise.
    // 20, not more: this library's 18 streams, the null stream and one of the host's each get their own, and the process
    // stays below what the device runs side by side.  A process that holds MORE queues than that (measured on MI355X /
    // ROCm 7.2: 20 and 22 fine, 24 and up not) is time-sliced by the device's scheduler -- running waves are switched out
    // and back in -- which halves the frame rate (round 4).  A PERFORMANCE hint only: nothing the library computes depends on
    // it (rounds 5-6: the one kernel path that was not safe under time-slicing is no longer taken by default).
    setenv("GPU_MAX_HW_QUEUES", TPT_DEFAULT_HW_QUEUES, 0);
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
    if (int rc_ = flushShardDeferred()) return rc_;
    if (requireInit()) return -1;
    if (discardLookahead()) return -2;
    HIPCHK(hipStreamSynchronize(g.stream));
    g.stream = hipStream ? reinterpret_cast<hipStream_t>(hipStream) : g.ownStream;
    return 0;
}

int tptSetSamplesPerPixel(int spp)
{
    if (int rc_ = flushShardDeferred()) return rc_;
    if (spp < 1 || spp > 65536) return fail("tptSetSamplesPerPixel: spp out of range");
    if (spp == g.spp) return 0; // (a setter that changes nothing must not invalidate frames traced ahead)
    g.spp = spp;
    g.configEpoch++;
    return 0;
}
int tptSetConfig(int lightSampling, float animateSmoothing, int mitsubaCompare)
{
    if (int rc_ = flushShardDeferred()) return rc_;
    const int config = (lightSampling ? CFG_LIGHT_SAMPLING : 0) | (mitsubaCompare ? CFG_MITSUBA_COMPARE : 0);
    if (config == g.config && animateSmoothing == g.animateSmoothing) return 0;
    g.config = config;
    g.animateSmoothing = animateSmoothing;
    g.configEpoch++;
    return 0;
}
int tptSetSeedMode(int mode)
{
    if (int rc_ = flushShardDeferred()) return rc_;
    if (mode != SEED_ROW_SERIAL && mode != SEED_PER_PIXEL) return fail("tptSetSeedMode: 0 (ROW_SERIAL) or 1 (PER_PIXEL)");
    if (mode == g.seedMode) return 0;
    g.seedMode = mode;
    g.configEpoch++;
    return 0;
}
int tptSetFoldMode(int mode)
{
    if (int rc_ = flushShardDeferred()) return rc_;
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


int tptSetFrameOverlap(int frames)
{
    if (int rc_ = flushShardDeferred()) return rc_;
    if (frames < 1 || frames > Context::kMaxOverlap) return fail("tptSetFrameOverlap: 1..16");
    int rc = drainPipeline();
    if (rc) return rc;
    g.overlap = frames;
    return 0;
}

int tptSetKernelVariant(int hitSpheres, int persistent, int ldsScene)
{
    if (int rc_ = flushShardDeferred()) return rc_;
    if (hitSpheres < 0 || hitSpheres > 4)
        return fail("tptSetKernelVariant: hitSpheres 0 (two-phase) 1 (simple) 2 (two-phase, no groups) 3 (two-phase, VALU filter) 4 (two-phase, groups' bounds on the matrix cores)");
    if (hitSpheres == 4 && !tptQueueGroupMatrixBounds())
        return fail("tptSetKernelVariant: hitSpheres 4 (a grouped scene's bounds on the matrix cores) is compiled into the hooks build only "
                    "(libtoypathtracer_hip_hooks.so): waves that have run that path are not safe in a time-sliced process, DESIGN.md 2.2");
    if (persistent != 0 && persistent != 1 && persistent != 3)
        return fail("tptSetKernelVariant: persistent 3 (path queues, the default), 1 (lane refill) or 0 (one thread per pixel: the lane-refill kernel with re-filling off); the lane-sorting kernel (2) was removed in round 3");
    g.hs = hitSpheres == 1 ? HS_SIMPLE : HS_TWO_PHASE;
    const int allow = hitSpheres == 2 ? 0 : 1, matrix = hitSpheres == 3 ? 0 : 1; // 3: the packed VALU filter everywhere (no matrix-core table)
    const int groupMatrix = hitSpheres == 4 ? 1 : 0; // 4: as 0, and the bounds of a GROUPED scene on the matrix cores too (opt-in)
    if (allow != g.allowGroups || matrix != g.useMatrix || groupMatrix != g.groupMatrix) {
        g.allowGroups = allow;
        g.useMatrix = matrix;
        g.groupMatrix = groupMatrix;
        g.sceneDirty = true; // the staged scene set carries (or not) the grouped arrays / the matrix table
    }
    g.persist = persistent; // 3 = path queues (default), 1 = lane-refill kernel, 0 = the same kernel with re-filling off (one thread per pixel: the north_star's shape, kept for A/B runs)
    g.configEpoch++;
    g.ldsScene = ldsScene < 0 ? -1 : (ldsScene ? 1 : 0);
    return 0;
}

int tptSetScene(const void* spheres, const void* materials, int count)
{
    if (int rc_ = flushShardDeferred()) return rc_;
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
    if (int rc_ = flushShardDeferred()) return rc_;
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
    // sharded frames that were accepted but deferred depend on the camera and the scene as they are NOW: out they go before either changes
    if (g.shard.pendCount > 0 && (screenWidth != g.shard.pendW || screenHeight != g.shard.pendH || (testFlags & TPT_FLAG_ANIMATE)))
        if (int rc_ = flushShardDeferred()) return rc_;
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

int tptGetSceneInfo(int* outSpheres, int* outGroups, int* outBoundsOnMatrixCores)
{
    if (requireInit()) return -1;
    if (g.sceneDirty || g.packed.nSpheres != (int)g.spheres.size()) packScene(g.spheres, g.mats, g.packed);
    const PackedScene& P = g.packed;
    const bool grouped = g.allowGroups && P.nGroups > 0; // (the same decisions stageScene takes for the next upload)
    if (outSpheres) *outSpheres = P.nSpheres;
    if (outGroups) *outGroups = grouped ? P.nGroups : 0;
    if (outBoundsOnMatrixCores) *outBoundsOnMatrixCores = (grouped && g.useMatrix && g.groupMatrix && tptQueueGroupMatrixBounds() && P.gmxTiles > 0) ? 1 : 0;
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

} // extern "C"

// ---------------------------------------------------------------- the reference's C++ Test API (Test.h:10-17)
// Same names, signatures and C++ linkage, so `nm` shows the very symbols Test.cpp exports
// (_Z14InitializeTestv, _Z12ShutdownTestv, _Z10UpdateTestfiiij, _Z8DrawTestfiiiPfRij,
// _Z14GetObjectCountRiS_S_S_, _Z12GetSceneDescPvS_S_S_Pi).  The reference functions return void and
// have no error channel: by default a failure is reported on stderr and the process aborts -- there is no CPU path to fall back to, and
// a frame that silently was not rendered is worse than a stop.  A host that wants to decide itself installs a handler
// (tptSetErrorHandler): it is called with the entry point's name and tptGetLastError()'s text, and the call then RETURNS without
// having rendered (DrawTest leaves the buffer alone and reports 0 rays).
static tptErrorHandler g_errorHandler = nullptr;
extern "C" int tptSetErrorHandler(tptErrorHandler handler)
{
    g_errorHandler = handler;
    return 0;
}
static bool dieOn(int rc, const char* where)
{
    if (!rc) return false;
    if (g_errorHandler) {
        g_errorHandler(where, tptGetLastError());
        return true;
    }
    fprintf(stderr, "toypathtracer_hip: %s failed: %s\n", where, tptGetLastError());
    abort();
}
void InitializeTest() { dieOn(tptInitialize(), "InitializeTest"); }
void ShutdownTest() { dieOn(tptShutdown(), "ShutdownTest"); }
void UpdateTest(float time, int frameCount, int screenWidth, int screenHeight, unsigned testFlags)
{
    dieOn(tptUpdate(time, frameCount, screenWidth, screenHeight, testFlags), "UpdateTest");
}
void DrawTest(float time, int frameCount, int screenWidth, int screenHeight, float* backbuffer, int& outRayCount, unsigned testFlags)
{
    if (dieOn(tptDraw(time, frameCount, screenWidth, screenHeight, backbuffer, &outRayCount, testFlags), "DrawTest")) outRayCount = 0;
}
void GetObjectCount(int& outCount, int& outObjectSize, int& outMaterialSize, int& outCamSize)
{
    dieOn(tptGetObjectCount(&outCount, &outObjectSize, &outMaterialSize, &outCamSize), "GetObjectCount");
}
void GetSceneDesc(void* outObjects, void* outMaterials, void* outCam, void* outEmissives, int* outEmissiveCount)
{
    dieOn(tptGetSceneDesc(outObjects, outMaterials, outCam, outEmissives, outEmissiveCount), "GetSceneDesc");
}

