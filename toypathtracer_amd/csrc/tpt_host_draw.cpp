// tpt_host_draw.cpp -- DrawTest on the caller's HOST backbuffer (Test.cpp:344-367, synchronous): look-ahead, row-serial batches, banded copies; the display conversion
// (one of the host runtime's translation units: tpt_context.h lists them)
#include "tpt_context.h"

using namespace tpt;
using namespace tpth;

namespace tpth {

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


} // namespace tpth

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
    if (int rc_ = flushShardDeferred()) return rc_;
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

} // extern "C"
