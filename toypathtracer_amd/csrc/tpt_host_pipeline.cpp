// tpt_host_pipeline.cpp -- one frame through the pipeline: plan, per-slot buffers, trace launch on its own stream, the ordered blend; tail helpers; tptDrawDevice / tptDrawDeviceBatch and the blocking calls (replaces the fan-out / join of DrawTest, Test.cpp:344-367)
// (one of the host runtime's translation units: tpt_context.h lists them)
#include "tpt_context.h"

using namespace tpt;
using namespace tpth;

namespace tpth {

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
        // ... and no ticket may keep pointing into a buffer that is gone (frames traced ahead, an open stream batch, the row-serial
        // batches): drop them all -- a later takeAhead / batch serve would blend from freed memory, and every later enqueueTrace
        // would be refused with "colour slots are held by frames traced ahead".  (Everything was drained before the buffers grew.)
        g.rsb[0].used = g.rsb[1].used = false;
        for (int k = 0; k < 4; ++k) g.ahead[k].used = false;
        g.sbatch.used = false;
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
    // grouped scene on the path-queue kernel: the second level of the bounds filter reads the groups' pair records per lane -- from LDS
    // when they fit the area the grouped instantiation's smaller path pool leaves (<= 544 groups), else from global memory; the host
    // that asked for the FLAT filter (hitSpheres variant 3: the A/B) or for the matrix cores (variant 4) gets neither
    a.ldsGroupPairs = -1;
    if (P.queued && !P.ldsScene && a.scene.nGroups > 0 && a.scene.gmxTiles == 0 && a.scene.nSuperPairs > 0 && g.useMatrix)
        a.ldsGroupPairs = tptQueueGroupPairsInLds(a.scene.nGroups, a.scene.nSuperPairs);
    P.lds = P.queued ? tptQueueLdsBytes(a, P.ldsScene) : ldsV1;
    if (P.queued && a.ldsGroupPairs > 0 && 160 * 1024 / (P.lds + 256) < 2) {
        // the groups' bounds in LDS would cost the second workgroup per CU (many lights beside them): second level from global memory
        a.ldsGroupPairs = 0;
        P.lds = tptQueueLdsBytes(a, P.ldsScene);
    }
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
    a.noRefill = (g.persist == 0 && !P.rowSerial && !P.queued) ? 1 : 0; // (one thread per pixel: tptSetKernelVariant persistent 0)
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


// First half of a frame: plan, buffers, trace kernel on the slot's stream.  `frameRays`: where the kernel adds its ray
// count (the context's counter, or a per-slot one for frames that are traced ahead of their DrawTest call).
int enqueueTrace(int frameCount, int w, int h, unsigned testFlags, unsigned long long* frameRays, TraceTicket& T, int batch, int rayStride)
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


} // namespace tpth

extern "C" {

int tptDrawDevice(float time, int frameCount, int w, int h, float* deviceTile, unsigned testFlags)
{
    (void)time; // stored but never read by the reference either (Test.cpp:257,347)
    // (a host that mixes sharded and plain frames on one context: sharded frames accepted earlier go out first, in call order.  The
    //  sharded path itself comes through here with nothing pending.)
    if (int rc_ = flushShardDeferred()) return rc_;
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
    if (int rc_ = flushShardDeferred()) return rc_; // (see tptDrawDevice)
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
    if (int rc_ = flushShardDeferred()) return rc_;
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
    if (int rc_ = flushShardDeferred()) return rc_;
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
    if (int rc_ = flushShardDeferred()) return rc_;
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
    if (int rc_ = flushShardDeferred()) return rc_;
    if (requireInit()) return -1;
    HIPCHK(hipEventRecord(g.ev1, g.stream));
    if (int rc = launchTailHelpers()) return rc;
    HIPCHK(hipEventSynchronize(g.ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, g.ev0, g.ev1));
    if (outMs) *outMs = ms;
    return 0;
}

} // extern "C"
