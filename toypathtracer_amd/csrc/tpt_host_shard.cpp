// tpt_host_shard.cpp -- multi-GPU inside the library: one process per GPU, RCCL loaded on first use, one gather per frame
// (one of the host runtime's translation units: tpt_context.h lists them)
#include "tpt_context.h"

using namespace tpt;
using namespace tpth;

namespace tpth {
// The sharded frames tptDrawSharded has accepted but not issued (exchange interval): as one batch, now.  Called by every entry point that
// is about to change what a frame depends on, and by the ones that wait.
int flushShardDeferred()
{
    Context::Shard& S = g.shard;
    if (!S.active || S.pendCount <= 0) return 0;
    const int first = S.pendFirst, n = S.pendCount;
    S.pendCount = 0; // (first: the calls below pass through entry points that flush)
    const int keep = S.exchangeEvery;
    S.exchangeEvery = 1; // issue exactly these n frames, nothing deferred again
    const int rc = tptDrawShardedBatch(S.pendTime, first, n, S.pendW, S.pendH, S.lastImage, S.pendFlags);
    S.exchangeEvery = keep;
    return rc;
}
} // namespace tpth

extern "C" {
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

namespace {
// gather + de-interleave of snapshot k, behind whatever g.stream holds now
int enqueueExchange(int k)
{
    Context::Shard& S = g.shard;
    HIPCHK(hipEventRecord(S.evSnap[k], g.stream));
    HIPCHK(hipStreamWaitEvent(S.commStream, S.evSnap[k], 0));
    const size_t count = shardSnapshotPixels(S.padRows, S.w) * 4;
    if (S.loopback) HIPCHK(hipMemcpyAsync(S.gathered, S.send[k], count * sizeof(float), hipMemcpyDeviceToDevice, S.commStream));
    else NCCLCHK(S.Gather(S.send[k], S.gathered, count, ncclFloat32, 0, S.comm, S.commStream));
    if (S.rank == 0) HIPCHK(tptLaunchAssemble(S.gathered, S.lastImage, S.w, S.h, S.stripeRows, S.nRanks, S.padRows, S.commStream));
    HIPCHK(hipEventRecord(S.evSent[k], S.commStream));
    S.sentRecorded[k] = true;
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
    S.pendCount = 0; // (frames accepted but never waited for are dropped with the communicator: the collective needs every rank)
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
    S.lastImage = deviceImageOnRoot;
    // One frame per call: issue it now, or defer it until k frames of the same configuration can go out as one batch (EVERY rank
    // takes the same decision: it depends on the frame's shape, the flags and the call sequence only).
    int every = S.exchangeEvery;
    if (every <= 0) {
        const long long samples = (long long)shardPadRows(h, S.stripeRows, S.nRanks) * w * g.spp;
        every = (S.nRanks <= 1 || (testFlags & TPT_FLAG_ANIMATE) || samples >= 2400000) ? 1 : (samples >= 1200000 ? 2 : samples >= 600000 ? 4 : 8);
    }
    if (every > kMaxBatch) every = kMaxBatch;
    if (S.pendCount > 0 && !(nFrames == 1 && every > 1 && frameCount == S.pendFirst + S.pendCount && w == S.pendW && h == S.pendH && testFlags == S.pendFlags)) {
        int rc = flushShardDeferred(); // not the continuation of what is pending: that goes out first
        if (rc) return rc;
    }
    if (nFrames == 1 && every > 1) {
        if (S.pendCount == 0) { S.pendFirst = frameCount; S.pendW = w; S.pendH = h; S.pendFlags = testFlags; S.pendTime = time; }
        if (++S.pendCount < every) return 0;
        frameCount = S.pendFirst; nFrames = S.pendCount; time = S.pendTime; // the k-th frame: the whole batch goes out now
        S.pendCount = 0;
    }
    const int k = shardRingSlot(S.frames, Context::Shard::kRing);
    S.frames++;
    // the snapshot this frame's resolve kernel writes must have left the GPU (gather of the frame that used it last)
    if (S.sentRecorded[k]) HIPCHK(hipStreamWaitEvent(g.stream, S.evSent[k], 0));
    const size_t tileFloats = shardCounterPixel(S.padRows, w) * 4;
    int rc = tptSetTileMirror(S.send[k], S.send[k] + tileFloats); // blended tile -> snapshot, ray counter -> first 8 bytes of the extra row
    if (rc) return rc;
    if ((rc = nFrames > 1 ? tptDrawDeviceBatch(time, frameCount, nFrames, w, h, S.tile, testFlags) : tptDrawDevice(time, frameCount, w, h, S.tile, testFlags))) return rc;
    return enqueueExchange(k);
}

// How many consecutive frames tptDrawSharded collects into one launch + blend + exchange: 0 = automatic (1 for tiles of 2.4 M samples
// or more and for animated scenes, 2 / 4 / 8 below 2.4 / 1.2 / 0.6 M), k = 1..32.  Every rank must choose the same.
int tptSetShardExchangeInterval(int k)
{
    if (k < 0 || k > kMaxBatch) return fail("tptSetShardExchangeInterval: 0 (automatic) or 1..32 frames");
    if (int rc = flushShardDeferred()) return rc;
    g.shard.exchangeEvery = k;
    return 0;
}

// Waits for every exchange enqueued so far; on rank 0 *outTotalRays = sum over the ranks of their ray counters as of the last
// gathered frame (exact 64-bit integers: they travel bit-cast in the float payload), elsewhere this rank's own.
int tptShardedFinish(int64_t* outTotalRays)
{
    if (requireInit()) return -1;
    Context::Shard& S = g.shard;
    if (!S.active) return fail("tptShardedFinish: call tptCommInit first");
    if (int rc = flushShardDeferred()) return rc; // frames accepted but not issued yet go out now
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

} // extern "C"
