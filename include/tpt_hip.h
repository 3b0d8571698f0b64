/* tpt_hip.h -- C ABI of the MI355X (gfx950, HIP) implementation of ToyPathTracer's
 * Trace / HitWorld / Scatter hot path.
 *
 * Drop-in boundary: the first block mirrors, one to one, the reference's "Test API"
 * (/root/reference/Cpp/Source/Test.h:10-17) that every reference host links against
 * (Cpp/Windows/TestWin.cpp:76,258,265,315-316; Cpp/Apple/Renderer.mm:155,181,225,234;
 * Cpp/Emscripten/main.cpp:59-60).  The shared library ALSO exports those six functions with the
 * reference's C++ linkage and exact signatures (see tpt_test_api.h), so a host compiled against the
 * reference's own Test.h links against libtoypathtracer_hip.so instead of Test.cpp+Maths.cpp+enkiTS
 * without source changes.  Precedent for an extern "C" veneer over this API in the reference:
 * Cpp/Emscripten/main.cpp:46-61.
 *
 * Everything is plain C: pointers, ints, floats.  No HIP / torch types appear in any signature
 * (streams and device buffers travel as void* / float*).  All functions return 0 on success and a
 * negative code on failure (tptGetLastError() has the text); nothing throws across the boundary.
 * The reference API itself has no error channel (all void): the C++-linkage wrappers print the error
 * to stderr and abort() -- there is NO CPU fallback.
 *
 * Not thread-safe / not re-entrant, like the reference (global scheduler + global scene,
 * Test.cpp:13,34,46,66-69,237).  One context per process == one GPU per process.
 */
#ifndef TPT_HIP_H
#define TPT_HIP_H
#include <stdint.h>

/* The library is built with -fvisibility=hidden: exactly the functions declared here (and the six C++ symbols of
 * tpt_test_api.h) are exported, nothing else (tests/test_abi.py compares `nm -D` with these headers). */
#if defined(__GNUC__)
#define TPT_API __attribute__((visibility("default")))
#else
#define TPT_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- TestFlags, Test.h:4-8 */
enum { TPT_FLAG_ANIMATE = 1 << 0, TPT_FLAG_PROGRESSIVE = 1 << 1 };

/* ================= 1. the reference Test API (Test.h:10-17), C spelling ================= */

/* InitializeTest(), Test.h:10 / Test.cpp:240-246.  Picks the HIP device (env TPT_DEVICE, else
 * LOCAL_RANK, else 0), creates the stream, uploads the built-in 46-sphere scene. */
TPT_API int tptInitialize(void);
/* ShutdownTest(), Test.h:11 / Test.cpp:248-253. */
TPT_API int tptShutdown(void);
/* UpdateTest(time, frameCount, screenWidth, screenHeight, testFlags), Test.h:13 / Test.cpp:302-342:
 * animate spheres 1 and 8 if TPT_FLAG_ANIMATE, derive 1/r and r^2, emissive list, camera;
 * uploads the scene arrays when they changed. */
TPT_API int tptUpdate(float time, int frameCount, int screenWidth, int screenHeight, unsigned testFlags);
/* DrawTest(time, frameCount, w, h, backbuffer, outRayCount, testFlags), Test.h:14 / Test.cpp:344-367.
 * `backbuffer` is a HOST pointer to w*h*4 floats, read-modify-written in place (RGB blended with the
 * previous contents, alpha untouched); synchronous: on return the frame and *outRayCount are final.
 * With row sharding active (tptSetRowShard) only this rank's rows are touched. */
TPT_API int tptDraw(float time, int frameCount, int screenWidth, int screenHeight, float* backbuffer, int* outRayCount,
            unsigned testFlags);
/* The host-pointer path above keeps the reference's contract (synchronous, the caller's buffer read-modify-written in
 * place); three things make it fast, none changes a byte of the result:
 *  - the upload of the previous image, the blend and the download are done in four row bands on two streams, so that a
 *    band's blend and download do not wait for the whole upload.  The caller's memory is NOT page-locked: it is the
 *    caller's to free between calls, and both directions run at link speed from pageable memory (measured: 0.27-0.30 ms
 *    each for 1280x720; they do not overlap -- a copy on pageable memory returns when it is done);
 *  - tptSetHostBufferMode(1): the caller promises that nobody but DrawTest writes the backbuffer between calls (true of
 *    every reference host: TestWin.cpp:73-74,315-316; Renderer.mm:225; Emscripten/main.cpp:59-60) -- the device-resident
 *    accumulation tile is then the source of truth and the buffer is uploaded once per buffer / size / frameCount == 0
 *    instead of every frame.  Default 0: upload every frame, exactly as the reference's semantics demand;
 *  - tptSetHostLookahead(n), default 2: after DrawTest(f) the library traces frames f+1 .. f+n AHEAD, guessing that the
 *    host goes on with the same size / flags / scene (what every reference host does); the next DrawTest then only
 *    blends and downloads.  A frame alone on the GPU is bound by its longest paths (1.0 ms at 1280x720x4); with three in
 *    flight the pipeline delivers one every 0.55 ms.  A wrong guess (other frame number, size, flags, scene, spp, ...) only
 *    costs GPU time: the frames traced ahead are dropped and the frame is traced again.  Never used with kFlagAnimate.
 *    In seed mode 0 (the reference's own pixels) any n > 0 means: this frame and the 31 after it as one batched launch, the
 *    batch after that as soon as this one is being served.
 *    The same look-ahead serves tptDrawDevice for a SYNCHRONOUS caller -- one whose previous frame has already been blended
 *    when its next call arrives, twice in a row, for consecutive frames of one configuration (a caller that streams frames
 *    never meets that and is unaffected): 0.98 -> ~0.55 ms per 1280x720x4 frame for a host that waits for every frame. */
TPT_API int tptSetHostBufferMode(int hostBufferOnlyWrittenByDrawTest);
TPT_API int tptSetHostLookahead(int frames);
/* how many frames were found traced ahead when their DrawTest / tptDrawDevice call arrived (monotonic; diagnostics, tests) */
TPT_API int tptGetLookaheadHits(long long* outHits);
/* GetObjectCount / GetSceneDesc, Test.h:16-17 / Test.cpp:369-384: sizes are 20 / 36 / 88 bytes and
 * the copies are byte-compatible with the reference's Sphere / Material / Camera structs. */
TPT_API int tptGetObjectCount(int* outCount, int* outObjectSize, int* outMaterialSize, int* outCamSize);
TPT_API int tptGetSceneDesc(void* outObjects, void* outMaterials, void* outCam, void* outEmissives, int* outEmissiveCount);

/* ================= 2. what the reference fixes at compile time, as run-time state ================= */

/* DO_SAMPLES_PER_PIXEL, Config.h:22 (default 4). */
TPT_API int tptSetSamplesPerPixel(int spp);
/* DO_LIGHT_SAMPLING (default 1), DO_ANIMATE_SMOOTHING (default 0.9f), DO_MITSUBA_COMPARE (default 0), Config.h:23-25 /
 * Test.cpp:95,143-145,209-214,226-227,273-274,312-313.  Without light sampling Lambert hits shoot no shadow rays and
 * emission is never suppressed; "Mitsuba compare" is the reference's only correctness method (readme.md:30): metal
 * roughness 0, constant sky (0.15, 0.21, 0.3), aperture 0 (takes effect at the next tptUpdate). */
TPT_API int tptSetConfig(int lightSampling, float animateSmoothing, int mitsubaCompare);
/* RNG seeding.  0 = ROW_SERIAL: one XorShift stream per image row carried along x (Test.cpp:280);
 * bit-identical to the reference CPU image.  A frame alone is parallel over rows only (720 lanes of work), but rows AND
 * frames are independent streams: for a static scene DrawTest / tptDraw trace the next 32 frames ahead as ONE launch (rows x
 * frames lanes) and serve them one by one (3.7 ms instead of 60-90 ms per 1280x720x4 frame), and tptDrawDeviceBatch takes up
 * to 32 frames per call (8-10 Gray/s).
 * 1 = PER_PIXEL (default): one stream per pixel, the reference's own GPU formula
 * (Cpp/Windows/ComputeShader.hlsl:380); parallel over pixels. */
TPT_API int tptSetSeedMode(int mode);
/* Colour fold.  0 = RECURSIVE (default): matE + lightE + attenuation*Trace(...) nesting of
 * Test.cpp:216, bit-identical colours.  1 = FORWARD: radiance += throughput*e (same rays, colours
 * equal up to rounding, no LDS bounce stack). */
TPT_API int tptSetFoldMode(int mode);
/* Replace the static scene tables (Test.cpp:13-31, 46-64).  spheres: count x 20 B {center xyz, radius,
 * invRadius(ignored)}; materials: count x 36 B {int type; albedo xyz; emissive xyz; roughness; ri}.
 * count <= 0 or NULL restores the built-in scene. */
TPT_API int tptSetScene(const void* spheres, const void* materials, int count);
/* Camera ctor arguments (Maths.h:418; defaults Test.cpp:309-319).  NULL lookFrom restores defaults. */
TPT_API int tptSetCamera(const float* lookFrom, const float* lookAt, float vfovDegrees, float aperture, float focusDist);

/* ================= 3. device-resident / multi-GPU path ================= */

/* Use an existing HIP stream (hipStream_t passed as void*, e.g. torch.cuda.current_stream().cuda_stream).
 * NULL -> the context's own stream. */
TPT_API int tptSetStream(void* hipStream);
/* Row sharding for one-process-per-GPU rendering: the image's rows are dealt out in stripes of
 * `stripeRows` rows, round-robin over `numParts` ranks; this context renders the stripes of `part`
 * into a COMPACT local tile (tptLocalRowCount(h) rows).  Seeds depend on the global (x,y) only, so
 * the union of the tiles is bit-identical to a 1-GPU render.  (0,1,0) or numParts<=1 disables. */
TPT_API int tptSetRowShard(int stripeRows, int numParts, int part);
TPT_API int tptLocalRowCount(int screenHeight);
/* global image row of local tile row `localRow` */
TPT_API int tptLocalRowToGlobal(int localRow);
/* Asynchronous draw into a DEVICE buffer holding this rank's tile: localRows*w*4 floats, the
 * accumulation buffer stays resident in HBM across frames.  Enqueued on the context's stream;
 * returns immediately.  Ray counts accumulate in a device counter (tptRayCounterRead). */
TPT_API int tptDrawDevice(float time, int frameCount, int screenWidth, int screenHeight, float* deviceTile, unsigned testFlags);
/* Several frames per launch: frames firstFrame .. firstFrame + nFrames - 1 of the scene and camera as of the last tptUpdate
 * (what the reference's main loop renders while nothing moves: TestWin.cpp:313-316 with kFlagAnimate off), traced by ONE
 * kernel launch and blended into the tile in frame order by one more.  Bit-identical to nFrames tptDrawDevice calls; a
 * launch's fixed costs (pool ramp-up and drain, no launch shorter than its longest pixel, queue latencies) are paid once
 * per batch instead of once per frame -- what bounds small frames and tiles of a sharded frame.  Path-queue kernel only
 * (the default); frames up to 8192 x 8192; kFlagAnimate is refused (the scene changes every frame). */
TPT_API int tptDrawDeviceBatch(float time, int firstFrame, int nFrames, int screenWidth, int screenHeight, float* deviceTile, unsigned testFlags);
/* Frame pipelining of the asynchronous path: the trace kernels of up to `frames` consecutive tptDrawDevice
 * calls may be in flight at once (each on its own internal stream, writing its own per-frame colour
 * buffer); the progressive blend into the tile (Test.cpp:293-295) is a separate, ordered kernel on the
 * context's stream, so results are bit-identical to frames=1.  1..16, default 16: the tail of frame f (a few long
 * paths) overlaps the following frames, and each launch takes only its share of the machine (2/frames-in-flight of
 * the resident workgroups; a caller that synchronises every frame gets whole-machine launches).
 * Needs one hardware queue per in-flight kernel: tptInitialize sets GPU_MAX_HW_QUEUES=20 if the HIP runtime has
 * not been initialised yet (ROCm's default of 4 makes 3 streams slower than 2; more than ~22 queues in one process are
 * time-sliced by the device: slower, and see INTEGRATION.md "Streams are not free"), then MEASURES how many streams really
 * run side by side and clamps the pipeline to that (tptGetPipelineInfo).  Twice as many frames may be ENQUEUED ahead
 * (frames f and f + frames share a stream). */
TPT_API int tptSetFrameOverlap(int frames);
/* Stream batching (default ON since round 4; tptSetStreamBatching(0) or env TPT_STREAM_BATCH=0 turns it off): a caller that streams consecutive frames of one static configuration with SMALL frames -- fewer
 * than 2.4 M samples (rows x width x spp): tiles of a sharded frame, 640x360 -- is bound by the latency of a launch (no launch is
 * shorter than its longest pixel's sequential samples), not by arithmetic.  For such a caller tptDrawDevice / tptDrawSharded trace
 * the frames of the next 1-7 calls in the SAME launch (2 / 4 / 8 frames per launch for halves / quarters / eighths of 1280x720x4)
 * and every later call only blends its own colour plane: each frame is still delivered, in order, with its own ray count (the
 * counter and the mirrored snapshot are exact per frame), bit-identical to one launch per frame.  A call that does not continue
 * the sequence (other frame number, size, flags, scene, ...) drops the unserved planes: GPU time only. */
TPT_API int tptSetStreamBatching(int enable);
/* Display conversion of a device-resident FULL image (w*h float4, row 0 = bottom) into w*h RGBA8 in device memory,
 * top row first: the reference's own conversion for its C++ path, Cpp/Emscripten/main.cpp:63-79
 * (c8 = min(sqrtf(c)*255, 255), alpha 255).  Enqueued on the context's stream; 4x less data to download than
 * the float buffer. */
TPT_API int tptDisplayRGBA8(const float* deviceTile, int screenWidth, int screenHeight, unsigned char* deviceRGBA);
/* Synchronise the stream and return the monotonic total of rays traced by this context. */
TPT_API int tptRayCounterRead(int64_t* outTotalRays);
/* Let the caller own the ray counter: `deviceU64` points to one zero-initialised 64-bit word in device
 * memory (e.g. a torch int64 tensor) that the kernels atomically add to; NULL -> the internal counter.
 * Lets the multi-GPU host sum-reduce the counters with RCCL without a host round trip. */
TPT_API int tptSetRayCounter(void* deviceU64);
/* Sharded hosts: from the next tptDrawDevice on, the progressive blend also writes every blended pixel of the tile to
 * `deviceMirror` (same size and layout as the tile) and the current ray-counter value to the 8 bytes at
 * `deviceCounterOut` (may be NULL) -- the snapshot handed to the collective while later frames keep accumulating into
 * the tile -- in the SAME kernel, so the frame's dependency chain stays one kernel long.  NULL turns it off.  The
 * pointers are read at enqueue time; call again to rotate buffers.  The counter written is the context's RUNNING TOTAL at
 * the moment of the blend, not a per-frame count: with later frames already tracing it includes their rays so far, and
 * is exact for "all frames up to f" only once nothing later is in flight (the last frame's snapshot after a synchronise,
 * which is what tptShardedFinish and sharding.finish() read). */
TPT_API int tptSetTileMirror(float* deviceMirror, void* deviceCounterOut);
TPT_API int tptSynchronize(void);

/* ---- multi-GPU inside the library: one process per GPU, RCCL over xGMI (replaces the row fan-out / join of
 * DrawTest, Test.cpp:357-361, across GPUs).  A C++ host needs nothing but these five calls and a way to hand 128 bytes from
 * rank 0 to the other processes (pipe, file, MPI, ...): examples/multi_gpu_host.cpp.
 *   rank 0: tptCommGetUniqueId(id);  every rank: tptInitialize (TPT_DEVICE / LOCAL_RANK picks the GPU), tptCommInit(id, n, rank, 8);
 *   per frame, every rank: tptUpdate(...); tptDrawSharded(time, f, w, h, imageOnRank0, flags);   (asynchronous)
 *   tptShardedFinish(&rays);  ...  tptCommDestroy() / tptShutdown().
 * The image's rows are dealt out in stripes of `stripeRows` rows round-robin over the ranks (cost is not uniform in y); each
 * rank keeps its compact accumulation tile resident; per frame exactly ONE collective -- ncclGather (rccl.h:745) of the
 * blended tile plus one row whose first 8 bytes carry the rank's 64-bit ray counter -- on a communication stream, from a
 * ring of 4 snapshots written by the blend kernel itself, so it overlaps the tracing of the next frames; rank 0
 * de-interleaves the gathered tiles into `deviceImageOnRoot` (w*h*4 floats, device memory; ignored on other ranks).  RNG seeds
 * depend on the global (x, y) only: the assembled image is bit-identical to a 1-GPU render.  librccl is dlopen()ed by
 * tptCommInit / tptCommGetUniqueId; a single-GPU host never loads it. */
#define TPT_COMM_ID_BYTES 128
TPT_API int tptCommGetUniqueId(void* outId128);
TPT_API int tptCommInit(const void* id128, int nRanks, int rank, int stripeRows);
/* Measurement aid, no RCCL: this process plays rank 0 of an nRanks-way run alone -- same tile, snapshot ring, events and
 * assemble kernel, a device copy of its own slice in place of the gather (the other ranks' rows of the image stay zero).
 * Shows what one GPU sustains as rank 0 of N (bench.py --emulate-ranks N).  Paired with tptCommDestroy like tptCommInit. */
TPT_API int tptCommInitLoopback(int nRanks, int stripeRows);
/* size of the communicator and this process's rank as RCCL reports them (ncclCommCount / ncclCommUserRank), loopback flag */
TPT_API int tptCommInfo(int* outRanks, int* outRank, int* outLoopback);
TPT_API int tptCommDestroy(void);
TPT_API int tptDrawSharded(float time, int frameCount, int screenWidth, int screenHeight, float* deviceImageOnRoot, unsigned testFlags);
/* How many consecutive frames tptDrawSharded collects into ONE trace launch + blend + exchange.  0 (default) = automatic: 1 when a
 * rank's tile is 2.4 M samples per frame or more (rows x width x spp) and for animated scenes; 2 / 4 / 8 below 2.4 / 1.2 / 0.6 M -- with small tiles the
 * chain behind a frame (trace launch, blend + snapshot, gather, de-interleave: four dispatches beside a machine full of trace
 * workgroups) bounds the frame rate, not the arithmetic.  k = 1..32 = the host's choice.  A frame that is collected is issued when the
 * k-th arrives, when anything it depends on is about to change (every setter, tptUpdate with another size), or when the caller waits
 * (tptShardedFinish, tptSynchronize, tptRayCounterRead): same bits as frame-by-frame calls; the image on rank 0 is current after a
 * batch has gone out and after tptShardedFinish.  EVERY rank must choose the same. */
TPT_API int tptSetShardExchangeInterval(int everyKFrames);
/* nFrames (1..32) consecutive frames per call, traced by one launch per rank (tptDrawDeviceBatch) and followed by ONE exchange:
 * rank 0's image is that of the batch's last frame.  Same bits as nFrames tptDrawSharded calls. */
TPT_API int tptDrawShardedBatch(float time, int firstFrame, int nFrames, int screenWidth, int screenHeight, float* deviceImageOnRoot, unsigned testFlags);
/* waits for every exchange enqueued so far; rank 0: sum of all ranks' ray counters as of the last gathered frame, other
 * ranks: their own.  Counters are running totals since tptInitialize / tptSetRayCounter (before the first sharded frame:
 * this rank's own running total), so callers take differences. */
TPT_API int tptShardedFinish(int64_t* outTotalRays);

/* hipEvent bracket on the context's stream, for kernel-only timing (as the reference times its
 * Dispatch with timestamp queries, TestWin.cpp:299-302). */
TPT_API int tptTimerBegin(void);
TPT_API int tptTimerEnd(float* outMilliseconds); /* synchronises */

/* Per-launch timing of the trace kernel: between Begin and End every tptDrawDevice brackets its trace
 * launch with a hipEvent pair recorded on the stream that launch goes to (the internal trace streams when
 * frames overlap).  End synchronises and returns the SUM of the individual launch durations and their
 * number -- the same per-dispatch durations a rocprofv3 kernel trace reports. */
TPT_API int tptKernelTimingBegin(int maxLaunches);
TPT_API int tptKernelTimingEnd(float* outSumMilliseconds, int* outLaunches);

/* ================= 4. tuning and diagnostics (unit-test entry points: tpt_test_hooks.h, a separate build) ================= */

/* hitSpheres: 0 = two-phase (default: a conservative filter + the reference's exact test for what passes.  The filter runs
 * on the matrix cores for scenes of <= 64 spheres that binary16 operands can carry, as packed FP32 on the VALU otherwise;
 * scenes of 256 spheres or more are traversed through compact groups of <= 8 spheres with bounding spheres, the groups' own
 * bounds through a two-level packed filter (super-groups of 8 groups, then the groups of what passes) -- same hits, same
 * tie-break, several times faster on the 4096-sphere scene), 1 = simple loop (exact test for every sphere), 2 = two-phase
 * without grouping (brute force over all spheres, the reference's cost model), 3 = two-phase with the FLAT packed VALU filter
 * everywhere (no matrix-core table, no second level over the groups), 4 = as 0 with the bounds of a GROUPED scene on the matrix cores as well -- compiled into the
 * hooks build only (the product library refuses it): in a process the device time-slices (more than ~22 hardware queues, or
 * anything else on the GPU) waves that have run that path lose a hit in ~1e-9 of their rays, and the two-level
 * VALU filter is no slower.  persistent: 3 = path queues in LDS (default; per-pixel seeds, recursive fold, two-phase
 * only -- anything else falls back to 1), 1 = persistent waves with lane refill, 0 = ONE THREAD PER PIXEL: the same kernel with re-filling
 * off -- a wave takes an 8x8 tile and every lane keeps its pixel until the tile is done (the shape of the reference's compute shaders,
 * ComputeShader.hlsl:353-395, and of BASELINE.json's north_star; with hitSpheres 1 also its brute-force loop): kept as a live A/B,
 * 4-6 x slower than the default.  ldsScene:
 * 1 = stage sphere records and materials in LDS (default when they fit), 0 = read them from global memory, -1 = auto.
 * All variants produce identical bits. */
TPT_API int tptSetKernelVariant(int hitSpheres, int persistent, int ldsScene);
/* What the next launch will do with the scene: its sphere count, the number of sphere groups (0: the scene is traversed flat -- up to 255
 * spheres, or a scene the grouping refuses), and whether the groups' bounding spheres are filtered on the matrix cores (1: only after
 * tptSetKernelVariant(4, ..)) or by the packed VALU filter (0: the default). */
TPT_API int tptGetSceneInfo(int* outSpheres, int* outGroups, int* outBoundsOnMatrixCores);
/* kernel resource facts for DESIGN/bench: occupancy (blocks/CU), LDS bytes/block, grid size of the last launch */
TPT_API int tptGetLaunchInfo(int* outBlocksPerCU, int* outLdsBytes, int* outGridBlocks, int* outNumCUs);
/* Facts about the frame pipeline: hardware queues the runtime really runs side by side for this process (measured at
 * tptInitialize with one spinning wave per trace stream: GPU_MAX_HW_QUEUES only counts if it was set before the HIP
 * runtime started), the frames-in-flight limit that results (tptSetFrameOverlap's value clamped to what those queues can
 * carry), the deepest pipeline the caller has built so far (decides the grid of a launch), and how often the per-slot
 * buffers were (re-)allocated (once per frame shape; never on the steady-state path). */
TPT_API int tptGetPipelineInfo(int* outHwQueues, int* outOverlapEffective, int* outStreamDepth, int* outSlotReservations);
TPT_API const char* tptGetLastError(void);
/* The reference's six functions (include/tpt_test_api.h == Test.h:10-17) return void: when one of them fails -- no GPU, a HIP error, DrawTest
 * before UpdateTest -- the library by default prints the message and abort()s (there is no CPU path to fall back to; a frame that silently
 * was not rendered is worse than a stop).  A host that wants to decide itself installs a handler: it is called on the calling thread with
 * the entry point's name and the message, and the failed call then RETURNS without effect (DrawTest leaves the buffer alone and reports 0
 * rays).  NULL restores the default.  The tpt* functions never abort: they return a negative code. */
typedef void (*tptErrorHandler)(const char* where, const char* message);
TPT_API int tptSetErrorHandler(tptErrorHandler handler);
TPT_API const char* tptGetDeviceName(void);

#ifdef __cplusplus
}
#endif
#endif
