// tpt_host_hooks.cpp -- unit-test / profiling entry points of include/tpt_test_hooks.h (compiled into the hooks build only)
// (one of the host runtime's translation units: tpt_context.h lists them)
#include "tpt_context.h"

using namespace tpt;
using namespace tpth;

extern "C" {
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
    if (a.scene.nGroups <= 0 || (a.scene.gmxTiles <= 0 && a.scene.nSuperPairs <= 0)) return fail("tptTestGroupFilter: the current scene is not grouped (fewer than 256 spheres, a group too loose, or hit-spheres variant 2)");
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

int tptTestSetDealCapacities(int superGroupEntries, int groupEntries, int survivorEntries)
{
    if (requireInit()) return -1;
    if (int rc = tptSynchronize()) return rc; // (no launch in flight may see the sizes change)
    if (tptSetDealCapacitiesForTest(superGroupEntries, groupEntries, survivorEntries) != hipSuccess)
        return fail("tptTestSetDealCapacities: each size between 64 and its compiled value, or 0, 0, 0 for the compiled ones");
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
