/* tpt_test_hooks.h -- unit-test and profiling entry points.  NOT part of the product: libtoypathtracer_hip.so does not
 * export them.  toypathtracer_amd/csrc/build.sh builds the same sources a second time with -DTPT_TEST_HOOKS into
 * libtoypathtracer_hip_hooks.so (everything of tpt_hip.h plus the functions below); the GPU suite loads that build for its
 * math / HitSpheres / filter unit tests only, every render test goes through the product library. */
#ifndef TPT_TEST_HOOKS_H
#define TPT_TEST_HOOKS_H
#include "tpt_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* device math unit tests: op 0 sqrt 1 div 2 sin 3 cos 4 pow5 5 rnd01^16 6 schlick 7 normalize.x 8 / 9 sin / cos of the sincos pair 10 tdivSafeNum(a, b) 11 tdivByPi(a) (tpt_math.h's short correctly rounded divisions); host arrays */
TPT_API int tptTestMath(int op, const float* a, const float* b, float* out, int n);
/* intersect n host rays ([n][6] = origin, UNIT direction -- the reference asserts it, Maths.h:337; the two-phase
 * filter's error bound assumes it) with the current scene on the GPU */
TPT_API int tptTestHitSpheres(int hitSpheres, const float* rays, int* outId, float* outT, int n);
/* self-check: tpt_math.h's fast correctly-rounded sqrt (op 0) / 1.0f / sqrtf (op 1) against the compiler's correctly rounded
 * expansions for EVERY binary32 bit pattern in [lo, hi] on the device; mismatch count + the first offending inputs */
TPT_API int tptTestMathExhaustive(int op, unsigned lo, unsigned hi, unsigned long long* outMismatches, unsigned* outFirst8);
/* phase 1 of HitSpheres as the path-queue kernel runs it for scenes of <= 64 spheres: on the matrix cores
 * (v_mfma_f32_32x32x16_f16 over an 11-term expansion of the filter's discriminant, every f32 factor split into two binary16
 * pieces).  n host rays -> candidate masks (sphere p at bit 63 - p; may be NULL) and / or the nearest hit through the filter
 * + the exact test of its candidates (outId / outT; may be NULL). */
TPT_API int tptTestMatrixFilter(const float* rays, unsigned long long* outMask, int* outId, float* outT, int n);
/* the matrix-core filter over the GROUP BOUNDS of the current grouped scene (>= 256 spheres) against the reference's
 * discriminant for every member sphere: violations = (ray, member) pairs with discr > 0 whose group the filter dropped (must be
 * 0); touched = groups kept, exact = members with discr > 0, both summed over the rays.  n host rays ([n][6], unit direction). */
TPT_API int tptTestGroupFilter(const float* rays, int n, unsigned long long* outViolations, unsigned long long* outTouched, unsigned long long* outExact);
/* the entry areas of the grouped traversal's three dealt stages (tpt_kernels.hip dealThreeStage: (path, super-group) entries per round,
 * (path, group) entries waiting, survivors waiting) shrunk at run time: each between 64 and its compiled size; 0, 0, 0 restores them.
 * With 64-entry areas every overflow path runs (further rounds, entries served in place): same bits, a fraction of the rate. */
TPT_API int tptTestSetDealCapacities(int superGroupEntries, int groupEntries, int survivorEntries);
/* profiling builds only (-DTPT_STATS): 128 counters, wave-level entries [i] / lane counts [32+i] of the
 * state machine's blocks (enum ST_* in tpt_trace.h); the shipped build returns an error. */
TPT_API int tptDebugStats(unsigned long long* out128, int reset);
/* per-chunk accumulated ray counts and the chunk order table of the last launch (cost-ordered work distribution
 * of the persistent kernel); either pointer may be NULL; returns the number of chunks copied */
TPT_API int tptDebugChunkOrder(unsigned* outCost, unsigned* outOrder, int capacity);
#ifdef __cplusplus
}
#endif
#endif
