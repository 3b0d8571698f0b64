// tools/filter_stats.cpp -- CPU measurement tool, not part of the product.  What does phase 2 of HitSpheres get to do?
// Traces a frame of the default scene with the product's own per-lane code compiled for the host (as tests/lane_emu.cpp
// does) and, for every ray, counts: candidates of the matrix-core filter (phase1MatrixHRef, what the device's MFMA filter
// passes up to its accumulation order), candidates of the packed VALU filter (memberFilter), spheres with a positive
// exact discriminant (what the reference's loop gets past `discr > 0`), and spheres with a hit in (tMin, tMax).
// Per ray kind (camera / bounce, shadow).  Also the wave-level view: rays in arrival order grouped by 64 of one kind,
// trips = the largest candidate count in the group (what a wave waits for).
// Build: g++ -O2 -std=c++17 -ffp-contract=off -I toypathtracer_amd/csrc tools/filter_stats.cpp -o /tmp/filter_stats
#include "tpt_scene.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using namespace tpt;

struct Acc {
    long long rays = 0, matrix = 0, valu = 0, exact = 0, hits = 0, hist[12] = {}, afterPlane = 0, afterBehind = 0, afterSelf = 0, selfKnown = 0, selfCulled = 0, selfViolations = 0;
    std::vector<unsigned char> perRayMatrix, perRayExact, perRayPlane, perRayBehind, perRaySelf;
};

int main(int argc, char** argv)
{
    const int w = argc > 1 ? atoi(argv[1]) : 320, h = argc > 2 ? atoi(argv[2]) : 180, spp = argc > 3 ? atoi(argv[3]) : 4;
    std::vector<SpherePOD> S;
    std::vector<MaterialPOD> M;
    defaultScene(S, M);
    PackedScene P;
    packScene(S, M, P);
    SceneView sv = viewOf(P);
    if (sv.mxR1 < 0) { printf("no matrix table\n"); return 1; }
    CameraPOD cam = makeCamera(defaultCameraSetup(), float(w) / float(h));
    FrameConsts fc = makeFrameConsts(cam, w, h, spp, 0, 1u /* progressive */, SEED_PER_PIXEL, CFG_LIGHT_SAMPLING, 0.9f);
    f4 stackMem[TPT_MAX_DEPTH];
    BounceStack stack;
    stack.base = stackMem; stack.stride = 1; stack.spill = stackMem + 3; stack.spillStride = 1; stack.fastLevels = 3;
    Acc acc[2];
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            Lane L;
            L.rays = 0; L.active = false;
            laneBeginPixel(L, fc, x, y, y * w + x, true);
            int self = -1; // the sphere the ray starts on (round 5: the self cull of the path-queue kernel)
            for (;;) {
                if (L.needCamera) { laneCamera<FOLD_RECURSIVE>(L, fc); self = -1; }
                const f3 o = L.orig, d = L.dir;
                Acc& A = acc[L.kind == KIND_SHADOW ? 1 : 0];
                const uint64_t mm = phase1MatrixHRef(sv.amatH, sv.mxR1, sv.nSpheres, o, d);
                const f3 dk = mk3(d.x * TPT_P1_K, d.y * TPT_P1_K, d.z * TPT_P1_K);
                int nv = 0, ne = 0, nh = 0, np = 0, nbh = 0, nself = 0;
                for (int i = 0; i < sv.nSpheres; ++i) {
                    const f4 s = sv.sph4[i];
                    nv += memberFilter(s, o, dk);
                    const float coX = s.x - o.x, coY = s.y - o.y, coZ = s.z - o.z;
                    const float nb = coX * d.x + coY * d.y + coZ * d.z;
                    const float c = coX * coX + coY * coY + coZ * coZ - s.w;
                    const float discr = nb * nb - c;
                    // two cheap "the whole sphere lies behind the origin" tests on top of the filter's candidates (margins as a
                    // conservative filter would need them, relative 2^-10): (P) behind the origin's plane: co.d + r < 0, linear in
                    // the ray; (B) the line meets the sphere behind the origin: nb < 0 and |co|^2 > r^2
                    if ((mm >> (63 - i)) & 1ull) {
                        const float r = sqrtf(s.w);
                        const bool plane = nb + r * 1.001f < -1e-4f * (fabsf(nb) + r);
                        const bool behind = nb < -1e-4f * sqrtf(coX * coX + coY * coY + coZ * coZ) && c > 1e-3f * s.w;
                        np += !plane;
                        nbh += !behind;
                    }
                    if (i == self) {
                        // the self cull: nb <= 0 (the ray leaves the sphere), not absurdly large, and the origin at most a hair inside:
                        // then t = nb + sqrt(discr) <= tMin whatever the roundings (derivation: tpt_trace.h, selfCull)
                        A.selfKnown++;
                        const bool cull = nb <= 0.0f && nb >= -1000.0f && c >= 0.0005f * nb;
                        if (cull) {
                            A.selfCulled++;
                            nself += ((mm >> (63 - i)) & 1ull) ? 1 : 0;
                            if (discr > 0) {
                                const float sq = tsqrt(discr);
                                float t = nb - sq;
                                if (t <= TPT_MIN_T) t = nb + sq;
                                if (t > TPT_MIN_T) A.selfViolations++;
                            }
                        }
                    }
                    if (discr > 0) {
                        ++ne;
                        const float sq = tsqrt(discr);
                        float t = nb - sq;
                        if (t <= TPT_MIN_T) t = nb + sq;
                        nh += t > TPT_MIN_T && t < TPT_MAX_T;
                    }
                }
                const int nm = __builtin_popcountll(mm);
                A.afterPlane += np; A.afterBehind += nbh; A.afterSelf += nm - nself;
                A.perRaySelf.push_back((unsigned char)(nm - nself));
                A.perRayPlane.push_back((unsigned char)np); A.perRayBehind.push_back((unsigned char)nbh);
                A.rays++; A.matrix += nm; A.valu += nv; A.exact += ne; A.hits += nh;
                A.hist[nm < 11 ? nm : 11]++;
                A.perRayMatrix.push_back((unsigned char)nm);
                A.perRayExact.push_back((unsigned char)ne);
                float t;
                const int id = hitSpheres<HS_MATRIX>(sv, o, d, TPT_MIN_T, TPT_MAX_T, t);
                L.rays++;
                if (L.kind == KIND_MAIN && id >= 0) self = id; // (a shadow ray's result does not move the path)
                if (lanePost<FOLD_RECURSIVE>(L, id, t, sv, fc, stack)) break;
            }
        }
    const char* names[2] = {"camera + bounce rays", "shadow rays"};
    for (int k = 0; k < 2; ++k) {
        const Acc& A = acc[k];
        if (!A.rays) continue;
        printf("%-22s %9lld rays: candidates per ray  matrix filter %.3f  VALU filter %.3f  |  exact discr > 0: %.3f  hit in (tMin, tMax): %.3f\n",
               names[k], A.rays, (double)A.matrix / A.rays, (double)A.valu / A.rays, (double)A.exact / A.rays, (double)A.hits / A.rays);
        printf("    candidates left after culling spheres behind the origin's plane: %.3f; behind the origin on the line: %.3f\n", (double)A.afterPlane / A.rays, (double)A.afterBehind / A.rays);
        printf("    self cull: the start sphere is known for %.1f %% of the rays, culled for %.1f %% (violations: %lld); candidates left %.3f\n",
               100.0 * A.selfKnown / A.rays, 100.0 * A.selfCulled / A.rays, A.selfViolations, (double)A.afterSelf / A.rays);
        {
            long long g2 = 0, tr = 0, trs = 0, g2s = 0;
            for (size_t i = 0; i + 64 <= A.perRaySelf.size(); i += 64) { int mx = 0; for (int j = 0; j < 64; ++j) mx = A.perRaySelf[i + j] > mx ? A.perRaySelf[i + j] : mx; tr += mx; g2++; }
            const size_t n2 = A.perRaySelf.size() / 64 * 64, st2 = n2 / 64;
            for (size_t g = 0; g < st2; ++g) { int mx = 0; for (int j = 0; j < 64; ++j) { const size_t idx = g + (size_t)j * st2; mx = A.perRaySelf[idx] > mx ? A.perRaySelf[idx] : mx; } trs += mx; g2s++; }
            if (g2) printf("    phase-2 trips per 64 rays with the self cull: coherent %.2f scattered %.2f\n", (double)tr / g2, (double)trs / g2s);
        }
        printf("    matrix-filter candidates per ray, histogram 0..10, 11+: ");
        for (int i = 0; i < 12; ++i) printf("%.1f%% ", 100.0 * A.hist[i] / A.rays);
        printf("\n");
        // wave view: groups of 64 rays in arrival order (the queue kernel batches rays of one class; arrival order here is
        // pixel order, i.e. spatially coherent -- an optimistic stand-in for the queues' mix)
        long long groups = 0, tripsM = 0, tripsE = 0, tripsP = 0, tripsB = 0, tripsPs = 0, tripsBs = 0;
        for (size_t i = 0; i + 64 <= A.perRayMatrix.size(); i += 64) {
            int mxm = 0, mxe = 0, mxp = 0, mxb = 0;
            for (int j = 0; j < 64; ++j) { mxm = A.perRayMatrix[i + j] > mxm ? A.perRayMatrix[i + j] : mxm; mxe = A.perRayExact[i + j] > mxe ? A.perRayExact[i + j] : mxe;
                mxp = A.perRayPlane[i + j] > mxp ? A.perRayPlane[i + j] : mxp; mxb = A.perRayBehind[i + j] > mxb ? A.perRayBehind[i + j] : mxb; }
            tripsM += mxm; tripsE += mxe; tripsP += mxp; tripsB += mxb; groups++;
        }
        // and a shuffled grouping (stride permutation): incoherent rays per wave, the pessimistic end
        long long tripsMs = 0, tripsEs = 0, groupsS = 0;
        const size_t n = A.perRayMatrix.size() / 64 * 64, stride = n / 64;
        for (size_t g = 0; g < stride; ++g) {
            int mxm = 0, mxe = 0, mxp = 0, mxb = 0;
            for (int j = 0; j < 64; ++j) { const size_t idx = g + (size_t)j * stride; mxm = A.perRayMatrix[idx] > mxm ? A.perRayMatrix[idx] : mxm; mxe = A.perRayExact[idx] > mxe ? A.perRayExact[idx] : mxe;
                mxp = A.perRayPlane[idx] > mxp ? A.perRayPlane[idx] : mxp; mxb = A.perRayBehind[idx] > mxb ? A.perRayBehind[idx] : mxb; }
            tripsMs += mxm; tripsEs += mxe; tripsPs += mxp; tripsBs += mxb; groupsS++;
        }
        {   // rays binned by candidate count before they are intersected (0 / 1 / 2 / 3+): waves of one bin, arrival order within
            long long trips = 0, n64 = 0, cntBin[4] = {};
            for (int bin = 0; bin < 4; ++bin) {
                int fill = 0, mx = 0;
                for (size_t i = 0; i < A.perRayMatrix.size(); ++i) {
                    const int c = A.perRayMatrix[i];
                    if ((c < 3 ? c : 3) != bin) continue;
                    cntBin[bin]++;
                    mx = c > mx ? c : mx;
                    if (++fill == 64) { trips += mx; n64++; fill = 0; mx = 0; }
                }
                if (fill) { trips += mx; n64++; }
            }
            printf("    binned by candidate count (0 / 1 / 2 / 3+: %.1f / %.1f / %.1f / %.1f %% of the rays): %.2f trips per 64 rays\n",
                   100.0 * cntBin[0] / A.rays, 100.0 * cntBin[1] / A.rays, 100.0 * cntBin[2] / A.rays, 100.0 * cntBin[3] / A.rays,
                   (double)trips * 64.0 / A.rays);
        }
        if (groups)
            printf("    phase-2 trips per 64 rays (largest count in the group): coherent groups %.2f (an exact filter: %.2f), scattered groups %.2f (exact: %.2f)\n"
                   "    with the plane cull: coherent %.2f scattered %.2f; with the behind-the-origin cull: coherent %.2f scattered %.2f\n",
                   (double)tripsM / groups, (double)tripsE / groups, (double)tripsMs / groupsS, (double)tripsEs / groupsS,
                   (double)tripsP / groups, (double)tripsPs / groupsS, (double)tripsB / groups, (double)tripsBs / groupsS);
    }
    return 0;
}
