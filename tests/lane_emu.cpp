// tests/lane_emu.cpp -- TEST HARNESS ONLY (built into tests/_build/, never part of the product).
//
// Compiles the per-lane device logic of toypathtracer_amd/csrc (tpt_math.h, tpt_trace.h,
// tpt_scene.h) for the HOST and runs one lane at a time over an image, so that the `-m "not gpu"`
// suite can check the flattened Trace/Scatter state machine, the two-phase HitSpheres and the scene
// packing against the oracle without a GPU.  What it cannot cover (wave-level refill, LDS staging,
// atomics, the device compiler) is covered by the `-m gpu` parity tests.
#include "tpt_scene.h"
#include "tpt_shard.h"
#include <stdint.h>
#include <vector>

using namespace tpt;

static int g_emuConfig = CFG_LIGHT_SAMPLING;
static float g_emuSmoothing = 0.9f;

extern "C" {

// Config.h:23-25 as run-time switches for the renders that follow (tptSetConfig's twin)
void emu_set_config(int lightSampling, float animateSmoothing, int mitsubaCompare)
{
    g_emuConfig = (lightSampling ? CFG_LIGHT_SAMPLING : 0) | (mitsubaCompare ? CFG_MITSUBA_COMPARE : 0);
    g_emuSmoothing = animateSmoothing;
}

// spheres/mats: reference layouts (20 B / 36 B).  cam: 88 B.  Returns ray count.
int64_t emu_render_ex(const void* spheres, const void* mats, int count, const void* cam, int w, int h, int y0, int y1,
                      int spp, int frame, unsigned flags, int seedMode, int hs, int fold, float* backbuffer, int* perPixelRays);

int64_t emu_render(const void* spheres, const void* mats, int count, const void* cam, int w, int h, int y0, int y1,
                   int spp, int frame, unsigned flags, int seedMode, int hs, int fold, float* backbuffer)
{
    return emu_render_ex(spheres, mats, count, cam, w, h, y0, y1, spp, frame, flags, seedMode, hs, fold, backbuffer, nullptr);
}

// perPixelRays (optional, w*h ints): number of laneStep calls (= rays) each pixel took -- used by tools/ to model
// wave scheduling on the CPU.
int64_t emu_render_ex(const void* spheres, const void* mats, int count, const void* cam, int w, int h, int y0, int y1,
                      int spp, int frame, unsigned flags, int seedMode, int hs, int fold, float* backbuffer, int* perPixelRays)
{
    std::vector<SpherePOD> S((const SpherePOD*)spheres, (const SpherePOD*)spheres + count);
    std::vector<MaterialPOD> M((const MaterialPOD*)mats, (const MaterialPOD*)mats + count);
    PackedScene P;
    packScene(S, M, P);
    SceneView sv = viewOf(P);
    if (hs == 2) sv.nGroups = 0; // two-phase, brute force even for a large scene
    if (hs == 3) sv.mxR1 = -1;   // the packed VALU filter everywhere (tptSetKernelVariant(3, ., .))
    CameraPOD c;
    memcpy(&c, cam, sizeof(c));
    FrameConsts fc = makeFrameConsts(c, w, h, spp, frame, flags, seedMode, g_emuConfig, g_emuSmoothing);
    f4 stackMem[TPT_MAX_DEPTH];
    BounceStack stack;
    stack.base = stackMem;
    stack.stride = 1;
    stack.spill = stackMem + 3; // exercise the two-level stack: levels 0-2 "fast", 3-9 "spill"
    stack.spillStride = 1;
    stack.fastLevels = 3;
    int64_t rays = 0;
    for (int y = y0; y < y1; ++y) {
        Lane L;
        L.rays = 0;
        L.active = false;
        for (int x = 0; x < w; ++x) {
            laneBeginPixel(L, fc, x, y, y * w + x, seedMode == SEED_PER_PIXEL || x == 0);
            const uint32_t raysBefore = L.rays;
            for (;;) {
                bool done;
                if (hs == HS_SIMPLE)
                    done = fold == FOLD_FORWARD ? laneStep<HS_SIMPLE, FOLD_FORWARD>(L, sv, fc, stack)
                                                : laneStep<HS_SIMPLE, FOLD_RECURSIVE>(L, sv, fc, stack);
                else if (hs == 0 && sv.mxR1 >= 0) // as the product: phase 1 = the matrix-core filter (its host restatement) when the scene has a table
                    done = fold == FOLD_FORWARD ? laneStep<HS_MATRIX, FOLD_FORWARD>(L, sv, fc, stack)
                                                : laneStep<HS_MATRIX, FOLD_RECURSIVE>(L, sv, fc, stack);
                else
                    done = fold == FOLD_FORWARD ? laneStep<HS_TWO_PHASE_GROUPS, FOLD_FORWARD>(L, sv, fc, stack)
                                                : laneStep<HS_TWO_PHASE_GROUPS, FOLD_RECURSIVE>(L, sv, fc, stack);
                if (done) break;
            }
            {
                float* px = backbuffer + (size_t)L.pix * 4;
                f3 c = blendPixel(ld3(px), lanePixelColour(L, fc), fc.lerpFac);
                px[0] = c.x; px[1] = c.y; px[2] = c.z;
            }
            if (perPixelRays) perPixelRays[y * w + x] = (int)(L.rays - raysBefore);
        }
        rays += L.rays;
    }
    return rays;
}

// The path-queue kernel's class code (tpt_trace.h: qCamera, qLambertBegin, qLightRay, qLightShade, qLambertE, qMetal, qDielectric,
// qEndTerm, qFold, qStackPush) driven sequentially, one path at a time, in the order tptTraceQueueKernel applies it to a path:
// intersect, classify, class code, whole light loop of a Lambert hit in place, a Metal whose scattered ray points into the surface
// ends with the record as it stands.  PER_PIXEL seeds, recursive fold (what that kernel implements).  hs as emu_render.
int64_t emu_render_queue_classes(const void* spheres, const void* mats, int count, const void* cam, int w, int h, int spp, int frame,
                                 unsigned flags, int hs, float* backbuffer)
{
    std::vector<SpherePOD> S((const SpherePOD*)spheres, (const SpherePOD*)spheres + count);
    std::vector<MaterialPOD> M((const MaterialPOD*)mats, (const MaterialPOD*)mats + count);
    PackedScene P;
    packScene(S, M, P);
    SceneView sv = viewOf(P);
    if (hs == 2) sv.nGroups = 0;
    if (hs == 3) sv.mxR1 = -1;
    CameraPOD c;
    memcpy(&c, cam, sizeof(c));
    const FrameConsts fc = makeFrameConsts(c, w, h, spp, frame, flags, SEED_PER_PIXEL, g_emuConfig, g_emuSmoothing);
    auto hit = [&](f3 o, f3 d, float& t) {
        if (hs == HS_SIMPLE) return hitSpheres<HS_SIMPLE>(sv, o, d, TPT_MIN_T, TPT_MAX_T, t);
        if (hs == 0 && sv.mxR1 >= 0) return hitSpheres<HS_MATRIX>(sv, o, d, TPT_MIN_T, TPT_MAX_T, t);
        return hitSpheres<HS_TWO_PHASE_GROUPS>(sv, o, d, TPT_MIN_T, TPT_MAX_T, t);
    };
    const bool fastDiv = (sv.flags & SCENE_LIGHT_R2_DIV_SAFE) != 0;
    f4 level0, spill[TPT_MAX_DEPTH];
    QStack stack;
    stack.l0 = &level0;
    stack.spill = spill;
    stack.stride = 1;
    int64_t rays = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint32_t rng = pixelSeed(SEED_PER_PIXEL, x, y, frame);
            f3 col = mk3(0, 0, 0), ro, rd;
            for (int sample = 0; sample < spp; ++sample) {
                qCamera(fc, x, y, rng, ro, rd);
                int depth = 0, recId = -1;
                bool doMatE = true;
                for (;;) {
                    float t;
                    const int hitId = hit(ro, rd, t);
                    ++rays;
                    int cls = -1; // -1: END
                    if (hitId >= 0 && depth < TPT_MAX_DEPTH) cls = (int)f2u(sv.mats[hitId * 3].w);
                    if (hitId >= 0) ro = ro + rd * t;
                    recId = hitId;
                    bool ended = cls != MAT_LAMBERT && cls != MAT_METAL && cls != MAT_DIELECTRIC;
                    if (cls == MAT_DIELECTRIC) {
                        f3 e;
                        rd = qDielectric(sv, fc, ro, rd, recId, doMatE, rng, e);
                        qStackPush(stack, depth, e, -1);
                        depth++;
                        doMatE = true;
                    } else if (cls == MAT_METAL) {
                        f3 e, nd;
                        if (qMetal(sv, fc, ro, rd, recId, doMatE, rng, e, nd)) {
                            qStackPush(stack, depth, e, recId);
                            depth++;
                            doMatE = true;
                            rd = nd;
                        } else {
                            ended = true; // Test.cpp:218-221: the record as it stands goes to the END class
                        }
                    } else if (cls == MAT_LAMBERT) {
                        QLambert lam;
                        qLambertBegin(sv, ro, rd, recId, rng, lam);
                        const int nShadow = (fc.config & CFG_LIGHT_SAMPLING) ? sv.nLights : 0;
                        for (int j = 0; j < nShadow; ++j) {
                            const f4 l1 = sv.lights[j * 2 + 1];
                            const int lightId = (int)f2u(l1.w);
                            if (lightId == recId) continue; // Test.cpp:100
                            const f3 d2 = qLightRay(sv.lights[j * 2], ro, rng, lam.cosAMax, fastDiv);
                            float ts;
                            const int id = hit(ro, d2, ts);
                            ++rays;
                            if (id == lightId) qLightShade(l1, d2, lam);
                        }
                        qStackPush(stack, depth, qLambertE(sv, recId, doMatE, lam), recId);
                        depth++;
                        doMatE = !(fc.config & CFG_LIGHT_SAMPLING);
                        rd = lam.sdir;
                    }
                    if (ended) {
                        col = col + qFold(sv, qEndTerm(sv, fc, rd, recId), depth, stack);
                        break;
                    }
                }
            }
            float* px = backbuffer + ((size_t)y * w + x) * 4;
            const f3 out = blendPixel(ld3(px), col * fc.invSpp, fc.lerpFac);
            px[0] = out.x; px[1] = out.y; px[2] = out.z;
        }
    return rays;
}

// default scene / camera as the product builds them (compared with the reference's GetSceneDesc)
int emu_default_scene(void* spheres, void* mats, int capacity)
{
    std::vector<SpherePOD> S;
    std::vector<MaterialPOD> M;
    defaultScene(S, M);
    if ((int)S.size() > capacity) return -1;
    PackedScene P;
    packScene(S, M, P); // fills invRadius
    memcpy(spheres, S.data(), S.size() * sizeof(SpherePOD));
    memcpy(mats, M.data(), M.size() * sizeof(MaterialPOD));
    return (int)S.size();
}
void emu_default_camera(void* cam, int w, int h)
{
    CameraPOD c = makeCamera(defaultCameraSetup(), float(w) / float(h));
    memcpy(cam, &c, sizeof(c));
}
// {nGroups, nGroupPairs, nBig} of the grouped representation packScene builds for this scene (0 groups: flat)
void emu_group_info(const void* spheres, const void* mats, int count, int* out3)
{
    std::vector<SpherePOD> S((const SpherePOD*)spheres, (const SpherePOD*)spheres + count);
    std::vector<MaterialPOD> M((const MaterialPOD*)mats, (const MaterialPOD*)mats + count);
    PackedScene P;
    packScene(S, M, P);
    out3[0] = P.nGroups; out3[1] = P.nGroupPairs; out3[2] = P.nBig;
}
// tdivByPi's 3-instruction fast form (q0 = a Y, r = fma(-pi, q0, a), q = fma(r, Y, q0), Y = RN(1 / kPI)) against the IEEE
// quotient a / kPI for EVERY significand of a, at the given binary exponent: the number of mismatches (plain fmaf on the host
// is what v_fma_f32 computes).  The guard of tdivByPi hands arguments outside {+0} u [2^-100, 2^126) to the IEEE division.
long long emu_div_pi_mismatches(int exponent, unsigned* firstBad)
{
    long long bad = 0;
    for (uint32_t m = 0; m < (1u << 23); ++m) {
        const float a = ldexpf(u2f(0x3f800000u | m), exponent);
        const float want = a / TPT_PI, got = tdivByPiFast(a);
        if (f2u(want) != f2u(got)) {
            if (!bad && firstBad) *firstBad = f2u(a);
            ++bad;
        }
    }
    return bad;
}
// ---- the C-ABI exchange (tptDrawSharded / tptShardedFinish, tpt_host.cpp) replayed on the CPU with the product's own index
// functions (tpt_shard.h): every rank "renders" its tile -- pixel (x, gy) carries the code gy * w + x, written through the
// kernel's local -> global row map exactly as mapItem / the colour store address it -- the resolve kernel's snapshot layout
// ([padRows + 1][w] pixels, the counter bit-cast into the first 8 bytes of the extra row), the gather (rank-major concatenation:
// what ncclGather delivers on the root), tptAssembleKernel's inverse map and tptShardedFinish's counter reads.
//   out:  image[h * w]    the assembled codes (must be gy * w + x everywhere)
//         info[0..3]      padRows, sum of the counters as tptShardedFinish reads them (low 32 bits), ring slot used, errors found
//         rowOfRank0[h]   for every image row: its row in the flattened receive buffer (sharding.py's rowmap)
// counterOf(r) = (r + 1) * 1000003 + frames is what rank r's counter row carries.
long long emu_shard_exchange(int w, int h, int S, int N, int ring, unsigned long long frames, long long* image, long long* info, long long* rowOfRank0)
{
    const int padRows = shardPadRows(h, S, N);
    const size_t snap = shardSnapshotPixels(padRows, w);
    std::vector<f4> gathered(snap * (size_t)N);
    for (size_t i = 0; i < gathered.size(); ++i) gathered[i].x = gathered[i].y = gathered[i].z = gathered[i].w = -1.0f;
    long long errors = 0;
    int covered = 0;
    for (int r = 0; r < N; ++r) {
        const int rows = shardLocalRows(h, S, N, r);
        covered += rows;
        if (rows > padRows) ++errors;
        f4* send = gathered.data() + snap * (size_t)r; // (the gather: rank r's snapshot lands at slot r of the root's buffer)
        // the launch's stripe constants as enqueueTrace sets them
        const int stripeRows = N > 1 ? S : (h > 0 ? h : 1), stripeStride = N > 1 ? S * N : stripeRows, stripeOffset = N > 1 ? S * r : 0;
        for (int ly = 0; ly < rows; ++ly) {
            const int gy = shardKernelLocalToGlobal(ly, stripeRows, stripeStride, stripeOffset);
            if (gy != shardLocalToGlobal(ly, S, N, r) || gy < 0 || gy >= h) ++errors;
            if (shardKernelGlobalToLocal(gy, stripeRows, stripeStride, stripeOffset) != ly) ++errors;
            if (shardOwner(gy, S, N) != r || shardGlobalToLocal(gy, S, N) != ly) ++errors;
            for (int x = 0; x < w; ++x) {
                f4 v;
                v.x = (float)(gy * w + x); // (exact below 2^24)
                v.y = (float)gy; v.z = (float)x; v.w = 1.0f;
                send[(size_t)ly * w + x] = v;
            }
        }
        const unsigned long long counter = (unsigned long long)(r + 1) * 1000003ull + frames + (1ull << 40); // (needs both words)
        memcpy(&send[shardCounterPixel(padRows, w)], &counter, 8);
    }
    if (covered != h) ++errors;
    for (int gy = 0; gy < h; ++gy) {
        for (int x = 0; x < w; ++x) {
            const f4 v = gathered[shardGatheredPixel(x, gy, w, S, N, padRows)];
            image[(size_t)gy * w + x] = (long long)v.x;
            if (v.y != (float)gy || v.z != (float)x) ++errors;
        }
        rowOfRank0[gy] = (long long)(shardGatheredPixel(0, gy, w, S, N, padRows) / (size_t)w);
    }
    unsigned long long total = 0;
    for (int r = 0; r < N; ++r) { // tptShardedFinish on rank 0
        unsigned long long v = 0;
        memcpy(&v, &gathered[(size_t)r * snap + shardCounterPixel(padRows, w)], 8);
        total += v;
    }
    info[0] = padRows;
    info[1] = (long long)total;
    info[2] = shardRingSlot(frames, ring);
    info[3] = errors;
    return errors;
}
// The matrix-core filter over the GROUP BOUNDS of a grouped scene (buildGroupMatrixTable), through its host restatement
// (phase1MatrixHRef, slackShift 1), against the reference's discriminant of every member (Maths.cpp:171-178):
// out[0] = (ray, member) pairs with discr > 0 whose group the filter dropped (must be 0), out[1] = groups kept (summed over the
// rays), out[2] = members with discr > 0, out[3] = tiles of 64 groups (0: the scene has no table).
void emu_group_matrix_check(const void* spheres, const void* mats, int count, const float* rays, int nRays, long long* out)
{
    std::vector<SpherePOD> S((const SpherePOD*)spheres, (const SpherePOD*)spheres + count);
    std::vector<MaterialPOD> M((const MaterialPOD*)mats, (const MaterialPOD*)mats + count);
    PackedScene P;
    packScene(S, M, P);
    out[0] = out[1] = out[2] = 0;
    out[3] = P.gmxTiles;
    for (int i = 0; i < nRays && P.gmxTiles > 0; ++i) {
        const f3 o = mk3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), d = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
        for (int t = 0; t < P.gmxTiles; ++t) {
            const int left = P.nGroups - t * 64, nEnt = left < 64 ? left : 64;
            const uint64_t m = phase1MatrixHRef(P.gmatH.data() + (size_t)t * TPT_MXH_TABLE_DWORDS, 16, nEnt, o, d, nullptr, nullptr, 1);
            out[1] += __builtin_popcountll(m);
            for (int q = 0; q < nEnt; ++q) {
                const bool kept = (m >> (63 - q)) & 1ull;
                for (int j = 0; j < TPT_GROUP; ++j) {
                    const f4 s = P.gsph[(size_t)(t * 64 + q) * TPT_GROUP + j];
                    const float coX = s.x - o.x, coY = s.y - o.y, coZ = s.z - o.z;
                    const float nb = coX * d.x + coY * d.y + coZ * d.z;
                    const float c = coX * coX + coY * coY + coZ * coZ - s.w;
                    const float discr = nb * nb - c;
                    if (discr > 0) {
                        out[2]++;
                        if (!kept) out[0]++;
                    }
                }
            }
        }
    }
}
// The two-level packed VALU filter over the groups' bounds in its HALF-LINE form (tpt_trace.h phase1PairT<true>: what the path-queue
// kernel's three-stage dealing evaluates) against the reference's WHOLE acceptance of every member (Maths.cpp:171-190: positive
// discriminant and a root beyond tMin): out[0] = accepted (ray, member) pairs whose group or super-group the filter dropped (must be 0),
// out[1] = groups kept by the half-line form, out[2] = groups kept by the line form, out[3] = accepted pairs, out[4] = groups the
// half-line form keeps that the line form drops (must be 0: it only ever drops more).
static void groupHalfCheckRay(const PackedScene& P, f3 o, f3 d, long long* out)
{
    {
        const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
        const float gx = d.x * TPT_PG_K, gy = d.y * TPT_PG_K, gz = d.z * TPT_PG_K;
        const v2f dx = {gx, gx}, dy = {gy, gy}, dz = {gz, gz};
        for (int sg = 0; sg < P.nSupers; ++sg) {
            uint32_t mh = 0, ml = 0;
            phase1PairT<true>(P.spairs.data() + (size_t)(sg / 2) * 8, ox, oy, oz, dx, dy, dz, mh);
            phase1PairT<false>(P.spairs.data() + (size_t)(sg / 2) * 8, ox, oy, oz, dx, dy, dz, ml);
            const bool sgHalf = !((mh >> (1 - (sg & 1))) & 1u), sgLine = !((ml >> (1 - (sg & 1))) & 1u);
            for (int q = 0; q < TPT_SUPER / 2; ++q) {
                const float* rec = P.gpairs.data() + ((size_t)sg * (TPT_SUPER / 2) + q) * 8;
                uint32_t gh = 0, gl = 0;
                phase1PairT<true>(rec, ox, oy, oz, dx, dy, dz, gh);
                phase1PairT<false>(rec, ox, oy, oz, dx, dy, dz, gl);
                for (int hIdx = 0; hIdx < 2; ++hIdx) {
                    const int g = sg * TPT_SUPER + q * 2 + hIdx;
                    if (g >= P.nGroups) continue;
                    const bool keptHalf = sgHalf && !((gh >> (1 - hIdx)) & 1u), keptLine = sgLine && !((gl >> (1 - hIdx)) & 1u);
                    out[1] += keptHalf;
                    out[2] += keptLine;
                    out[4] += keptHalf && !keptLine;
                    for (int j = 0; j < TPT_GROUP; ++j) {
                        const f4 s = P.gsph[(size_t)g * TPT_GROUP + j];
                        const float coX = s.x - o.x, coY = s.y - o.y, coZ = s.z - o.z;
                        const float nb = coX * d.x + coY * d.y + coZ * d.z;
                        const float c = coX * coX + coY * coY + coZ * coZ - s.w;
                        const float discr = nb * nb - c;
                        if (discr > 0) {
                            const float sq = tsqrt(discr);
                            float t = nb - sq;
                            if (t <= TPT_MIN_T) t = nb + sq;
                            if (t > TPT_MIN_T) {
                                out[3]++;
                                if (!keptHalf) out[0]++;
                            }
                        }
                    }
                }
            }
        }
    }
}
void emu_group_half_check(const void* spheres, const void* mats, int count, const float* rays, int nRays, long long* out)
{
    std::vector<SpherePOD> S((const SpherePOD*)spheres, (const SpherePOD*)spheres + count);
    std::vector<MaterialPOD> M((const MaterialPOD*)mats, (const MaterialPOD*)mats + count);
    PackedScene P;
    packScene(S, M, P);
    for (int k = 0; k < 5; ++k) out[k] = 0;
    for (int i = 0; i < nRays && P.nSuperPairs > 0; ++i)
        groupHalfCheckRay(P, mk3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]), out);
}
// The same check for rays made to sit ON THE EDGE of the half-line rule: origins at (1 + delta) x the bound's radius from the centre of
// a group's or a super-group's bound (delta from -1e-3 to +1e-2 through 0 and through the rule's own margin 2^-12), directions
// tangential, a hair to either side of tangential, pointing away, and random.  perBound rays per bound and delta.
void emu_group_half_adversarial(const void* spheres, const void* mats, int count, unsigned seed, int perBound, long long* out)
{
    std::vector<SpherePOD> S((const SpherePOD*)spheres, (const SpherePOD*)spheres + count);
    std::vector<MaterialPOD> M((const MaterialPOD*)mats, (const MaterialPOD*)mats + count);
    PackedScene P;
    packScene(S, M, P);
    for (int k = 0; k < 5; ++k) out[k] = 0;
    if (P.nSuperPairs <= 0) return;
    uint32_t st = seed | 1u;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 17; st ^= st << 5; return (double)(st & 0xffffffu) / 16777216.0; };
    auto unit = [&](double* v) {
        double n2;
        do { for (int a = 0; a < 3; ++a) v[a] = 2.0 * rnd() - 1.0; n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; } while (n2 > 1.0 || n2 < 1e-4);
        const double inv = 1.0 / sqrt(n2);
        for (int a = 0; a < 3; ++a) v[a] *= inv;
    };
    const double deltas[] = {-1e-3, -1e-5, -1e-7, 0.0, 1e-7, 1e-5, 1.2e-4, 2.44e-4, 3.7e-4, 4.9e-4, 6e-4, 1e-3, 1e-2};
    auto sweep = [&](const std::vector<float>& recs, int nBounds, int stride) {
        for (int b = 0; b < nBounds; b += stride) {
            const float* rec = recs.data() + (size_t)(b / 2) * 8;
            const double C[3] = {rec[0 + (b & 1)], rec[2 + (b & 1)], rec[4 + (b & 1)]}, nsq = rec[6 + (b & 1)];
            if (!(nsq < 0) || !(nsq > -1e30)) continue; // padding / always-a-candidate records
            const double R = sqrt(-nsq / (1.0 + 1.0 / 4096.0));
            for (double dl : deltas)
                for (int k = 0; k < perBound; ++k) {
                    double u[3], t[3];
                    unit(u);
                    unit(t);
                    const double ut = u[0] * t[0] + u[1] * t[1] + u[2] * t[2];
                    for (int a = 0; a < 3; ++a) t[a] -= ut * u[a]; // tangential part
                    const double tn = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
                    if (tn < 1e-3) continue;
                    const double tilt = (k % 5 == 0) ? 1.0 : (k % 5 == 1) ? 0.0 : (k % 5 == 2) ? 1e-4 : (k % 5 == 3) ? -1e-4 : (2.0 * rnd() - 1.0);
                    double dd[3], n2 = 0;
                    for (int a = 0; a < 3; ++a) { dd[a] = t[a] / tn + tilt * u[a]; n2 += dd[a] * dd[a]; }
                    const double inv = 1.0 / sqrt(n2);
                    const f3 o = mk3((float)(C[0] + u[0] * R * (1.0 + dl)), (float)(C[1] + u[1] * R * (1.0 + dl)), (float)(C[2] + u[2] * R * (1.0 + dl)));
                    const f3 d = normalize(mk3((float)(dd[0] * inv), (float)(dd[1] * inv), (float)(dd[2] * inv)));
                    groupHalfCheckRay(P, o, d, out);
                }
        }
    };
    sweep(P.gpairs, P.nGroups, P.nGroups > 600 ? 7 : 1);
    sweep(P.spairs, P.nSupers, 1);
}
float emu_sinf(float x) { return tsinf(x); }
float emu_cosf(float x) { return tcosf(x); }
float emu_pow5f(float x) { return tpow5f(x); }
// tsincosf (one evaluation of each polynomial, signs applied to the binary32 results) against tsinf / tcosf (glibc's branch
// structure) for every float whose bit pattern lies in [loBits, hiBits], both signs: number of arguments whose sine or cosine differ
long long emu_sincos_pair_mismatches(unsigned loBits, unsigned hiBits, float* firstBad)
{
    long long bad = 0;
    for (unsigned long long b = loBits; b <= hiBits; ++b) {
        for (unsigned sgn = 0; sgn < 2; ++sgn) {
            const float y = u2f((uint32_t)b | (sgn << 31));
            float sn, cs;
            tsincosf(y, sn, cs);
            if (f2u(sn) != f2u(tsinf(y)) || f2u(cs) != f2u(tcosf(y))) {
                if (!bad && firstBad) *firstBad = y;
                ++bad;
            }
        }
    }
    return bad;
}
void emu_sincos_pair(float y, float* outSin, float* outCos) { tsincosf(y, *outSin, *outCos); }

// HitSpheres alone: n rays [n][6] = origin, unit direction; hs as tptSetKernelVariant numbers it: 0 = the product's default
// (conservative filter + exact test: matrix-core filter's restatement for scenes with a table, groups for large scenes),
// 1 = simple loop (the reference's arithmetic for every sphere), 2 = no groups, 3 = packed VALU filter everywhere.
// ids/ts must be identical between all of them.
void emu_hit_spheres(const void* spheres, const void* mats, int count, int hs, const float* rays, int n, int* outId, float* outT)
{
    std::vector<SpherePOD> S((const SpherePOD*)spheres, (const SpherePOD*)spheres + count);
    std::vector<MaterialPOD> M((const MaterialPOD*)mats, (const MaterialPOD*)mats + count);
    PackedScene P;
    packScene(S, M, P);
    SceneView sv = viewOf(P);
    if (hs == 2) sv.nGroups = 0;
    if (hs == 3) sv.mxR1 = -1;
    for (int i = 0; i < n; ++i) {
        f3 o = mk3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), d = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
        float t;
        outId[i] = hs == 1                       ? hitSpheres<HS_SIMPLE>(sv, o, d, TPT_MIN_T, TPT_MAX_T, t)
                   : (hs == 0 && sv.mxR1 >= 0) ? hitSpheres<HS_MATRIX>(sv, o, d, TPT_MIN_T, TPT_MAX_T, t)
                                               : hitSpheres<HS_TWO_PHASE_GROUPS>(sv, o, d, TPT_MIN_T, TPT_MAX_T, t);
        outT[i] = t;
    }
}

// Candidate masks of the matrix-core filter's host restatement (phase1MatrixHRef: same table, same ray slots, f32
// accumulation in slot order) and, per (ray, sphere), the EXACT slot sum and the sum of the slot products' magnitudes in
// binary64 (outSum / outAbs, [n][nSpheres], optional): the GPU test holds the device's sign bits against them within the
// error model's bound.  Returns mxR1 (< 0: no table).
int emu_matrix_masks(const void* spheres, const void* mats, int count, const float* rays, int n, unsigned long long* outMask, double* outSum, double* outAbs)
{
    std::vector<SpherePOD> S((const SpherePOD*)spheres, (const SpherePOD*)spheres + count);
    std::vector<MaterialPOD> M((const MaterialPOD*)mats, (const MaterialPOD*)mats + count);
    PackedScene P;
    packScene(S, M, P);
    if (P.mxR1 < 0) return -1;
    for (int i = 0; i < n; ++i) {
        f3 o = mk3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), d = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
        outMask[i] = phase1MatrixHRef(P.amatH.data(), P.mxR1, P.nSpheres, o, d, outSum ? outSum + (size_t)i * count : nullptr,
                                      outAbs ? outAbs + (size_t)i * count : nullptr);
    }
    return P.mxR1;
}

} // extern "C"
