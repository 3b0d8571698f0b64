// tpt_trace.h -- per-lane path-tracing logic of the MI355X renderer (HitSpheres, Scatter, Trace).
//
// Replaces the reference's L2 hot path (Cpp/Source/Test.cpp: HitWorld :76, Scatter :83-193,
// Trace :195-234, TraceRowJob :266-300; Maths.cpp HitSpheres :165-202; Camera::GetRay Maths.h:437)
// with a design made for a 64-lane wavefront instead of a recursive CPU call tree:
//
//  * Trace's recursion and Scatter's shadow-ray loop are flattened into ONE state machine per lane
//    (`laneStep`): every step intersects exactly one ray (camera, bounce or shadow ray) with the
//    whole scene and then advances the lane's path.  All lanes of a wave therefore share the
//    expensive part (the sphere loop) every step, whatever their path depth or material.
//  * HitSpheres is two-phase.  Phase 1 is branch-free and wave-uniform: for every sphere pair the
//    discriminant is computed with packed fp32 ops on scene data held in SGPRs (s_load), and only
//    its sign bit is kept (v_alignbit into a 64-bit candidate mask per lane).  Phase 2 walks the few
//    set bits of each lane (typically 1-4 of 46), gathers that sphere from LDS, recomputes the
//    identical discriminant and does sqrt / nearest-hit bookkeeping.  Scanning candidates in
//    ascending index with a strict `t < hitT` reproduces the reference's tie-break (lowest id).
//  * Results are bit-identical to the reference CPU scalar path: same operation order, no FMA
//    contraction, explicit RNG draw order, libm-free sin/cos/pow5 (tpt_math.h).
//
// Plain C++ shared by the gfx950 kernels (tpt_kernels.hip) and by the host-compiled lane-logic
// test in tests/ (never by the shipped library on the CPU).
#pragma once
#include "tpt_math.h"

namespace tpt {

#if defined(__clang__)
typedef float v2f __attribute__((ext_vector_type(2)));
#else
typedef float v2f __attribute__((vector_size(8)));
#endif
struct alignas(16) f4 {
    float x, y, z, w;
};

#define TPT_MIN_T 0.001f // kMinT, Test.cpp:71
#define TPT_MAX_T 1.0e7f // kMaxT, Test.cpp:72
#define TPT_MAX_DEPTH 10 // kMaxDepth, Test.cpp:73

enum { MAT_LAMBERT = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2 }; // Test.cpp:38
enum { SEED_ROW_SERIAL = 0, SEED_PER_PIXEL = 1 };            // Test.cpp:280 / ComputeShader.hlsl:380
enum { FOLD_RECURSIVE = 0, FOLD_FORWARD = 1 };               // Test.cpp:216 nesting / front-to-back
enum { HS_TWO_PHASE = 0, HS_SIMPLE = 1, HS_TWO_PHASE_GROUPS = 2, HS_MATRIX = 3 }; // _GROUPS: two-phase that also understands grouped scenes; _MATRIX: phase 1 on the matrix cores

// Camera: byte-for-byte the reference layout (Maths.h:444-449, 88 B) so GetSceneDesc can memcpy it.
struct CameraPOD {
    float origin[3], lowerLeftCorner[3], horizontal[3], vertical[3], uu[3], vv[3], ww[3];
    float lensRadius;
};

// Device-side view of the scene (all arrays built on the host by tptUpdate, see tpt_host.cpp).
//   pairs  : phase-1 stream, one 32-B record per sphere PAIR {cx0,cx1, cy0,cy1, cz0,cz1, sq0,sq1};
//            wave-uniform reads -> scalar loads; odd counts are padded with sq = -inf (never hit)
//   sph4   : {cx,cy,cz,sqRadius} per sphere for the phase-2 gather (staged into LDS by the kernel)
//   invR   : 1/radius per sphere (Maths.h:359)
//   mats   : 3 x f4 per sphere {albedo.xyz,type} {emissive.xyz,roughness} {ri, 1/ri, schlick's r0^2, -} (the last two: what
//            Scatter / schlick derive from ri alone, Test.cpp:168 / Maths.h:329-330, computed once on the host with the same IEEE operations)
//   lights : 2 x f4 per emissive sphere {cx,cy,cz,radius} {emissive.xyz, id}
struct SceneView {
    const float* pairs;
    const f4* sph4;
    const float* invR;
    const f4* mats;
    const f4* lights;
    int nSpheres, nPairs, nLights;
    // Grouped scenes (packScene builds them for >= TPT_GROUP_MIN_SPHERES spheres; nGroups == 0: flat brute force).
    // Small spheres are split (median cuts) into compact groups of <= TPT_GROUP; `gpairs` holds the groups'
    // bounding spheres in the pair-record format of phase 1, `gsph`/`gid` the members {centre, r^2} and their
    // original indices (padding: r^2 = -inf, id -1); the few big spheres (ground, lights) stay in a flat list.
    const float* gpairs;
    const f4* gsph;
    const int* gid;
    const f4* bsph;
    const int* bid;
    int nGroups, nGroupPairs, nBig;
    // second level: bounding spheres of TPT_SUPER consecutive groups each, pair-record format (packScene); 0 pairs: none
    const float* spairs;
    int nSuperPairs;
    // Matrix-core form of the phase-1 filter (scenes of <= 64 spheres in binary16 range; see phase1MatrixH): amatH =
    // [2][2][64][4] dwords, the A operands of the 2 x 2 v_mfma_f32_32x32x16_f16 a ray tile needs; mxR1 = rows per half of the
    // second sphere tile that hold spheres (0, 4, ..., 16); mxR1 < 0: no table.
    const uint32_t* amatH;
    int mxR1;
    // the same for the bounding spheres of a grouped scene: gmatH = [gmxTiles][2][2][64][4] dwords, one tile pair per 64 groups
    // (buildGroupMatrixTable); gmxTiles == 0: no table, the packed VALU filter runs over gpairs
    const uint32_t* gmatH;
    int gmxTiles;
    int flags; // SCENE_* bits (packScene)
};
enum { SCENE_LIGHT_R2_DIV_SAFE = 1 }; // every light's radius^2 lies in [2^-60, 2^60]: Scatter's r^2 / d^2 may take tdivSafeNum (tpt_math.h)
#ifndef TPT_GROUP
#define TPT_GROUP 8 /* members per group, <= 32.  8 since round 4: with the bounds on the matrix cores small groups are cheap to reject and cheaper
                       to visit -- 4096-sphere scene: 7.9 / 7.7 / 7.8 / 7.8 / 7.05 / 3.6 Gray/s at 4 / 6 / 8 / 12 / 16 / 32 (profiles/r04/r04_run15-16.log) */
#endif
#define TPT_GROUP_MIN_SPHERES 256
#ifndef TPT_SUPER
#define TPT_SUPER 8 /* groups per super-group: 8 or 16 (host packing and kernels alike; the three-stage dealing keeps one candidate bit per group of a super-group in a 32-bit word) */
#endif
static_assert(TPT_SUPER == 8 || TPT_SUPER == 16, "super-groups of 8 or 16 groups");

struct FrameConsts {
    CameraPOD cam;
    int width, height;
    int spp, frame;
    float invWidth, invHeight;
    float lerpFac;  // Test.cpp:272-276, computed on the host
    float invSpp;   // 1.0f / float(spp), Test.cpp:291
    int seedMode;
    int config;     // CFG_* bits: the reference's compile-time switches Config.h:24-25 at run time
};
enum { CFG_LIGHT_SAMPLING = 1, CFG_MITSUBA_COMPARE = 2 }; // DO_LIGHT_SAMPLING (default on), DO_MITSUBA_COMPARE (default off)

TPT_HD f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// Profiling build only (-DTPT_STATS, tools/): wave-level entry counts [i] and lane counts [32+i] per block.
#if defined(__HIPCC__) && defined(TPT_STATS)
__device__ unsigned long long g_tptStats[128];
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(TPT_STATS) && TPT_STATS < 2
#define TPT_STAT(i)                                                                  \
    do {                                                                             \
        unsigned long long m_ = __ballot(1);                                         \
        if ((int)__ffsll((long long)m_) - 1 == (int)(threadIdx.x & 63)) {            \
            atomicAdd(&g_tptStats[i], 1ull);                                         \
            atomicAdd(&g_tptStats[32 + (i)], (unsigned long long)__popcll(m_));      \
        }                                                                            \
    } while (0)
#else
#define TPT_STAT(i) \
    do {            \
    } while (0)
#endif
// -DTPT_STATS=2: only a handful of cheap event counters (slots 100..), so that the build still runs at full speed
#if defined(__HIP_DEVICE_COMPILE__) && defined(TPT_STATS)
#define TPT_COUNT(slot, n) atomicAdd(&g_tptStats[slot], (unsigned long long)(n))
#else
#define TPT_COUNT(slot, n) do { } while (0)
#endif
enum { ST_STEP = 0, ST_PHASE2 = 1, ST_CAMERA = 2, ST_SHADOW = 3, ST_SKY = 4, ST_HIT = 5, ST_LAMBERT = 6, ST_METAL = 7,
       ST_DIELECTRIC = 8, ST_LIGHTGEN = 9, ST_BOUNCE = 10, ST_FINISH = 11, ST_REFILL = 12, ST_CHUNK = 13, ST_PIXELDONE = 14,
       ST_DISKLOOP = 15, ST_SPHERELOOP = 16 };

TPT_HD uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> sh);
#endif
}

// ---------------------------------------------------------------- HitSpheres (Maths.cpp:165-202)
// exact per-sphere test shared by both variants; expression order is the reference's (:171-190)
TPT_HD void testSphere(f4 s, int i, f3 o, f3 d, float tMin, float& hitT, int& id)
{
    float coX = s.x - o.x;
    float coY = s.y - o.y;
    float coZ = s.z - o.z;
    float nb = coX * d.x + coY * d.y + coZ * d.z;
    float c = coX * coX + coY * coY + coZ * coZ - s.w;
    float discr = nb * nb - c;
    if (discr > 0) {
        float discrSq = tsqrt(discr);
        float t = nb - discrSq;
        if (t <= tMin) t = nb + discrSq;
        if (t > tMin && t < hitT) {
            id = i;
            hitT = t;
        }
    }
}

TPT_HD int hitSpheresSimple(const SceneView& sv, f3 o, f3 d, float tMin, float tMax, float& outT)
{
    float hitT = tMax;
    int id = -1;
    for (int i = 0; i < sv.nSpheres; ++i) testSphere(sv.sph4[i], i, o, d, tMin, hitT, id);
    outT = hitT;
    return id;
}

// The pair records are read-only for the whole launch and every lane reads the same address: on the device the
// pointer is retyped to the CONSTANT address space so the compiler emits scalar loads (s_load_dwordx8/x16 into
// SGPRs, fed to the VALU as scalar operands) instead of 64 identical vector loads.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const float __attribute__((address_space(4))) * PairPtr;
#else
typedef const float* PairPtr;
#endif
TPT_HD PairPtr pairPtr(const float* p)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (PairPtr)(p);
#pragma clang diagnostic pop
#else
    return p;
#endif
}

// Phase 1 is a CONSERVATIVE filter: it may pass a sphere the ray misses (phase 2 repeats the reference's exact
// arithmetic, discr > 0 test included, for everything that passes) but must never drop one the reference hits.
// That frees it from the reference's operation order: FMA chains, 11 packed instructions per pair instead of 16.
// With S = |co|^2 and A = sum |co_i d_i| <= |co||d|, the reference's rounded discriminant (16 roundings) and this one
// (10 roundings) are both within 13 u (S + r^2) of the real value nb^2 - S + r^2 (u = 2^-24, |d|^2 <= 1 + 1e-5), so
//   D_ref > 0  =>  D_here > -26 u (S + r^2)  =>  nb^2 (1 + 2^-17) - S + r^2 (1 + 2^-16) > 0   [a 2.5x wider margin],
// which is what is evaluated: the record carries -r^2 (1 + 2^-16) (packScene), and the direction is scaled by
// TPT_P1_K = 1 + 2^-18 once per ray (K^2 >= 1 + 2^-17; its own rounding, 1 u on nb, is covered by the slack).
// (Coordinates are assumed to stay below ~1e18 so S does not overflow; padding records carry +inf -> never pass.)
#define TPT_P1_K 1.000003814697265625f /* 1 + 2^-18 */
TPT_HD float fma1(float a, float b, float c)
{
    return __builtin_fmaf(a, b, c);
}
TPT_HD v2f fma2(v2f a, v2f b, v2f c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_fma(a, b, c);
#else
    v2f r = {__builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1])};
    return r;
#endif
}
// HALF: the half-line form for BOUNDS (groups, super-groups: records carrying -R^2 (1 + 2^-12) of a sphere that holds every member
// sphere; never for the spheres' own records).  The line test above keeps a bound the ray's LINE passes through, wherever: on the
// 4096-sphere scene a quarter of the candidates it keeps lie wholly behind the ray's origin (profiles/r06/r06_run34.log).  A bound is
// dropped as well when its centre is behind the origin (nb < 0) AND the origin is outside it by a margin, w = e' + 2^-11 (1.01) nsq > 0
// with e' = S - R'^2, i.e. S > R'^2 (1 + 2^-11).  Why that is safe: for t >= 0 and nb <= 0 every point of the ray is at least sqrt(S)
// from the centre (|o + t d - C|^2 = S - 2 t nb + t^2 >= S; a computed nb < 0 whose real value is a rounding error above zero costs
// 9 u^2 S).  A member the reference accepts at t > tMin > 0 has its hit point within sqrt(r^2 + O(30 u)(S_m + r^2)) of its own
// centre -- the reference's rounded roots of its rounded discriminant --, hence within a + that of C, whose square exceeds R^2 by at
// most 26 u (S + R^2)(1 + rho) <= 1 690 u (S + R^2) (rho <= 64: the argument of "Group bounds are looser ..." below) -- so such a
// point exists only if S (1 - 1 690 u) <= R^2 (1 + 1 690 u), i.e. S <= R^2 (1 + 2^-12.3).  The test asks for S > R^2 (1 + 2^-12)(1 +
// 2^-11) = R^2 (1 + 2^-10.4): 3.6 x that.  (A member whose centre is ahead while the bound's is behind is covered: the argument is
// about the hit POINT, which lies inside the bound.  Records of bounds that are "always a candidate" carry -inf: w = -inf, kept.
// Padding records carry +inf: dropped by the line test.)  In sign bits: dropped = sign(discr) | (sign(nb) & sign(-w)).
#define TPT_HALF_C 0.00049316406f /* 2^-11 x 1.01 */
template <bool HALF>
TPT_HD void phase1PairT(PairPtr rec, v2f ox, v2f oy, v2f oz, v2f dx, v2f dy, v2f dz, uint32_t& m)
{
    v2f cx = {rec[0], rec[1]}, cy = {rec[2], rec[3]}, cz = {rec[4], rec[5]}, nsq = {rec[6], rec[7]};
    v2f coX = cx - ox;
    v2f coY = cy - oy;
    v2f coZ = cz - oz;
    v2f nb = fma2(coZ, dz, fma2(coY, dy, coX * dx));
    v2f e = fma2(coZ, coZ, fma2(coY, coY, fma2(coX, coX, nsq))); // S - r^2 (1 + 2^-16)
    v2f discr = fma2(nb, nb, -e);
    if (HALF) {
        const v2f hc = {-TPT_HALF_C, -TPT_HALF_C};
        const v2f wn = fma2(nsq, hc, -e); // -w
        m = alignbit(m, (f2u(nb[0]) & f2u(wn[0])) | f2u(discr[0]), 31);
        m = alignbit(m, (f2u(nb[1]) & f2u(wn[1])) | f2u(discr[1]), 31);
    } else {
        m = alignbit(m, f2u(discr[0]), 31); // m = (m << 1) | sign(discr)
        m = alignbit(m, f2u(discr[1]), 31);
    }
}
TPT_HD void phase1Pair(PairPtr rec, v2f ox, v2f oy, v2f oz, v2f dx, v2f dy, v2f dz, uint32_t& m) { phase1PairT<false>(rec, ox, oy, oz, dx, dy, dz, m); }

// Candidate mask of up to 32 pair records (64 spheres): sphere k of the chunk sits at bit (63 - k).
template <bool HALF>
TPT_HD uint64_t phase1ChunkT(PairPtr rec, int cnt, v2f ox, v2f oy, v2f oz, v2f dx, v2f dy, v2f dz);
TPT_HD uint64_t phase1Chunk(PairPtr rec, int cnt, v2f ox, v2f oy, v2f oz, v2f dx, v2f dy, v2f dz) { return phase1ChunkT<false>(rec, cnt, ox, oy, oz, dx, dy, dz); }
template <bool HALF>
TPT_HD uint64_t phase1ChunkT(PairPtr rec, int cnt, v2f ox, v2f oy, v2f oz, v2f dx, v2f dy, v2f dz)
{
    const int c0 = cnt < 16 ? cnt : 16, c1 = cnt - c0;
    uint32_t m0 = 0, m1 = 0;
#pragma unroll 4
    for (int p = 0; p < c0; ++p) phase1PairT<HALF>(rec + p * 8, ox, oy, oz, dx, dy, dz, m0);
#pragma unroll 4
    for (int p = 0; p < c1; ++p) phase1PairT<HALF>(rec + (16 + p) * 8, ox, oy, oz, dx, dy, dz, m1);
    // candidates = sign bit clear (value >= +0)
    uint32_t cand0 = ~m0 << (32 - 2 * c0);
    uint32_t cand1 = c1 ? (~m1 << (32 - 2 * c1)) : 0u;
    return ((uint64_t)cand0 << 32) | cand1;
}

// Measured alternatives (profiles/r01/run8*.log, run9*.log): scalar VOP2 arithmetic instead of packed VOP3P is
// a wash (packed ops issue at half rate on gfx950); software-pipelining the scalar loads one 4-pair block
// ahead costs 64 more SGPRs, spills, and is 8-15 % slower than letting 3-4 waves per SIMD hide the latency.
// profiling build (-DTPT_STATS=2): s_memtime around the two phases, [120] phase-1 ticks, [121] phase-2 ticks,
// [122] phase-2 trips of a wave, [123] lanes busy in them (first active lane adds for the wave)
#if defined(__HIP_DEVICE_COMPILE__) && defined(TPT_STATS) && TPT_STATS >= 2
#define TPT_HS_STAMP(v)                    \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_waitcnt(0);         \
    const unsigned long long v = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0)
__shared__ unsigned long long g_hsLds[4]; // per-workgroup sums, flushed to g_tptStats[120..123] when the workgroup ends
#define TPT_HS_FIRST() ((int)__ffsll((long long)__ballot(1)) - 1 == (int)(threadIdx.x & 63))
#define TPT_HS_TRIP() (hsTrips_++)
#define TPT_HS_ADD(a, b, c)                                                                                         \
    do {                                                                                                            \
        const unsigned long long act_ = __ballot(1);                                                                \
        unsigned sum_ = hsTrips_, max_ = hsTrips_;                                                                  \
        for (int off_ = 32; off_ > 0; off_ >>= 1) {                                                                 \
            const unsigned os_ = __shfl_xor(sum_, off_, 64), om_ = __shfl_xor(max_, off_, 64);                      \
            const bool ok_ = (act_ >> ((threadIdx.x & 63) ^ off_)) & 1ull;                                          \
            sum_ += ok_ ? os_ : 0u;                                                                                 \
            max_ = (ok_ && om_ > max_) ? om_ : max_;                                                                \
        }                                                                                                           \
        if (TPT_HS_FIRST()) {                                                                                       \
            atomicAdd(&g_hsLds[0], (b) - (a));                                                                 \
            atomicAdd(&g_hsLds[1], (c) - (b));                                                                 \
            atomicAdd(&g_hsLds[2], (unsigned long long)max_);                                                  \
            atomicAdd(&g_hsLds[3], (unsigned long long)sum_);                                                  \
        }                                                                                                           \
    } while (0)
#else
#define TPT_HS_STAMP(v) do { } while (0)
#define TPT_HS_TRIP() do { } while (0)
#define TPT_HS_ADD(a, b, c) do { } while (0)
#endif
TPT_HD int hitSpheresTwoPhase(const SceneView& sv, f3 o, f3 d, float tMin, float tMax, float& outT)
{
    float hitT = tMax;
    int id = -1;
    const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
    const float kx = d.x * TPT_P1_K, ky = d.y * TPT_P1_K, kz = d.z * TPT_P1_K; // phase 1 only (see phase1Pair)
    const v2f dx = {kx, kx}, dy = {ky, ky}, dz = {kz, kz};
    for (int pb = 0; pb < sv.nPairs; pb += 32) { // chunks of 64 spheres
        int cnt = sv.nPairs - pb;
        if (cnt > 32) cnt = 32;
        unsigned hsTrips_ = 0;
        (void)hsTrips_;
        TPT_HS_STAMP(t0_);
        uint64_t cand = phase1Chunk(pairPtr(sv.pairs + (size_t)pb * 8), cnt, ox, oy, oz, dx, dy, dz);
        TPT_HS_STAMP(t1_);
        while (cand) {
            int k = __builtin_clzll(cand);
            cand &= ~(0x8000000000000000ull >> k);
            int i = pb * 2 + k;
            TPT_STAT(ST_PHASE2);
            TPT_HS_TRIP();
            testSphere(sv.sph4[i], i, o, d, tMin, hitT, id);
        }
        TPT_HS_STAMP(t2_);
        TPT_HS_ADD(t0_, t1_, t2_);
    }
    outT = hitT;
    return id;
}

// ---------------------------------------------------------------- phase 1 on the matrix cores (scenes of <= 64 spheres)
// The filter's discriminant is bilinear in a sphere vector and a ray vector once it is expanded around the coordinate
// origin instead of the ray origin:
//   D = (co.d)^2 - |co|^2 + r^2,  co = c - o
//     = (c.d)^2 + c.(2 o - 2 (o.d) d) + (r^2 - |c|^2) + ((o.d)^2 - |o|^2)
//     = sum_k a_k(sphere) b_k(ray),  k = 0..10:
//        a = { cx^2, cy^2, cz^2, 2 cx cy, 2 cx cz, 2 cy cz,  2 cx, 2 cy, 2 cz,  r^2 - |c|^2 + m_s,  1 }
//        b = { dx^2, dy^2, dz^2,   dx dy,   dx dz,   dy dz,   ex,   ey,   ez,                  1,  q },
//   e = o - (o.d) d,  q = (o.d)^2 - |o|^2 (1 - 2^-13),  m_s = 2^-13 |c|^2 + 2^-14 r^2 + 2^-20.
// That is a [spheres x K] x [K x rays] product, and the matrix pipe -- which this VALU-bound kernel otherwise leaves idle --
// runs it at 16x the rate of the f32 vector unit if the operands are binary16.  They are made so WITHOUT giving up f32
// accuracy: every factor is split into two binary16 pieces, x = hi + lo (hi = x truncated to binary16, lo = x - hi truncated
// likewise), and a product a b becomes a_hi b_hi + a_hi b_lo + a_lo b_hi -- exact binary16 x binary16 products accumulated
// in f32 by v_mfma_f32_32x32x16_f16; only a_lo b_lo is dropped.  32 K-slots per (sphere, ray) = two MFMAs per 32 x 32 tile:
//   slots 2t, 2t+1      (t = 0..8): A = {a_hi[t], a_hi[t]},        B = {b_hi[t], b_lo[t]}
//   slots 18+2u, 19+2u  (u = 0..3): A = {a_lo[2u], a_lo[2u+1]},    B = {b_hi[2u], b_hi[2u+1]}
//   slots 26, 27:                   A = {a_lo[8], 1},              B = {b_hi[8], q_hi}
//   slots 28, 29:                   A = {1, a9_hi},                B = {q_lo, 1}
//   slots 30, 31:                   A = {a9_lo, 0},                B = {1, 0}
// What is left for the VALU per 64 rays: 17 instructions for b, 35 to split and pack (v_and / v_sub / v_cvt_pkrtz), 8 lane
// swaps to lay b out as B operands, one v_alignbit per (sphere, ray) for the sign, one swap to bring a ray's two row groups
// together -- ~115 instead of the packed filter's 276 (configs[1]: 43.3 -> 55.2 Gray/s, configs[2]: 49.5 -> 65.1).
//
// Still a CONSERVATIVE filter (phase 2 re-tests everything that passes with the reference's arithmetic).  What has to hold:
// D_ref > 0 (the reference's rounded discriminant, within 13 u (|co|^2 + r^2) of the real one, u = 2^-24: phase1Pair)  =>
// the sum the MFMA delivers is >= 0.  With T = sum |a_k b_k| <= 2 (|c| + |o|)^2 + r^2, the delivered sum differs from
// D + m (m = m_s + 2^-13 |o|^2) by at most
//    2 u T                     a_k rounded once from binary64; b_k: one f32 rounding each (products of d) ...
//  + 17 u (|c| + |o|)^2        ... and e, q: the 3-rounding dot products o.d, |o|^2 feeding them
//  + 48 u T                    the split: |x - hi - lo| <= 2^-20 |x| = 16 u |x| on either side, and the dropped a_lo b_lo <= 16 u |a b|
//  + 2 u (3 |c|^2 + 4 |c| + 2 |o| + 5)   binary16 subnormals: hi / lo below 2^-14 are kept to 2^-24 absolute (the conversion
//                              and the MFMA honour subnormals on this device: tools/exhaustive/mfma_f16_probe.hip)
//  + 64 u T                    the MFMA's own accumulation (32 products + the carried sum; measured on the device <= 2 ulp
//                              of the largest term per instruction, i.e. ~8 u T; 64 u T is the bound the tests enforce:
//                              test_matrix_filter_sign_agrees_with_the_exact_slot_sum)
// <= 245 u (|c| + |o|)^2 + 114 u r^2 + ..., and with the reference's 13 u: < 520 u (|c|^2 + |o|^2) + 130 u r^2 + 10 u,
// against the slack m = 2048 u (|c|^2 + |o|^2) + 1024 u r^2 + 16 u: a factor 3.9 to spare (tests/adversarial_filter.cpp
// searches near-tangent configurations over six decades of scale for a miss of the WORST CASE, exact slot sum minus 64 u T).
// For a 0.5-radius sphere eight units from the origin the slack inflates r^2 by 3 %.  Rays the binary16 range cannot carry
// (|o|^2 >= 60000, or a direction that is not finite and about unit length) keep every sphere as a candidate; spheres it
// cannot carry (|a_k| >= 60000, e.g. a 1000-unit ground sphere) or scenes of more than 64 spheres have no table and run the
// packed VALU filter (phase1Pair).
#define TPT_MX_K 11
#define TPT_MXH_TERMS 9 /* product terms a_t b_t, t = 0..8; b_9 = 1 and a_10 = 1 are the two special terms */
#define TPT_MXH_SLOTS 32
#define TPT_MXH_TABLE_DWORDS (2 * 2 * 64 * 4) /* [sphere tile][k step][lane][4]: the A operands as the lanes read them */
TPT_HD void matrixRaySide(f3 o, f3 d, float* b, int slackShift = 0) // slackShift 1: the group-bound table's doubled slack (2^-12 |o|^2)
{
    b[0] = d.x * d.x; b[1] = d.y * d.y; b[2] = d.z * d.z;
    b[3] = d.x * d.y; b[4] = d.x * d.z; b[5] = d.y * d.z;
    const float od = fma1(o.z, d.z, fma1(o.y, d.y, o.x * d.x));
    b[6] = fma1(-od, d.x, o.x); b[7] = fma1(-od, d.y, o.y); b[8] = fma1(-od, d.z, o.z);
    const float oo = fma1(o.z, o.z, fma1(o.y, o.y, o.x * o.x));
    b[9] = 1.0f;
    b[10] = fma1(od, od, -(oo * (slackShift ? 0.999755859375f : 0.9998779296875f))); // 1 - 2^-13 (groups: 1 - 2^-12)
}
// rays the binary16 operands can carry: |o|^2 < 60000 (so |e|, |q| fit) and a finite direction of about unit length
TPT_HD bool matrixRayInRange(f3 o, const float* b)
{
    const float oo = fma1(o.z, o.z, fma1(o.y, o.y, o.x * o.x)), dd = b[0] + b[1] + b[2];
    return oo < 60000.0f && dd < 2.0f;
}
TPT_HD float f16hi(float x) { return u2f(f2u(x) & 0xffffe000u); } // x truncated to 11 significant bits (= binary16 for |x| >= 2^-14)
// binary16 bit pattern of x rounded toward zero (x finite, |x| < 65520) and its value -- what v_cvt_pkrtz_f16_f32 delivers,
// subnormals included -- for the sphere-side table and the host restatement
TPT_HD uint32_t f16rtz(float x)
{
    const uint32_t u = f2u(x), sign = (u >> 16) & 0x8000u;
    const int e = (int)((u >> 23) & 0xffu) - 127;
    const uint32_t m = (u & 0x7fffffu) | 0x800000u;
    if (e < -24 || (u & 0x7fffffffu) == 0u) return sign;
    if (e < -14) return sign | (m >> (-1 - e)); // subnormal: multiples of 2^-24
    return sign | ((uint32_t)(e + 15) << 10) | ((u >> 13) & 0x3ffu);
}
TPT_HD float f16val(uint32_t h)
{
    const int e = (int)((h >> 10) & 31u), f = (int)(h & 0x3ffu);
    const float v = e == 0 ? (float)f * 5.9604644775390625e-08f : (float)(f | 0x400) * u2f((uint32_t)(e - 25 + 127) << 23); // f 2^-24 : (1.f) 2^(e-15)
    return (h & 0x8000u) ? -v : v;
}
// the 32 B-side slot values of a ray, as the device packs them
TPT_HD void matrixRaySlots(const float* b, float* slot)
{
    float hi[TPT_MXH_TERMS];
    for (int t = 0; t < TPT_MXH_TERMS; ++t) {
        hi[t] = f16hi(b[t]);
        slot[2 * t] = f16val(f16rtz(hi[t]));
        slot[2 * t + 1] = f16val(f16rtz(b[t] - hi[t]));
    }
    for (int u = 0; u < 4; ++u) {
        slot[18 + 2 * u] = slot[4 * u];      // b_hi[2u]
        slot[19 + 2 * u] = slot[4 * u + 2];  // b_hi[2u + 1]
    }
    const float qh = f16hi(b[10]);
    slot[26] = slot[16];
    slot[27] = f16val(f16rtz(qh));
    slot[28] = f16val(f16rtz(b[10] - qh));
    slot[29] = 1.0f;
    slot[30] = 1.0f;
    slot[31] = 0.0f;
}
// position of sphere p in the table: tile mt, row i of the A operand.  The accumulator register r of lane l holds row
// 8 (r / 4) + 4 (l / 32) + r % 4 of column l % 32: a lane's 16 + R1 sign bits, taken in register order through tile 0 then
// tile 1, belong to spheres g n + 0 .. g n + n - 1 (g = l / 32, n = 16 + R1) -- so that the assembled 64-bit mask lists the
// spheres in ascending index (phase 2's tie-break depends on that order).
TPT_HD void matrixSlot(int p, int R1, int& mt, int& row)
{
    const int n = 16 + R1, g = p / n, q = p % n;
    mt = q < 16 ? 0 : 1;
    const int r = q < 16 ? q : q - 16;
    row = 8 * (r / 4) + 4 * g + r % 4;
}
// A-side slot value of the sphere in (mt, row) from the table
TPT_HD float matrixTableSlot(const uint32_t* amatH, int mt, int row, int slot)
{
    const int j = slot / 16, within = slot % 16, lane = row + 32 * (within / 8), w = (within % 8) / 2;
    const uint32_t d = amatH[((mt * 2 + j) * 64 + lane) * 4 + w];
    return f16val((slot & 1) ? (d >> 16) : (d & 0xffffu));
}
// Host restatement of phase1MatrixH for the CPU tests: same table, same ray slots, the 32 products accumulated in f32 one
// after the other (the MFMA's own order and internal width are the hardware's: within the bound above of each other, so the
// two masks may differ where the sum is within ~100 u T of zero -- both are conservative).  outSum / outAbs (optional, per
// sphere): the exact slot sum and sum of magnitudes in binary64, for the tests of the error model.
TPT_HD uint64_t phase1MatrixHRef(const uint32_t* amatH, int R1, int nSpheres, f3 o, f3 d, double* outSum = nullptr, double* outAbs = nullptr, int slackShift = 0)
{
    float b[TPT_MX_K], slot[TPT_MXH_SLOTS];
    matrixRaySide(o, d, b, slackShift);
    matrixRaySlots(b, slot);
    const bool ok = matrixRayInRange(o, b);
    uint64_t cand = 0;
    for (int p = 0; p < nSpheres; ++p) {
        int mt, row;
        matrixSlot(p, R1, mt, row);
        float c = 0.0f;
        double cs = 0.0, ca = 0.0;
        for (int k = 0; k < TPT_MXH_SLOTS; ++k) {
            const float av = matrixTableSlot(amatH, mt, row, k);
            c = c + av * slot[k]; // (a product of two binary16 values is exact in f32)
            cs += (double)av * (double)slot[k];
            ca += (double)av * (double)slot[k] < 0 ? -((double)av * (double)slot[k]) : (double)av * (double)slot[k];
        }
        if (outSum) outSum[p] = cs;
        if (outAbs) outAbs[p] = ca;
        if (!ok || (f2u(c) >> 31) == 0u) cand |= 0x8000000000000000ull >> p;
    }
    return cand;
}
// The ray side of the filter as MFMA B operands: what the lane's ray contributes to ray tile 0 / 1 of the two k steps.
struct MatrixRayOps {
    uint32_t B0[2][4], B1[2][4];
    bool ok; // the ray is in binary16 range (else: every sphere stays a candidate)
};
#if defined(__HIPCC__) && !defined(__HIP_DEVICE_COMPILE__)
// (host pass of a .hip file: declarations only)
__device__ uint64_t phase1MatrixH(const uint32_t* ldsA, int R1, int nSpheres, f3 o, f3 d);
__device__ void matrixRayOperands(f3 o, f3 d, int slackShift, MatrixRayOps& R);
__device__ uint64_t matrixApply(const uint4* A, int R1, int nEntries, const MatrixRayOps& R);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __fp16 v2h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pkh(float a, float b) // {binary16(a), binary16(b)} (round toward zero), a in the low half
{
    const v2h h = __builtin_amdgcn_cvt_pkrtz(a, b);
    uint32_t u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}
__device__ __forceinline__ void matrixRayOperands(f3 o, f3 d, int slackShift, MatrixRayOps& R)
{
    float b[TPT_MX_K];
    matrixRaySide(o, d, b, slackShift);
    R.ok = matrixRayInRange(o, b);
    uint32_t V[16];
    float hi[TPT_MXH_TERMS];
#pragma unroll
    for (int t = 0; t < TPT_MXH_TERMS; ++t) {
        hi[t] = f16hi(b[t]);
        V[t] = pkh(hi[t], b[t] - hi[t]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) V[9 + u] = pkh(hi[2 * u], hi[2 * u + 1]);
    const float qh = f16hi(b[10]);
    V[13] = pkh(hi[8], qh);
    V[14] = pkh(b[10] - qh, 1.0f);
    V[15] = 0x00003c00u; // {1, 0}
    // B operands: MFMA j of ray tile 0 / 1 reads slots 16 j .. 16 j + 7 from lanes 0..31 and 16 j + 8 .. 16 j + 15 from lanes
    // 32..63 -- of the SAME ray: one half-swap per register pair hands every lane's upper slots to its partner lane
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            auto sw = __builtin_amdgcn_permlane32_swap(V[8 * j + r], V[8 * j + 4 + r], false, false);
            R.B0[j][r] = sw[0];
            R.B1[j][r] = sw[1];
        }
}
// Candidate mask (entry p at bit 63 - p) of the ray in this lane against ONE table of up to 64 entries (spheres, or the
// bounding spheres of 64 groups).  Must be called by ALL 64 lanes of the wave (lanes without a ray ignore the result; a lane's
// operands reach its own column only, so whatever it feeds cannot disturb another ray).  A: the table's A operands,
// [sphere tile][k step][lane] uint4, in LDS or global memory.
__device__ __forceinline__ uint64_t matrixApply(const uint4* A, int R1, int nEntries, const MatrixRayOps& R)
{
    const int lane = (int)(threadIdx.x & 63u);
    uint32_t W0 = 0, W1 = 0; // sign bits of this lane's accumulator rows, ray tile 0 / 1
    auto asH = [](const uint32_t* p) {
        v8h h;
        __builtin_memcpy(&h, p, 16);
        return h;
    };
    {
        v16f c0 = {0}, c1 = {0};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint4 a4 = A[j * 64 + lane];
            const uint32_t aw[4] = {a4.x, a4.y, a4.z, a4.w};
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asH(aw), asH(R.B0[j]), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asH(aw), asH(R.B1[j]), c1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            W0 = alignbit(W0, f2u(c0[r]), 31);
            W1 = alignbit(W1, f2u(c1[r]), 31);
        }
    }
    if (R1 > 0) {
        v16f c0 = {0}, c1 = {0};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint4 a4 = A[(2 + j) * 64 + lane];
            const uint32_t aw[4] = {a4.x, a4.y, a4.z, a4.w};
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asH(aw), asH(R.B0[j]), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(asH(aw), asH(R.B1[j]), c1, 0, 0, 0);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            if (R1 > 4 * r4) {
#pragma unroll
                for (int r = 4 * r4; r < 4 * r4 + 4; ++r) {
                    W0 = alignbit(W0, f2u(c0[r]), 31);
                    W1 = alignbit(W1, f2u(c1[r]), 31);
                }
            }
        }
    }
    // rays 0..31 sit in ray tile 0, whose rows are split between lane j (row group 0) and lane j + 32 (row group 1);
    // rays 32..63 likewise in tile 1: one swap gives every lane both row groups of its own ray
    auto sw = __builtin_amdgcn_permlane32_swap(W0, W1, false, false);
    const uint32_t G0 = sw[0], G1 = sw[1];
    const int n = 16 + R1; // entries per row group
    const uint64_t rejected = ((uint64_t)G0 << (64 - n)) | ((uint64_t)G1 << (64 - 2 * n));
    const uint64_t valid = ~0ull << (64 - nEntries); // (padding rows end at -inf, but a NaN ray must not keep them either)
    return R.ok ? (~rejected & valid) : valid;
}
__device__ __forceinline__ uint64_t phase1MatrixH(const uint32_t* ldsA, int R1, int nSpheres, f3 o, f3 d)
{
    MatrixRayOps R;
    matrixRayOperands(o, d, 0, R);
    return matrixApply(reinterpret_cast<const uint4*>(ldsA), R1, nSpheres, R);
}
#endif
// phase 2 over a candidate mask (sphere p at bit 63 - p): the reference's arithmetic, ascending index
TPT_HD int hitSpheresCandidates(const SceneView& sv, uint64_t cand, f3 o, f3 d, float tMin, float tMax, float& outT)
{
    float hitT = tMax;
    int id = -1;
    unsigned hsTrips_ = 0;
    (void)hsTrips_;
    TPT_HS_STAMP(t1_);
    while (cand) {
        const int i = __builtin_clzll(cand);
        cand &= ~(0x8000000000000000ull >> i);
        TPT_STAT(ST_PHASE2);
        TPT_HS_TRIP();
        testSphere(sv.sph4[i], i, o, d, tMin, hitT, id);
    }
    TPT_HS_STAMP(t2_);
    TPT_HS_ADD(t1_, t1_, t2_);
    outT = hitT;
    return id;
}

// ---------------------------------------------------------------- grouped HitWorld (large scenes)
// Same result as the flat loops, sphere for sphere: every sphere the reference's test accepts is still tested with
// the reference's arithmetic (testSphere's), only the ORDER differs -- so the winner among equal t is chosen
// explicitly (lowest original index, what ascending order + strict t < hitT gives the reference).
TPT_HD void testSphereTie(f4 s, int i, f3 o, f3 d, float tMin, float& hitT, int& id)
{
    float coX = s.x - o.x;
    float coY = s.y - o.y;
    float coZ = s.z - o.z;
    float nb = coX * d.x + coY * d.y + coZ * d.z;
    float c = coX * coX + coY * coY + coZ * coZ - s.w;
    float discr = nb * nb - c;
    if (discr > 0) {
        float discrSq = tsqrt(discr);
        float t = nb - discrSq;
        if (t <= tMin) t = nb + discrSq;
        if (t > tMin && (t < hitT || (t == hitT && i < id))) {
            id = i;
            hitT = t;
        }
    }
}
// per-lane version of phase1Pair's conservative filter for one sphere (dk = direction scaled by TPT_P1_K): the value whose SIGN BIT
// decides (clear: the sphere stays a candidate)
TPT_HD float memberFilterValue(f4 s, f3 o, f3 dk)
{
    float coX = s.x - o.x;
    float coY = s.y - o.y;
    float coZ = s.z - o.z;
    float nb = fma1(coZ, dk.z, fma1(coY, dk.y, coX * dk.x));
    float e = fma1(coZ, coZ, fma1(coY, coY, fma1(coX, coX, s.w * -1.0000152587890625f))); // S - r^2 (1 + 2^-16); padding: +inf
    return fma1(nb, nb, -e);
}
TPT_HD bool memberFilter(f4 s, f3 o, f3 dk) { return (f2u(memberFilterValue(s, o, dk)) >> 31) == 0u; }
// Group bounds are looser than sphere bounds on purpose.  A member the reference accepts lies within
// sqrt(r^2 + 13 u (S + r^2)) of the ray's line (the reference's own rounding error), and its centre at most a = |c - C|
// from the group centre C: against R = max(a + r) the squared distance to C overshoots R^2 by up to
// 13 u (S_i + r^2)(1 + a / r) <= 26 u (S + R^2)(1 + rho), rho = max a / r.  Together with the filter's own 13 u (S + R^2)
// that must stay below the slack tau (S + R^2) the filter grants; with the direction scaled by 1 + 2^-13 (square
// >= 1 + 2^-12) and the record carrying -R^2 (1 + 2^-12), tau = 2^-13 = 2 048 u (same algebra as phase1Pair), i.e.
// rho <= 77.  packScene groups a scene only if rho <= 64 for every group and rounds R up.
#define TPT_PG_K 1.0001220703125f /* 1 + 2^-13: (1 + 2^-13)^2 > 1 + 2^-12 */
// Per-lane FIFO of up to 8 member slots (16 bits each) in two 64-bit words: exact tests are deferred so that the lanes of
// a wave run them together (tested right where they are found, 3.5 of 64 lanes were busy per trip on the stress scene).
struct SlotQueue {
    uint64_t a, b;
    int n;
};
TPT_HD void sqPush(SlotQueue& q, uint32_t slot)
{
    q.b = (q.b << 16) | (q.a >> 48);
    q.a = (q.a << 16) | (uint64_t)(slot & 0xffffu);
    q.n++;
}
TPT_HD uint32_t sqPop(SlotQueue& q) // most recent first; the order is irrelevant (explicit tie-break)
{
    const uint32_t slot = (uint32_t)(q.a & 0xffffull);
    q.a = (q.a >> 16) | (q.b << 48);
    q.b >>= 16;
    q.n--;
    return slot;
}
TPT_HD void sqDrain(SlotQueue& q, const SceneView& sv, f3 o, f3 d, float tMin, float& hitT, int& id)
{
    while (q.n > 0) {
        const uint32_t slot = sqPop(q);
        TPT_STAT(ST_PHASE2);
        testSphereTie(sv.gsph[slot], sv.gid[slot], o, d, tMin, hitT, id);
    }
}
TPT_HD int hitSpheresGrouped(const SceneView& sv, f3 o, f3 d, float tMin, float tMax, float& outT)
{
    float hitT = tMax;
    int id = -1;
    // the big spheres: exact test each (a handful)
    for (int b = 0; b < sv.nBig; ++b) testSphereTie(sv.bsph[b], sv.bid[b], o, d, tMin, hitT, id);
    const f3 dk = mk3(d.x * TPT_P1_K, d.y * TPT_P1_K, d.z * TPT_P1_K);
    const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
    const float gx = d.x * TPT_PG_K, gy = d.y * TPT_PG_K, gz = d.z * TPT_PG_K;
    const v2f dx = {gx, gx}, dy = {gy, gy}, dz = {gz, gz};
    SlotQueue q;
    q.a = q.b = 0ull;
    q.n = 0;
    const bool slots16 = sv.nGroups * TPT_GROUP <= 65536; // else: no deferral (slots would not fit 16 bits)
    for (int pb0 = 0; pb0 < sv.nGroupPairs; pb0 += 128) {
        // wave-uniform filter on the bounding spheres of up to 256 groups: four 64-bit candidate masks per lane
        uint64_t cm0 = 0, cm1 = 0, cm2 = 0, cm3 = 0;
        {
            const int left = sv.nGroupPairs - pb0;
            cm0 = phase1Chunk(pairPtr(sv.gpairs + (size_t)pb0 * 8), left < 32 ? left : 32, ox, oy, oz, dx, dy, dz);
            if (left > 32) cm1 = phase1Chunk(pairPtr(sv.gpairs + (size_t)(pb0 + 32) * 8), left - 32 < 32 ? left - 32 : 32, ox, oy, oz, dx, dy, dz);
            if (left > 64) cm2 = phase1Chunk(pairPtr(sv.gpairs + (size_t)(pb0 + 64) * 8), left - 64 < 32 ? left - 64 : 32, ox, oy, oz, dx, dy, dz);
            if (left > 96) cm3 = phase1Chunk(pairPtr(sv.gpairs + (size_t)(pb0 + 96) * 8), left - 96 < 32 ? left - 96 : 32, ox, oy, oz, dx, dy, dz);
        }
        // per lane: ONE loop over the groups this ray's line touches, whichever mask word they sit in (a loop per word
        // costs the sum of the per-word maxima over the lanes instead of the maximum of the sums)
        for (;;) {
            const int sel = cm0 ? 0 : cm1 ? 1 : cm2 ? 2 : 3;
            const uint64_t w = cm0 ? cm0 : cm1 ? cm1 : cm2 ? cm2 : cm3;
            if (!w) break;
            const int k = __builtin_clzll(w);
            const uint64_t keep = ~(0x8000000000000000ull >> k);
            cm0 &= sel == 0 ? keep : ~0ull;
            cm1 &= sel == 1 ? keep : ~0ull;
            cm2 &= sel == 2 ? keep : ~0ull;
            cm3 &= sel == 3 ? keep : ~0ull;
            const int base = ((pb0 + sel * 32) * 2 + k) * TPT_GROUP;
            const f4* mem = sv.gsph + base;
            TPT_STAT(ST_SPHERELOOP); // profiling build: group visits
            uint32_t mm = 0;
#pragma unroll 4
            for (int j = 0; j < TPT_GROUP; ++j) mm |= (memberFilter(mem[j], o, dk) ? 1u : 0u) << j;
            while (mm) {
                const int j = __builtin_ctz(mm);
                mm &= mm - 1u;
                if (slots16) {
                    if (q.n == 8) sqDrain(q, sv, o, d, tMin, hitT, id);
                    sqPush(q, (uint32_t)(base + j));
                } else {
                    TPT_STAT(ST_PHASE2);
                    testSphereTie(mem[j], sv.gid[base + j], o, d, tMin, hitT, id);
                }
            }
        }
    }
    sqDrain(q, sv, o, d, tMin, hitT, id);
    outT = hitT;
    return id;
}

template <int HS>
TPT_HD int hitSpheres(const SceneView& sv, f3 o, f3 d, float tMin, float tMax, float& outT)
{
    if (HS == HS_SIMPLE) return hitSpheresSimple(sv, o, d, tMin, tMax, outT);
#if !defined(__HIP_DEVICE_COMPILE__)
    // host tests: the matrix filter's restatement (on the device the kernels call phase1MatrixH wave-wide themselves)
    if (HS == HS_MATRIX && sv.mxR1 >= 0) return hitSpheresCandidates(sv, phase1MatrixHRef(sv.amatH, sv.mxR1, sv.nSpheres, o, d), o, d, tMin, tMax, outT);
#endif
    // (kernels that stage the scene in LDS are instantiated without the grouped code: such scenes are small and never
    //  grouped, and the extra registers cost the 46-sphere kernel 4 %)
    if (HS == HS_TWO_PHASE_GROUPS && sv.nGroups > 0) return hitSpheresGrouped(sv, o, d, tMin, tMax, outT);
    return hitSpheresTwoPhase(sv, o, d, tMin, tMax, outT);
}

// ---------------------------------------------------------------- camera (Maths.h:437-442) / sky (Test.cpp:229-231)
TPT_HD void cameraGetRay(const CameraPOD& c, float s, float t, uint32_t& state, f3& orig, f3& dir)
{
    f3 rd = c.lensRadius * randomInUnitDisk(state);
    f3 offset = ld3(c.uu) * rd.x + ld3(c.vv) * rd.y;
    orig = ld3(c.origin) + offset;
    dir = normalize(ld3(c.lowerLeftCorner) + s * ld3(c.horizontal) + t * ld3(c.vertical) - ld3(c.origin) - offset);
}
TPT_HD f3 sky(f3 dir)
{
    float t = 0.5f * (dir.y + 1.0f);
    return ((1.0f - t) * mk3(1.0f, 1.0f, 1.0f) + t * mk3(0.5f, 0.7f, 1.0f)) * 0.3f;
}

// ---------------------------------------------------------------- the per-lane state machine
enum { KIND_MAIN = 0, KIND_SHADOW = 1 };

struct Lane {
    uint32_t rng;
    int x, y;     // pixel in full-image coordinates (y = 0 is the bottom row, Test.cpp:287)
    int pix;      // float4 index into the (local) backbuffer
    int sample, depth;
    int kind, j, hitId, hitType;
    bool active, needCamera, doMatE;
    f3 orig, dir;                              // the ray the next step intersects
    f3 sdir, nl, albedo, lightE, matE;         // Lambert light-loop context (Test.cpp:89-134)
    float cosAMax;
    f3 col;                                    // sum over samples (Test.cpp:283-290)
    f3 radiance, throughput;                   // FOLD_FORWARD
    int sp;                                    // FOLD_RECURSIVE: entries on the bounce stack
    int item;                                  // persistent kernel: work-item index of the current pixel (for the chunk cost statistics)
    int frameIdx;                              //   batched row-serial launch: which frame of the batch the lane's row belongs to
    uint32_t rays0;                            //   and the lane's ray count when the pixel started
    uint32_t rays;
};

// Bounce stack for FOLD_RECURSIVE: entry d = {matE+lightE (xyz), attenuation id}.  attId >= 0 ->
// attenuation = albedo of material attId, -1 -> (1,1,1) (dielectric, Test.cpp:158).
// Device: LDS, laid out [level][thread]; host test: plain array.
struct BounceStack {
    f4* base;       // levels [0, fastLevels): LDS on the device (host test: plain array)
    int stride;     // elements between consecutive levels
    f4* spill;      // levels [fastLevels, TPT_MAX_DEPTH): global memory (only ~3 % of the samples get there)
    int spillStride;
    int fastLevels;
    TPT_HD void push(int level, f3 e, int attId) const
    {
        f4 v;
        v.x = e.x; v.y = e.y; v.z = e.z; v.w = u2f((uint32_t)attId);
        if (level < fastLevels)
            base[level * stride] = v;
        else
            spill[(level - fastLevels) * spillStride] = v;
    }
    TPT_HD f4 get(int level) const
    {
        if (level < fastLevels) return base[level * stride];
        return spill[(level - fastLevels) * spillStride];
    }
};

TPT_HD uint32_t pixelSeed(int seedMode, int x, int y, int frame)
{
    if (seedMode == SEED_PER_PIXEL) // ComputeShader.hlsl:380
        return ((uint32_t)x * 1973u + (uint32_t)y * 9277u + (uint32_t)frame * 26699u) | 1u;
    return ((uint32_t)y * 9781u + (uint32_t)frame * 6271u) | 1u; // Test.cpp:280 (row start)
}

// Start a pixel.  In ROW_SERIAL mode only the first pixel of a row reseeds (reseed = true).
TPT_HD void laneBeginPixel(Lane& L, const FrameConsts& fc, int x, int y, int pix, bool reseed)
{
    L.x = x; L.y = y; L.pix = pix;
    if (reseed) L.rng = pixelSeed(fc.seedMode, x, y, fc.frame);
    L.sample = 0;
    L.col = mk3(0, 0, 0);
    L.needCamera = true;
    L.active = true;
}

// New sample: camera ray (Test.cpp:286-289).
template <int FOLD>
TPT_HD void laneCamera(Lane& L, const FrameConsts& fc)
{
    TPT_STAT(ST_CAMERA);
    float u = ((float)L.x + rnd01(L.rng)) * fc.invWidth;
    float v = ((float)L.y + rnd01(L.rng)) * fc.invHeight;
    cameraGetRay(fc.cam, u, v, L.rng, L.orig, L.dir);
    L.depth = 0;
    L.doMatE = true;
    L.kind = KIND_MAIN;
    L.needCamera = false;
    if (FOLD == FOLD_FORWARD) {
        L.radiance = mk3(0, 0, 0);
        L.throughput = mk3(1, 1, 1);
    } else {
        L.sp = 0;
    }
}

// Everything after HitWorld for one ray of this lane: Scatter / light sampling / bounce / fold.
// Returns true when the lane's pixel is complete (L.col then holds the sum over spp samples).
// POS_GIVEN: L.orig already holds the hit position orig + dir * t (the path-queue kernel computes it right after the
// intersection and keeps it instead of {orig, t} in the path record)
template <int FOLD, bool POS_GIVEN = false>
TPT_HD bool lanePost(Lane& L, const int id, const float t, const SceneView& sv, const FrameConsts& fc, const BounceStack& stack)
{
    bool lightLoop = false, finish = false, bounce = false;
    f3 term = mk3(0, 0, 0), newDir = mk3(0, 0, 0), bounceE = mk3(0, 0, 0), atten = mk3(1, 1, 1);
    int attId = -1;

    if (L.kind == KIND_SHADOW) {
        // ---- shadow ray came back (Test.cpp:123-132)
        TPT_STAT(ST_SHADOW);
        f4 l0 = sv.lights[L.j * 2], l1 = sv.lights[L.j * 2 + 1];
        if (id == (int)f2u(l1.w)) {
            float omega = 2 * TPT_PI * (1 - L.cosAMax);
            float dln = dot(L.dir, L.nl);
            float mx = 0.0f < dln ? dln : 0.0f; // std::max(0.0f, dln)
            L.lightE = L.lightE + (L.albedo * mk3(l1.x, l1.y, l1.z)) * (mx * omega / TPT_PI);
        }
        (void)l0;
        L.j++;
        lightLoop = true;
    } else if (id < 0) {
        TPT_STAT(ST_SKY);
        term = (fc.config & CFG_MITSUBA_COMPARE) ? mk3(0.15f, 0.21f, 0.3f) : sky(L.dir); // Test.cpp:226-231
        finish = true;
    } else {
        // ---- hit: finish HitSpheres (Maths.cpp:195-197), then Scatter (Test.cpp:83-193)
        TPT_STAT(ST_HIT);
        f4 s = sv.sph4[id];
        f3 pos = POS_GIVEN ? L.orig : L.orig + L.dir * t;
        f3 normal = (pos - mk3(s.x, s.y, s.z)) * sv.invR[id];
        f4 m0 = sv.mats[id * 3], m1 = sv.mats[id * 3 + 1];
        int type = (int)f2u(m0.w);
        f3 albedo = mk3(m0.x, m0.y, m0.z), matE = mk3(m1.x, m1.y, m1.z);
        if (L.depth >= TPT_MAX_DEPTH) { // Test.cpp:207 -> :220
            term = matE;
            finish = true;
        } else if (type == MAT_LAMBERT) { // Test.cpp:86-136
            TPT_STAT(ST_LAMBERT);
            f3 target = pos + normal + randomUnitVector(L.rng);
            L.sdir = normalize(target - pos);
            L.albedo = albedo;
            L.nl = dot(normal, L.dir) < 0 ? normal : -normal; // Test.cpp:129 (uses r_in.dir)
            L.lightE = mk3(0, 0, 0);
            L.matE = L.doMatE ? matE : mk3(0, 0, 0); // Test.cpp:210
            L.hitId = id;
            L.j = (fc.config & CFG_LIGHT_SAMPLING) ? 0 : sv.nLights; // Test.cpp:95: the light loop is compiled out without it
            L.orig = pos;
            lightLoop = true;
        } else if (type == MAT_METAL) { // Test.cpp:137-150
            TPT_STAT(ST_METAL);
            f3 refl = reflect(L.dir, normal);
            const float roughness = (fc.config & CFG_MITSUBA_COMPARE) ? 0.0f : m1.w; // Test.cpp:143-145 (the samples are drawn either way)
            newDir = normalize(refl + roughness * randomInUnitSphere(L.rng));
            if (dot(newDir, normal) > 0) {
                bounce = true;
                atten = albedo;
                attId = id;
                bounceE = L.doMatE ? matE : mk3(0, 0, 0);
                L.orig = pos;
            } else {
                term = matE; // Test.cpp:220 (emission NOT zeroed on scatter failure)
                finish = true;
            }
        } else if (type == MAT_DIELECTRIC) { // Test.cpp:151-186
            TPT_STAT(ST_DIELECTRIC);
            float ri = sv.mats[id * 3 + 2].x;
            f3 rdir = L.dir;
            f3 refl = reflect(rdir, normal);
            f3 outwardN, refr = mk3(0, 0, 0);
            float nint, cosine, reflProb;
            float dn = dot(rdir, normal);
            if (dn > 0) {
                outwardN = -normal;
                nint = ri;
                cosine = ri * dn;
            } else {
                outwardN = normal;
                nint = 1.0f / ri;
                cosine = -dn;
            }
            if (refract(rdir, outwardN, nint, refr))
                reflProb = schlick(cosine, ri);
            else
                reflProb = 1;
            if (rnd01(L.rng) < reflProb)
                newDir = normalize(refl);
            else
                newDir = normalize(refr);
            bounce = true;
            atten = mk3(1, 1, 1);
            attId = -1;
            bounceE = L.doMatE ? matE : mk3(0, 0, 0);
            L.orig = pos;
        } else { // Test.cpp:187-191
            term = matE;
            finish = true;
        }
        L.hitType = type;
    }

    // ---- Lambert light sampling loop (Test.cpp:96-133), one emissive sphere per step
    if (lightLoop) {
        while (L.j < sv.nLights && (int)f2u(sv.lights[L.j * 2 + 1].w) == L.hitId) L.j++; // skip self (:100)
        if (L.j < sv.nLights) {
            TPT_STAT(ST_LIGHTGEN);
            f4 l0 = sv.lights[L.j * 2];
            f3 sc = mk3(l0.x, l0.y, l0.z);
            f3 sw = normalize(sc - L.orig);
            f3 su = normalize(cross((sw.x < 0 ? -sw.x : sw.x) > 0.01f ? mk3(0, 1, 0) : mk3(1, 0, 0), sw));
            f3 sv_ = cross(sw, su);
            L.cosAMax = tsqrt(1.0f - l0.w * l0.w / sqLength(L.orig - sc));
            float eps1 = rnd01(L.rng), eps2 = rnd01(L.rng);
            float cosA = 1.0f - eps1 + eps1 * L.cosAMax;
            float sinA = tsqrt(1.0f - cosA * cosA);
            float phi = 2 * TPT_PI * eps2;
            float sn, cs;
            tsincosf(phi, sn, cs);
            L.dir = su * (cs * sinA) + sv_ * (sn * sinA) + sw * cosA;
            L.kind = KIND_SHADOW;
        } else {
            bounce = true;
            newDir = L.sdir;
            atten = L.albedo;
            attId = L.hitId;
            bounceE = L.matE;
            L.hitType = MAT_LAMBERT;
        }
    }

    // ---- Scatter succeeded: Test.cpp:210-216
    if (bounce) {
        TPT_STAT(ST_BOUNCE);
        if (L.hitType == MAT_LAMBERT) bounceE = bounceE + L.lightE; // matE + lightE (lightE == 0 otherwise)
        L.doMatE = (L.hitType != MAT_LAMBERT) || !(fc.config & CFG_LIGHT_SAMPLING); // Test.cpp:209-214: only with light sampling
        if (FOLD == FOLD_FORWARD) {
            L.radiance = L.radiance + L.throughput * bounceE;
            L.throughput = L.throughput * atten;
        } else {
            stack.push(L.sp, bounceE, attId);
            L.sp++;
        }
        L.dir = newDir;
        L.depth++;
        L.kind = KIND_MAIN;
    }

    // ---- path ended: fold the colour, next sample or pixel done
    if (finish) {
        TPT_STAT(ST_FINISH);
        f3 c;
        if (FOLD == FOLD_FORWARD) {
            c = L.radiance + L.throughput * term;
        } else {
            c = term;
            // matE + lightE + attenuation * Trace(...), Test.cpp:216, innermost level first.
            const int sp = L.sp;
            if (stack.fastLevels <= 1) {
                // Stack (almost) entirely in global memory (path-queue kernel: level 0 in LDS, the rest global; lane-sorting
                // kernel: all global).  The records of ALL levels are requested before the first one is used: a loop that
                // loads one level per trip pays one memory latency per level, and a wave runs as many trips as its deepest
                // lane -- that loop alone was a third of all wave time.  Two groups of five keep the register cost at 20.
#pragma unroll
                for (int g5 = 1; g5 >= 0; --g5) {
                    f4 ent[5];
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        ent[k].x = ent[k].y = ent[k].z = ent[k].w = 0.0f;
                        const int lvl = g5 * 5 + k;
                        if (lvl < sp) ent[k] = lvl < stack.fastLevels ? stack.base[lvl * stack.stride] : stack.spill[(lvl - stack.fastLevels) * stack.spillStride];
                    }
#pragma unroll
                    for (int k = 4; k >= 0; --k) {
                        if (g5 * 5 + k < sp) {
                            const f4 e = ent[k];
                            int a = (int)f2u(e.w);
                            f3 at = mk3(1, 1, 1);
                            if (a >= 0) {
                                f4 m0 = sv.mats[a * 3];
                                at = mk3(m0.x, m0.y, m0.z);
                            }
                            c = mk3(e.x, e.y, e.z) + at * c;
                        }
                    }
                }
            } else {
                for (int lv = sp - 1; lv >= 0; --lv) { // LDS levels: a plain loop
                    f4 e = stack.get(lv);
                    int a = (int)f2u(e.w);
                    f3 at = mk3(1, 1, 1);
                    if (a >= 0) {
                        f4 m0 = sv.mats[a * 3];
                        at = mk3(m0.x, m0.y, m0.z);
                    }
                    c = mk3(e.x, e.y, e.z) + at * c;
                }
            }
        }
        L.col = L.col + c;
        L.sample++;
        if (L.sample < fc.spp)
            L.needCamera = true;
        else
            return true;
    }
    return false;
}

// One step: intersect one ray, advance the path (camera ray if a new sample starts, HitWorld, lanePost).
template <int HS, int FOLD>
TPT_HD bool laneStep(Lane& L, const SceneView& sv, const FrameConsts& fc, const BounceStack& stack)
{
    TPT_STAT(ST_STEP);
    if (L.needCamera) laneCamera<FOLD>(L, fc);
    // ---- HitWorld (Test.cpp:76; counted at Test.cpp:122 and :199)
    float t;
    const int id = hitSpheres<HS>(sv, L.orig, L.dir, TPT_MIN_T, TPT_MAX_T, t);
    L.rays++;
    return lanePost<FOLD>(L, id, t, sv, fc, stack);
}

// ---------------------------------------------------------------- class code of the path-queue kernel
// The same Scatter / Trace logic as lanePost, cut into one straight-line function per material class: in the path-queue
// kernel a whole batch needs the SAME code (the class is wave-uniform), so nothing of the generic state machine's flag
// merging is left -- each function reads what its class needs and returns the bounce.  (lanePost stays the code of the
// lane-refill fallback kernel; tests/lane_emu.cpp runs both against the oracle.)
// Bounce stack of a path in the queue kernel: level 0 in the path record (LDS), levels 1..9 in global memory.
// (on the device level 0 is addressed as LDS explicitly: a store through a pointer that is LDS for level 0 and global memory
//  otherwise compiles to FLAT instructions, which take the vector-memory path even when they land in LDS)
#if defined(__HIP_DEVICE_COMPILE__)
typedef f4 __attribute__((address_space(3))) * LdsF4Ptr;
#else
typedef f4* LdsF4Ptr;
#endif
struct QStack {
    LdsF4Ptr l0;
    f4* spill;  // level k >= 1 at spill[(k - 1) * stride]
    int stride;
};
TPT_HD void qStackPush(const QStack& s, int level, f3 e, int attId)
{
    f4 v;
    v.x = e.x; v.y = e.y; v.z = e.z; v.w = u2f((uint32_t)attId);
    if (level == 0)
        *s.l0 = v;
    else
        s.spill[(level - 1) * s.stride] = v;
}
// hit normal, Maths.cpp:196-197
TPT_HD f3 qNormal(const SceneView& sv, int id, f3 pos)
{
    const f4 s = sv.sph4[id];
    return (pos - mk3(s.x, s.y, s.z)) * sv.invR[id];
}
// camera ray of the next sample of pixel (x, y), Test.cpp:286-288
TPT_HD void qCamera(const FrameConsts& fc, int x, int y, uint32_t& rng, f3& o, f3& d)
{
    TPT_STAT(ST_CAMERA);
    const float u = ((float)x + rnd01(rng)) * fc.invWidth;
    const float v = ((float)y + rnd01(rng)) * fc.invHeight;
    cameraGetRay(fc.cam, u, v, rng, o, d);
}
// what a path that ends contributes before the fold: sky (Test.cpp:226-231) or the un-zeroed emission of the sphere it
// stopped on (depth cap / failed scatter / unknown material, Test.cpp:207,218-221,187-191)
TPT_HD f3 qEndTerm(const SceneView& sv, const FrameConsts& fc, f3 dir, int id)
{
    if (id < 0) {
        TPT_STAT(ST_SKY);
        return (fc.config & CFG_MITSUBA_COMPARE) ? mk3(0.15f, 0.21f, 0.3f) : sky(dir);
    }
    const f4 m1 = sv.mats[id * 3 + 1];
    return mk3(m1.x, m1.y, m1.z);
}
// matE + lightE + attenuation * Trace(...), Test.cpp:216, innermost level first.  All levels are requested before the
// first is used (one memory latency, not one per level); two groups of five keep the register cost at 20.
TPT_HD f3 qFold(const SceneView& sv, f3 term, int depth, const QStack& s)
{
    TPT_STAT(ST_FINISH);
    f3 c = term;
    const int stride = uniformHere(s.stride); // (the eight multiples of it below are made here, once per fold, not held for the whole kernel)
#pragma unroll
    for (int g5 = 1; g5 >= 0; --g5) {
        f4 ent[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            ent[k].x = ent[k].y = ent[k].z = ent[k].w = 0.0f;
            const int lvl = g5 * 5 + k;
            if (lvl < depth) ent[k] = lvl == 0 ? *s.l0 : s.spill[(lvl - 1) * stride];
        }
#pragma unroll
        for (int k = 4; k >= 0; --k) {
            if (g5 * 5 + k < depth) {
                const f4 e = ent[k];
                const int a = (int)f2u(e.w);
                f3 at = mk3(1, 1, 1);
                if (a >= 0) {
                    const f4 m0 = sv.mats[a * 3];
                    at = mk3(m0.x, m0.y, m0.z);
                }
                c = mk3(e.x, e.y, e.z) + at * c;
            }
        }
    }
    return c;
}
// Dielectric, Test.cpp:151-186: always scatters.  e = what this level adds (matE + lightE, lightE = 0), attenuation (1,1,1).
TPT_HD f3 qDielectric(const SceneView& sv, const FrameConsts& fc, f3 pos, f3 rdir, int id, bool doMatE, uint32_t& rng, f3& e)
{
    TPT_STAT(ST_DIELECTRIC);
    (void)fc;
    const f3 normal = qNormal(sv, id, pos);
    const f4 m1 = sv.mats[id * 3 + 1];
    const f4 m2 = sv.mats[id * 3 + 2]; // {ri, 1.0f / ri, r0^2}: the division of Test.cpp:168 and schlick's r0 (Maths.h:329-330) depend on the material only
    const float ri = m2.x;
    const f3 refl = reflect(rdir, normal);
    f3 outwardN, refr = mk3(0, 0, 0);
    float nint, cosine, reflProb;
    const float dn = dot(rdir, normal);
    if (dn > 0) {
        outwardN = -normal;
        nint = ri;
        cosine = ri * dn;
    } else {
        outwardN = normal;
        nint = m2.y;
        cosine = -dn;
    }
    if (refract(rdir, outwardN, nint, refr))
        reflProb = schlickR0(cosine, m2.z);
    else
        reflProb = 1;
    const f3 pick = rnd01(rng) < reflProb ? refl : refr;
    e = (doMatE ? mk3(m1.x, m1.y, m1.z) : mk3(0, 0, 0)) + mk3(0, 0, 0); // matE + lightE, Test.cpp:216
    return normalize(pick);
}
// Metal, Test.cpp:137-150: false = the scattered ray points into the surface (the path ends with the sphere's emission).
TPT_HD bool qMetal(const SceneView& sv, const FrameConsts& fc, f3 pos, f3 rdir, int id, bool doMatE, uint32_t& rng, f3& e, f3& newDir)
{
    TPT_STAT(ST_METAL);
    const f3 normal = qNormal(sv, id, pos);
    const f4 m1 = sv.mats[id * 3 + 1];
    const f3 refl = reflect(rdir, normal);
    const float roughness = (fc.config & CFG_MITSUBA_COMPARE) ? 0.0f : m1.w; // Test.cpp:143-145 (the samples are drawn either way)
    newDir = normalize(refl + roughness * randomInUnitSphere(rng));
    e = (doMatE ? mk3(m1.x, m1.y, m1.z) : mk3(0, 0, 0)) + mk3(0, 0, 0);
    return dot(newDir, normal) > 0;
}
// Lambert, Test.cpp:86-136, in three pieces so that the kernel can intersect the shadow rays in between:
//   qLambertBegin: the bounce direction (drawn first, Test.cpp:89-91) and the shading frame;
//   qLightRay:     light j's sample direction, Test.cpp:102-120 (false: j is the sphere itself, Test.cpp:100);
//   qLightShade:   Test.cpp:123-132 once the shadow ray's nearest hit is known.
struct QLambert {
    f3 sdir, nl, albedo, lightE;
    float cosAMax;
};
TPT_HD void qLambertBegin(const SceneView& sv, f3 pos, f3 rdir, int id, uint32_t& rng, QLambert& q)
{
    TPT_STAT(ST_LAMBERT);
    const f3 normal = qNormal(sv, id, pos);
    const f4 m0 = sv.mats[id * 3];
    const f3 target = pos + normal + randomUnitVector(rng);
    q.sdir = normalize(target - pos);
    q.albedo = mk3(m0.x, m0.y, m0.z);
    q.nl = dot(normal, rdir) < 0 ? normal : -normal; // Test.cpp:129 (uses r_in.dir)
    q.lightE = mk3(0, 0, 0);
    q.cosAMax = 0.0f;
}
TPT_HD f3 qLightRay(const f4 l0, f3 pos, uint32_t& rng, float& cosAMax, bool fastDiv)
{
    TPT_STAT(ST_LIGHTGEN);
    const f3 sc = mk3(l0.x, l0.y, l0.z);
    const f3 toL = sc - pos;
    const float d2 = dot(toL, toL); // = sqLength(pos - sc) bit for bit: (a - b) and (b - a) differ in sign only, the squares are equal
    const f3 sw = toL * trsqrt2(d2); // normalize(sc - pos)
    const f3 su = normalize(cross((sw.x < 0 ? -sw.x : sw.x) > 0.01f ? mk3(0, 1, 0) : mk3(1, 0, 0), sw));
    const f3 sv_ = cross(sw, su);
    const float r2 = l0.w * l0.w;
    cosAMax = tsqrt(1.0f - (fastDiv ? tdivSafeNum(r2, d2) : r2 / d2)); // (fastDiv is wave-uniform: a scene flag)
    const float eps1 = rnd01(rng), eps2 = rnd01(rng);
    const float cosA = 1.0f - eps1 + eps1 * cosAMax;
    const float sinA = tsqrt(1.0f - cosA * cosA);
    const float phi = 2 * TPT_PI * eps2;
    float sn, cs;
    tsincosf(phi, sn, cs);
    return su * (cs * sinA) + sv_ * (sn * sinA) + sw * cosA;
}
// what the Lambert level adds to the fold: matE (unless the previous bounce already sampled the lights, Test.cpp:210) + lightE
TPT_HD f3 qLambertE(const SceneView& sv, int id, bool doMatE, const QLambert& q)
{
    const f4 m1 = sv.mats[id * 3 + 1];
    return (doMatE ? mk3(m1.x, m1.y, m1.z) : mk3(0, 0, 0)) + q.lightE;
}
TPT_HD void qLightShade(const f4 l1, f3 l, QLambert& q)
{
    TPT_STAT(ST_SHADOW);
    const float omega = 2 * TPT_PI * (1 - q.cosAMax);
    const float dln = dot(l, q.nl);
    const float mx = 0.0f < dln ? dln : 0.0f; // std::max(0.0f, dln)
    q.lightE = q.lightE + (q.albedo * mk3(l1.x, l1.y, l1.z)) * tdivByPi(mx * omega);
}

// Pixel complete: the frame's colour of this pixel, averaged over the samples (Test.cpp:291).
TPT_HD f3 lanePixelColour(const Lane& L, const FrameConsts& fc) { return L.col * fc.invSpp; }

// Progressive accumulation (Test.cpp:293-295): blend with the previous frame's RGB.  The reference reads
// `prev` even when lerpFac == 0 (NaN/Inf in the buffer propagate), so does this.  Alpha is never touched
// (Maths.h:38 store() writes 3 floats).
TPT_HD f3 blendPixel(f3 prev, f3 col, float lerpFac) { return prev * lerpFac + col * (1 - lerpFac); }

} // namespace tpt
