// tpt_scene.h -- host-side scene state: the reference's Sphere / Material / Camera layouts, the
// built-in 46-sphere scene, UpdateTest's per-frame preparation and the packing into the arrays the
// kernels read (SceneView, tpt_trace.h).
//
// Replaces Cpp/Source/Test.cpp:13-69 (scene tables, emissive list, camera), UpdateTest :302-342,
// Maths.h Sphere :354-364, SpheresSoA :368-404, Camera ctor :418-435, GetSceneDesc :369-384.
// Host-only plain C++ (no HIP types) so the lane-logic test in tests/ can reuse it.
#pragma once
#include "tpt_trace.h"
#include <math.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace tpt {

// ---- layout contract of the reference (sizes asserted by its GPU hosts, TestWin.cpp:132-134)
struct SpherePOD { // Maths.h:354-364, 20 B
    float cx, cy, cz, radius, invRadius;
};
struct MaterialPOD { // Test.cpp:36-44, 36 B
    int type;
    float albedo[3], emissive[3], roughness, ri;
};
static_assert(sizeof(SpherePOD) == 20, "Sphere layout (TestWin.cpp:132)");
static_assert(sizeof(MaterialPOD) == 36, "Material layout (TestWin.cpp:133)");
static_assert(sizeof(CameraPOD) == 88, "Camera layout (TestWin.cpp:134)");

struct CameraSetup { // arguments of the Camera ctor, Maths.h:418
    float lookFrom[3], lookAt[3], vup[3];
    float vfov, aperture, focusDist;
};

inline CameraSetup defaultCameraSetup() // Test.cpp:309-319
{
    CameraSetup c;
    c.lookFrom[0] = 0; c.lookFrom[1] = 2; c.lookFrom[2] = 3;
    c.lookAt[0] = 0; c.lookAt[1] = 0; c.lookAt[2] = 0;
    c.vup[0] = 0; c.vup[1] = 1; c.vup[2] = 0;
    c.vfov = 60;
    float aperture = 0.1f;
    aperture *= 0.2f; // DO_BIG_SCENE, Test.cpp:317-319
    c.aperture = aperture;
    c.focusDist = 3;
    return c;
}

inline void st3(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

inline CameraPOD makeCamera(const CameraSetup& s, float aspect) // Maths.h:418-435 (tanf from the host libm, as the reference)
{
    CameraPOD cam;
    cam.lensRadius = s.aperture / 2;
    float theta = s.vfov * TPT_PI / 180;
    float halfHeight = tanf(theta / 2);
    float halfWidth = aspect * halfHeight;
    f3 org = ld3(s.lookFrom);
    f3 w = normalize(ld3(s.lookFrom) - ld3(s.lookAt));
    f3 u = normalize(cross(ld3(s.vup), w));
    f3 v = cross(w, u);
    st3(cam.origin, org);
    st3(cam.ww, w);
    st3(cam.uu, u);
    st3(cam.vv, v);
    st3(cam.lowerLeftCorner, org - (halfWidth * s.focusDist) * u - (halfHeight * s.focusDist) * v - s.focusDist * w);
    st3(cam.horizontal, (2 * halfWidth * s.focusDist) * u);
    st3(cam.vertical, (2 * halfHeight * s.focusDist) * v);
    return cam;
}

// ---- the built-in scene: same data as Test.cpp:13-31 (spheres) and :46-64 (materials)
inline void defaultScene(std::vector<SpherePOD>& S, std::vector<MaterialPOD>& M)
{
    S.clear();
    M.clear();
    auto sphere = [&](float x, float y, float z, float r) {
        SpherePOD s = {x, y, z, r, 0.0f};
        S.push_back(s);
    };
    auto mat = [&](int type, float r, float g, float b, float er, float eg, float eb, float rough, float ri) {
        MaterialPOD m = {type, {r, g, b}, {er, eg, eb}, rough, ri};
        M.push_back(m);
    };
    const float grey[9] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f, 0.9f};
    const float hue[9][3] = {{0.8f, 0.1f, 0.1f}, {0.8f, 0.5f, 0.1f}, {0.8f, 0.8f, 0.1f}, {0.4f, 0.8f, 0.1f}, {0.1f, 0.8f, 0.1f},
                             {0.1f, 0.8f, 0.5f}, {0.1f, 0.8f, 0.8f}, {0.1f, 0.1f, 0.8f}, {0.5f, 0.1f, 0.8f}};
    sphere(0, -100.5f, -1, 100);
    mat(MAT_LAMBERT, 0.8f, 0.8f, 0.8f, 0, 0, 0, 0, 0);
    const float featAlb[6][3] = {{0.8f, 0.4f, 0.4f}, {0.4f, 0.8f, 0.4f}, {0.4f, 0.4f, 0.8f}, {0.4f, 0.8f, 0.4f}, {0.4f, 0.8f, 0.4f}, {0.4f, 0.8f, 0.4f}};
    const int featType[6] = {MAT_LAMBERT, MAT_LAMBERT, MAT_METAL, MAT_METAL, MAT_METAL, MAT_METAL};
    const float featRough[6] = {0, 0, 0, 0, 0.2f, 0.6f};
    for (int k = 0; k < 6; ++k) {
        sphere((float)(2 - 2 * (k % 3)), 0, k < 3 ? -1.0f : 1.0f, 0.5f);
        mat(featType[k], featAlb[k][0], featAlb[k][1], featAlb[k][2], 0, 0, 0, featRough[k], 0);
    }
    sphere(0.5f, 1, 0.5f, 0.5f);
    mat(MAT_DIELECTRIC, 0.4f, 0.4f, 0.4f, 0, 0, 0, 0, 1.5f);
    sphere(-1.5f, 1.5f, 0.f, 0.3f);
    mat(MAT_LAMBERT, 0.8f, 0.6f, 0.2f, 30, 25, 15, 0, 0);
    for (int row = 0; row < 4; ++row)
        for (int k = 0; k < 9; ++k) {
            sphere((float)(4 - k), 0, (float)(-3 - row), 0.5f);
            if (row == 0) mat(MAT_LAMBERT, grey[k], grey[k], grey[k], 0, 0, 0, 0, 0);
            if (row == 1) mat(MAT_METAL, grey[k], grey[k], grey[k], 0, 0, 0, 0, 0);
            if (row == 2) mat(MAT_METAL, hue[k][0], hue[k][1], hue[k][2], 0, 0, 0, 0, 0);
            if (row == 3) mat(k < 8 ? MAT_LAMBERT : MAT_METAL, hue[k][0], hue[k][1], hue[k][2], 0, 0, 0, 0, 0);
        }
    sphere(1.5f, 1.5f, -2, 0.3f);
    mat(MAT_LAMBERT, 0.1f, 0.2f, 0.5f, 3, 10, 20, 0, 0);
}

// ---- packed arrays the kernels read
struct PackedScene {
    std::vector<float> pairs; // [nPairs][8]: {cx0,cx1, cy0,cy1, cz0,cz1, -r0^2 (1+2^-16), -r1^2 (1+2^-16)}
    std::vector<f4> sph4;     // [nPairs*2]
    std::vector<float> invR;  // [nPairs*2]
    std::vector<f4> mats;     // [n][3]
    std::vector<f4> lights;   // [nLights][2]
    std::vector<int> emissive; // ids, as GetSceneDesc exports them (Test.cpp:382)
    int nSpheres = 0, nPairs = 0, nLights = 0;
    // grouped representation for large scenes (SceneView, hitSpheresGrouped); nGroups == 0: not grouped
    std::vector<float> gpairs; // [nGroupPairs][8] bounding spheres of the groups, pair-record format
    std::vector<f4> gsph;      // [nGroups * TPT_GROUP] members {centre, r^2}, padding r^2 = -inf
    std::vector<int> gid;      // original sphere index of every member slot, padding -1
    std::vector<f4> bsph;      // the big spheres {centre, r^2} ...
    std::vector<int> bid;      // ... and their original indices (ascending)
    int nGroups = 0, nGroupPairs = 0, nBig = 0;
    // second level over the groups: super-group k bounds groups [TPT_SUPER k, TPT_SUPER (k + 1)) -- consecutive leaves of the median
    // splits, i.e. one subtree; same pair-record format and slack as gpairs (a super-group too loose for the filter's slack
    // carries R^2 = +inf: always a candidate)
    std::vector<float> spairs; // [nSuperPairs][8]
    int nSupers = 0, nSuperPairs = 0;
    std::vector<uint32_t> amatH; // [2][2][64][4] A operands of the matrix-core filter (phase1MatrixH); empty: not available
    int mxR1 = -1;
    std::vector<uint32_t> gmatH; // grouped scenes: [tiles of 64 groups][2][2][64][4] A operands for the group bounds; empty: not available
    int gmxTiles = 0;
    int flags = 0;               // SCENE_* bits (tpt_trace.h)
};

// a_k of one sphere for the matrix-core filter (tpt_trace.h, phase1MatrixH): binary64, rounded once; a_9 carries the
// sphere side of the slack, m_s = 2^-13 |c|^2 + 2^-14 r^2 + 2^-20
// (slackShift 1: the group-bound table, twice the slack -- see buildGroupMatrixTable)
inline void matrixSphereSide(float fcx, float fcy, float fcz, float fr2, float* a, int slackShift = 0)
{
    const double cx = fcx, cy = fcy, cz = fcz, r2 = fr2;
    const double cc = cx * cx + cy * cy + cz * cz, k = slackShift ? 2.0 : 1.0;
    const double v[TPT_MX_K] = {cx * cx, cy * cy, cz * cz, 2 * cx * cy, 2 * cx * cz, 2 * cy * cz, 2 * cx, 2 * cy, 2 * cz,
                                (r2 - cc) + k * cc / 8192.0 + k * r2 / 16384.0 + 1.0 / 1048576.0, 1.0};
    for (int k = 0; k < TPT_MX_K; ++k) a[k] = (float)v[k];
}

// Sphere side of the matrix-core filter: per sphere the 32 K-slot values (tpt_trace.h: slot table at phase1MatrixH), two
// binary16 per dword, laid out as the lanes of v_mfma_f32_32x32x16_f16 read their A operand:
// amatH[((mt * 2 + j) * 64 + lane) * 4 + w] holds slots 16 j + 8 (lane / 32) + 2 w, + 1 of the sphere in row lane % 32 of
// sphere tile mt.  Scenes of up to 64 spheres whose a_k all fit binary16 (|a_k| < 60000); otherwise no table (mxR1 = -1)
// and the packed VALU filter runs.
// one 64-entry tile pair (4 KB = 1024 dwords at T): padding rows first, then entry q = 0 .. n - 1 from a[q][0..10]
inline void matrixPut(uint32_t* T, int mt, int row, int slot, uint32_t h)
{
    const int j = slot / 16, within = slot % 16, lane = row + 32 * (within / 8), w = (within % 8) / 2;
    uint32_t& d = T[(size_t)((mt * 2 + j) * 64 + lane) * 4 + w];
    d = (slot & 1) ? ((d & 0x0000ffffu) | (h << 16)) : ((d & 0xffff0000u) | h);
}
inline void matrixPadRows(uint32_t* T)
{
    // padding rows: a9_hi = -inf (slot 29, whose B value is 1), everything else 0: the sum is -inf, sign set, never a candidate
    for (int mt = 0; mt < 2; ++mt)
        for (int row = 0; row < 32; ++row) matrixPut(T, mt, row, 29, 0xfc00u);
}
inline bool matrixPutEntry(uint32_t* T, int q, int R1, const float* a) // false: binary16 cannot carry this entry
{
    int mt, row;
    matrixSlot(q, R1, mt, row);
    uint32_t hi[10], lo[10];
    for (int k = 0; k < 10; ++k) {
        if (!(fabsf(a[k]) < 60000.0f)) return false; // (also NaN)
        hi[k] = f16rtz(a[k]);
        lo[k] = f16rtz(a[k] - f16val(hi[k]));
    }
    const uint32_t one = 0x3c00u;
    for (int t = 0; t < TPT_MXH_TERMS; ++t) {
        matrixPut(T, mt, row, 2 * t, hi[t]);
        matrixPut(T, mt, row, 2 * t + 1, hi[t]);
    }
    for (int u = 0; u < 4; ++u) {
        matrixPut(T, mt, row, 18 + 2 * u, lo[2 * u]);
        matrixPut(T, mt, row, 19 + 2 * u, lo[2 * u + 1]);
    }
    matrixPut(T, mt, row, 26, lo[8]);
    matrixPut(T, mt, row, 27, one);
    matrixPut(T, mt, row, 28, one);
    matrixPut(T, mt, row, 29, hi[9]);
    matrixPut(T, mt, row, 30, lo[9]);
    matrixPut(T, mt, row, 31, 0u);
    return true;
}
inline void buildMatrixTable(const std::vector<SpherePOD>& S, PackedScene& P)
{
    P.amatH.clear();
    P.mxR1 = -1;
    const int n = (int)S.size();
    if (n < 1 || n > 64) return;
    int R1 = 0;
    if (n > 32) R1 = ((n - 32 + 1) / 2 + 3) / 4 * 4; // rows per half of tile 1, multiple of 4: capacity 2 (16 + R1) >= n
    std::vector<uint32_t> T(TPT_MXH_TABLE_DWORDS, 0u);
    matrixPadRows(T.data());
    for (int p = 0; p < n; ++p) {
        float a[TPT_MX_K];
        matrixSphereSide(S[p].cx, S[p].cy, S[p].cz, S[p].radius * S[p].radius, a); // r^2 as the exact test sees it (Test.cpp:329)
        if (!matrixPutEntry(T.data(), p, R1, a)) return; // binary16 cannot carry this sphere
    }
    P.amatH.swap(T);
    P.mxR1 = R1;
}

// The same filter over the BOUNDING SPHERES of a grouped scene (hitSpheresGroupedDeal): one 4-KB tile pair per 64 groups, full
// tiles (R1 = 16: group q of a tile at mask bit 63 - q, the order phase1Chunk delivers).  A group bound has to pass whenever the
// reference accepts one of its members, which costs slack on top of the filter's own error (tpt_trace.h, comment at
// TPT_PG_K): the member's line distance exceeds the bound by up to 26 u (S + R^2)(1 + rho), rho = max |c - C| / r over the
// members, S = |C - o|^2 <= 2 (|C|^2 + |o|^2); with the matrix form's own 490 u (|C|^2 + |o|^2) + 114 u R^2 (phase1MatrixH) that is
//   [52 (1 + rho) + 490] u (|C|^2 + |o|^2) + [26 (1 + rho) + 114] u R^2  <=  2206 u (...) + 972 u R^2   for rho <= 32,
// against the doubled slack m = 4096 u (|C|^2 + |o|^2) + 2048 u R^2 + 16 u of this table (slackShift 1 on both sides): a factor
// 1.85 to spare.  Scenes with a looser group (rho > 32), a bound binary16 cannot carry or more than 65536 groups get no table
// and the packed VALU filter.
inline void buildGroupMatrixTable(PackedScene& P, const std::vector<float>& C3, const std::vector<double>& R, double rhoMax)
{
    P.gmatH.clear();
    P.gmxTiles = 0;
    const int nGroups = (int)R.size();
    if (nGroups < 1 || nGroups > 65536 || !(rhoMax <= 32.0)) return;
    const int tiles = (nGroups + 63) / 64;
    std::vector<uint32_t> T((size_t)tiles * TPT_MXH_TABLE_DWORDS, 0u);
    for (int t = 0; t < tiles; ++t) matrixPadRows(T.data() + (size_t)t * TPT_MXH_TABLE_DWORDS);
    for (int g = 0; g < nGroups; ++g) {
        float a[TPT_MX_K];
        const float r2 = (float)(R[g] * R[g] * (1.0 + 1.0e-6)); // (rounded up: the bound must not shrink)
        matrixSphereSide(C3[(size_t)g * 3], C3[(size_t)g * 3 + 1], C3[(size_t)g * 3 + 2], r2, a, 1);
        if (!matrixPutEntry(T.data() + (size_t)(g / 64) * TPT_MXH_TABLE_DWORDS, g % 64, 16, a)) return;
    }
    P.gmatH.swap(T);
    P.gmxTiles = tiles;
}

// Large scenes: compact groups of <= TPT_GROUP small spheres (median splits) with bounding spheres, big spheres kept apart.
// Leaves P ungrouped (nGroups = 0) when the scene is small or a bound would be too loose for the filter's slack
// (tpt_trace.h, hitSpheresGrouped).
inline void buildGroups(const std::vector<SpherePOD>& S, PackedScene& P)
{
    P.gpairs.clear(); P.gsph.clear(); P.gid.clear(); P.bsph.clear(); P.bid.clear(); P.gmatH.clear(); P.spairs.clear();
    P.nGroups = P.nGroupPairs = P.nBig = 0; P.gmxTiles = 0; P.nSupers = P.nSuperPairs = 0;
    const int n = (int)S.size();
    if (n < TPT_GROUP_MIN_SPHERES) return;
    std::vector<float> radii(n);
    for (int i = 0; i < n; ++i) {
        radii[i] = fabsf(S[i].radius);
        if (!(radii[i] < 1e30f) || !(fabsf(S[i].cx) < 1e30f) || !(fabsf(S[i].cy) < 1e30f) || !(fabsf(S[i].cz) < 1e30f)) return; // inf / NaN: flat
    }
    std::vector<float> sorted = radii;
    std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
    const float median = sorted[n / 2];
    std::vector<int> small, big;
    for (int i = 0; i < n; ++i) (radii[i] > 3.0f * median ? big : small).push_back(i);
    if ((int)big.size() > 32 || (int)small.size() < TPT_GROUP) return;
    // kd-style median splits (widest axis of the centres' box) down to <= TPT_GROUP members: compact groups whatever
    // the distribution.  `order` ends up holding the groups back to back, `cuts` their boundaries.
    std::vector<int> order = small;
    std::vector<std::pair<int, int>> work, leaves; // [first, last) ranges of `order`
    work.push_back(std::make_pair(0, (int)order.size()));
    while (!work.empty()) {
        const std::pair<int, int> rg = work.back();
        work.pop_back();
        const int cntR = rg.second - rg.first;
        if (cntR <= TPT_GROUP) {
            leaves.push_back(rg);
            continue;
        }
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int k = rg.first; k < rg.second; ++k) {
            const double c[3] = {S[order[k]].cx, S[order[k]].cy, S[order[k]].cz};
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c[a]); hi[a] = std::max(hi[a], c[a]); }
        }
        int axis = 0;
        for (int a = 1; a < 3; ++a) if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
        // left part: a multiple of TPT_GROUP nearest to half, so the leaves come out full
        int left = ((cntR / 2 + TPT_GROUP / 2) / TPT_GROUP) * TPT_GROUP;
        if (left <= 0 || left >= cntR) left = cntR / 2;
        auto key = [&](int i) { return axis == 0 ? S[i].cx : axis == 1 ? S[i].cy : S[i].cz; };
        std::nth_element(order.begin() + rg.first, order.begin() + rg.first + left, order.begin() + rg.second,
                         [&](int x, int y) { return key(x) < key(y) || (key(x) == key(y) && x < y); });
        work.push_back(std::make_pair(rg.first, rg.first + left));
        work.push_back(std::make_pair(rg.first + left, rg.second));
    }
    std::sort(leaves.begin(), leaves.end());
    // bounds; a group whose bound is too loose for the filter's slack (rho > 64: tiny spheres far apart) is dissolved
    // into the flat list of big spheres
    const float negInf = u2f(0xff800000u), posInf = u2f(0x7f800000u);
    struct GroupRec { float C[3]; double R; std::vector<int> mem; };
    std::vector<GroupRec> groups;
    double rhoMax = 0;
    for (size_t l = 0; l < leaves.size(); ++l) {
        GroupRec G;
        for (int k = leaves[l].first; k < leaves[l].second; ++k) G.mem.push_back(order[k]);
        std::sort(G.mem.begin(), G.mem.end());
        double c[3] = {0, 0, 0};
        for (int i : G.mem) { c[0] += S[i].cx; c[1] += S[i].cy; c[2] += S[i].cz; }
        for (int a = 0; a < 3; ++a) G.C[a] = (float)(c[a] / (double)G.mem.size()); // the centre the kernel will see
        double R = 0, rho = 0;
        for (int i : G.mem) {
            const double dxx = (double)S[i].cx - G.C[0], dyy = (double)S[i].cy - G.C[1], dzz = (double)S[i].cz - G.C[2];
            const double a = sqrt(dxx * dxx + dyy * dyy + dzz * dzz), r = radii[i];
            R = std::max(R, a + r);
            rho = std::max(rho, r > 0 ? a / r : 1e300);
        }
        if (!(rho <= 64.0)) {
            for (int i : G.mem) big.push_back(i);
            continue;
        }
        G.R = R * 1.00001;
        rhoMax = std::max(rhoMax, rho);
        groups.push_back(G);
    }
    if ((int)big.size() > 64 || groups.empty()) return;
    std::sort(big.begin(), big.end());
    const int nGroups = (int)groups.size();
    const int nGroupPairs = (nGroups + 1) / 2;
    // (records for whole super-groups: the second level of the bounds filter reads TPT_SUPER / 2 pair records per super-group)
    const int nGroupPairsPadded = ((nGroups + TPT_SUPER - 1) / TPT_SUPER) * (TPT_SUPER / 2);
    std::vector<float> gpairs((size_t)nGroupPairsPadded * 8, 0.0f);
    std::vector<f4> gsph((size_t)nGroups * TPT_GROUP);
    std::vector<int> gid((size_t)nGroups * TPT_GROUP, -1);
    for (size_t k = 0; k < gsph.size(); ++k) { f4 v = {0, 0, 0, negInf}; gsph[k] = v; }
    for (int g = 0; g < nGroupPairsPadded * 2; ++g) {
        float* rec = &gpairs[(size_t)(g / 2) * 8];
        if (g >= nGroups) { // padding group: never a candidate
            rec[6 + (g & 1)] = posInf;
            continue;
        }
        const GroupRec& G = groups[g];
        for (size_t k = 0; k < G.mem.size(); ++k) {
            const int i = G.mem[k];
            f4 v = {S[i].cx, S[i].cy, S[i].cz, S[i].radius * S[i].radius}; // r^2 exactly as packScene / Test.cpp:329
            gsph[(size_t)g * TPT_GROUP + k] = v;
            gid[(size_t)g * TPT_GROUP + k] = i;
        }
        rec[0 + (g & 1)] = G.C[0];
        rec[2 + (g & 1)] = G.C[1];
        rec[4 + (g & 1)] = G.C[2];
        rec[6 + (g & 1)] = (float)(-(G.R * G.R) * (1.0 + 1.0 / 4096.0));
    }
    P.gpairs.swap(gpairs); P.gsph.swap(gsph); P.gid.swap(gid);
    for (int i : big) {
        f4 v = {S[i].cx, S[i].cy, S[i].cz, S[i].radius * S[i].radius};
        P.bsph.push_back(v);
        P.bid.push_back(i);
    }
    P.nGroups = nGroups; P.nGroupPairs = nGroupPairs; P.nBig = (int)big.size();
    {
        // super-groups: TPT_SUPER consecutive groups each (the leaves are in the order of the median splits, so they are one
        // subtree -- spatially compact).  The bound is a sphere around the mean of the MEMBER SPHERES' centres that holds every
        // member sphere; the filter's slack argument (tpt_trace.h, "Group bounds are looser ...") is the groups' own with a = the
        // member's distance from the super-group's centre, so it needs rho = max a / r <= 64 here as well -- a super-group that
        // fails it is always a candidate (-R^2 = -inf).
        const int nSupers = (nGroups + TPT_SUPER - 1) / TPT_SUPER, nSuperPairs = (nSupers + 1) / 2;
        std::vector<float> spairs((size_t)nSuperPairs * 8, 0.0f);
        for (int k = 0; k < nSuperPairs * 2; ++k) {
            float* rec = &spairs[(size_t)(k / 2) * 8];
            if (k >= nSupers) { rec[6 + (k & 1)] = posInf; continue; }
            double c[3] = {0, 0, 0};
            size_t cnt = 0;
            for (int g = k * TPT_SUPER; g < (k + 1) * TPT_SUPER && g < nGroups; ++g)
                for (int i : groups[g].mem) { c[0] += S[i].cx; c[1] += S[i].cy; c[2] += S[i].cz; ++cnt; }
            float C[3];
            for (int a = 0; a < 3; ++a) C[a] = (float)(c[a] / (double)cnt);
            double R = 0, rho = 0;
            for (int g = k * TPT_SUPER; g < (k + 1) * TPT_SUPER && g < nGroups; ++g)
                for (int i : groups[g].mem) {
                    const double dxx = (double)S[i].cx - C[0], dyy = (double)S[i].cy - C[1], dzz = (double)S[i].cz - C[2];
                    const double a = sqrt(dxx * dxx + dyy * dyy + dzz * dzz), r = radii[i];
                    R = std::max(R, a + r);
                    rho = std::max(rho, r > 0 ? a / r : 1e300);
                }
            R *= 1.00001;
            rec[0 + (k & 1)] = C[0];
            rec[2 + (k & 1)] = C[1];
            rec[4 + (k & 1)] = C[2];
            rec[6 + (k & 1)] = rho <= 64.0 ? (float)(-(R * R) * (1.0 + 1.0 / 4096.0)) : negInf;
        }
        P.spairs.swap(spairs);
        P.nSupers = nSupers; P.nSuperPairs = nSuperPairs;
    }
    {
        std::vector<float> C3((size_t)nGroups * 3);
        std::vector<double> Rg((size_t)nGroups);
        for (int g = 0; g < nGroups; ++g) {
            for (int a = 0; a < 3; ++a) C3[(size_t)g * 3 + a] = groups[g].C[a];
            Rg[g] = groups[g].R;
        }
        buildGroupMatrixTable(P, C3, Rg, rhoMax);
    }
}

// UpdateTest's scene half (Test.cpp:321-339): derived data, SoA, emissive list -- in kernel layout.
inline void packScene(std::vector<SpherePOD>& S, const std::vector<MaterialPOD>& M, PackedScene& P)
{
    const int n = (int)S.size();
    const int nPairs = (n + 1) / 2, nPad = nPairs * 2;
    P.nSpheres = n;
    P.nPairs = nPairs;
    P.pairs.assign((size_t)nPairs * 8, 0.0f);
    f4 zero = {0, 0, 0, 0};
    P.sph4.assign((size_t)nPad, zero);
    P.invR.assign((size_t)nPad, 0.0f);
    P.mats.assign((size_t)n * 3, zero);
    P.lights.clear();
    P.emissive.clear();
    P.flags = SCENE_LIGHT_R2_DIV_SAFE;
    const float negInf = u2f(0xff800000u);
    for (int i = 0; i < nPad; ++i) {
        float cx = 0, cy = 0, cz = 0, sq = negInf; // padding sphere: filter value = -inf, never a candidate
        if (i < n) {
            S[i].invRadius = 1.0f / S[i].radius; // Sphere::UpdateDerivedData, Maths.h:359
            cx = S[i].cx; cy = S[i].cy; cz = S[i].cz;
            sq = S[i].radius * S[i].radius; // Test.cpp:329
            P.invR[i] = S[i].invRadius;
            f4 v = {cx, cy, cz, sq};
            P.sph4[i] = v;
            const MaterialPOD& m = M[i];
            f4 m0 = {m.albedo[0], m.albedo[1], m.albedo[2], u2f((uint32_t)m.type)};
            f4 m1 = {m.emissive[0], m.emissive[1], m.emissive[2], m.roughness};
            float r0 = (1 - m.ri) / (1 + m.ri); // schlick, Maths.h:329-330; 1.0f / ri: Test.cpp:168 -- the reference's own operations, done once
            r0 = r0 * r0;
            f4 m2 = {m.ri, 1.0f / m.ri, r0, 0};
            P.mats[(size_t)i * 3] = m0;
            P.mats[(size_t)i * 3 + 1] = m1;
            P.mats[(size_t)i * 3 + 2] = m2;
            if (m.emissive[0] > 0 || m.emissive[1] > 0 || m.emissive[2] > 0) { // Test.cpp:334
                if (!tdivInRange(S[i].radius * S[i].radius)) P.flags &= ~SCENE_LIGHT_R2_DIV_SAFE;
                f4 l0 = {cx, cy, cz, S[i].radius};
                f4 l1 = {m.emissive[0], m.emissive[1], m.emissive[2], u2f((uint32_t)i)};
                P.lights.push_back(l0);
                P.lights.push_back(l1);
                P.emissive.push_back(i);
            }
        } else {
            f4 v = {cx, cy, cz, sq};
            P.sph4[i] = v;
        }
        float* rec = &P.pairs[(size_t)(i / 2) * 8];
        rec[0 + (i & 1)] = cx;
        rec[2 + (i & 1)] = cy;
        rec[4 + (i & 1)] = cz;
        // phase 1's conservative filter (tpt_trace.h, phase1Pair) wants -r^2 (1 + 2^-16); padding: +inf
        rec[6 + (i & 1)] = (float)(-(double)sq * (1.0 + 1.0 / 65536.0));
    }
    P.nLights = (int)P.emissive.size();
    buildGroups(S, P);
    buildMatrixTable(S, P);
}

inline SceneView viewOf(const PackedScene& P)
{
    SceneView sv;
    sv.pairs = P.pairs.data();
    sv.sph4 = P.sph4.data();
    sv.invR = P.invR.data();
    sv.mats = P.mats.data();
    sv.lights = P.lights.data();
    sv.nSpheres = P.nSpheres;
    sv.nPairs = P.nPairs;
    sv.nLights = P.nLights;
    sv.gpairs = P.gpairs.data();
    sv.gsph = P.gsph.data();
    sv.gid = P.gid.data();
    sv.bsph = P.bsph.data();
    sv.bid = P.bid.data();
    sv.nGroups = P.nGroups;
    sv.nGroupPairs = P.nGroupPairs;
    sv.nBig = P.nBig;
    sv.spairs = P.spairs.data();
    sv.nSuperPairs = P.nSuperPairs;
    sv.amatH = P.amatH.empty() ? nullptr : P.amatH.data();
    sv.mxR1 = P.mxR1;
    sv.gmatH = P.gmatH.empty() ? nullptr : P.gmatH.data();
    sv.gmxTiles = P.gmxTiles;
    sv.flags = P.flags;
    return sv;
}

inline FrameConsts makeFrameConsts(const CameraPOD& cam, int w, int h, int spp, int frame, unsigned flags, int seedMode,
                                   int config = CFG_LIGHT_SAMPLING, float animateSmoothing = 0.9f)
{
    FrameConsts fc;
    fc.cam = cam;
    fc.width = w;
    fc.height = h;
    fc.spp = spp;
    fc.frame = frame;
    fc.invWidth = 1.0f / w;   // Test.cpp:270
    fc.invHeight = 1.0f / h;  // Test.cpp:271
    float lerpFac = float(frame) / float(frame + 1); // Test.cpp:272
    if (flags & 1u) lerpFac *= animateSmoothing;     // kFlagAnimate * DO_ANIMATE_SMOOTHING (Config.h:23: 0.9f), Test.cpp:273-274
    if (!(flags & 2u)) lerpFac = 0;                  // !kFlagProgressive, Test.cpp:275-276
    fc.lerpFac = lerpFac;
    fc.invSpp = 1.0f / float(spp); // Test.cpp:291
    fc.seedMode = seedMode;
    fc.config = config;
    return fc;
}

} // namespace tpt
