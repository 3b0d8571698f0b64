// tests/adversarial_filter.cpp -- TEST HARNESS ONLY.  Randomised adversarial search: does the conservative phase-1 filter
// (memberFilter = the per-lane form of phase1Pair), the group-bound filter of hitSpheresGrouped or the matrix-core filter
// (phase1MatrixH, worst case of its error model) ever reject a sphere
// the reference's discriminant accepts?  Near-tangent rays, centre distances 1e-2..1e4, radii down to 1e-3 of that,
// offsets 1e-9..1e-2 radii on both sides, group bounds with |c - C| / r up to 64.  Exit code 1 on any miss.
#include <stdio.h>
#include <stdlib.h>
#include <omp.h>
#include "tpt_scene.h"
using namespace tpt;
static inline double urand(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) * (1.0 / 9007199254740992.0); }
int main(int argc, char** argv) {
    long long N = argc > 1 ? atoll(argv[1]) : 200000000LL;
    long long bad = 0, refpos = 0, filtpos = 0, badGroup = 0, badMatrix = 0, matrixpos = 0, matrixTested = 0;
#pragma omp parallel reduction(+:bad,refpos,filtpos,badGroup,badMatrix,matrixpos,matrixTested)
    {
        uint64_t s = 0x9E3779B97F4A7C15ull * (omp_get_thread_num() + 1);
#pragma omp for schedule(static)
        for (long long i = 0; i < N; ++i) {
            double scale = pow(10.0, -2 + 6 * urand(s));                 // centre distance scale 1e-2 .. 1e4
            double r = scale * pow(10.0, -3 + 3 * urand(s));              // radius 1e-3..1 of that
            double c[3], o[3], n[3], t[3];
            for (int a = 0; a < 3; ++a) { c[a] = scale * (2 * urand(s) - 1); n[a] = 2 * urand(s) - 1; t[a] = 2 * urand(s) - 1; }
            double nl = sqrt(n[0]*n[0]+n[1]*n[1]+n[2]*n[2]); for (int a = 0; a < 3; ++a) n[a] /= nl;
            double tn = t[0]*n[0]+t[1]*n[1]+t[2]*n[2]; for (int a = 0; a < 3; ++a) t[a] -= tn * n[a];
            double tl = sqrt(t[0]*t[0]+t[1]*t[1]+t[2]*t[2]); for (int a = 0; a < 3; ++a) t[a] /= tl;
            double eps = pow(10.0, -9 + 7 * urand(s)) * (urand(s) < 0.5 ? -1 : 1);
            double L = scale * pow(10.0, -2 + 3 * urand(s));              // distance walked back along the tangent
            for (int a = 0; a < 3; ++a) o[a] = c[a] + n[a] * r * (1 + eps) - t[a] * L;
            f3 of = mk3((float)o[0], (float)o[1], (float)o[2]);
            f3 df = normalize(mk3((float)t[0], (float)t[1], (float)t[2]));
            float rf = (float)r;
            f4 sp = {(float)c[0], (float)c[1], (float)c[2], rf * rf};
            // reference discriminant (testSphere's arithmetic)
            float coX = sp.x - of.x, coY = sp.y - of.y, coZ = sp.z - of.z;
            float nb = coX * df.x + coY * df.y + coZ * df.z;
            float cc = coX * coX + coY * coY + coZ * coZ - sp.w;
            float discr = nb * nb - cc;
            bool ref = discr > 0;
            f3 dk = mk3(df.x * TPT_P1_K, df.y * TPT_P1_K, df.z * TPT_P1_K);
            bool filt = memberFilter(sp, of, dk);
            refpos += ref; filtpos += filt;
            if (ref && !filt) bad++;
            // matrix-core filter (phase1MatrixH): the sphere's 32 A-side slot values as packScene builds them, the ray's as the
            // device packs them; the MFMA's accumulation is the hardware's, so the WORST CASE the error model allows is
            // tested: exact slot sum (binary64) minus 64 u x (sum of the slot products' magnitudes) must not be negative.
            // Spheres / rays outside binary16 range have no table / keep every candidate: nothing to miss there.
            {
                float am[TPT_MX_K], bm[TPT_MX_K], bs[TPT_MXH_SLOTS], as[TPT_MXH_SLOTS];
                matrixSphereSide(sp.x, sp.y, sp.z, sp.w, am);
                matrixRaySide(of, df, bm);
                bool inRange = matrixRayInRange(of, bm);
                for (int k = 0; k < 10; ++k) inRange = inRange && fabsf(am[k]) < 60000.0f;
                if (inRange) {
                    matrixRaySlots(bm, bs);
                    float hi[10], lo[10];
                    for (int k = 0; k < 10; ++k) { hi[k] = f16val(f16rtz(am[k])); lo[k] = f16val(f16rtz(am[k] - hi[k])); }
                    for (int t = 0; t < TPT_MXH_TERMS; ++t) { as[2 * t] = hi[t]; as[2 * t + 1] = hi[t]; }
                    for (int u = 0; u < 4; ++u) { as[18 + 2 * u] = lo[2 * u]; as[19 + 2 * u] = lo[2 * u + 1]; }
                    as[26] = lo[8]; as[27] = 1.0f; as[28] = 1.0f; as[29] = hi[9]; as[30] = lo[9]; as[31] = 0.0f;
                    double sum = 0.0, mag = 0.0;
                    for (int k = 0; k < TPT_MXH_SLOTS; ++k) { const double pr = (double)as[k] * (double)bs[k]; sum += pr; mag += fabs(pr); }
                    const bool mf = sum - 64.0 * 5.9604644775390625e-08 * mag >= 0.0;
                    matrixpos += mf; matrixTested++;
                    if (ref && !mf) badMatrix++;
                }
            }
            // group filter: a bounding sphere R = a + r around a centre displaced by a (rho = a / r up to 64)
            double rho = 64 * urand(s), a = rho * r, R = (a + r) * 1.00001;
            double u[3] = {2 * urand(s) - 1, 2 * urand(s) - 1, 2 * urand(s) - 1};
            double ul = sqrt(u[0]*u[0]+u[1]*u[1]+u[2]*u[2]);
            f4 gs = {(float)(c[0] + a * u[0] / ul), (float)(c[1] + a * u[1] / ul), (float)(c[2] + a * u[2] / ul), 0};
            float gcoX = gs.x - of.x, gcoY = gs.y - of.y, gcoZ = gs.z - of.z;
            float gdx = df.x * TPT_PG_K, gdy = df.y * TPT_PG_K, gdz = df.z * TPT_PG_K;
            float gnb = fma1(gcoZ, gdz, fma1(gcoY, gdy, gcoX * gdx));
            float nsq = (float)(-(R * R) * (1.0 + 1.0 / 4096.0));
            float ge = fma1(gcoZ, gcoZ, fma1(gcoY, gcoY, fma1(gcoX, gcoX, nsq)));
            float gv = fma1(gnb, gnb, -ge);
            bool gf = (f2u(gv) >> 31) == 0u;
            if (ref && !gf) badGroup++;
        }
    }
    printf("trials %lld  reference accepts %lld  filter passes %lld  matrix filter passes %lld of %lld in binary16 range  FILTER MISSES %lld  GROUP FILTER MISSES %lld  MATRIX FILTER MISSES %lld\n",
           N, refpos, filtpos, matrixpos, matrixTested, bad, badGroup, badMatrix);
    return bad || badGroup || badMatrix || matrixTested * 4 < N ? 1 : 0;
}
