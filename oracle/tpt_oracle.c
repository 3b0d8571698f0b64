/* oracle/tpt_oracle.c -- TEST INFRASTRUCTURE ONLY (checker; never the thing measured or shipped).
 *
 * Plain-C restatement of the reference's CPU scalar path, function by function.  Every function
 * names the reference file:line it follows (paths relative to /root/reference/Cpp/Source).
 * It adds what the reference cannot offer because of its compile-time macros and static tables:
 * runtime spp / scene / camera / row range, the PER_PIXEL seed mode, a libm-free math mode and the
 * forward-accumulation fold.  In (ROW_SERIAL, RECURSIVE) mode with the default scene it must
 * reproduce the pristine reference build bit for bit -- pinned by tests/test_oracle_pin.py against
 * oracle/_ref/libtpt_ref.so and the golden vectors of BASELINE.md section 2.
 *
 * Bit-parity hazards handled explicitly:
 *   H1  RNG draw order inside float3(rnd,rnd[,rnd]) ctor calls is GCC's right-to-left.
 *   H2  built with -ffp-contract=off, no fast-math; every expression keeps the reference's
 *       association.
 *   H3  libm: MATH_LIBM calls the host sinf/cosf/powf, MATH_TPT the restated glibc algorithms.
 */
#include "tpt_oracle.h"
#include "tpt_oracle_math.h"
#include <math.h>
#include <stdlib.h>
#if defined(TPTO_PTHREADS)
#include <pthread.h>
#include <stdatomic.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

#define kPI 3.1415926f /* Maths.h:9 */
static const float kMinT = 0.001f; /* Test.cpp:71 */
static const float kMaxT = 1.0e7f; /* Test.cpp:72 */
#define kMaxDepth 10               /* Test.cpp:73 */

/* ---------------------------------------------------------------- float3 (Maths.h:250-286, scalar branch) */
typedef struct { float x, y, z; } f3;
static inline f3 mk(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 add(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 muls(f3 a, float b) { return mk(a.x * b, a.y * b, a.z * b); }   /* float3 * float */
static inline f3 smul(float a, f3 b) { return mk(a * b.x, a * b.y, a * b.z); }   /* float * float3 */
static inline f3 neg(f3 a) { return mk(-a.x, -a.y, -a.z); }
static inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* Maths.h:277 */
static inline f3 cross(f3 a, f3 b)                                                /* Maths.h:278-285 */
{
    return mk(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
static inline float length3(f3 v) { return sqrtf(dot(v, v)); }                    /* Maths.h:299 */
static inline float sqLength(f3 v) { return dot(v, v); }                          /* Maths.h:300 */
static inline f3 normalize(f3 v) { return muls(v, 1.0f / length3(v)); }           /* Maths.h:301 */
static inline f3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

typedef struct { int math_mode; } MathCtx;
static inline float m_sinf(int mode, float x) { return mode == TPTO_MATH_TPT ? tptm_sinf(x) : sinf(x); }
static inline float m_cosf(int mode, float x) { return mode == TPTO_MATH_TPT ? tptm_cosf(x) : cosf(x); }
static inline float m_pow5(int mode, float x) { return mode == TPTO_MATH_TPT ? tptm_pow5f(x) : powf(x, 5); }

/* ---------------------------------------------------------------- RNG (Maths.cpp:5-47) */
uint32_t tpto_xorshift32(uint32_t* state) /* Maths.cpp:5-13 */
{
    uint32_t x = *state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 15;
    *state = x;
    return x;
}
float tpto_random_float01(uint32_t* state) /* Maths.cpp:15-18 */
{
    return (tpto_xorshift32(state) & 0xFFFFFF) / 16777216.0f;
}
#define RND(s) tpto_random_float01(s)

static f3 RandomInUnitDisk(uint32_t* state) /* Maths.cpp:20-28; H1: y is drawn first */
{
    f3 p;
    do {
        float ry = RND(state);
        float rx = RND(state);
        p = sub(smul(2.0f, mk(rx, ry, 0)), mk(1, 1, 0));
    } while (dot(p, p) >= 1.0f);
    return p;
}
static f3 RandomInUnitSphere(uint32_t* state) /* Maths.cpp:30-37; H1: z, y, x */
{
    f3 p;
    do {
        float rz = RND(state);
        float ry = RND(state);
        float rx = RND(state);
        p = sub(smul(2.0f, mk(rx, ry, rz)), mk(1, 1, 1));
    } while (sqLength(p) >= 1.0f);
    return p;
}
static f3 RandomUnitVector(uint32_t* state, int mm) /* Maths.cpp:39-47 */
{
    float z = RND(state) * 2.0f - 1.0f;
    float a = RND(state) * 2.0f * kPI;
    float r = sqrtf(1.0f - z * z);
    float x = r * m_cosf(mm, a);
    float y = r * m_sinf(mm, a);
    return mk(x, y, z);
}

/* ---------------------------------------------------------------- scene */
typedef struct { f3 orig, dir; } Ray;           /* Maths.h:334-343 */
typedef struct { f3 pos, normal; float t; } Hit; /* Maths.h:346-351 */

typedef struct {
    const TptoSphere* spheres; /* AoS, used by light sampling (Test.cpp:102-111) */
    const TptoMaterial* mats;
    int count;
    /* SoA (Maths.h:368-404), filled like UpdateTest does (Test.cpp:321-331) */
    float *cx, *cy, *cz, *sqR, *invR;
    int* emissive; /* s_EmissiveSpheres, Test.cpp:67 (sized to the scene here) */
    int emissiveCount;
    const TptoCamera* cam;
    int math_mode, fold_mode;
    int no_light_sampling, mitsuba_compare; /* Config.h:24-25 as run-time switches */
} Scene;

static int HitSpheres(const Scene* sc, const Ray* r, float tMin, float tMax, Hit* outHit) /* Maths.cpp:165-202 */
{
    float hitT = tMax;
    int id = -1;
    for (int i = 0; i < sc->count; ++i) {
        float coX = sc->cx[i] - r->orig.x;
        float coY = sc->cy[i] - r->orig.y;
        float coZ = sc->cz[i] - r->orig.z;
        float nb = coX * r->dir.x + coY * r->dir.y + coZ * r->dir.z;
        float c = coX * coX + coY * coY + coZ * coZ - sc->sqR[i];
        float discr = nb * nb - c;
        if (discr > 0) {
            float discrSq = sqrtf(discr);
            float t = nb - discrSq;
            if (t <= tMin) t = nb + discrSq;
            if (t > tMin && t < hitT) {
                id = i;
                hitT = t;
            }
        }
    }
    if (id != -1) {
        outHit->pos = add(r->orig, muls(r->dir, hitT)); /* Ray::pointAt, Maths.h:339 */
        outHit->normal = muls(sub(outHit->pos, mk(sc->cx[id], sc->cy[id], sc->cz[id])), sc->invR[id]);
        outHit->t = hitT;
        return id;
    }
    return -1;
}

static inline f3 reflect3(f3 v, f3 n) { return sub(v, smul(2 * dot(v, n), n)); } /* Maths.h:310-313 */
static inline int refract3(f3 v, f3 n, float nint, f3* out)                      /* Maths.h:315-326 */
{
    float dt = dot(v, n);
    float discr = 1.0f - nint * nint * (1 - dt * dt);
    if (discr > 0) {
        *out = sub(smul(nint, sub(v, muls(n, dt))), muls(n, sqrtf(discr)));
        return 1;
    }
    return 0;
}
static inline float schlick(float cosine, float ri, int mm) /* Maths.h:327-332 */
{
    float r0 = (1 - ri) / (1 + ri);
    r0 = r0 * r0;
    return r0 + (1 - r0) * m_pow5(mm, 1 - cosine);
}

/* Test.cpp:83-193.  matId plays the role of the `&mat == &smat` identity test (Test.cpp:100). */
static int Scatter(const Scene* sc, int matId, const Ray* r_in, const Hit* rec, f3* attenuation, Ray* scattered,
                   f3* outLightE, int64_t* inoutRayCount, uint32_t* state)
{
    const TptoMaterial* mat = &sc->mats[matId];
    const int mm = sc->math_mode;
    *outLightE = mk(0, 0, 0);
    if (mat->type == TPTO_LAMBERT) {
        f3 target = add(add(rec->pos, rec->normal), RandomUnitVector(state, mm));
        scattered->orig = rec->pos;
        scattered->dir = normalize(sub(target, rec->pos));
        f3 matAlbedo = ld3(mat->albedo);
        *attenuation = matAlbedo;
        for (int j = 0; j < (sc->no_light_sampling ? 0 : sc->emissiveCount); ++j) { /* Test.cpp:96-133 (#if DO_LIGHT_SAMPLING) */
            int i = sc->emissive[j];
            if (i == matId) continue; /* skip self */
            const TptoMaterial* smat = &sc->mats[i];
            const TptoSphere* s = &sc->spheres[i];
            f3 scn = mk(s->cx, s->cy, s->cz);
            f3 sw = normalize(sub(scn, rec->pos));
            f3 su = normalize(cross(fabsf(sw.x) > 0.01f ? mk(0, 1, 0) : mk(1, 0, 0), sw));
            f3 sv = cross(sw, su);
            float cosAMax = sqrtf(1.0f - s->radius * s->radius / sqLength(sub(rec->pos, scn)));
            float eps1 = RND(state), eps2 = RND(state);
            float cosA = 1.0f - eps1 + eps1 * cosAMax;
            float sinA = sqrtf(1.0f - cosA * cosA);
            float phi = 2 * kPI * eps2;
            f3 l = add(add(muls(su, m_cosf(mm, phi) * sinA), muls(sv, m_sinf(mm, phi) * sinA)), muls(sw, cosA));
            Hit lightHit;
            ++*inoutRayCount;
            Ray sr = {rec->pos, l};
            int hitID = HitSpheres(sc, &sr, kMinT, kMaxT, &lightHit);
            if (hitID != -1 && hitID == i) {
                float omega = 2 * kPI * (1 - cosAMax);
                f3 rdir = r_in->dir;
                f3 nl = dot(rec->normal, rdir) < 0 ? rec->normal : neg(rec->normal);
                f3 smatEmissive = ld3(smat->emissive);
                float d = dot(l, nl);
                float mx = 0.0f < d ? d : 0.0f; /* std::max(0.0f, d) */
                *outLightE = add(*outLightE, muls(mul(matAlbedo, smatEmissive), mx * omega / kPI));
            }
        }
        return 1;
    } else if (mat->type == TPTO_METAL) { /* Test.cpp:137-150 */
        f3 refl = reflect3(r_in->dir, rec->normal);
        float roughness = mat->roughness;
        if (sc->mitsuba_compare) roughness = 0; /* Test.cpp:143-145 */
        scattered->orig = rec->pos;
        scattered->dir = normalize(add(refl, smul(roughness, RandomInUnitSphere(state))));
        *attenuation = ld3(mat->albedo);
        return dot(scattered->dir, rec->normal) > 0;
    } else if (mat->type == TPTO_DIELECTRIC) { /* Test.cpp:151-186 */
        f3 outwardN;
        f3 rdir = r_in->dir;
        f3 refl = reflect3(rdir, rec->normal);
        float nint;
        *attenuation = mk(1, 1, 1);
        f3 refr = mk(0, 0, 0);
        float reflProb;
        float cosine;
        if (dot(rdir, rec->normal) > 0) {
            outwardN = neg(rec->normal);
            nint = mat->ri;
            cosine = mat->ri * dot(rdir, rec->normal);
        } else {
            outwardN = rec->normal;
            nint = 1.0f / mat->ri;
            cosine = -dot(rdir, rec->normal);
        }
        if (refract3(rdir, outwardN, nint, &refr))
            reflProb = schlick(cosine, mat->ri, mm);
        else
            reflProb = 1;
        scattered->orig = rec->pos;
        if (RND(state) < reflProb)
            scattered->dir = normalize(refl);
        else
            scattered->dir = normalize(refr);
    } else { /* Test.cpp:187-191 */
        *attenuation = mk(1, 0, 1);
        return 0;
    }
    return 1;
}

static inline f3 Sky(const Ray* r) /* Test.cpp:229-231 */
{
    float t = 0.5f * (r->dir.y + 1.0f);
    return muls(add(smul(1.0f - t, mk(1.0f, 1.0f, 1.0f)), smul(t, mk(0.5f, 0.7f, 1.0f))), 0.3f);
}

static f3 Trace(const Scene* sc, const Ray* r, int depth, int64_t* inoutRayCount, uint32_t* state, int doMaterialE) /* Test.cpp:195-234 */
{
    Hit rec;
    ++*inoutRayCount;
    int id = HitSpheres(sc, r, kMinT, kMaxT, &rec);
    if (id != -1) {
        Ray scattered;
        f3 attenuation, lightE;
        const TptoMaterial* mat = &sc->mats[id];
        f3 matE = ld3(mat->emissive);
        if (depth < kMaxDepth && Scatter(sc, id, r, &rec, &attenuation, &scattered, &lightE, inoutRayCount, state)) {
            if (!sc->no_light_sampling) { /* Test.cpp:209-214 */
                if (!doMaterialE) matE = mk(0, 0, 0);
                doMaterialE = (mat->type != TPTO_LAMBERT);
            }
            f3 rest = Trace(sc, &scattered, depth + 1, inoutRayCount, state, doMaterialE);
            return add(add(matE, lightE), mul(attenuation, rest));
        }
        return matE;
    }
    return sc->mitsuba_compare ? mk(0.15f, 0.21f, 0.3f) : Sky(r); /* Test.cpp:226-231 */
}

/* Same paths as Trace, colour folded front-to-back (what a GPU megakernel naturally does). */
static f3 TraceForward(const Scene* sc, Ray r, int64_t* inoutRayCount, uint32_t* state)
{
    f3 radiance = mk(0, 0, 0), throughput = mk(1, 1, 1);
    int doMaterialE = 1;
    for (int depth = 0;; ++depth) {
        Hit rec;
        ++*inoutRayCount;
        int id = HitSpheres(sc, &r, kMinT, kMaxT, &rec);
        if (id == -1) return add(radiance, mul(throughput, sc->mitsuba_compare ? mk(0.15f, 0.21f, 0.3f) : Sky(&r)));
        Ray scattered;
        f3 attenuation, lightE;
        const TptoMaterial* mat = &sc->mats[id];
        f3 matE = ld3(mat->emissive);
        if (depth < kMaxDepth && Scatter(sc, id, &r, &rec, &attenuation, &scattered, &lightE, inoutRayCount, state)) {
            if (!sc->no_light_sampling) {
                if (!doMaterialE) matE = mk(0, 0, 0);
                doMaterialE = (mat->type != TPTO_LAMBERT);
            }
            radiance = add(radiance, mul(throughput, add(matE, lightE)));
            throughput = mul(throughput, attenuation);
            r = scattered;
        } else {
            return add(radiance, mul(throughput, matE));
        }
    }
}

static Ray CameraGetRay(const TptoCamera* c, float s, float t, uint32_t* state) /* Maths.h:437-442 */
{
    f3 rd = smul(c->lensRadius, RandomInUnitDisk(state));
    f3 offset = add(muls(ld3(c->uu), rd.x), muls(ld3(c->vv), rd.y));
    Ray r;
    r.orig = add(ld3(c->origin), offset);
    r.dir = normalize(sub(sub(add(add(ld3(c->lowerLeftCorner), smul(s, ld3(c->horizontal))), smul(t, ld3(c->vertical))),
                              ld3(c->origin)), offset));
    return r;
}

/* Test.cpp:266-300 for rows [start,end) */
static int64_t TraceRows(const Scene* sc, const TptoParams* p, int start, int end, float* backbufferBase)
{
    float* backbuffer = backbufferBase + (size_t)start * p->width * 4;
    float invWidth = 1.0f / p->width;
    float invHeight = 1.0f / p->height;
    float lerpFac = (float)p->frame / (float)(p->frame + 1);
    if (p->flags & TPTO_FLAG_ANIMATE) lerpFac *= p->has_animate_smoothing ? p->animate_smoothing : 0.9f; /* DO_ANIMATE_SMOOTHING, Config.h:23 */
    if (!(p->flags & TPTO_FLAG_PROGRESSIVE)) lerpFac = 0;
    int64_t rayCount = 0;
    for (uint32_t y = (uint32_t)start; y < (uint32_t)end; ++y) {
        uint32_t state = (y * 9781u + (uint32_t)p->frame * 6271u) | 1u; /* Test.cpp:280 */
        for (int x = 0; x < p->width; ++x) {
            if (p->seed_mode == TPTO_SEED_PER_PIXEL) /* ComputeShader.hlsl:380 */
                state = ((uint32_t)x * 1973u + y * 9277u + (uint32_t)p->frame * 26699u) | 1u;
            f3 col = mk(0, 0, 0);
            for (int s = 0; s < p->spp; s++) {
                float u = ((float)x + RND(&state)) * invWidth;
                float v = ((float)y + RND(&state)) * invHeight;
                Ray r = CameraGetRay(sc->cam, u, v, &state);
                f3 c = sc->fold_mode == TPTO_FOLD_FORWARD ? TraceForward(sc, r, &rayCount, &state)
                                                         : Trace(sc, &r, 0, &rayCount, &state, 1);
                col = add(col, c);
            }
            col = muls(col, 1.0f / (float)p->spp);
            f3 prev = mk(backbuffer[0], backbuffer[1], backbuffer[2]);
            col = add(muls(prev, lerpFac), muls(col, 1 - lerpFac));
            backbuffer[0] = col.x;
            backbuffer[1] = col.y;
            backbuffer[2] = col.z;
            backbuffer += 4;
        }
    }
    return rayCount;
}

#if defined(TPTO_PTHREADS)
typedef struct { const Scene* sc; const TptoParams* p; float* bb; int y1; atomic_int next; atomic_llong rays; } RowPool;
static void* rowWorker(void* arg)
{
    RowPool* pool = (RowPool*)arg;
    long long rays = 0;
    for (;;) {
        const int y = atomic_fetch_add(&pool->next, 4);
        if (y >= pool->y1) break;
        rays += TraceRows(pool->sc, pool->p, y, y + 4 < pool->y1 ? y + 4 : pool->y1, pool->bb);
    }
    atomic_fetch_add(&pool->rays, rays);
    return NULL;
}
#endif

int64_t tpto_render(const TptoSphere* spheres, const TptoMaterial* mats, int count, const TptoCamera* cam,
                    const TptoParams* p, float* backbuffer)
{
    Scene* sc = (Scene*)calloc(1, sizeof(Scene));
    if (count < 0) count = 0;
    sc->emissive = (int*)malloc(sizeof(int) * (size_t)(count > 0 ? count : 1));
    sc->spheres = spheres;
    sc->mats = mats;
    sc->count = count;
    sc->cam = cam;
    sc->math_mode = p->math_mode;
    sc->fold_mode = p->fold_mode;
    sc->no_light_sampling = p->no_light_sampling;
    sc->mitsuba_compare = p->mitsuba_compare;
    float* soa = (float*)malloc(sizeof(float) * 5 * (size_t)(count > 0 ? count : 1));
    sc->cx = soa; sc->cy = soa + count; sc->cz = soa + 2 * count; sc->sqR = soa + 3 * count; sc->invR = soa + 4 * count;
    for (int i = 0; i < count; ++i) { /* Test.cpp:321-339 */
        sc->cx[i] = spheres[i].cx;
        sc->cy[i] = spheres[i].cy;
        sc->cz[i] = spheres[i].cz;
        sc->sqR[i] = spheres[i].radius * spheres[i].radius;
        sc->invR[i] = spheres[i].invRadius;
        if (mats[i].emissive[0] > 0 || mats[i].emissive[1] > 0 || mats[i].emissive[2] > 0)
            sc->emissive[sc->emissiveCount++] = i;
    }
    int64_t rays = 0;
    int y0 = p->y0 < 0 ? 0 : p->y0, y1 = p->y1 > p->height ? p->height : p->y1;
#if defined(TPTO_PTHREADS)
    /* the same row fan-out on plain pthreads + C11 atomics (ThreadSanitizer understands these; it does not understand libgomp's
       barriers and reports every access around an OpenMP region): the build oracle/Makefile makes for `oracle_soak_tsan` */
    {
        RowPool pool;
        pool.sc = sc; pool.p = p; pool.bb = backbuffer; pool.y1 = y1;
        atomic_init(&pool.next, y0);
        atomic_init(&pool.rays, 0);
        int nt = p->threads > 0 ? p->threads : 8;
        if (nt > 256) nt = 256;
        pthread_t th[256];
        for (int i = 0; i < nt; ++i) pthread_create(&th[i], NULL, rowWorker, &pool);
        for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
        rays = atomic_load(&pool.rays);
    }
#else
#ifdef _OPENMP
    int nt = p->threads > 0 ? p->threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays) num_threads(nt)
#endif
    for (int y = y0; y < y1; ++y) rays += TraceRows(sc, p, y, y + 1, backbuffer);
#endif
    free(soa);
    free(sc->emissive);
    free(sc);
    return rays;
}

int tpto_hit_spheres(const TptoSphere* spheres, int count, const float orig[3], const float dir[3], float tMin,
                     float tMax, float* outT, float outPos[3], float outNormal[3])
{
    Scene sc;
    float* soa = (float*)malloc(sizeof(float) * 5 * (size_t)(count > 0 ? count : 1));
    sc.count = count;
    sc.cx = soa; sc.cy = soa + count; sc.cz = soa + 2 * count; sc.sqR = soa + 3 * count; sc.invR = soa + 4 * count;
    for (int i = 0; i < count; ++i) {
        sc.cx[i] = spheres[i].cx; sc.cy[i] = spheres[i].cy; sc.cz[i] = spheres[i].cz;
        sc.sqR[i] = spheres[i].radius * spheres[i].radius; sc.invR[i] = spheres[i].invRadius;
    }
    Ray r = {ld3(orig), ld3(dir)};
    Hit h = {{0, 0, 0}, {0, 0, 0}, 0};
    int id = HitSpheres(&sc, &r, tMin, tMax, &h);
    free(soa);
    if (outT) *outT = h.t;
    if (outPos) { outPos[0] = h.pos.x; outPos[1] = h.pos.y; outPos[2] = h.pos.z; }
    if (outNormal) { outNormal[0] = h.normal.x; outNormal[1] = h.normal.y; outNormal[2] = h.normal.z; }
    return id;
}

/* ---------------------------------------------------------------- default scene / camera */
static void set_sphere(TptoSphere* s, float x, float y, float z, float r)
{
    s->cx = x; s->cy = y; s->cz = z; s->radius = r; s->invRadius = 0.0f; /* Maths.h:357 */
}
static void set_mat(TptoMaterial* m, int type, float r, float g, float b, float er, float eg, float eb, float rough, float ri)
{
    m->type = type;
    m->albedo[0] = r; m->albedo[1] = g; m->albedo[2] = b;
    m->emissive[0] = er; m->emissive[1] = eg; m->emissive[2] = eb;
    m->roughness = rough; m->ri = ri;
}

int tpto_default_scene(TptoSphere* S, TptoMaterial* M, int capacity) /* data of Test.cpp:13-31 and 46-64 */
{
    if (capacity < 46) return -1;
    static const float grey[9] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f, 0.9f};
    static const float rainbow[9][3] = {{0.8f, 0.1f, 0.1f}, {0.8f, 0.5f, 0.1f}, {0.8f, 0.8f, 0.1f},
                                        {0.4f, 0.8f, 0.1f}, {0.1f, 0.8f, 0.1f}, {0.1f, 0.8f, 0.5f},
                                        {0.1f, 0.8f, 0.8f}, {0.1f, 0.1f, 0.8f}, {0.5f, 0.1f, 0.8f}};
    int n = 0;
    set_sphere(&S[n++], 0, -100.5f, -1, 100); /* ground */
    for (int zi = 0; zi < 2; ++zi)
        for (int xi = 0; xi < 3; ++xi) set_sphere(&S[n++], (float)(2 - 2 * xi), 0, zi ? 1.0f : -1.0f, 0.5f);
    set_sphere(&S[n++], 0.5f, 1, 0.5f, 0.5f);   /* glass */
    set_sphere(&S[n++], -1.5f, 1.5f, 0.f, 0.3f); /* light */
    for (int row = 0; row < 4; ++row)
        for (int xi = 0; xi < 9; ++xi) set_sphere(&S[n++], (float)(4 - xi), 0, (float)(-3 - row), 0.5f);
    set_sphere(&S[n++], 1.5f, 1.5f, -2, 0.3f); /* light */

    int m = 0;
    set_mat(&M[m++], TPTO_LAMBERT, 0.8f, 0.8f, 0.8f, 0, 0, 0, 0, 0);
    set_mat(&M[m++], TPTO_LAMBERT, 0.8f, 0.4f, 0.4f, 0, 0, 0, 0, 0);
    set_mat(&M[m++], TPTO_LAMBERT, 0.4f, 0.8f, 0.4f, 0, 0, 0, 0, 0);
    set_mat(&M[m++], TPTO_METAL, 0.4f, 0.4f, 0.8f, 0, 0, 0, 0, 0);
    set_mat(&M[m++], TPTO_METAL, 0.4f, 0.8f, 0.4f, 0, 0, 0, 0, 0);
    set_mat(&M[m++], TPTO_METAL, 0.4f, 0.8f, 0.4f, 0, 0, 0, 0.2f, 0);
    set_mat(&M[m++], TPTO_METAL, 0.4f, 0.8f, 0.4f, 0, 0, 0, 0.6f, 0);
    set_mat(&M[m++], TPTO_DIELECTRIC, 0.4f, 0.4f, 0.4f, 0, 0, 0, 0, 1.5f);
    set_mat(&M[m++], TPTO_LAMBERT, 0.8f, 0.6f, 0.2f, 30, 25, 15, 0, 0);
    for (int i = 0; i < 9; ++i) set_mat(&M[m++], TPTO_LAMBERT, grey[i], grey[i], grey[i], 0, 0, 0, 0, 0);
    for (int i = 0; i < 9; ++i) set_mat(&M[m++], TPTO_METAL, grey[i], grey[i], grey[i], 0, 0, 0, 0, 0);
    for (int i = 0; i < 9; ++i) set_mat(&M[m++], TPTO_METAL, rainbow[i][0], rainbow[i][1], rainbow[i][2], 0, 0, 0, 0, 0);
    for (int i = 0; i < 9; ++i)
        set_mat(&M[m++], i < 8 ? TPTO_LAMBERT : TPTO_METAL, rainbow[i][0], rainbow[i][1], rainbow[i][2], 0, 0, 0, 0, 0);
    set_mat(&M[m++], TPTO_LAMBERT, 0.1f, 0.2f, 0.5f, 3, 10, 20, 0, 0);
    tpto_update_derived(S, n);
    return n;
}

void tpto_update_derived(TptoSphere* spheres, int count) /* Sphere::UpdateDerivedData, Maths.h:359 */
{
    for (int i = 0; i < count; ++i) spheres[i].invRadius = 1.0f / spheres[i].radius;
}

void tpto_animate(TptoSphere* s, float time) /* Test.cpp:304-308 */
{
    s[1].cy = cosf(time) + 1.0f;
    s[8].cz = sinf(time) * 0.3f;
}

static void st3(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

void tpto_camera(TptoCamera* cam, const float lookFrom[3], const float lookAt[3], const float vup[3], float vfov,
                 float aspect, float aperture, float focusDist) /* Maths.h:418-435 */
{
    cam->lensRadius = aperture / 2;
    float theta = vfov * kPI / 180;
    float halfHeight = tanf(theta / 2);
    float halfWidth = aspect * halfHeight;
    f3 org = ld3(lookFrom);
    f3 w = normalize(sub(ld3(lookFrom), ld3(lookAt)));
    f3 u = normalize(cross(ld3(vup), w));
    f3 v = cross(w, u);
    st3(cam->origin, org);
    st3(cam->ww, w);
    st3(cam->uu, u);
    st3(cam->vv, v);
    st3(cam->lowerLeftCorner,
        sub(sub(sub(org, smul(halfWidth * focusDist, u)), smul(halfHeight * focusDist, v)), smul(focusDist, w)));
    st3(cam->horizontal, smul(2 * halfWidth * focusDist, u));
    st3(cam->vertical, smul(2 * halfHeight * focusDist, v));
}

void tpto_default_camera(TptoCamera* cam, int width, int height) /* Test.cpp:309-319,341 */
{
    const float lookfrom[3] = {0, 2, 3}, lookat[3] = {0, 0, 0}, vup[3] = {0, 1, 0};
    float distToFocus = 3;
    float aperture = 0.1f;
    aperture *= 0.2f;
    tpto_camera(cam, lookfrom, lookat, vup, 60, (float)width / (float)height, aperture, distToFocus);
}

/* ---------------------------------------------------------------- math pinning helpers */
float tpto_sinf(float x) { return tptm_sinf(x); }
float tpto_cosf(float x) { return tptm_cosf(x); }
float tpto_pow5f(float x) { return tptm_pow5f(x); }

int64_t tpto_check_sincos_vs_libm(void)
{
    int64_t bad = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : bad)
#endif
    for (uint32_t k = 0; k < (1u << 24); ++k) {
        float r = (float)k / 16777216.0f;
        float a1 = r * 2.0f * kPI; /* Maths.cpp:42 */
        float a2 = 2 * kPI * r;    /* Test.cpp:115 */
        bad += tptm_asu(tptm_sinf(a1)) != tptm_asu(sinf(a1));
        bad += tptm_asu(tptm_cosf(a1)) != tptm_asu(cosf(a1));
        bad += tptm_asu(tptm_sinf(a2)) != tptm_asu(sinf(a2));
        bad += tptm_asu(tptm_cosf(a2)) != tptm_asu(cosf(a2));
    }
    return bad;
}

int64_t tpto_check_pow5_vs_libm(uint32_t stride)
{
    int64_t bad = 0;
    if (stride == 0) stride = 1;
    int64_t n = (0x3f800000u - 0x2b800000u) / stride;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : bad)
#endif
    for (int64_t j = 0; j <= n; ++j) {
        uint32_t u = 0x2b800000u + (uint32_t)j * stride; /* [2^-40, 1] */
        float x = tptm_asf(u), xn = tptm_asf(u | 0x80000000u);
        bad += tptm_asu(tptm_pow5f(x)) != tptm_asu(powf(x, 5));
        bad += tptm_asu(tptm_pow5f(xn)) != tptm_asu(powf(xn, 5));
    }
    bad += tptm_asu(tptm_pow5f(0.0f)) != tptm_asu(powf(0.0f, 5));
    return bad;
}

uint32_t tpto_fnv1a(const void* data, uint64_t bytes)
{
    const uint8_t* p = (const uint8_t*)data;
    uint32_t h = 0x811c9dc5u;
    for (uint64_t i = 0; i < bytes; ++i) h = (h ^ p[i]) * 16777619u;
    return h;
}
