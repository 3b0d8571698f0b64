/* oracle/tpt_oracle_math.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Platform-independent restatement of the three libm functions the reference's hot path calls:
 *   sinf/cosf : Maths.cpp:44-45 (RandomUnitVector), Test.cpp:116 (light sampling)
 *   powf(x,5) : Maths.h:331 (schlick)
 * The reference gets them from the host libm, i.e. glibc 2.35 on the box the goldens were made on.
 * glibc is an un-vendored dependency of the reference (not under /root/reference); its published
 * algorithm (glibc 2.35 sysdeps/ieee754/flt-32/s_sincosf.h, s_sinf.c, s_cosf.c, e_powf.c, derived
 * from ARM optimized-routines) is restated here on IEEE-754 binary64 operations only, so the very
 * same sequence can be executed by the HIP kernel.  Constants were read out of this container's
 * libm.so.6 data tables.  Pinning (tests/test_oracle_math.py, via tpto_check_* in tpt_oracle.c):
 *   - sin/cos: bit-identical to libm for ALL 2^24 arguments the path can produce in each of the
 *     two forms `r*2.0f*kPI` (Maths.cpp:42) and `2*kPI*r` (Test.cpp:115), r = k/2^24;
 *   - pow5:   bit-identical to libm powf(x,5) for ALL floats in [2^-40,1] and [-1,-2^-40]
 *     (671 M values; the path evaluates it on x = 1-cosine in [-0.5,1]).
 * The fma() calls are where glibc's x86-64 FMA ifunc variant contracts; with plain mul+add pow5
 * differs from libm on 2 inputs per 335 M, sin/cos on none.
 */
#ifndef TPT_ORACLE_MATH_H
#define TPT_ORACLE_MATH_H
#include <stdint.h>
#include <string.h>

static inline uint32_t tptm_asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float tptm_asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t tptm_asu64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double tptm_asd(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
#define TPTM_FMA(a, b, c) __builtin_fma((a), (b), (c))

/* ---- sinf / cosf (s_sincosf.h: reduce_fast, sinf_poly; __sincosf_table) ---- */
static inline uint32_t tptm_abstop12(float x) { return (tptm_asu(x) >> 20) & 0x7ff; }

/* n odd -> cosine polynomial, n even -> sine polynomial; neg -> table[1] (negated cosine). */
static inline float tptm_sincos_poly(double x, double x2, int neg, int n)
{
    static const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    static const double C0 = 1.0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
                        C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = TPTM_FMA(x2, S3, S2);
        double x7 = x3 * x2;
        double s = TPTM_FMA(x3, S1, x);
        return (float)TPTM_FMA(x7, s1, s);
    } else {
        double sg = neg ? -1.0 : 1.0;
        double x4 = x2 * x2;
        double c2 = TPTM_FMA(x2, sg * C4, sg * C3);
        double c1 = TPTM_FMA(x2, sg * C1, sg * C0);
        double x6 = x4 * x2;
        double c = TPTM_FMA(x4, sg * C2, c1);
        return (float)TPTM_FMA(x6, c2, c);
    }
}

static inline double tptm_reduce_fast(double x, int* np)
{
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    double r = x * HPI_INV;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return TPTM_FMA(-(double)n, HPI, x);
}

/* valid for |y| < 120 (the path only produces [0, 2*pi]) */
static inline float tptm_sinf(float y)
{
    double x = y;
    int n;
    if (tptm_abstop12(y) < tptm_abstop12(0x1.921FB6p-1f)) {
        if (tptm_abstop12(y) < tptm_abstop12(0x1p-12f)) return y;
        return tptm_sincos_poly(x, x * x, 0, 0);
    }
    x = tptm_reduce_fast(x, &n);
    double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return tptm_sincos_poly(x * s, x * x, (n & 2) != 0, n);
}

static inline float tptm_cosf(float y)
{
    double x = y;
    int n;
    if (tptm_abstop12(y) < tptm_abstop12(0x1.921FB6p-1f)) {
        if (tptm_abstop12(y) < tptm_abstop12(0x1p-12f)) return 1.0f;
        return tptm_sincos_poly(x, x * x, 0, 1);
    }
    x = tptm_reduce_fast(x, &n);
    double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return tptm_sincos_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
}

/* ---- powf(x, 5.0f) (e_powf.c: log2_inline, exp2_inline; __powf_log2_data, __exp2f_data) ---- */
static inline double tptm_log2_inline(uint32_t ix)
{
    static const double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
        {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
        {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
        {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
        {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
        {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
    static const double A[5] = {0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2,
                                -0x1.7154748bef6c8p-1, 0x1.71547652ab82bp+0};
    uint32_t tmp = ix - 0x3f330000u;
    int i = (tmp >> (23 - 4)) % 16;
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)top >> 23;
    double invc = T[i][0], logc = T[i][1];
    double z = (double)tptm_asf(iz);
    double r = TPTM_FMA(z, invc, -1.0);
    double y0 = logc + (double)k;
    double r2 = r * r;
    double y = TPTM_FMA(A[0], r, A[1]);
    double p = TPTM_FMA(A[2], r, A[3]);
    double r4 = r2 * r2;
    double q = TPTM_FMA(A[4], r, y0);
    q = TPTM_FMA(p, r2, q);
    y = TPTM_FMA(y, r4, q);
    return y;
}

static inline float tptm_exp2_inline(double xd, uint32_t sign_bias)
{
    static const uint64_t T[32] = {
        0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
        0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
        0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
        0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
        0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
        0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
        0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
        0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
    static const double C[3] = {0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1};
    const double SHIFT = 0x1.8p+47; /* 0x1.8p52 / 32 */
    double kd = xd + SHIFT;
    uint64_t ki = tptm_asu64(kd);
    kd -= SHIFT;
    double r = xd - kd;
    uint64_t t = T[ki % 32];
    uint64_t ski = ki + sign_bias;
    t += ski << (52 - 5);
    double s = tptm_asd(t);
    double z = TPTM_FMA(C[0], r, C[1]);
    double r2 = r * r;
    double y = TPTM_FMA(C[2], r, 1.0);
    y = TPTM_FMA(z, r2, y);
    y = y * s;
    return (float)y;
}

/* == glibc powf(x, 5.0f) for every finite x (errno / fp exceptions not modelled) */
static inline float tptm_pow5f(float x)
{
    uint32_t sign_bias = 0;
    uint32_t ix = tptm_asu(x);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (2 * ix - 1 >= 2u * 0x7f800000u - 1) { /* x is +-0, inf or nan */
            float x2 = x * x;
            if (ix & 0x80000000u) x2 = -x2;
            return x2;
        }
        if (ix & 0x80000000u) { /* finite x < 0, y = 5 is an odd integer */
            sign_bias = 1u << 16;
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) { /* subnormal */
            ix = tptm_asu(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    double logx = tptm_log2_inline(ix);
    double ylogx = 5.0 * logx;
    if (((tptm_asu64(ylogx) >> 47) & 0xffff) >= (tptm_asu64(126.0) >> 47)) {
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
    }
    return tptm_exp2_inline(ylogx, sign_bias);
}

#endif
