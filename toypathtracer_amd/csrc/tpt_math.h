// tpt_math.h -- arithmetic layer of the MI355X path tracer (float3, RNG, libm-free sin/cos/pow5).
//
// Replaces the reference's L1 layer (Cpp/Source/Maths.h scalar float3 :250-286, helpers :299-332,
// Maths.cpp RNG :5-47).  Design rule: every value a branch can depend on is computed with IEEE-754
// correctly-rounded binary32 add/sub/mul/div/sqrt in the reference's association (the library is
// built with -ffp-contract=off, no fast-math; hipcc's default correctly-rounded fp32 div/sqrt
// expansions are used), and the three libm calls of the path (sinf, cosf, powf(x,5)) are evaluated
// with glibc 2.35's published binary64 algorithms so that CPU and GPU agree bit for bit.
//
// The header is plain C++ that compiles both as gfx950 device code (hipcc) and as host code
// (g++/clang); the host compile is used ONLY by tests/ to exercise the per-lane logic without a
// GPU -- the shipped library never runs it on the CPU.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TPT_HD __host__ __device__ __forceinline__
#else
#define TPT_HD inline __attribute__((always_inline))
#endif

namespace tpt {

// A wave-uniform integer the compiler may not carry in a scalar register across the kernel's main loop: on the device the value is
// laundered through an empty asm at the point of use, so what is derived from it (stride x level, division magic, ...) is made
// again there instead of being hoisted out of the loop and spilled (the path-queue kernel had 54-63 SGPRs spilled to VGPR lanes,
// most of them such products needed once per pixel or per sample).
TPT_HD int uniformHere(int v)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TPT_NO_UNIFORM_HERE)
    asm volatile("" : "+s"(v));
#endif
    return v;
}

#define TPT_PI 3.1415926f // kPI, Maths.h:9

// ---------------------------------------------------------------- bit casts
TPT_HD uint32_t f2u(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
#endif
}
TPT_HD float u2f(uint32_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}
TPT_HD uint64_t d2u(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(d);
#else
    uint64_t u;
    __builtin_memcpy(&u, &d, 8);
    return u;
#endif
}
TPT_HD double u2d(uint64_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double d;
    __builtin_memcpy(&d, &u, 8);
    return d;
#endif
}
// Correctly rounded sqrtf.  hipcc's own expansion (-fhip-fp32-correctly-rounded-divide-sqrt, the default) costs 17 VALU
// instructions, 7 of them for denormal scaling and the inf / nan / zero fix-up.  For 2^-96 <= x <= 2^96 -- everything a
// path ever feeds it -- five do: y = v_rsq_f32(x), s0 = x y, one fused residual correction s0 + (x - s0^2) (y / 2).
// PROVEN BY EXHAUSTION, not by argument: tools/exhaustive/exhaustive_math.hip and test_gpu_math.py::test_fast_sqrt_*
// compare it with the compiler's expansion for every one of the 2^32 binary32 inputs on the device (0 mismatches; outside
// the guarded range the expansion itself runs).  The guard is two integer instructions.
#define TPT_SQRT_LO 0x0f800000u /* 2^-96 */
#define TPT_SQRT_HI 0x6f800000u /* 2^96 */
TPT_HD float tsqrt(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y;
    const float r = __builtin_fmaf(-s0, s0, x);
    float s = __builtin_fmaf(r, 0.5f * y, s0);
    if (__builtin_expect(__float_as_uint(x) - TPT_SQRT_LO > TPT_SQRT_HI - TPT_SQRT_LO, 0)) s = __builtin_sqrtf(x);
    return s;
#else
    return sqrtf(x);
#endif
}
// 1.0f / sqrtf(x), both roundings (what normalize() multiplies by, Maths.h:301): the fast sqrt above, then
// r = v_rcp_f32(L) and ONE fused Newton step r + r (1 - L r).  Exhaustively equal to 1.0f / sqrtf(x) as hipcc expands it
// (27 instructions) for all x in the guarded range (same harness; the reciprocal step alone holds for every L that is a
// square root of a normal number).  8 instructions + the guard.
TPT_HD float trsqrt2(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y;
    const float r = __builtin_fmaf(-s0, s0, x);
    const float L = __builtin_fmaf(r, 0.5f * y, s0);
    const float q = __builtin_amdgcn_rcpf(L);
    float inv = __builtin_fmaf(__builtin_fmaf(-L, q, 1.0f), q, q);
    if (__builtin_expect(__float_as_uint(x) - TPT_SQRT_LO > TPT_SQRT_HI - TPT_SQRT_LO, 0)) inv = 1.0f / __builtin_sqrtf(x);
    return inv;
#else
    return 1.0f / sqrtf(x);
#endif
}
TPT_HD double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Correctly rounded binary32 division in 6 instructions instead of hipcc's 11 (v_div_scale x 2, v_rcp, 5 fma / mul,
// v_div_fmas, v_div_fixup): Markstein's sequence  y0 = v_rcp_f32(b); e = fma(-b, y0, 1); y1 = fma(e, y0, y0); q0 = a y1;
// r = fma(-b, q0, a); q = fma(r, y1, q0).  PROVEN BY EXHAUSTION on the device, not by argument: tools/exhaustive/
// exhaustive_div.hip compares it with the compiler's expansion for ALL 2^23 x 2^23 pairs of significands (42 s of an
// MI355X, profiles/r04/r04_run1.log: 0 mismatches; the 4-instruction form without the reciprocal refinement fails for
// 47 045 pairs).  A quotient's significand depends on the operands' significands only as long as no intermediate leaves the
// normal range, so the result holds for every a, b with exponents in [-60, 60] (r = a - b q0 ~ 2^-24 a stays normal); a
// generic guard on both operands would cost what the sequence saves, so the two hot call sites use forms whose guard is one
// range check:
//   tdivSafeNum(a, b): a is known (checked on the host) to lie in [2^-60, 2^60]; b is checked here -- Scatter's r^2 / d^2;
//   tdivByPi(a):       b = kPI; a >= +0 (a product of a clamped cosine and a solid angle), 0 or >= 2^-100 takes the
//                      3-instruction form with the correctly rounded reciprocal of kPI as a constant (every significand of a
//                      checked on the host: tests/test_lane_logic.py::test_division_by_pi_all_significands).
// Everything else, and the host build, runs the plain IEEE division.
#define TPT_DIV_LO 0x21800000u /* 2^-60 */
#define TPT_DIV_HI 0x5d800000u /* 2^60 */
TPT_HD bool tdivInRange(float x) { return f2u(x) - TPT_DIV_LO <= TPT_DIV_HI - TPT_DIV_LO; } // (positive, finite, |exponent| <= 60)
TPT_HD float tdivSafeNum(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    const float q0 = a * y1;
    const float r = __builtin_fmaf(-b, q0, a);
    float q = __builtin_fmaf(r, y1, q0);
    if (__builtin_expect(!tdivInRange(b), 0)) q = a / b;
    return q;
#else
    return a / b;
#endif
}
#define TPT_INV_PI 0.318309903144836425781f /* RN(1 / 3.1415926f): 0x3ea2f984 */
TPT_HD float tdivByPiFast(float a) // the 3-instruction form alone (host-checkable: plain fmaf)
{
    const float q0 = a * TPT_INV_PI;
    const float r = __builtin_fmaf(-TPT_PI, q0, a);
    return __builtin_fmaf(r, TPT_INV_PI, q0);
}
TPT_HD float tdivByPi(float a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float q = tdivByPiFast(a);
    // +0 and [2^-100, 2^126) take the fast form; negative, tiny, huge and non-finite arguments the IEEE expansion
    const uint32_t bits = f2u(a);
    const bool fast = bits == 0u || bits - 0x0d800000u < 0x7e800000u - 0x0d800000u;
    if (__builtin_expect(!fast, 0)) q = a / TPT_PI;
    return q;
#else
    return a / TPT_PI;
#endif
}

// ---------------------------------------------------------------- float3
struct f3 {
    float x, y, z;
};
TPT_HD f3 mk3(float x, float y, float z)
{
    f3 r;
    r.x = x; r.y = y; r.z = z;
    return r;
}
TPT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
TPT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
TPT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
TPT_HD f3 operator*(f3 a, float b) { return mk3(a.x * b, a.y * b, a.z * b); }
TPT_HD f3 operator*(float a, f3 b) { return mk3(a * b.x, a * b.y, a * b.z); }
TPT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
TPT_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // Maths.h:277
TPT_HD f3 cross(f3 a, f3 b)                                                 // Maths.h:278-285
{
    return mk3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
TPT_HD float sqLength(f3 v) { return dot(v, v); }
TPT_HD float length(f3 v) { return tsqrt(dot(v, v)); }
TPT_HD f3 normalize(f3 v) { return v * trsqrt2(dot(v, v)); } // Maths.h:301: v * (1.0f / length(v)) -- reciprocal, then multiply
TPT_HD f3 reflect(f3 v, f3 n) { return v - (2 * dot(v, n)) * n; } // Maths.h:310-313
TPT_HD bool refract(f3 v, f3 n, float nint, f3& out)              // Maths.h:315-326
{
    float dt = dot(v, n);
    float discr = 1.0f - nint * nint * (1 - dt * dt);
    if (discr > 0) {
        out = nint * (v - n * dt) - n * tsqrt(discr);
        return true;
    }
    return false;
}

// ---------------------------------------------------------------- RNG (Maths.cpp:5-18)
TPT_HD uint32_t xorshift32(uint32_t& state)
{
    uint32_t x = state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 15;
    state = x;
    return x;
}
TPT_HD float rnd01(uint32_t& state) { return (float)(xorshift32(state) & 0xFFFFFF) / 16777216.0f; }

// ---------------------------------------------------------------- sinf/cosf for |y| < 120
// glibc 2.35 sysdeps/ieee754/flt-32/s_sincosf.h (reduce_fast, sinf_poly) + s_sinf.c/s_cosf.c,
// fma at glibc's x86-64 FMA-variant contraction points.  Bit-identical to libm on every argument
// the path produces (pinned exhaustively on the oracle side: oracle/tpt_oracle_math.h).
TPT_HD uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ff; }

TPT_HD float sincos_poly(double x, double x2, bool neg, int n)
{
    if ((n & 1) == 0) {
        const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
        double x3 = x * x2;
        double s1 = dfma(x2, S3, S2);
        double x7 = x3 * x2;
        double s = dfma(x3, S1, x);
        return (float)dfma(x7, s1, s);
    } else {
        const double C0 = 1.0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                     C4 = 0x1.99343027bf8c3p-16;
        double sg = neg ? -1.0 : 1.0;
        double x4 = x2 * x2;
        double c2 = dfma(x2, sg * C4, sg * C3);
        double c1 = dfma(x2, sg * C1, sg * C0);
        double x6 = x4 * x2;
        double c = dfma(x4, sg * C2, c1);
        return (float)dfma(x6, c2, c);
    }
}
TPT_HD double reduce_fast(double x, int& n)
{
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    double r = x * HPI_INV;
    n = ((int32_t)r + 0x800000) >> 24;
    return dfma(-(double)n, HPI, x);
}
TPT_HD float tsinf(float y)
{
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sincos_poly(x, x * x, false, 0);
    }
    int n;
    x = reduce_fast(x, n);
    double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return sincos_poly(x * s, x * x, (n & 2) != 0, n);
}
TPT_HD float tcosf(float y)
{
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return sincos_poly(x, x * x, false, 1);
    }
    int n;
    x = reduce_fast(x, n);
    double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return sincos_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
}
// sin and cos of the same angle share the argument reduction (glibc's sincosf does the same and
// returns the same bits as sinf/cosf).
TPT_HD void tsincosf(float y, float& outSin, float& outCos)
{
    if (abstop12(y) < abstop12(0x1p-12f)) {
        outSin = y;
        outCos = 1.0f;
        return;
    }
    // glibc branches on |y| < pi/4 to skip the argument reduction; the reduction of such an argument is the identity (n = 0,
    // x - 0 * pi/2 = x, sign +1, same polynomials on the same x and x^2), so ONE path serves both and a wave whose lanes
    // straddle pi/4 -- nearly every wave: the angle is uniform in [0, 2 pi) -- no longer runs the polynomials twice.
    //
    // Each polynomial is evaluated ONCE, on the unsigned operands, and the signs are applied to the binary32 results:
    //  * sinf_poly is odd in x and cosf_poly's sign factor multiplies every coefficient, so flipping the sign of the input
    //    (sine) or of all coefficients (cosine) flips the sign of every intermediate -- products, fmas and the final
    //    conversion round to nearest, which is symmetric -- and the result is the negated result, bit for bit;
    //  * sin and cos of the pair take the two polynomials on the same (x, x^2): which one is the sine depends on the
    //    quadrant's parity, so both are computed and swapped (written as two calls the compiler evaluated each polynomial
    //    in both branches of every wave: 31 binary64 operations per pair, now 14).
    // Same bits: tests/test_lane_logic.py::test_sincos_pair_all_floats (every float with |y| < 120 against tsinf / tcosf on the
    // host) and test_gpu_math.py::test_sincos_pair_equals_sinf_cosf_on_the_whole_path_domain (all 2^24 arguments of both
    // call forms on the device).
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    const double x0 = y;
    const double r = x0 * HPI_INV;
    const int32_t q = (int32_t)r + 0x800000;
    const int n = q >> 24;
    const double x = dfma(-(double)n, HPI, x0);
    const double x2 = x * x;
    // sinf_poly(x, n even)
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const double x3 = x * x2;
    const double s1 = dfma(x2, S3, S2);
    const double x7 = x3 * x2;
    const double sp = dfma(x3, S1, x);
    const float sinPoly = (float)dfma(x7, s1, sp);
    // cosf_poly (n odd), sign factor +1
    const double C0 = 1.0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16;
    const double x4 = x2 * x2;
    const double c2 = dfma(x2, C4, C3);
    const double c1 = dfma(x2, C1, C0);
    const double x6 = x4 * x2;
    const double cp = dfma(x4, C2, c1);
    const float cosPoly = (float)dfma(x6, c2, cp);
    // signs: the sine polynomial's argument is x * s with s = -1 in quadrants 1 and 2 (n & 3), the cosine polynomial's
    // factor is -1 when n & 2; the quadrant's parity says which polynomial is the sine
    const uint32_t sinFlip = (((uint32_t)n + 1u) & 2u) << 30; // n & 3 in {1, 2}
    const uint32_t cosFlip = ((uint32_t)n & 2u) << 30;
    const float a = u2f(f2u(sinPoly) ^ sinFlip), b = u2f(f2u(cosPoly) ^ cosFlip);
    const bool odd = (n & 1) != 0;
    outSin = odd ? b : a;
    outCos = odd ? a : b;
}

// ---------------------------------------------------------------- powf(x, 5.0f)
// glibc 2.35 sysdeps/ieee754/flt-32/e_powf.c (log2_inline, exp2_inline, tables __powf_log2_data and
// __exp2f_data), specialised to y = 5.  Bit-identical to libm for all floats in [2^-40,1] and
// [-1,-2^-40] (pinned on the oracle side).  Used by schlick (Maths.h:327-332).
#if defined(__HIP_DEVICE_COMPILE__)
#define TPT_TABLE_QUAL __constant__
#else
#define TPT_TABLE_QUAL static
#endif
TPT_TABLE_QUAL const double kPowLog2Tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
TPT_TABLE_QUAL const uint64_t kExp2Tab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b,
    0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb,
    0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429,
    0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad,
    0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540};

TPT_HD double pow_log2_inline(uint32_t ix)
{
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
                 A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp+0;
    uint32_t tmp = ix - 0x3f330000u;
    int i = (int)((tmp >> (23 - 4)) % 16);
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)top >> 23;
    double invc = kPowLog2Tab[i][0], logc = kPowLog2Tab[i][1];
    double z = (double)u2f(iz);
    double r = dfma(z, invc, -1.0);
    double y0 = logc + (double)k;
    double r2 = r * r;
    double y = dfma(A0, r, A1);
    double p = dfma(A2, r, A3);
    double r4 = r2 * r2;
    double q = dfma(A4, r, y0);
    q = dfma(p, r2, q);
    y = dfma(y, r4, q);
    return y;
}
TPT_HD float pow_exp2_inline(double xd, uint32_t sign_bias)
{
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    const double SHIFT = 0x1.8p+47;
    double kd = xd + SHIFT;
    uint64_t ki = d2u(kd);
    kd -= SHIFT;
    double r = xd - kd;
    uint64_t t = kExp2Tab[ki % 32];
    uint64_t ski = ki + sign_bias;
    t += ski << (52 - 5);
    double s = u2d(t);
    double z = dfma(C0, r, C1);
    double r2 = r * r;
    double y = dfma(C2, r, 1.0);
    y = dfma(z, r2, y);
    y = y * s;
    return (float)y;
}
TPT_HD float tpow5f(float x)
{
    uint32_t sign_bias = 0;
    uint32_t ix = f2u(x);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (2 * ix - 1 >= 2u * 0x7f800000u - 1) { // +-0, inf, nan
            float x2 = x * x;
            if (ix & 0x80000000u) x2 = -x2;
            return x2;
        }
        if (ix & 0x80000000u) { // finite x < 0; y = 5 is an odd integer
            sign_bias = 1u << 16;
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) { // subnormal
            ix = f2u(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    double logx = pow_log2_inline(ix);
    double ylogx = 5.0 * logx;
    if (((d2u(ylogx) >> 47) & 0xffff) >= (d2u(126.0) >> 47)) {
        if (ylogx > 0x1.fffffffd1d571p+6) return u2f(sign_bias ? 0xff800000u : 0x7f800000u);
        if (ylogx <= -150.0) return u2f(sign_bias ? 0x80000000u : 0u);
    }
    return pow_exp2_inline(ylogx, sign_bias);
}
TPT_HD float schlickR0(float cosine, float r0sq) // the same with r0^2 = ((1 - ri) / (1 + ri))^2 taken from the material record (packScene)
{
    return r0sq + (1 - r0sq) * tpow5f(1 - cosine);
}
TPT_HD float schlick(float cosine, float ri) // Maths.h:327-332
{
    float r0 = (1 - ri) / (1 + ri);
    r0 = r0 * r0;
    return r0 + (1 - r0) * tpow5f(1 - cosine);
}

// ---------------------------------------------------------------- samplers (Maths.cpp:20-47)
// The reference builds float3(rnd,rnd[,rnd]) inside one expression; with GCC (the compiler of the
// pinned goldens) the LAST constructor argument is evaluated first.  Spelled out here.
TPT_HD f3 randomInUnitDisk(uint32_t& state)
{
    f3 p;
    do {
        float ry = rnd01(state);
        float rx = rnd01(state);
        p = 2.0f * mk3(rx, ry, 0) - mk3(1, 1, 0);
    } while (dot(p, p) >= 1.0f);
    return p;
}
TPT_HD f3 randomInUnitSphere(uint32_t& state)
{
    f3 p;
    do {
        float rz = rnd01(state);
        float ry = rnd01(state);
        float rx = rnd01(state);
        p = 2.0f * mk3(rx, ry, rz) - mk3(1, 1, 1);
    } while (sqLength(p) >= 1.0f);
    return p;
}
TPT_HD f3 randomUnitVector(uint32_t& state)
{
    float z = rnd01(state) * 2.0f - 1.0f;
    float a = rnd01(state) * 2.0f * TPT_PI;
    float r = tsqrt(1.0f - z * z);
    float sn, cs;
    tsincosf(a, sn, cs);
    float x = r * cs;
    float y = r * sn;
    return mk3(x, y, z);
}

} // namespace tpt
