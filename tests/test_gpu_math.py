"""GPU parity of the arithmetic layer (tpt_math.h on gfx950) against the oracle, bit for bit:
correctly-rounded sqrt/div expansions, libm-free sin/cos/pow5, RNG, schlick, normalize."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def rnd_floats(rng, n, lo_exp=-30, hi_exp=30, signed=False):
    m = rng.uniform(1.0, 2.0, n)
    e = rng.integers(lo_exp, hi_exp, n)
    x = (m * np.exp2(e)).astype(np.float32)
    if signed:
        x *= rng.choice(np.float32([-1, 1]), n)
    return x


def test_sqrt_correctly_rounded(tpt_hooks):
    rng = np.random.default_rng(0)
    x = np.concatenate([rnd_floats(rng, 1 << 20, -60, 60), np.float32([0, 1, 2, 4, 1e-38, 1e-40, 1e-45, 3.4e38, 0.25])])
    assert np.array_equal(bits(tpt_hooks.test_math(0, x)), bits(np.sqrt(x)))


@pytest.mark.parametrize("op", [0, 1], ids=["sqrt", "normalize_scale"])
def test_fast_sqrt_paths_equal_the_compilers_expansion_for_all_2_to_32_inputs(tpt_hooks, op):
    """tpt_math.h's 5-instruction sqrt and 8-instruction 1.0f / sqrtf (guarded to [2^-96, 2^96], the compiler's expansion
    outside) against hipcc's correctly rounded sqrt / divide for EVERY binary32 bit pattern.  Together with
    test_sqrt_correctly_rounded / test_schlick_rng_normalize (the compiler's expansion == the host's IEEE results) this
    pins the fast paths to IEEE for all inputs."""
    bad, first = tpt_hooks.test_math_exhaustive(op)
    assert bad == 0, ["%08x" % v for v in first]


def test_div_correctly_rounded(tpt_hooks):
    rng = np.random.default_rng(1)
    a = rnd_floats(rng, 1 << 20, -40, 40, signed=True)
    b = rnd_floats(rng, 1 << 20, -40, 40, signed=True)
    assert np.array_equal(bits(tpt_hooks.test_math(1, a, b)), bits(a / b))


def test_sincos_bit_exact_on_path_domain(tpt_hooks, oracle):
    k = np.arange(0, 1 << 24, 7, dtype=np.uint32)
    r = k.astype(np.float32) / np.float32(16777216.0)
    for a in (r * np.float32(2.0) * np.float32(3.1415926), np.float32(2 * np.float32(3.1415926)) * r):
        a = a.astype(np.float32)
        want_s = np.array([oracle.lib.tpto_sinf(v) for v in a[::97]], np.float32)
        want_c = np.array([oracle.lib.tpto_cosf(v) for v in a[::97]], np.float32)
        got_s, got_c = tpt_hooks.test_math(2, a), tpt_hooks.test_math(3, a)
        assert np.array_equal(bits(got_s[::97]), bits(want_s))
        assert np.array_equal(bits(got_c[::97]), bits(want_c))
        # the oracle's math is pinned to libm exhaustively (tests/test_oracle_math.py); on this host numpy's
        # float32 sin/cos are not glibc's, so the full-array check is against a vectorised copy of the same
        # algorithm in float64 numpy
        assert np.all(np.abs(got_s.astype(np.float64) - np.sin(a.astype(np.float64))) < 1.2e-7)
        assert np.all(np.abs(got_c.astype(np.float64) - np.cos(a.astype(np.float64))) < 1.2e-7)


def test_sincos_pair_equals_sinf_cosf_on_the_whole_path_domain(tpt_hooks):
    """tsincosf (one argument-reduction path for every angle) against tsinf / tcosf (glibc's branch structure, pinned to libm
    above) for ALL 2^24 arguments rnd01 * 2 * kPI can take, in both association orders the path uses (Maths.cpp:42, Test.cpp:116)."""
    k = np.arange(0, 1 << 24, dtype=np.uint32)
    r = k.astype(np.float32) / np.float32(16777216.0)
    for a in (r * np.float32(2.0) * np.float32(3.1415926), np.float32(2 * np.float32(3.1415926)) * r):
        a = a.astype(np.float32)
        assert np.array_equal(bits(tpt_hooks.test_math(8, a)), bits(tpt_hooks.test_math(2, a)))
        assert np.array_equal(bits(tpt_hooks.test_math(9, a)), bits(tpt_hooks.test_math(3, a)))


def test_pow5_bit_exact(tpt_hooks, oracle):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-0.5, 1.0, 200000).astype(np.float32), rnd_floats(rng, 50000, -39, 0, signed=True),
                        np.float32([0.0, 1.0, -0.5, -1.0, 1e-7, 0.999999])])
    want = np.array([oracle.lib.tpto_pow5f(v) for v in x], np.float32)
    assert np.array_equal(bits(tpt_hooks.test_math(4, x)), bits(want))


def test_schlick_rng_normalize(tpt_hooks, oracle):
    import ctypes as C
    rng = np.random.default_rng(3)
    # RNG: 16th draw of the stream seeded with the input bits | 1
    seeds = rng.integers(1, 1 << 31, 4096, dtype=np.uint32)
    got = tpt_hooks.test_math(5, seeds.view(np.float32))
    want = np.empty(len(seeds), np.float32)
    for i, s in enumerate(seeds):
        st = C.c_uint32(int(s) | 1)
        for _ in range(16):
            v = oracle.lib.tpto_random_float01(C.byref(st))
        want[i] = v
    assert np.array_equal(bits(got), bits(want))
    # schlick(cosine, ri) = r0 + (1-r0)*pow5(1-cosine)  (Maths.h:327-332)
    cosine = rng.uniform(0, 1.5, 50000).astype(np.float32)
    ri = np.full_like(cosine, 1.5)
    r0 = (np.float32(1) - ri) / (np.float32(1) + ri)
    r0 = r0 * r0
    p5 = np.array([oracle.lib.tpto_pow5f(np.float32(1) - c) for c in cosine], np.float32)
    assert np.array_equal(bits(tpt_hooks.test_math(6, cosine, ri)), bits(r0 + (np.float32(1) - r0) * p5))
    # normalize(x,y,1).x = x * (1/sqrt(x*x+y*y+1))
    x = rng.uniform(-3, 3, 50000).astype(np.float32)
    y = rng.uniform(-3, 3, 50000).astype(np.float32)
    want = x * (np.float32(1.0) / np.sqrt(x * x + y * y + np.float32(1) * np.float32(1)))
    assert np.array_equal(bits(tpt_hooks.test_math(7, x, y)), bits(want))


def test_short_divisions_equal_ieee(tpt_hooks):
    """tpt_math.h's 6-instruction division (tdivSafeNum: numerator in [2^-60, 2^60], any divisor -- out-of-range divisors take
    the compiler's expansion) and the 3-instruction division by kPI (tdivByPi) against the host's IEEE quotient, bit for bit:
    random significands over the whole guarded exponent range, the guard's edges, zeros, denormals, huge and negative values.
    (All 2^46 significand pairs: tools/exhaustive/exhaustive_div.hip, 42 s on the device, profiles/r04/r04_run1.log; all 2^23
    significands of a / kPI: tests/test_lane_logic.py.)"""
    rng = np.random.default_rng(7)
    n = 1 << 21
    a = rnd_floats(rng, n, -60, 60)
    b = np.concatenate([rnd_floats(rng, n - 4096, -60, 60), rnd_floats(rng, 2048, -126, -60), rnd_floats(rng, 2040, 60, 127),
                        np.float32([0.0, -0.0, 1e-45, 1e-40, -1.5, 3.4e38, np.inf, 2.0 ** -60])])
    with np.errstate(divide="ignore", invalid="ignore", over="ignore", under="ignore"):
        want = a / b
    got = tpt_hooks.test_math(10, a, b)
    ok = bits(got) == bits(want)
    assert ok.all(), [(float(x), float(y)) for x, y in zip(a[~ok][:4], b[~ok][:4])]
    x = np.concatenate([rnd_floats(rng, n, -110, 4), rnd_floats(rng, 4096, -126, -99), rnd_floats(rng, 4096, 100, 127), rnd_floats(rng, 4096, -20, 3, signed=True),
                        np.float32([0.0, -0.0, 1e-45, 1e-39, 2.0 ** -100, np.nextafter(np.float32(2.0 ** -100), np.float32(0)), 6.2831855, 3.1415926, 3.4e38, np.inf])])
    with np.errstate(over="ignore", under="ignore"):
        want = x / np.float32(3.1415926)
    got = tpt_hooks.test_math(11, x)
    ok = bits(got) == bits(want)
    assert ok.all(), [float(v) for v in x[~ok][:8]]
