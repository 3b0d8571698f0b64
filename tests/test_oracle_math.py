"""The libm-free sin/cos/pow5 restatement (oracle/tpt_oracle_math.h, mirrored by
toypathtracer_amd/csrc/tpt_math.h) is bit-identical to the host libm on the path's input domain."""
import numpy as np


def test_sincos_bit_identical_to_libm_on_all_2p24_arguments(oracle):
    # every r = k/2^24 in both call forms: r*2.0f*kPI (Maths.cpp:42) and 2*kPI*r (Test.cpp:115)
    assert oracle.lib.tpto_check_sincos_vs_libm() == 0


def test_pow5_bit_identical_to_libm_strided(oracle):
    # all floats in [2^-40,1] and [-1,-2^-40] with stride 5 (the exhaustive stride-1 run, 671 M values,
    # was done once when the tables were extracted: 0 mismatches)
    assert oracle.lib.tpto_check_pow5_vs_libm(5) == 0


def test_special_values(oracle):
    lib = oracle.lib
    assert lib.tpto_pow5f(0.0) == 0.0
    assert lib.tpto_pow5f(1.0) == 1.0
    assert lib.tpto_pow5f(-0.5) == -0.03125
    assert lib.tpto_sinf(0.0) == 0.0 and lib.tpto_cosf(0.0) == 1.0
    assert abs(lib.tpto_sinf(np.float32(1.5707964)) - 1.0) < 1e-7
