// tools/ubench_valu.hip -- VALU issue-rate microbenchmark for gfx950 (design input for the phase-1
// sphere loop: is packed fp32 (v_pk_mul/add_f32) faster than scalar fp32 per sphere test?).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f, m2 = 0.9999999f;
    unsigned long long mask = 0x5555555555555555ull;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    v2f pm = {m, m};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, dm = 1.0000001;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // v_mul_f32, 8 independent chains x 8
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (MODE == 1) { // v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));)
        } else if (MODE == 2) { // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (MODE == 3) { // v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                              "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));)
        } else if (MODE == 4) { // v_fma_f64
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                              "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dm));)
        } else if (MODE == 5) { // v_alignbit_b32
            REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %1, %1, %2, 31\n v_alignbit_b32 %2, %2, %3, 31\n v_alignbit_b32 %3, %3, %4, 31\n"
                              "v_alignbit_b32 %4, %4, %5, 31\n v_alignbit_b32 %5, %5, %6, 31\n v_alignbit_b32 %6, %6, %7, 31\n v_alignbit_b32 %7, %7, %0, 31\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));)
        } else if (MODE == 6) { // v_sqrt_f32 (transcendental unit)
            REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                              "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 7) { // v_pk_add_f32 with an SGPR-pair operand (how the compiler feeds scene data)
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(1.0000001));)
        } else if (MODE == 8) { // mixed: v_mul_f32 + v_add_f32 dependent pairs (the scalar sphere test's shape)
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %0\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %2\n"
                              "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %4\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %6\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (MODE == 9) { // v_mul_f32 in the 8-byte VOP3 encoding (same two sources as mode 0)
            REP8(asm volatile("v_mul_f32_e64 %0, %0, %8\n v_mul_f32_e64 %1, %1, %8\n v_mul_f32_e64 %2, %2, %8\n v_mul_f32_e64 %3, %3, %8\n"
                              "v_mul_f32_e64 %4, %4, %8\n v_mul_f32_e64 %5, %5, %8\n v_mul_f32_e64 %6, %6, %8\n v_mul_f32_e64 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (MODE == 10) { // v_fmac_f32 (VOP2: dst += a * b -- three register reads in a 4-byte encoding)
            REP8(asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                              "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(m2));)
        } else if (MODE == 11) { // v_fma_f32 with three DISTINCT source registers (mode 2 reads one register twice)
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(m2));)
        } else if (MODE == 12) { // v_mul_f32 with a 32-bit literal (VOP2 + literal dword: 8 bytes)
            REP8(asm volatile("v_mul_f32_e32 %0, 0x3f800001, %0\n v_mul_f32_e32 %1, 0x3f800001, %1\n v_mul_f32_e32 %2, 0x3f800001, %2\n v_mul_f32_e32 %3, 0x3f800001, %3\n"
                              "v_mul_f32_e32 %4, 0x3f800001, %4\n v_mul_f32_e32 %5, 0x3f800001, %5\n v_mul_f32_e32 %6, 0x3f800001, %6\n v_mul_f32_e32 %7, 0x3f800001, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 13) { // v_cndmask_b32, VOP2 (mask in vcc)
            REP8(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
        } else if (MODE == 14) { // v_cndmask_b32, VOP3 (mask in an SGPR pair)
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                              "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "s"(mask));)
        } else if (MODE == 15) { // v_mov_b32 (VOP1)
            REP8(asm volatile("v_mov_b32_e32 %0, %1\n v_mov_b32_e32 %1, %2\n v_mov_b32_e32 %2, %3\n v_mov_b32_e32 %3, %4\n"
                              "v_mov_b32_e32 %4, %5\n v_mov_b32_e32 %5, %6\n v_mov_b32_e32 %6, %7\n v_mov_b32_e32 %7, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 16) { // v_readlane_b32 (a spilled SGPR coming back) + v_writelane_b32 (one going out): pairs
            REP8(asm volatile("v_readlane_b32 s20, %0, 3\n v_writelane_b32 %1, s20, 5\n v_readlane_b32 s21, %2, 3\n v_writelane_b32 %3, s21, 5\n"
                              "v_readlane_b32 s22, %4, 3\n v_writelane_b32 %5, s22, 5\n v_readlane_b32 s23, %6, 3\n v_writelane_b32 %7, s23, 5\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : : "s20", "s21", "s22", "s23");)
        } else if (MODE == 17) { // integer VOP2 (v_xor_b32 / v_lshlrev_b32: the XorShift32 stream)
            REP8(asm volatile("v_lshlrev_b32_e32 %1, 13, %0\n v_xor_b32_e32 %0, %0, %1\n v_lshrrev_b32_e32 %3, 17, %2\n v_xor_b32_e32 %2, %2, %3\n"
                              "v_lshlrev_b32_e32 %5, 5, %4\n v_xor_b32_e32 %4, %4, %5\n v_lshlrev_b32_e32 %7, 13, %6\n v_xor_b32_e32 %6, %6, %7\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));)
        } else if (MODE == 18) { // v_sub_f32 with a source modifier (|x| / -x force the VOP3 encoding)
            REP8(asm volatile("v_mul_f32_e64 %0, %0, -%8\n v_mul_f32_e64 %1, %1, -%8\n v_mul_f32_e64 %2, %2, -%8\n v_mul_f32_e64 %3, %3, -%8\n"
                              "v_mul_f32_e64 %4, %4, -%8\n v_mul_f32_e64 %5, %5, -%8\n v_mul_f32_e64 %6, %6, -%8\n v_mul_f32_e64 %7, %7, -%8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (MODE == 19) { // two v_fmac_f32 against one v_pk_fma_f32 (mode 3): the same flops unpacked
            REP8(asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                              "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n"
                              "v_fmac_f32_e32 %0, %9, %8\n v_fmac_f32_e32 %1, %9, %8\n v_fmac_f32_e32 %2, %9, %8\n v_fmac_f32_e32 %3, %9, %8\n"
                              "v_fmac_f32_e32 %4, %9, %8\n v_fmac_f32_e32 %5, %9, %8\n v_fmac_f32_e32 %6, %9, %8\n v_fmac_f32_e32 %7, %9, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(m2));)
        } else if (MODE == 20) { // v_cndmask_b32_e32 with vcc WRITTEN by the SALU once per block (mode 13 never defines it)
            REP8(asm volatile("s_mov_b64 vcc, %9\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "s"(mask) : "vcc");)
        } else if (MODE == 21) { // v_cmp_lt_f32 -> vcc, then v_cndmask_b32_e32 on it: the pair as the compiler emits it (one pair = 2 instructions)
            REP8(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %8\n s_nop 1\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cmp_lt_f32_e32 vcc, %1, %8\n s_nop 1\n v_cndmask_b32_e32 %1, %1, %8, vcc\n"
                              "v_cmp_lt_f32_e32 vcc, %2, %8\n s_nop 1\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cmp_lt_f32_e32 vcc, %3, %8\n s_nop 1\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
        } else if (MODE == 22) { // the same pair through an SGPR pair (v_cmp_lt_f32_e64 s[20:21] / v_cndmask_b32_e64)
            REP8(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %8\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %1, %8\n s_nop 1\n v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n"
                              "v_cmp_lt_f32_e64 s[20:21], %2, %8\n s_nop 1\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %3, %8\n s_nop 1\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "s20", "s21", "s22", "s23");)
        } else if (MODE == 23) { // v_cndmask_b32_e32 between plain multiplies (1 in 4): is the cost of mode 13 a throughput or a stall?
            REP8(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_mul_f32_e32 %1, %1, %8\n v_mul_f32_e32 %2, %2, %8\n v_mul_f32_e32 %3, %3, %8\n"
                              "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_mul_f32_e32 %5, %5, %8\n v_mul_f32_e32 %6, %6, %8\n v_mul_f32_e32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
        } else if (MODE == 24) { // v_lshlrev_b32 alone (independent)
            REP8(asm volatile("v_lshlrev_b32_e32 %0, 1, %0\n v_lshlrev_b32_e32 %1, 1, %1\n v_lshlrev_b32_e32 %2, 1, %2\n v_lshlrev_b32_e32 %3, 1, %3\n"
                              "v_lshlrev_b32_e32 %4, 1, %4\n v_lshlrev_b32_e32 %5, 1, %5\n v_lshlrev_b32_e32 %6, 1, %6\n v_lshlrev_b32_e32 %7, 1, %7\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));)
        } else if (MODE == 25) { // v_xor_b32 alone (independent)
            REP8(asm volatile("v_xor_b32_e32 %0, %0, %1\n v_xor_b32_e32 %1, %1, %2\n v_xor_b32_e32 %2, %2, %3\n v_xor_b32_e32 %3, %3, %4\n"
                              "v_xor_b32_e32 %4, %4, %5\n v_xor_b32_e32 %5, %5, %6\n v_xor_b32_e32 %6, %6, %7\n v_xor_b32_e32 %7, %7, %0\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));)
        } else if (MODE == 26) { // v_add_f32 / v_sub_f32 (independent)
            REP8(asm volatile("v_add_f32_e32 %0, %0, %8\n v_sub_f32_e32 %1, %1, %8\n v_add_f32_e32 %2, %2, %8\n v_sub_f32_e32 %3, %3, %8\n"
                              "v_add_f32_e32 %4, %4, %8\n v_sub_f32_e32 %5, %5, %8\n v_add_f32_e32 %6, %6, %8\n v_sub_f32_e32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (MODE == 27) { // v_cmp_lt_f32_e32 -> vcc alone
            REP8(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %8\n v_cmp_lt_f32_e32 vcc, %1, %8\n v_cmp_lt_f32_e32 vcc, %2, %8\n v_cmp_lt_f32_e32 vcc, %3, %8\n"
                              "v_cmp_lt_f32_e32 vcc, %4, %8\n v_cmp_lt_f32_e32 vcc, %5, %8\n v_cmp_lt_f32_e32 vcc, %6, %8\n v_cmp_lt_f32_e32 vcc, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
        } else if (MODE == 28) { // two v_cndmask_b32_e32 back to back, then two multiplies (the hit update of phase 2: t and id under one mask)
            REP8(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_mul_f32_e32 %2, %2, %8\n v_mul_f32_e32 %3, %3, %8\n"
                              "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_mul_f32_e32 %6, %6, %8\n v_mul_f32_e32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
        } else if (MODE == 29) { // the same two selects separated by a multiply each
            REP8(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_mul_f32_e32 %2, %2, %8\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_mul_f32_e32 %3, %3, %8\n"
                              "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_mul_f32_e32 %6, %6, %8\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_mul_f32_e32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
        } else if (MODE == 30) { // v_cmp -> vcc, TWO selects on it, one multiply (4 instructions)
            REP8(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %8\n s_nop 1\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_mul_f32_e32 %2, %2, %8\n"
                              "v_cmp_lt_f32_e32 vcc, %4, %8\n s_nop 1\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_mul_f32_e32 %6, %6, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
        } else if (MODE == 31) { // the same through v_bfi_b32 on a lane mask held in a register (one select makes the mask, v_bfi applies it)
            REP8(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %8\n s_nop 1\n v_cndmask_b32_e64 %3, 0, -1, vcc\n v_bfi_b32 %0, %3, %8, %0\n v_bfi_b32 %1, %3, %8, %1\n v_mul_f32_e32 %2, %2, %8\n"
                              "v_cmp_lt_f32_e32 vcc, %4, %8\n s_nop 1\n v_cndmask_b32_e64 %7, 0, -1, vcc\n v_bfi_b32 %4, %7, %8, %4\n v_bfi_b32 %5, %7, %8, %5\n v_mul_f32_e32 %6, %6, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y +
                                                 (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
}

template <int MODE>
static void run(const char* name, int wavesPerSimd, float* out)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = 4000;
    const int blocks = cus * wavesPerSimd; // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double instrPerWave = (double)iters * (MODE == 19 ? 128 : MODE == 31 ? 80 : 64);
    double waveInstr = instrPerWave * blocks * 4;
    double perSimdPerSec = waveInstr / (cus * 4) / (ms * 1e-3);
    printf("%-28s waves/SIMD %d  %8.3f ms  %7.3f G wave-instr/s/SIMD  (= %.2f cycles per wave-instr at 2.4 GHz)\n", name, wavesPerSimd, ms,
           perSimdPerSec / 1e9, 2.4e9 / perSimdPerSec);
}

int main()
{
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 64);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_mul_f32", w, out);
        run<1>("v_pk_mul_f32", w, out);
        run<2>("v_fma_f32", w, out);
        run<3>("v_pk_fma_f32", w, out);
        run<4>("v_fma_f64", w, out);
        run<5>("v_alignbit_b32", w, out);
        run<6>("v_sqrt_f32", w, out);
        run<7>("v_pk_add_f32 (sgpr src)", w, out);
        run<8>("v_mul+v_add dependent", w, out);
        // round 5: matched pairs -- is it the 8-byte encoding or the third register read that costs?
        run<9>("v_mul_f32_e64 (VOP3, 2 src)", w, out);
        run<18>("v_mul_f32_e64 with -src", w, out);
        run<12>("v_mul_f32_e32 + literal", w, out);
        run<10>("v_fmac_f32_e32 (VOP2)", w, out);
        run<11>("v_fma_f32 3 distinct src", w, out);
        run<19>("2x v_fmac_f32 (per pair)", w, out);
        run<13>("v_cndmask_b32_e32 (vcc)", w, out);
        run<14>("v_cndmask_b32_e64 (sgpr)", w, out);
        run<15>("v_mov_b32", w, out);
        run<16>("v_readlane+v_writelane", w, out);
        run<17>("v_lshl/v_xor int VOP2", w, out);
        run<24>("v_lshlrev_b32", w, out);
        run<25>("v_xor_b32", w, out);
        run<26>("v_add/v_sub_f32", w, out);
        run<27>("v_cmp_lt_f32_e32 -> vcc", w, out);
        run<20>("v_cndmask_e32, vcc by s_mov", w, out);
        run<21>("v_cmp + v_cndmask via vcc", w, out);
        run<22>("v_cmp + v_cndmask via sgpr", w, out);
        run<23>("1 v_cndmask_e32 + 3 v_mul", w, out);
        run<28>("2 v_cndmask_e32 adjacent+2mul", w, out);
        run<29>("2 v_cndmask_e32 apart + 2 mul", w, out);
        run<30>("cmp, 2 cndmask_e32, mul (x4)", w, out);
        run<31>("cmp, mask, 2 bfi, mul (x5)", w, out);
    }
    return 0;
}
