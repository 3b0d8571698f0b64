// tools/exhaustive/mfma_f16_probe.hip -- experiment harness (not product): facts about v_mfma_f32_32x32x16_f16 and
// v_cvt_pkrtz_f16_f32 on this device that the f16-split matrix filter's error bound depends on:
//  (1) does the conversion produce binary16 subnormals (or flush them)?   (2) does the MFMA honour subnormal inputs?
//  (3) the A / B / D lane layout (A = I-like probe with an asymmetric B)   (4) accumulation: is a sum of 16 products
//  whose exact value needs more than 24 bits rounded once (fused) or per product?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __fp16 v2h __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void probe(float* out, uint32_t* outU)
{
    const int lane = threadIdx.x;
    // (1) conversion of 2^-20 and of 3 * 2^-24
    v2h h = __builtin_amdgcn_cvt_pkrtz(0x1p-20f, 0x1.8p-23f);
    uint32_t hu;
    __builtin_memcpy(&hu, &h, 4);
    if (lane == 0) outU[0] = hu;
    // (2) A[i][k=0] = 1 for all rows (lanes 0..31 hold k = 0..7), B[k=0][j] = subnormal 2^-20 (0x0010)
    uint16_t a16[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b16[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 32) { a16[0] = 0x3c00; b16[0] = 0x0010; }
    v8h A, B;
    __builtin_memcpy(&A, a16, 16);
    __builtin_memcpy(&B, b16, 16);
    v16f c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
    // subnormal on the A side too
    if (lane < 32) { a16[0] = 0x0010; b16[0] = 0x3c00; }
    __builtin_memcpy(&A, a16, 16);
    __builtin_memcpy(&B, b16, 16);
    v16f c2 = {0};
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c2, 0, 0, 0);
    if (lane == 0) out[1] = c2[0];
    // (3) layout: A[i][k] = (i == k) ? 1 : 0 for k < 16 (rows 0..15), B[k][j] = k * 32 + j  -> D[i][j] = B[i][j] for i < 16
    for (int t = 0; t < 8; ++t) {
        const int k = 8 * (lane / 32) + t, i = lane % 32, j = lane % 32;
        _Float16 av = (i == k) ? (_Float16)1.0f : (_Float16)0.0f;
        _Float16 bv = (_Float16)(float)(k * 32 + j);
        __builtin_memcpy(&a16[t], &av, 2);
        __builtin_memcpy(&b16[t], &bv, 2);
    }
    __builtin_memcpy(&A, a16, 16);
    __builtin_memcpy(&B, b16, 16);
    v16f c3 = {0};
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c3, 0, 0, 0);
    // expected: register r of lane l = row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32
    int bad = 0;
    for (int r = 0; r < 16; ++r) {
        const int row = 8 * (r / 4) + 4 * (lane / 32) + r % 4, col = lane % 32;
        const float want = row < 16 ? (float)(row * 32 + col) : 0.0f;
        if (c3[r] != want) bad++;
    }
    outU[1 + lane] = (uint32_t)bad;
    // (4) accumulation: 16 products 1 * x_k with x = {2048, 1, 1, ..., 1} minus: exact sum 2048 + 15 needs 12 bits: fine in f32;
    //     use a sum that needs > 24 bits: x0 = 2^12 (4096), others (1 + 2^-10) each: exact = 4096 + 15 + 15 * 2^-10 = 4111.0146484375
    //     (f32 ulp at 4096 is 2^-11: 15 * 2^-10 is representable; so use smaller: others = 2^-13 * (1 + 2^-10)...)
    for (int t = 0; t < 8; ++t) {
        const int k = 8 * (lane / 32) + t;
        _Float16 av = (_Float16)1.0f;
        _Float16 bv = k == 0 ? (_Float16)4096.0f : (_Float16)(0x1p-13f * (1.0f + 0x1p-10f)); // 15 terms of 2^-13 + 2^-23 each: sum 15 * 2^-13 + 15 * 2^-23
        __builtin_memcpy(&a16[t], &av, 2);
        __builtin_memcpy(&b16[t], &bv, 2);
    }
    __builtin_memcpy(&A, a16, 16);
    __builtin_memcpy(&B, b16, 16);
    v16f c4 = {0};
    c4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c4, 0, 0, 0);
    if (lane == 0) out[2] = c4[0]; // ulp(4096) = 2^-11: each addend 2^-13 is a quarter ulp: per-product RNE keeps 4096; a fused sum gives 4096 + 15 * 2^-13 -> 4096.0018 -> rounds to 4096 + 2^-9 (4 ulp)
}
int main()
{
    float* out; uint32_t* outU;
    hipMalloc(&out, 64 * 4); hipMalloc(&outU, 80 * 4);
    hipMemset(out, 0, 256); hipMemset(outU, 0, 320);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, outU);
    float h[64]; uint32_t hu[80];
    hipMemcpy(h, out, 256, hipMemcpyDeviceToHost); hipMemcpy(hu, outU, 320, hipMemcpyDeviceToHost);
    printf("(1) cvt_pkrtz(2^-20, 3*2^-24) = %08x  (subnormals kept: 00030010; flushed: 00000000)\n", hu[0]);
    printf("(2) 1 * subnormal(2^-20) through the MFMA, B side: %g  A side: %g   (honoured: 9.53674e-07)\n", h[0], h[1]);
    int bad = 0; for (int l = 0; l < 64; ++l) bad += (int)hu[1 + l];
    printf("(3) layout mismatches (A row = lane %% 32, k = 8 (lane / 32) + t; B col = lane %% 32, same k; D row = 8 (r / 4) + 4 (lane / 32) + r %% 4): %d\n", bad);
    printf("(4) 4096 + 15 x (2^-13 + 2^-23) accumulated by one MFMA: %.10f  (rounded per product: 4096.0000000000; exact sum 4096.0018328; f32(exact) = 4096.0019531250)\n", h[2]);
    return 0;
}
