// Probe: how v_mfma_f32_16x16x32_f16 treats fp16 subnormal inputs and how it rounds the 32-term sum it adds to C.
//   hipcc -O2 --offload-arch=gfx950 scripts/f16_mfma_probe.hip -o build/f16_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using f4 = __attribute__((ext_vector_type(4))) float;
__global__ void k(const float* A, const float* B, const float* C, float* D) {   // A[16][32], B[32][16], C/D[16][16]
    const int l = threadIdx.x, n = l & 15, g = l >> 4;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[n * 32 + 8 * g + j]; b[j] = (_Float16)B[(8 * g + j) * 16 + n]; }
    f4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * g + r) * 16 + n];
    f4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + n] = d[r];
}
int main() {
    std::vector<float> A(512, 0.f), B(512, 0.f), C(256, 0.f), D(256);
    // row 0: subnormal fp16 a (2^-20) times 2^10 -> 2^-10 if denormal inputs are honoured
    A[0 * 32 + 0] = ldexpf(1.f, -20); B[0 * 16 + 0] = 1024.f;
    // row 1: 1 + 31 products of 2^-25: exact sum then RN -> 1 + 8 ulp; RZ -> 1 + 7 ulp; term-by-term fp32 adds -> 1
    A[1 * 32 + 0] = 1.f; B[0 * 16 + 1] = 1.f;
    for (int kk = 1; kk < 32; ++kk) { A[1 * 32 + kk] = ldexpf(1.f, -12); B[kk * 16 + 1] = ldexpf(1.f, -13); }
    // row 2: C = 1, 32 products of 2^-25 (sum 2^-20 = 8 ulp of 1): is the K-sum formed before it meets C?
    for (int kk = 0; kk < 32; ++kk) { A[2 * 32 + kk] = ldexpf(1.f, -12); B[kk * 16 + 2] = ldexpf(1.f, -13); }
    C[2 * 16 + 2] = 1.f;
    // row 3: products with 22 significant bits: (1 + 2^-10)(1 + 2^-10) = 1 + 2^-9 + 2^-20 exactly representable in fp32
    A[3 * 32 + 0] = 1.f + ldexpf(1.f, -10); B[0 * 16 + 3] = 1.f + ldexpf(1.f, -10);
    // row 4: sum needing more than 24 bits: 1.0 + (1+2^-10)^2 * 2^-12 -> exact = 1 + 2^-12 + 2^-21 + 2^-32
    A[4 * 32 + 0] = 1.f; B[0 * 16 + 4] = 1.f;
    A[4 * 32 + 1] = (1.f + ldexpf(1.f, -10)) * ldexpf(1.f, -6); B[1 * 16 + 4] = (1.f + ldexpf(1.f, -10)) * ldexpf(1.f, -6);
    // row 5: RN vs RZ on the final result: C = 1, one product 3 * 2^-25 (0.75 ulp of 1... ulp(1) = 2^-23): RN -> 1 + 1 ulp, RZ -> 1
    A[5 * 32 + 0] = 3.f * ldexpf(1.f, -12); B[0 * 16 + 5] = ldexpf(1.f, -13); C[5 * 16 + 5] = 1.f;
    // row 6: negative side: C = -1, same product -> RN -1 + ... ; C = 1, product -(3 * 2^-25): RN -> 1 - 1ulp', RZ (toward zero) -> 1 - ...
    A[6 * 32 + 0] = -3.f * ldexpf(1.f, -12); B[0 * 16 + 6] = ldexpf(1.f, -13); C[6 * 16 + 6] = 1.f;
    // row 7: fp16 overflow of an input: 70000 -> inf?
    A[7 * 32 + 0] = 70000.f; B[0 * 16 + 7] = 1.f;
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    const float ulp = ldexpf(1.f, -23);
    printf("row0 subnormal input: got %g (2^-10 = %g honoured, 0 = flushed)\n", D[0 * 16 + 0], ldexpf(1.f, -10));
    printf("row1 1 + 31 x 2^-25: (got - 1)/ulp = %g  (8: exact sum then RN; 7: RZ; 0: term-by-term fp32)\n", (D[1 * 16 + 1] - 1.f) / ulp);
    printf("row2 C=1 + 32 x 2^-25: (got - 1)/ulp = %g  (8 expected if the K-sum is formed exactly)\n", (D[2 * 16 + 2] - 1.f) / ulp);
    printf("row3 22-bit product: got - (1 + 2^-9 + 2^-20) = %g\n", D[3 * 16 + 3] - (1.f + ldexpf(1.f, -9) + ldexpf(1.f, -20)));
    printf("row4 1 + (1+2^-10)^2 2^-12: (got - 1 - 2^-12)/ulp = %g  (exact 4 + 2^-9 -> RN 4)\n", (D[4 * 16 + 4] - 1.f - ldexpf(1.f, -12)) / ulp);
    printf("row5 C=1 + 0.75 ulp: (got - 1)/ulp = %g  (1: RN, 0: RZ)\n", (D[5 * 16 + 5] - 1.f) / ulp);
    printf("row6 C=1 - 0.75 ulp(1): (got - 1)/ulp(0.5 side = 2^-24) = %g  (-2 + ... RN -> -1.5 ulp_lo?)\n", (D[6 * 16 + 6] - 1.f) / ldexpf(1.f, -24));
    printf("row7 70000 as fp16 input: got %g\n", D[7 * 16 + 7]);
    return 0;
}
