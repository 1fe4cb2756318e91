// Probe (not part of the library): fp32-grade GEMM on the bf16 MFMA by splitting every fp32 operand into three bf16
// pieces (x = x1 + x2 + x3, 8 significand bits each) and issuing the six products of order >= 2^-16:
//   a.b ~= a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1)          (dropped: a2b3, a3b2, a3b3 <= 2^-24 relative)
// Accuracy of one 32x32 output block over K = 128 against a float64 reference, next to the fp32 MFMA and to the 1- and
// 3-product forms; and the cycles per MFMA of the bare x6 stream with the activation split on the vector ALU.
//   hipcc -O3 --offload-arch=gfx950 scripts/x6_probe.hip -o /tmp/x6_probe && /tmp/x6_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// truncation split of 8 floats: piece = upper 16 bits, remainder exact
__device__ __forceinline__ void split3(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
    float r1[8], r2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned u = __builtin_bit_cast(unsigned, x[q]);
        r1[q] = x[q] - __builtin_bit_cast(float, u & 0xffff0000u);
        const unsigned u1 = __builtin_bit_cast(unsigned, r1[q]);
        r2[q] = r1[q] - __builtin_bit_cast(float, u1 & 0xffff0000u);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // v_perm_b32: bytes [3:2] of the odd element | bytes [3:2] of the even element
        p1[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x[2 * q + 1]), __builtin_bit_cast(unsigned, x[2 * q]), 0x07060302u);
        p2[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, r1[2 * q + 1]), __builtin_bit_cast(unsigned, r1[2 * q]), 0x07060302u);
        p3[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, r2[2 * q + 1]), __builtin_bit_cast(unsigned, r2[2 * q]), 0x07060302u);
    }
}

// A, B: [K][32] row-major fp32.  out[mode][32][32], mode 0 = fp32 MFMA, 1 = bf16 x1, 2 = x3, 3 = x6
__global__ void accuracy_kernel(const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ out) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc32 = {}, acc1 = {}, acc3 = {}, acc6 = {};
    for (int k = 0; k < K; k += 2) acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(k + h) * 32 + i], B[(k + h) * 32 + i], acc32, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        float a[8], b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            a[q] = A[(k0 + 8 * h + q) * 32 + i];
            b[q] = B[(k0 + 8 * h + q) * 32 + i];
        }
        u32x4 a1, a2, a3, b1, b2, b3;
        split3(a, a1, a2, a3);
        split3(b, b1, b2, b3);
#define BF(v) __builtin_bit_cast(bf16x8, v)
        acc1 = mfma16(BF(a1), BF(b1), acc1);
        acc3 = mfma16(BF(a2), BF(b1), acc3);
        acc3 = mfma16(BF(a1), BF(b2), acc3);
        acc3 = mfma16(BF(a1), BF(b1), acc3);
        acc6 = mfma16(BF(a3), BF(b1), acc6);
        acc6 = mfma16(BF(a2), BF(b2), acc6);
        acc6 = mfma16(BF(a1), BF(b3), acc6);
        acc6 = mfma16(BF(a2), BF(b1), acc6);
        acc6 = mfma16(BF(a1), BF(b2), acc6);
        acc6 = mfma16(BF(a1), BF(b1), acc6);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        out[0 * 1024 + row * 32 + i] = acc32[r];
        out[1 * 1024 + row * 32 + i] = acc1[r];
        out[2 * 1024 + row * 32 + i] = acc3[r];
        out[3 * 1024 + row * 32 + i] = acc6[r];
    }
}

// bare x6 stream: per k-step 4 output blocks x 6 MFMAs, B split on the VALU (kSplit) from the previous accumulators
template <bool kSplit>
__global__ __launch_bounds__(512, 2) void stream_kernel(const u32x4* __restrict__ w, int iters, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4] = {}, in[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) in[nb][r] = (float)(lane + r + nb) * 1e-3f;
    u32x4 a[4][3];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[nb][q] = w[lane + 64 * (3 * nb + q)];
    for (int it = 0; it < iters; ++it) {
#define KSTEP(ks)                                                                                   \
        {                                                                                           \
            float x[8];                                                                             \
            _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                         \
                const float v_ = in[(ks) >> 1][8 * ((ks) & 1) + q];                                 \
                const int bits = __builtin_bit_cast(int, v_);                                       \
                x[q] = __builtin_bit_cast(float, bits > 0 ? bits : 0);                              \
            }                                                                                       \
            u32x4 b1, b2, b3;                                                                       \
            if (kSplit) split3(x, b1, b2, b3);                                                      \
            else {                                                                                  \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                     \
                    b1[q] = __builtin_bit_cast(unsigned, x[q]);                                     \
                    b2[q] = __builtin_bit_cast(unsigned, x[q + 4]);                                 \
                    b3[q] = b1[q] ^ b2[q];                                                          \
                }                                                                                   \
            }                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) {                                      \
                acc[nb] = mfma16(BF(a[nb][2]), BF(b1), acc[nb]);                                        \
                acc[nb] = mfma16(BF(a[nb][1]), BF(b2), acc[nb]);                                        \
                acc[nb] = mfma16(BF(a[nb][0]), BF(b3), acc[nb]);                                        \
                acc[nb] = mfma16(BF(a[nb][1]), BF(b1), acc[nb]);                                        \
                acc[nb] = mfma16(BF(a[nb][0]), BF(b2), acc[nb]);                                        \
                acc[nb] = mfma16(BF(a[nb][0]), BF(b1), acc[nb]);                                        \
            }                                                                                       \
        }
        KSTEP(0) KSTEP(1) KSTEP(2) KSTEP(3) KSTEP(4) KSTEP(5) KSTEP(6) KSTEP(7)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            in[nb] = acc[nb];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += in[nb][r];
    if (s == 12345.678f) sink[0] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    const int K = 128;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> uw(-0.15f, 0.15f);
    std::normal_distribution<float> nx(0.0f, 1.0f);
    std::vector<float> A(K * 32), B(K * 32);
    for (auto& v : A) v = uw(rng);
    for (auto& v : B) v = std::fmax(nx(rng), 0.0f);
    float *dA, *dB, *dO;
    CK(hipMalloc(&dA, A.size() * 4));
    CK(hipMalloc(&dB, B.size() * 4));
    CK(hipMalloc(&dO, 4 * 1024 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(accuracy_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, dO);
    CK(hipDeviceSynchronize());
    std::vector<float> O(4 * 1024);
    CK(hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
    const char* names[4] = {"fp32 MFMA 32x32x2", "bf16 x1", "bf16 x3", "bf16 x6"};
    for (int m = 0; m < 4; ++m) {
        double maxabs = 0, maxref = 0, sum2 = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)A[k * 32 + i] * (double)B[k * 32 + j];
                const double e = std::fabs((double)O[m * 1024 + i * 32 + j] - ref);
                maxabs = std::fmax(maxabs, e);
                maxref = std::fmax(maxref, std::fabs(ref));
                sum2 += e * e;
            }
        printf("%-18s K=%d: max|err| %.3e  rms %.3e  (max|ref| %.3f)\n", names[m], K, maxabs, std::sqrt(sum2 / 1024), maxref);
    }
    // timing: 256 CUs x 1 workgroup of 8 waves (2 waves / SIMD)
    u32x4* dW;
    CK(hipMalloc(&dW, 64 * 12 * 16));
    CK(hipMemset(dW, 0x3c, 64 * 12 * 16));
    const int iters = 200;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int split = 0; split < 2; ++split) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (split) hipLaunchKernelGGL(stream_kernel<true>, dim3(256), dim3(512), 0, 0, dW, iters, dO);
            else hipLaunchKernelGGL(stream_kernel<false>, dim3(256), dim3(512), 0, 0, dW, iters, dO);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double mfmas = (double)iters * 8 * 24;                 // per wave
            const double flops = mfmas * 32768.0 * 256 * 8;
            printf("stream split=%d: %.3f ms, %.1f TFLOP/s bf16 (%.2f of 2.5 PF), %.1f ns per MFMA per SIMD (2 waves)\n", split, ms,
                   flops / ms / 1e9, flops / ms / 1e9 / 2500.0, ms * 1e6 / (mfmas * 2));
        }
    }
    return 0;
}
