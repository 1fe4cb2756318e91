// Probe (not part of the library): how many independent v_add_f32 hide behind one MFMA of a dependent chain, one wave per SIMD.
// For v_mfma_f32_32x32x2_f32 (16 passes) and v_mfma_f32_32x32x16_bf16 (8 passes); cycles per MFMA from s_memtime inside the wave.
//   hipcc -O3 --offload-arch=gfx950 scripts/filler_probe.hip -o build/filler_probe && build/filler_probe
#include <hip/hip_runtime.h>

#include <cstdio>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int kFill, bool kBf16, int kChains>
__global__ __launch_bounds__(256) void probe_kernel(unsigned long long* out, float* sink, int iters) {
    f32x16 acc[kChains];
    for (int c = 0; c < kChains; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    float f[16];
    for (int q = 0; q < 16; ++q) f[q] = (float)(threadIdx.x + q);
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    bf16x8 a8, b8;
    for (int q = 0; q < 8; ++q) {
        a8[q] = (__bf16)a;
        b8[q] = (__bf16)b;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = u % kChains;
            if (kBf16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[c], 0, 0, 0);
            else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < kFill; ++q) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(f[q % 16]) : "v"(b));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int c = 0; c < kChains; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int q = 0; q < 16; ++q) s += f[q];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int kFill, bool kBf16, int kChains>
void run(unsigned long long* d_out, float* d_sink) {
    const int iters = 2000, wgs = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe_kernel<kFill, kBf16, kChains>), dim3(wgs), dim3(256), 0, 0, d_out, d_sink, iters);
    hipDeviceSynchronize();
    static unsigned long long h[1024];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += (double)h[i];
    printf("%s chains %d fillers %2d: %.1f cycles per MFMA\n", kBf16 ? "32x32x16 bf16" : "32x32x2 f32  ", kChains, kFill, s / 1024 / (iters * 8.0));
}

int main() {
    unsigned long long* d_out;
    float* d_sink;
    hipMalloc(&d_out, 1024 * 8);
    hipMalloc(&d_sink, 4);
    run<0, false, 1>(d_out, d_sink);
    run<4, false, 1>(d_out, d_sink);
    run<8, false, 1>(d_out, d_sink);
    run<12, false, 1>(d_out, d_sink);
    run<16, false, 1>(d_out, d_sink);
    run<0, false, 2>(d_out, d_sink);
    run<8, false, 2>(d_out, d_sink);
    run<12, false, 2>(d_out, d_sink);
    run<0, true, 1>(d_out, d_sink);
    run<3, true, 1>(d_out, d_sink);
    run<5, true, 1>(d_out, d_sink);
    run<8, true, 1>(d_out, d_sink);
    run<0, true, 2>(d_out, d_sink);
    run<5, true, 2>(d_out, d_sink);
    run<8, true, 2>(d_out, d_sink);
    return 0;
}
