#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* v, unsigned* out) {
    const float v0 = v[2 * threadIdx.x], v1 = v[2 * threadIdx.x + 1];
    unsigned h, l;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(v0), "s"(0.015625f));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(v1), "s"(0.015625f));
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "s"(-64.0f), "v"(v0));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "s"(-64.0f), "v"(v1));
    out[2 * threadIdx.x] = h;
    out[2 * threadIdx.x + 1] = l;
}
static float h2f(unsigned short b) { _Float16 x; __builtin_memcpy(&x, &b, 2); return (float)x; }
int main() {
    float hv[128]; unsigned ho[128];
    for (int i = 0; i < 128; ++i) hv[i] = i < 64 ? (float)(i * 37 % 101) * 0.0371f + 0.001f * i : ((i & 1) ? -1.0f : 1.0f) * ldexpf(1.0f + 0.013f * i, -(i - 60));
    float* dv; unsigned* dout;
    hipMalloc(&dv, sizeof hv); hipMalloc(&dout, sizeof ho);
    hipMemcpy(dv, hv, sizeof hv, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dv, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        const float v0 = hv[2 * t], v1 = hv[2 * t + 1];
        const _Float16 e0 = (_Float16)(v0 / 64), e1 = (_Float16)(v1 / 64);
        const _Float16 r0 = (_Float16)(v0 - 64.0f * (float)e0), r1 = (_Float16)(v1 - 64.0f * (float)e1);
        const float g0 = h2f(ho[2 * t] & 0xffff), g1 = h2f(ho[2 * t] >> 16), q0 = h2f(ho[2 * t + 1] & 0xffff), q1 = h2f(ho[2 * t + 1] >> 16);
        if (g0 != (float)e0 || g1 != (float)e1 || q0 != (float)r0 || q1 != (float)r1) {
            if (bad < 6) printf("t=%d v=(%g,%g) hi got (%g,%g) want (%g,%g); lo got (%g,%g) want (%g,%g)\n", t, v0, v1, g0, g1, (float)e0, (float)e1, q0, q1, (float)r0, (float)r1);
            ++bad;
        }
    }
    printf("mismatches: %d of 64\n", bad);
    return 0;
}
