// Probe: LDS layout written by global_load_lds_dwordx3 / dwordx4 (which LDS byte does lane i's data land at?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int kBytes>
__global__ void k(const float* src, float* out) {
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.0f;
    __syncthreads();
    if (kBytes == 12)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + threadIdx.x * 3),
                                         (__attribute__((address_space(3))) void*)lds, 12, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + threadIdx.x * 4),
                                         (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(k<12>, dim3(1), dim3(64), 0, 0, d, o);
        else hipLaunchKernelGGL(k<16>, dim3(1), dim3(64), 0, 0, d, o);
        hipDeviceSynchronize();
        std::vector<float> r(1024);
        hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
        printf("size %d: first 24 LDS floats:", mode ? 16 : 12);
        for (int i = 0; i < 24; ++i) printf(" %g", r[i]);
        int last = 0; for (int i = 0; i < 1024; ++i) if (r[i] >= 0) last = i;
        int contiguous = 1; for (int i = 0; i <= last; ++i) if (r[i] != (float)i) contiguous = 0;
        printf("\n  last written float index %d, identity copy: %d\n", last, contiguous);
    }
    return 0;
}
