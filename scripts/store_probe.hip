// Probe (not part of the library): HBM write rate of the stash store pattern.  A wave writes 16 KiB tile blocks [128 features][32 samples]
// either as 64 dword stores (lane (j, h): sample j, features (r&3) + 8(r>>2) + 4h + 32 nb - the accumulator layout of store_tl) or as 16
// dwordx4 stores (lane l: feature 8q + l/8, samples 4(l%8)..+3 - what a transpose through LDS would allow).
//   hipcc -O3 --offload-arch=gfx950 scripts/store_probe.hip -o build/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int kMode>
__global__ __launch_bounds__(512) void store_kernel(float* __restrict__ dst, long n_tiles, int slots) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const long wave = (long)blockIdx.x * 8 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 8;
    for (long tile = wave; tile < n_tiles; tile += n_waves)
        for (int s = 0; s < slots; ++s) {
            float* base = dst + ((long)s * n_tiles + tile) * 4096;
            if (kMode == 0) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) base[(32 * nb + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = (float)(tile + r);
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const f32x4 v = {(float)tile, (float)q, 1.0f, 2.0f};
                    reinterpret_cast<f32x4*>(base)[64 * q + lane] = v;
                }
            }
        }
}

int main() {
    const long n_tiles = 16384;
    const int slots = 13;
    float* d;
    if (hipMalloc(&d, (size_t)slots * n_tiles * 16384) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(256), dim3(512), 0, 0, d, n_tiles, slots);
            else hipLaunchKernelGGL(store_kernel<1>, dim3(256), dim3(512), 0, 0, d, n_tiles, slots);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us for %.2f GB = %.2f TB/s\n", mode ? "16 x dwordx4 per block" : "64 x dword per block  ", ms * 1e3,
                   slots * n_tiles * 16384.0 / 1e9, slots * n_tiles * 16384.0 / ms / 1e9);
        }
    return 0;
}
