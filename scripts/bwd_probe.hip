// Probe (not part of the library): the Dense-layer backward kernel of train_ops.hip (dense_bwd_split8_kernel) alone on random
// tiles, timed with HIP events, with per-wave cycle totals of its phases (-DMVT_STAMP=1).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DMVT_STAMP=1 -Iinclude -Ithesis_clip_nerf_amd/csrc scripts/bwd_probe.hip -o /tmp/bwd_probe
#include "../thesis_clip_nerf_amd/csrc/train_ops.hip"

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void fill_kernel(float* p, long n, unsigned seed) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f);
    }
}

int main(int argc, char** argv) {
    const long n_tiles = argc > 1 ? atol(argv[1]) : 16384;
    const int wgs = argc > 2 ? atoi(argv[2]) : 512;
    float *g, *a, *r, *o, *w, *dW;
    const long tile_floats = 4096;
    CK(hipMalloc(&g, n_tiles * tile_floats * 4));
    CK(hipMalloc(&a, n_tiles * tile_floats * 4));
    CK(hipMalloc(&r, n_tiles * tile_floats * 4));
    CK(hipMalloc(&o, n_tiles * tile_floats * 4));
    CK(hipMalloc(&w, 16384 * 4));
    CK(hipMalloc(&dW, (16384 + 128) * 4));
    CK(hipMemset(dW, 0, (16384 + 128) * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, g, n_tiles * tile_floats, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, a, n_tiles * tile_floats, 2u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, r, n_tiles * tile_floats, 3u);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, w, 16384L, 4u);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int resid = 1; resid >= 0; --resid) {
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            CK(mvnerf::launch_dense_bwd_fused(g, a, w, resid ? r : nullptr, o, n_tiles, dW, dW + 16384, wgs, nullptr, 0));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)n_tiles * tile_floats * 4 * (resid ? 4 : 3);
            printf("tiles %ld wgs %d resid %d: %.1f us  (%.2f TB/s; %.2f us per tile per workgroup)\n", n_tiles, wgs, resid, ms * 1e3,
                   bytes / ms / 1e9, ms * 1e3 / ((double)n_tiles / wgs));
        }
#if MVT_STAMP
        // slots: 0 barrier-1 wait, 1 DMA issue + cut, 2 barrier-2 wait, 3 MFMA section, 4 vmcnt wait, 5 epilogue + loop, 6 whole kernel,
        // 7 role (0 = Z waves: dL/da, 1 = W waves: dW)
        const int n_wg = (int)(n_tiles < wgs / 2 ? n_tiles : wgs / 2), waves = n_wg * 8;
        std::vector<unsigned long long> st(256 * 8 * 8);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(mvnerf::g_bwd_stamp), st.size() * 8));
        const double tiles_per_wave = (double)n_tiles / n_wg;
        for (int role = 0; role < 2; ++role) {
            double s2[7] = {};
            int n = 0;
            for (int v = 0; v < waves; ++v)
                if ((int)st[v * 8 + 7] == role) {
                    ++n;
                    for (int q = 0; q < 7; ++q) s2[q] += (double)st[v * 8 + q];
                }
            if (!n) continue;
            printf("  %s waves (%d), cycles per tile: barrier-1 wait %.0f, DMA issue + cut %.0f, barrier-2 wait %.0f, MFMA section %.0f, vmcnt wait %.0f, "
                   "epilogue + loop %.0f, whole %.0f\n", role ? "W" : "Z", n, s2[0] / n / tiles_per_wave, s2[1] / n / tiles_per_wave,
                   s2[2] / n / tiles_per_wave, s2[3] / n / tiles_per_wave, s2[4] / n / tiles_per_wave, s2[5] / n / tiles_per_wave,
                   s2[6] / n / tiles_per_wave);
        }
#endif
    }
    return 0;
}
