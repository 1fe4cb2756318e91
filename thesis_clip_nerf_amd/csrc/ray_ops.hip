// Ray generation, stratified depths, alpha compositing and hierarchical resampling for gfx950.
// These are HBM-bound per-ray scans: one wavefront per ray, lane <-> sample, wave shuffles for the
// transmittance scan and reductions, wave-private LDS for the CDF search and the sort-merge.
// Reference lines are cited per kernel; see include/mvnerf_hip.h for the public contracts.
#include <hip/hip_runtime.h>

#include "mvnerf_kernels.h"
#include "mvnerf_math.h"

namespace mvnerf {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kRaysPerWG = 4;     // one wavefront per ray, 4 waves per workgroup

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---- a1/a2: get_rays / get_specific_rays (nerf_utils.py:15-35), float64 math (Q2), no +0.5 (Q1) ----
struct RayGenParams {
    double m[9];
    double origin[3];
};

__global__ void get_rays_kernel(RayGenParams rp, const float* __restrict__ u, const float* __restrict__ v, int n,
                                int width, int normalize, float* __restrict__ rays_o, float* __restrict__ rays_d,
                                double* __restrict__ rays_d64) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double pu, pv;
    if (u) {
        pu = (double)u[i];
        pv = (double)v[i];
    } else {                                 // meshgrid(indexing='xy') flattened row-major: n = v*W + u
        pu = (double)(float)(i % width);
        pv = (double)(float)(i / width);
    }
    double d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = (rp.m[3 * r] * pu + rp.m[3 * r + 1] * pv) + rp.m[3 * r + 2];
    if (normalize) {
        const double nrm = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
        for (int r = 0; r < 3; ++r) d[r] = d[r] / nrm;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        rays_d[3 * i + r] = (float)d[r];
        rays_o[3 * i + r] = (float)rp.origin[r];
        if (rays_d64) rays_d64[3 * i + r] = d[r];
    }
}

hipError_t launch_get_rays(const double* m9, const double* origin3, const float* u, const float* v, int n_rays,
                           int width, int normalize, float* rays_o, float* rays_d, double* rays_d64,
                           hipStream_t stream) {
    RayGenParams rp;
    for (int i = 0; i < 9; ++i) rp.m[i] = m9[i];
    for (int i = 0; i < 3; ++i) rp.origin[i] = origin3[i];
    const int threads = 256;
    hipLaunchKernelGGL(get_rays_kernel, dim3((n_rays + threads - 1) / threads), dim3(threads), 0, stream, rp, u, v,
                       n_rays, width, normalize, rays_o, rays_d, rays_d64);
    return hipGetLastError();
}

// ---- a4: stratified depths (nerf_utils.py:49-58) ----
__global__ void stratified_kernel(const float* __restrict__ u, long n, int n_samples, double near_, double far_,
                                  float* __restrict__ z) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    z[i] = stratified_z(near_, far_, n_samples, (int)(i % n_samples), u[i]);
}

hipError_t launch_stratified(const float* u, long n, int n_samples, double near_, double far_, float* z,
                             hipStream_t stream) {
    const int threads = 256;
    hipLaunchKernelGGL(stratified_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0, stream, u,
                       n, n_samples, near_, far_, z);
    return hipGetLastError();
}

// ---- a11/a12: volumetric_render (model_v0.py:89-100) ----
// Lane i owns the P = S/64 consecutive samples [i*P, i*P+P).  Transmittance = exclusive product scan
// of (1 - alpha + 1e-10): serial inside a lane, Hillis-Steele over lanes.
template <int P>
__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ z, const float* __restrict__ rgbs,
                                                        int n_rays, float* __restrict__ rgb, float* __restrict__ depth,
                                                        float* __restrict__ weights) {
    constexpr int S = 64 * P;
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * kRaysPerWG + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float* zr = z + (long)ray * S;
    const f32x4* cr = reinterpret_cast<const f32x4*>(rgbs) + (long)ray * S;

    float zl[P + 1];
    f32x4 c[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        zl[q] = zr[lane * P + q];
        c[q] = cr[lane * P + q];
    }
    zl[P] = __shfl_down(zl[0], 1);                   // first depth of the next lane
    float dist[P];
#pragma unroll
    for (int q = 0; q < P; ++q) dist[q] = zl[q + 1] - zl[q];
    // Q6: the last interval repeats the one before it (dists[..., -1:] appended, model_v0.py:92)
    if (P == 1) {
        const float prev = __shfl_up(dist[0], 1);
        if (lane == 63) dist[0] = prev;
    } else {
        if (lane == 63) dist[P - 1] = dist[P - 2];
    }
    float alpha[P], t[P];
    float prod = 1.0f;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        alpha[q] = sigma_to_alpha(c[q][3], dist[q]);
        t[q] = (1.0f - alpha[q]) + 1e-10f;
        prod = prod * t[q];
    }
    float incl = prod;                               // inclusive product scan over lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float up = __shfl_up(incl, off);
        if (lane >= off) incl = incl * up;
    }
    float trans = __shfl_up(incl, 1);
    if (lane == 0) trans = 1.0f;
    float sr = 0.0f, sg = 0.0f, sb = 0.0f, sd = 0.0f;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const float w = alpha[q] * trans;
        trans = trans * t[q];
        sr = sr + w * c[q][0];
        sg = sg + w * c[q][1];
        sb = sb + w * c[q][2];
        sd = sd + w * zl[q];
        if (weights) weights[(long)ray * S + lane * P + q] = w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sr = sr + __shfl_xor(sr, off);
        sg = sg + __shfl_xor(sg, off);
        sb = sb + __shfl_xor(sb, off);
        sd = sd + __shfl_xor(sd, off);
    }
    if (lane == 0) {
        rgb[3 * ray + 0] = sr;
        rgb[3 * ray + 1] = sg;
        rgb[3 * ray + 2] = sb;
        depth[ray] = sd;
    }
}

hipError_t launch_composite(const float* z, const float* rgbs, int n_rays, int S, float* rgb, float* depth,
                            float* weights, hipStream_t stream) {
    const dim3 grid((n_rays + kRaysPerWG - 1) / kRaysPerWG), block(256);
    switch (S / 64) {
        case 1: hipLaunchKernelGGL(composite_kernel<1>, grid, block, 0, stream, z, rgbs, n_rays, rgb, depth, weights); break;
        case 2: hipLaunchKernelGGL(composite_kernel<2>, grid, block, 0, stream, z, rgbs, n_rays, rgb, depth, weights); break;
        case 3: hipLaunchKernelGGL(composite_kernel<3>, grid, block, 0, stream, z, rgbs, n_rays, rgb, depth, weights); break;
        case 4: hipLaunchKernelGGL(composite_kernel<4>, grid, block, 0, stream, z, rgbs, n_rays, rgb, depth, weights); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- a13/a14: sample_pdf (nerf_utils.py:143-176) + sort-merge (model_v0.py:150-156), S = 64 ----
// The sums that decide the integer outputs (w_sum, cdf) are strictly sequential in fp32, as in the
// oracle: every lane runs the same 62-step loop over wave-private LDS (broadcast reads), so all
// lanes hold bit-identical values.  `above` is the reference's tf.scan: a count of cdf_j <= u.
// Core of sample_pdf for one ray on one wavefront.  On entry bins[0..62] and pdf[0..61] (= weights +
// 1e-5) are in wave-private LDS and fenced; lane l draws the sample for uniform u.
__device__ __forceinline__ float sample_pdf_core(const float* bins, float* pdf, float* cdf, int lane, float u,
                                                 int q7_mode, int& above, int& below) {
    constexpr int NB = 63, NW = 62;
    float wsum = 0.0f;
    for (int k = 0; k < NW; ++k) wsum = wsum + pdf[k];               // sequential
    if (fabsf(wsum) == 0.0f) wsum = 1.0f;                            // :146
    lds_fence();
    if (lane < NW) pdf[lane] = pdf[lane] / wsum;
    lds_fence();
    float run = 0.0f, mycdf = 0.0f;
    for (int k = 0; k < NW; ++k) {                                   // cdf_j = sum_{k<j} pdf_k, sequential
        run = run + pdf[k];
        if (lane == k + 1) mycdf = run;
    }
    if (lane < NB) cdf[lane] = mycdf;                                // cdf_0 = 0 (:149)
    lds_fence();

    above = 0;
    for (int jj = 0; jj < NB; ++jj) above += (u >= cdf[jj]) ? 1 : 0;  // tf.scan(greater_equal) (:156-160)
    below = above - 1;
    below = below < 0 ? 0 : (below > NB - 1 ? NB - 1 : below);       // :162
    // Q7: `above` may be NB (one past the end).  TF-GPU gather_nd yields 0 there; optional clamp.
    const bool oob = above >= NB;
    const int ia = oob ? NB - 1 : above;
    float cdf_a = cdf[ia], bins_a = bins[ia];
    if (oob && q7_mode == 0) {
        cdf_a = 0.0f;
        bins_a = 0.0f;
    }
    const float cdf_b = cdf[below], bins_b = bins[below];
    float den = cdf_a - cdf_b;
    if (den < 1e-5f) den = 1.0f;
    const float t = (u - cdf_b) / den;
    return bins_b + t * (bins_a - bins_b);
}

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ z, const float* __restrict__ weights,
                                                       const float* __restrict__ u_fine, int n_rays, int q7_mode,
                                                       float* __restrict__ z_all, float* __restrict__ z_fine,
                                                       int32_t* __restrict__ above_out,
                                                       int32_t* __restrict__ below_out,
                                                       int32_t* __restrict__ rank_out) {
    constexpr int S = 64, NB = 63, NW = 62;
    __shared__ float lds[kRaysPerWG][7 * 64];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * kRaysPerWG + wv;
    if (ray >= n_rays) return;
    float* bins = lds[wv];            // [64]  z_mid, 63 used
    float* pdf = bins + 64;           // [64]  stable weights then pdf, 62 used
    float* cdf = pdf + 64;            // [64]  63 used
    float* vals = cdf + 64;           // [128] coarse | fine depths
    float* sorted = vals + 128;       // [128]

    const long base = (long)ray * S;
    const float zi = z[base + lane];
    const float wi = weights[base + lane];
    const float znext = __shfl_down(zi, 1);
    if (lane < NB) bins[lane] = 0.5f * (znext + zi);                 // model_v0.py:150-151
    if (lane >= 1 && lane <= NW) pdf[lane - 1] = wi + 1e-5f;         // probs = weights[1:-1] (+1e-5, :144)
    vals[lane] = zi;
    lds_fence();
    int above, below;
    const float zf = sample_pdf_core(bins, pdf, cdf, lane, u_fine[base + lane], q7_mode, above, below);
    if (z_fine) z_fine[base + lane] = zf;
    if (above_out) above_out[base + lane] = above;
    if (below_out) below_out[base + lane] = below;

    // ascending sort of the 128 depths (tf.sort, model_v0.py:156) by rank; ties broken by index
    vals[64 + lane] = zf;
    lds_fence();
    int r0 = 0, r1 = 0;
    for (int k = 0; k < 128; ++k) {
        const float vk = vals[k];
        r0 += (vk < zi || (vk == zi && k < lane)) ? 1 : 0;
        r1 += (vk < zf || (vk == zf && k < 64 + lane)) ? 1 : 0;
    }
    sorted[r0] = zi;
    sorted[r1] = zf;
    if (rank_out) rank_out[base + lane] = r1;            // position of importance sample `lane` inside all_zs
    lds_fence();
    z_all[(long)ray * 128 + lane] = sorted[lane];
    z_all[(long)ray * 128 + 64 + lane] = sorted[64 + lane];
}

// sample_pdf (nerf_utils.py:143-176) on explicit bins (n,63), weights (n,62), u (n,64)
__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ bins_in,
                                                         const float* __restrict__ weights, const float* __restrict__ u,
                                                         int n_rays, int q7_mode, float* __restrict__ samples,
                                                         int32_t* __restrict__ above_out,
                                                         int32_t* __restrict__ below_out) {
    __shared__ float lds[kRaysPerWG][3 * 64];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * kRaysPerWG + wv;
    if (ray >= n_rays) return;
    float* bins = lds[wv];
    float* pdf = bins + 64;
    float* cdf = pdf + 64;
    if (lane < 63) bins[lane] = bins_in[(long)ray * 63 + lane];
    if (lane < 62) pdf[lane] = weights[(long)ray * 62 + lane] + 1e-5f;
    lds_fence();
    int above, below;
    const float zf = sample_pdf_core(bins, pdf, cdf, lane, u[(long)ray * 64 + lane], q7_mode, above, below);
    samples[(long)ray * 64 + lane] = zf;
    if (above_out) above_out[(long)ray * 64 + lane] = above;
    if (below_out) below_out[(long)ray * 64 + lane] = below;
}

hipError_t launch_sample_pdf(const float* bins, const float* weights, const float* u, int n_rays, int q7_mode,
                             float* samples, int32_t* above, int32_t* below, hipStream_t stream) {
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((n_rays + kRaysPerWG - 1) / kRaysPerWG), dim3(256), 0, stream, bins,
                       weights, u, n_rays, q7_mode, samples, above, below);
    return hipGetLastError();
}

hipError_t launch_resample(const float* z, const float* weights, const float* u_fine, int n_rays, int q7_mode,
                           float* z_all, float* z_fine, int32_t* above, int32_t* below, int32_t* fine_rank,
                           hipStream_t stream) {
    hipLaunchKernelGGL(resample_kernel, dim3((n_rays + kRaysPerWG - 1) / kRaysPerWG), dim3(256), 0, stream, z,
                       weights, u_fine, n_rays, q7_mode, z_all, z_fine, above, below, fine_rank);
    return hipGetLastError();
}

}  // namespace mvnerf
