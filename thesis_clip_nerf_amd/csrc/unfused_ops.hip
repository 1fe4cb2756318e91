// Stand-alone (unfused) forms of the per-sample operators of the render path, one kernel per
// reference function, built from the same scalar code (mvnerf_math.h) as the fused field kernel.
// They exist so that every reference function on the hot path has an op-level entry point and an
// op-level parity test; the renderer itself uses the fused kernels (field_eval.hip, ray_ops.hip).
// All of them are HBM-bound elementwise / gather kernels: one thread per output row, coalesced.
#include <hip/hip_runtime.h>

#include "mvnerf_kernels.h"
#include "mvnerf_math.h"

namespace mvnerf {

namespace {
constexpr int kThreads = 256;
inline dim3 grid_for(long n) { return dim3((unsigned)((n + kThreads - 1) / kThreads)); }
}  // namespace

// p = o + z*d  (nerf_utils.py:59-60, model_v0.py:157-158)
__global__ void points_on_rays_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                      const float* __restrict__ z, long n, int S, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long ray = i / S;
    const float zz = z[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) out[3 * i + c] = o[3 * ray + c] + zz * d[3 * ray + c];
}

// compute_pixel_in_image_mv (nerf_utils.py:64-81): world (B,N,3) -> pix (B,V,N,2), cam (B,V,N,4)
__global__ void project_points_kernel(const float* __restrict__ world, const float* __restrict__ k4,
                                      const float* __restrict__ einv, int B, int V, long N, float* __restrict__ pix,
                                      float* __restrict__ cam) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;        // over B*V*N
    if (i >= (long)B * V * N) return;
    const long n = i % N;
    const int bv = (int)(i / N);
    const int b = bv / V;
    const float* w = world + 3 * ((long)b * N + n);
    const float* E = einv + 16 * bv;
    float c[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = row_dot4(E, r, w[0], w[1], w[2], 1.0f);
    float px, py;
    pixel_from_cam(k4 + 16 * bv, c, &px, &py);
    pix[2 * i] = px;
    pix[2 * i + 1] = py;
#pragma unroll
    for (int r = 0; r < 4; ++r) cam[4 * i + r] = c[r];
}

// world_to_camera_direction_vector_mv (nerf_utils.py:84-105), Q3: w = 1.  d (B,R,3) -> (B,V,R,3)
__global__ void camera_directions_kernel(const float* __restrict__ d, const float* __restrict__ einv, int B, int V,
                                         int R, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * V * R) return;
    const int r = (int)(i % R);
    const int bv = (int)(i / R);
    const float* dd = d + 3 * ((long)(bv / V) * R + r);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[3 * i + c] = row_dot4(einv + 16 * bv, c, dd[0], dd[1], dd[2], 1.0f);
}

// position_encoding (nerf_utils.py:108-126): x (N,D) -> (N, D*2*n_freq), layout (d, k, {sin,cos})
__global__ void position_encoding_kernel(const float* __restrict__ x, long n_elems, int n_freq, float freq0,
                                         float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;        // over N*D*n_freq
    if (i >= n_elems * n_freq) return;
    const long e = i / n_freq;
    const int k = (int)(i % n_freq);
    const float f = ldexpf(freq0, k);                                    // pos_encoding_freq * 2^k (exact scaling)
    float s, c;
    sincos_f32(x[e] * f, &s, &c);
    out[2 * i] = s;
    out[2 * i + 1] = c;
}

// get_projection_features_mv (nerf_utils.py:277-285) = tfa interpolate_bilinear(indexing='xy') of the
// channel-concatenated grid [images (3) | features (256)].  pix (BV,Q,2) -> out (BV,Q,259), taps optional.
__global__ void bilinear_gather_kernel(const float* __restrict__ images, const float* __restrict__ features,
                                       const float* __restrict__ pix, int BV, long Q, int H, int W,
                                       float* __restrict__ out, int32_t* __restrict__ taps) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);    // one wave per query
    const int lane = threadIdx.x & 63;
    if (row >= (long)BV * Q) return;
    const int bv = (int)(row / Q);
    const Taps t = bilinear_taps(pix[2 * row], pix[2 * row + 1], H, W);
    const int tl = (bv * H + t.y0) * W + t.x0;
    if (taps && lane == 0) {
        taps[4 * row + 0] = tl;
        taps[4 * row + 1] = tl + 1;
        taps[4 * row + 2] = tl + W;
        taps[4 * row + 3] = tl + W + 1;
    }
    float* o = out + 259 * row;
    if (lane < 3) {
        const float* im = images + 3 * (long)tl + lane;
        o[lane] = bilerp(im[0], im[3], im[3 * W], im[3 * W + 3], t.ax, t.ay);
    }
    const float* f = features + 256 * (long)tl;
#pragma unroll
    for (int c = lane; c < 256; c += 64)
        o[3 + c] = bilerp(f[c], f[256 + c], f[256 * (long)W + c], f[256 * (long)W + 256 + c], t.ax, t.ay);
}

// sigma_to_alpha (nerf_utils.py:129-140)
__global__ void sigma_to_alpha_kernel(const float* __restrict__ sigma, const float* __restrict__ dists, long n,
                                      float* __restrict__ alpha) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) alpha[i] = sigma_to_alpha(sigma[i], dists[i]);
}

// RenderReadout (layers.py:392-397) on an explicit embedding: emb (N,128) -> rgbs (N,4).
// Wr/br in Keras layout (kernel[128,4], bias[4]); k-ordered fp32 FMA chain like the MFMA path.
__global__ void readout_kernel(const float* __restrict__ emb, const float* __restrict__ wr, const float* __restrict__ br,
                               long n, float* __restrict__ rgbs) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[4] = {br[0], br[1], br[2], br[3]};
    for (int k = 0; k < kHidden; ++k) {
        const float a = fmaxf(emb[kHidden * i + k], 0.0f);
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = fmaf(a, wr[4 * k + c], o[c]);
    }
    rgbs[4 * i + 0] = sigmoid_f32(o[0]);
    rgbs[4 * i + 1] = sigmoid_f32(o[1]);
    rgbs[4 * i + 2] = sigmoid_f32(o[2]);
    rgbs[4 * i + 3] = softplus_f32(o[3]);
}

// render_view epilogue (model_v0.py:275-281): rgb*255 clipped -> uint8; depth min-max normalised -> uint8
__global__ void depth_minmax_kernel(const float* __restrict__ depth, long n, float* __restrict__ mm) {
    __shared__ float smin[1024 / 64], smax[1024 / 64];
    float lo = INFINITY, hi = -INFINITY;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = depth[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = lo;
        smax[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (unsigned w = 1; w < blockDim.x / 64; ++w) {
            lo = fminf(lo, smin[w]);
            hi = fmaxf(hi, smax[w]);
        }
        mm[0] = lo;
        mm[1] = hi;
    }
}

__device__ __forceinline__ uint8_t to_u8(float v) {       // NumPy astype(uint8) on a value already in [0,255]; NaN -> 0
    return (v >= 0.0f && v <= 255.0f) ? (uint8_t)(int)v : (uint8_t)0;
}

__global__ void finish_view_kernel(const float* __restrict__ rgb, const float* __restrict__ depth,
                                   const float* __restrict__ mm, long n, uint8_t* __restrict__ rgb8,
                                   uint8_t* __restrict__ depth8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb8[3 * i + c] = to_u8(fminf(fmaxf(rgb[3 * i + c] * 255.0f, 0.0f), 255.0f));
    const float norm = (depth[i] - mm[0]) / (mm[1] - mm[0]);
    depth8[i] = to_u8(norm * 255.0f);
}

// ---- launchers --------------------------------------------------------------------------------
hipError_t launch_points_on_rays(const float* o, const float* d, const float* z, long n, int S, float* out, hipStream_t st) {
    hipLaunchKernelGGL(points_on_rays_kernel, grid_for(n), dim3(kThreads), 0, st, o, d, z, n, S, out);
    return hipGetLastError();
}
hipError_t launch_project_points(const float* world, const float* k4, const float* einv, int B, int V, long N,
                                 float* pix, float* cam, hipStream_t st) {
    hipLaunchKernelGGL(project_points_kernel, grid_for((long)B * V * N), dim3(kThreads), 0, st, world, k4, einv, B, V, N, pix, cam);
    return hipGetLastError();
}
hipError_t launch_camera_directions(const float* d, const float* einv, int B, int V, int R, float* out, hipStream_t st) {
    hipLaunchKernelGGL(camera_directions_kernel, grid_for((long)B * V * R), dim3(kThreads), 0, st, d, einv, B, V, R, out);
    return hipGetLastError();
}
hipError_t launch_position_encoding(const float* x, long n_elems, int n_freq, float freq0, float* out, hipStream_t st) {
    hipLaunchKernelGGL(position_encoding_kernel, grid_for(n_elems * n_freq), dim3(kThreads), 0, st, x, n_elems, n_freq, freq0, out);
    return hipGetLastError();
}
hipError_t launch_bilinear_gather(const float* images, const float* features, const float* pix, int BV, long Q, int H,
                                  int W, float* out, int32_t* taps, hipStream_t st) {
    const long rows = (long)BV * Q;
    hipLaunchKernelGGL(bilinear_gather_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, images, features, pix,
                       BV, Q, H, W, out, taps);
    return hipGetLastError();
}
hipError_t launch_sigma_to_alpha(const float* sigma, const float* dists, long n, float* alpha, hipStream_t st) {
    hipLaunchKernelGGL(sigma_to_alpha_kernel, grid_for(n), dim3(kThreads), 0, st, sigma, dists, n, alpha);
    return hipGetLastError();
}
hipError_t launch_readout(const float* emb, const float* wr, const float* br, long n, float* rgbs, hipStream_t st) {
    hipLaunchKernelGGL(readout_kernel, grid_for(n), dim3(kThreads), 0, st, emb, wr, br, n, rgbs);
    return hipGetLastError();
}
hipError_t launch_finish_view(const float* rgb, const float* depth, long n, float* minmax, uint8_t* rgb8, uint8_t* depth8,
                              hipStream_t st) {
    hipLaunchKernelGGL(depth_minmax_kernel, dim3(1), dim3(1024), 0, st, depth, n, minmax);
    hipLaunchKernelGGL(finish_view_kernel, grid_for(n), dim3(kThreads), 0, st, rgb, depth, minmax, n, rgb8, depth8);
    return hipGetLastError();
}

}  // namespace mvnerf
