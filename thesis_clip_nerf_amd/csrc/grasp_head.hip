// The per-point part of GraspReadout (delta_ngf/layers.py:8-42 as used by LanguageNeRF, lmvnerf/model_v4.py:261-263, 290-322) as three fused
// kernels - value, vector-Jacobian product, and the derivative of that product - so that one training step no longer runs ~400 torch
// launches on (64 512 x 64..256)-sized tensors for it.
//
//   head(a_1..a_4):  u_k = W_k a_k + b_k (128 -> 64),  h_k = elu(u_k),  c = [h_1 | h_2 | h_3 | h_4],  v = W_c c + b_c (256 -> 64),  y = elu(v)
//
// LanguageNeRF.train_step differentiates the prediction w.r.t. the grasp pose inside a second tape, so the read-out is needed three ways:
//   fwd      : y                                                             (both passes of a step)
//   vjp      : g_a_k = W_k^T ((W_c[:,k]^T (g_y . elu'(v))) . elu'(u_k))      (d prediction / d activations -> trunk VJP -> d pose)
//              + the per-point cotangents g_v, g_u from which the weight gradients are two skinny GEMMs (mvnerf_gemm_tn)
//   vjp_bwd  : given T_k = dL / d(g_a_k) (the trunk's forward-mode product of the pose-gradient loss), dL / d(g_y) and the per-point
//              tensors r, m, p from which dL / d(weights) are again mvnerf_gemm_tn products (formulas at the kernel)
//
// Mapping: one wavefront = 32 points; Y^T = W X^T on v_mfma_f32_32x32x2_f32 exactly as field_eval.hip - A = weights, B = activations
// (lane = point), D registers of one product are the B operands of the next (weights stored k-permuted, mvnerf_pack.h) - so a point's
// activations never leave the register file inside a pass.  fp32 throughout (the fp32 MFMA multiplies exactly).  The passes are small (0.8 -
// 1.5 GFLOP): the point is launch count and HBM round trips, not the matrix rate.
#include <hip/hip_runtime.h>

#include "../../include/mvnerf_hip.h"
#include "mvnerf_kernels.h"
#include "mvnerf_mfma.h"

namespace mvnerf {

namespace {

// ---- packed weights (floats): chunk = 1 KiB = [lane][4 k-steps]; set (KB, NBO): chunk index ((kb * 4 + t) * NBO + nbo) -----------------
//   P1_k (128 -> 64)  u_k / s_k   : A[i][kk] = W_k[32 nbo + i][kk]
//   P2_k ( 64 -> 64)  v, z terms  : A[i][kk] = W_c[32 nbo + i][64 k + kk]
//   P3_k ( 64 -> 64)  q_k, w_k    : A[i][kk] = W_c[kk][64 k + 32 nbo + i]
//   P4_k ( 64 -> 128) g_a_k       : A[i][kk] = W_k[kk][32 nbo + i]
// with kk = 32 kb + 8 t + 4 h + e (the accumulator order: register 4 t + e of block kb on lane half h)
constexpr int kP1 = 0, kP1Size = 8192;
constexpr int kP2 = kP1 + 4 * kP1Size, kP2Size = 4096;
constexpr int kP3 = kP2 + 4 * kP2Size, kP3Size = 4096;
constexpr int kP4 = kP3 + 4 * kP3Size, kP4Size = 8192;
constexpr int kHeadPacked = kP4 + 4 * kP4Size;                  // 98304 floats

__global__ void grasp_head_pack_kernel(const float* __restrict__ w4 /*(4,64,128)*/, const float* __restrict__ wc /*(64,256)*/,
                                       float* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kHeadPacked) return;
    int set, k, off;
    if (idx < kP2) { set = 1; k = idx / kP1Size; off = idx % kP1Size; }
    else if (idx < kP3) { set = 2; k = (idx - kP2) / kP2Size; off = (idx - kP2) % kP2Size; }
    else if (idx < kP4) { set = 3; k = (idx - kP3) / kP3Size; off = (idx - kP3) % kP3Size; }
    else { set = 4; k = (idx - kP4) / kP4Size; off = (idx - kP4) % kP4Size; }
    const int nbo_count = set == 4 ? 4 : 2;
    const int chunk = off / 256, lane = (off % 256) / 4, e = off % 4;
    const int i = lane & 31, h = lane >> 5;
    const int nbo = chunk % nbo_count, kt = chunk / nbo_count, kb = kt / 4, t = kt % 4;
    const int kk = 32 * kb + 8 * t + 4 * h + e, o = 32 * nbo + i;
    float val;
    if (set == 1) val = w4[(k * 64 + o) * 128 + kk];
    else if (set == 2) val = wc[o * 256 + 64 * k + kk];
    else if (set == 3) val = wc[kk * 256 + 64 * k + o];
    else val = w4[(k * 64 + kk) * 128 + o];
    dst[idx] = val;
}

// acc[nbo] += A^T-product over KB input blocks held in accumulator order.  The A chunks of step (kb, t) + 1 are requested before the
// MFMAs of step (kb, t) are issued and pinned there (the weights come from L2: a round trip is as long as a step's 4 x NBO MFMAs).
template <int KB, int NBO>
__device__ __forceinline__ void dense_blocks(const float* __restrict__ packed, int lane, const f32x16 (&in)[KB], f32x16 (&acc)[NBO]) {
    const f32x4* w = reinterpret_cast<const f32x4*>(packed) + lane;
    f32x4 a[NBO], an[NBO];
#pragma unroll
    for (int nbo = 0; nbo < NBO; ++nbo) a[nbo] = w[nbo * 64];
#pragma unroll
    for (int st = 0; st < KB * 4; ++st) {
        const int kb = st >> 2, t = st & 3;
        if (st + 1 < KB * 4) {
#pragma unroll
            for (int nbo = 0; nbo < NBO; ++nbo) an[nbo] = w[((st + 1) * NBO + nbo) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int nbo = 0; nbo < NBO; ++nbo) acc[nbo] = mfma(a[nbo][e], in[kb][4 * t + e], acc[nbo]);
        if (st + 1 < KB * 4) {
#pragma unroll
            for (int nbo = 0; nbo < NBO; ++nbo) a[nbo] = an[nbo];
        }
    }
}

// block nb (32 features) of row `point` of a row-major (N, F) tensor, in accumulator order: lane (j, h) register 4q + c = feature
// 32 nb + 8 q + 4 h + c
__device__ __forceinline__ f32x16 load_block(const float* __restrict__ rows, long point, int F, int nb, int h) {
    const f32x4* p = reinterpret_cast<const f32x4*>(rows + point * F + 32 * nb + 4 * h);
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t4 = p[2 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * q + c] = t4[c];
    }
    return v;
}

__device__ __forceinline__ void store_block(float* __restrict__ rows, long point, int F, int nb, int h, const f32x16& v) {
    f32x4* p = reinterpret_cast<f32x4*>(rows + point * F + 32 * nb + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t4 = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        p[2 * q] = t4;
    }
}

// bias (F floats, plain order) of block nb in accumulator order
__device__ __forceinline__ f32x16 bias_block(const float* __restrict__ bias, int nb, int h) {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = bias[32 * nb + (r & 3) + 8 * (r >> 2) + 4 * h];
    return v;
}

__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : expm1f(x); }
// derivatives of elu in terms of its OUTPUT e = elu(u): elu'(u) = u > 0 ? 1 : e + 1 ; elu''(u) = u > 0 ? 0 : e + 1   (e > 0 <=> u > 0)
__device__ __forceinline__ float delu(float e) { return e > 0.0f ? 1.0f : e + 1.0f; }
__device__ __forceinline__ float ddelu(float e) { return e > 0.0f ? 0.0f : e + 1.0f; }

// ---- value: acts (4, N, 128) -> c (N, 256) = [elu(u_k)], y (N, 64) ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grasp_head_fwd_kernel(const float* __restrict__ acts, const float* __restrict__ packed,
                                                              const float* __restrict__ b4, const float* __restrict__ bc, long N,
                                                              float* __restrict__ c_out, float* __restrict__ y_out) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile * 32 >= N) return;
    const long pt_raw = tile * 32 + j;
    const bool ok = pt_raw < N;
    const long pt = ok ? pt_raw : N - 1;
    f32x16 v[2] = {bias_block(bc, 0, h), bias_block(bc, 1, h)};
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        f32x16 a[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) a[kb] = load_block(acts + (long)k * N * 128, pt, 128, kb, h);
        f32x16 u[2] = {bias_block(b4 + 64 * k, 0, h), bias_block(b4 + 64 * k, 1, h)};
        dense_blocks<4, 2>(packed + kP1 + k * kP1Size, lane, a, u);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) u[nb][r] = elu1(u[nb][r]);
            if (ok) store_block(c_out, pt, 256, 2 * k + nb, h, u[nb]);
        }
        dense_blocks<2, 2>(packed + kP2 + k * kP2Size, lane, u, v);
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[nb][r] = elu1(v[nb][r]);
        if (ok) store_block(y_out, pt, 64, nb, h, v[nb]);
    }
}

// ---- vector-Jacobian product: g_y (N, 64) -> g_v (N, 64), q (N, 256), g_u (N, 256), g_acts (4, N, 128) ---------------------------------
//   g_v = g_y . elu'(v);  q_k = W_c[:,k]^T g_v;  g_u_k = q_k . elu'(u_k);  g_a_k = W_k^T g_u_k
//   (weight gradients afterwards: dW_c = g_v^T c, db_c = sum g_v, dW_k = g_u_k^T a_k, db_k = sum g_u_k)
__global__ __launch_bounds__(256) void grasp_head_vjp_kernel(const float* __restrict__ g_y, const float* __restrict__ c, const float* __restrict__ y,
                                                              const float* __restrict__ packed, long N, float* __restrict__ g_v_out,
                                                              float* __restrict__ q_out, float* __restrict__ g_u_out, float* __restrict__ g_acts) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile * 32 >= N) return;
    const long pt_raw = tile * 32 + j;
    const bool ok = pt_raw < N;
    const long pt = ok ? pt_raw : N - 1;
    f32x16 gv[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const f32x16 gy = load_block(g_y, pt, 64, nb, h), yy = load_block(y, pt, 64, nb, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) gv[nb][r] = gy[r] * delu(yy[r]);
        if (ok) store_block(g_v_out, pt, 64, nb, h, gv[nb]);
    }
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        f32x16 q[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) q[nb][r] = 0.0f;
        dense_blocks<2, 2>(packed + kP3 + k * kP3Size, lane, gv, q);
        f32x16 gu[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const f32x16 ck = load_block(c, pt, 256, 2 * k + nb, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) gu[nb][r] = q[nb][r] * delu(ck[r]);
            if (ok) {
                store_block(q_out, pt, 256, 2 * k + nb, h, q[nb]);
                store_block(g_u_out, pt, 256, 2 * k + nb, h, gu[nb]);
            }
        }
        f32x16 ga[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) ga[nb][r] = 0.0f;
        dense_blocks<2, 4>(packed + kP4 + k * kP4Size, lane, gu, ga);
        if (ok) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) store_block(g_acts + (long)k * N * 128, pt, 128, nb, h, ga[nb]);
        }
    }
}

// ---- derivative of the vector-Jacobian product.  L depends on g_a_k (k = 1..4) with dL/d(g_a_k) = T_k (N, 128):
//   s_k = W_k T_k;          r_k = s_k . elu'(u_k)                 (dL/dq_k)
//   z   = sum_k W_c[:,k] r_k                                       (dL/dg_v)
//   dL/dg_y = z . elu'(v);  m = z . g_y . elu''(v)                 (dL/dv through elu'(v))
//   w_k = W_c[:,k]^T m;     p_k = s_k . q_k . elu''(u_k) + w_k . elu'(u_k)      (dL/du_k: through elu'(u_k), and through c_k = elu(u_k))
//   weight gradients afterwards (mvnerf_gemm_tn):  dW_k = g_u_k^T T_k + p_k^T a_k,  db_k = sum p_k,  dW_c = g_v^T r + m^T c,  db_c = sum m
//   (dL/da_k is not formed: in LanguageNeRF.train_step the activations depend on the pose only, and only the read-out is trained.)
__global__ __launch_bounds__(256) void grasp_head_vjp_bwd_kernel(const float* __restrict__ t_acts, const float* __restrict__ g_y,
                                                                  const float* __restrict__ c, const float* __restrict__ y,
                                                                  const float* __restrict__ q_in, const float* __restrict__ packed, long N,
                                                                  float* __restrict__ out_gy, float* __restrict__ r_out,
                                                                  float* __restrict__ m_out, float* __restrict__ p_out) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile * 32 >= N) return;
    const long pt_raw = tile * 32 + j;
    const bool ok = pt_raw < N;
    const long pt = ok ? pt_raw : N - 1;
    f32x16 z[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[nb][r] = 0.0f;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        f32x16 tk[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) tk[kb] = load_block(t_acts + (long)k * N * 128, pt, 128, kb, h);
        f32x16 s[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[nb][r] = 0.0f;
        dense_blocks<4, 2>(packed + kP1 + k * kP1Size, lane, tk, s);
        f32x16 rk[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const f32x16 ck = load_block(c, pt, 256, 2 * k + nb, h), qk = load_block(q_in, pt, 256, 2 * k + nb, h);
            f32x16 p1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                rk[nb][r] = s[nb][r] * delu(ck[r]);
                p1[r] = s[nb][r] * qk[r] * ddelu(ck[r]);
            }
            if (ok) {
                store_block(r_out, pt, 256, 2 * k + nb, h, rk[nb]);
                store_block(p_out, pt, 256, 2 * k + nb, h, p1);            // first term of p_k; the second is added below
            }
        }
        dense_blocks<2, 2>(packed + kP2 + k * kP2Size, lane, rk, z);
    }
    f32x16 m[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const f32x16 gy = load_block(g_y, pt, 64, nb, h), yy = load_block(y, pt, 64, nb, h);
        f32x16 og;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            og[r] = z[nb][r] * delu(yy[r]);
            m[nb][r] = z[nb][r] * gy[r] * ddelu(yy[r]);
        }
        if (ok) {
            store_block(out_gy, pt, 64, nb, h, og);
            store_block(m_out, pt, 64, nb, h, m[nb]);
        }
    }
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        f32x16 w[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) w[nb][r] = 0.0f;
        dense_blocks<2, 2>(packed + kP3 + k * kP3Size, lane, m, w);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const f32x16 ck = load_block(c, pt, 256, 2 * k + nb, h);
            f32x16 pk = load_block(p_out, pt, 256, 2 * k + nb, h);           // this lane's own store from the first sweep
#pragma unroll
            for (int r = 0; r < 16; ++r) pk[r] = pk[r] + w[nb][r] * delu(ck[r]);
            if (ok) store_block(p_out, pt, 256, 2 * k + nb, h, pk);
        }
    }
}

}  // namespace

size_t grasp_head_packed_floats() { return (size_t)kHeadPacked; }

hipError_t launch_grasp_head_pack(const float* w4, const float* wc, float* packed, hipStream_t st) {
    hipLaunchKernelGGL(grasp_head_pack_kernel, dim3((kHeadPacked + 255) / 256), dim3(256), 0, st, w4, wc, packed);
    return hipGetLastError();
}

static unsigned head_grid(long N) { return (unsigned)(((N + 31) / 32 + 3) / 4); }

hipError_t launch_grasp_head_fwd(const float* acts, const float* packed, const float* b4, const float* bc, long N, float* c, float* y,
                                 hipStream_t st) {
    hipLaunchKernelGGL(grasp_head_fwd_kernel, dim3(head_grid(N)), dim3(256), 0, st, acts, packed, b4, bc, N, c, y);
    return hipGetLastError();
}

hipError_t launch_grasp_head_vjp(const float* g_y, const float* c, const float* y, const float* packed, long N, float* g_v, float* q, float* g_u,
                                 float* g_acts, hipStream_t st) {
    hipLaunchKernelGGL(grasp_head_vjp_kernel, dim3(head_grid(N)), dim3(256), 0, st, g_y, c, y, packed, N, g_v, q, g_u, g_acts);
    return hipGetLastError();
}

hipError_t launch_grasp_head_vjp_bwd(const float* t_acts, const float* g_y, const float* c, const float* y, const float* q, const float* packed,
                                     long N, float* out_gy, float* r, float* m, float* p, hipStream_t st) {
    hipLaunchKernelGGL(grasp_head_vjp_bwd_kernel, dim3(head_grid(N)), dim3(256), 0, st, t_acts, g_y, c, y, q, packed, N, out_gy, r, m, p);
    return hipGetLastError();
}

}  // namespace mvnerf

// ---- C ABI ----------------------------------------------------------------------------------------------------------------------------------
extern "C" {

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static int hs(hipError_t e, const char* who) { return e == hipSuccess ? 0 : mvnerf::api_fail((int)e, "%s: %s", who, hipGetErrorString(e)); }

size_t mvnerf_grasp_head_packed_floats(void) { return mvnerf::grasp_head_packed_floats(); }

int mvnerf_grasp_head_pack(const float* w4, const float* wc, float* packed, mvnerf_stream_t stream) {
    if (!w4 || !wc || !packed) return mvnerf::api_fail(MVNERF_E_ARG, "mvnerf_grasp_head_pack: null pointer");
    if (!al16(packed)) return mvnerf::api_fail(MVNERF_E_ALIGN, "mvnerf_grasp_head_pack: packed must be 16-byte aligned");
    return hs(mvnerf::launch_grasp_head_pack(w4, wc, packed, static_cast<hipStream_t>(stream)), "mvnerf_grasp_head_pack");
}

int mvnerf_grasp_head_fwd(const float* acts, const float* packed, const float* b4, const float* bc, long N, float* c, float* y,
                          mvnerf_stream_t stream) {
    if (!acts || !packed || !b4 || !bc || !c || !y) return mvnerf::api_fail(MVNERF_E_ARG, "mvnerf_grasp_head_fwd: null pointer");
    if (N <= 0) return mvnerf::api_fail(MVNERF_E_ARG, "mvnerf_grasp_head_fwd: N=%ld", N);
    if (!al16(acts) || !al16(packed) || !al16(c) || !al16(y))
        return mvnerf::api_fail(MVNERF_E_ALIGN, "mvnerf_grasp_head_fwd: acts, packed, c, y must be 16-byte aligned");
    return hs(mvnerf::launch_grasp_head_fwd(acts, packed, b4, bc, N, c, y, static_cast<hipStream_t>(stream)), "mvnerf_grasp_head_fwd");
}

int mvnerf_grasp_head_vjp(const float* g_y, const float* c, const float* y, const float* packed, long N, float* g_v, float* q, float* g_u,
                          float* g_acts, mvnerf_stream_t stream) {
    if (!g_y || !c || !y || !packed || !g_v || !q || !g_u || !g_acts) return mvnerf::api_fail(MVNERF_E_ARG, "mvnerf_grasp_head_vjp: null pointer");
    if (N <= 0) return mvnerf::api_fail(MVNERF_E_ARG, "mvnerf_grasp_head_vjp: N=%ld", N);
    if (!al16(g_y) || !al16(c) || !al16(y) || !al16(packed) || !al16(g_v) || !al16(q) || !al16(g_u) || !al16(g_acts))
        return mvnerf::api_fail(MVNERF_E_ALIGN, "mvnerf_grasp_head_vjp: every buffer must be 16-byte aligned");
    return hs(mvnerf::launch_grasp_head_vjp(g_y, c, y, packed, N, g_v, q, g_u, g_acts, static_cast<hipStream_t>(stream)), "mvnerf_grasp_head_vjp");
}

int mvnerf_grasp_head_vjp_bwd(const float* t_acts, const float* g_y, const float* c, const float* y, const float* q, const float* packed, long N,
                              float* out_gy, float* r, float* m, float* p, mvnerf_stream_t stream) {
    if (!t_acts || !g_y || !c || !y || !q || !packed || !out_gy || !r || !m || !p)
        return mvnerf::api_fail(MVNERF_E_ARG, "mvnerf_grasp_head_vjp_bwd: null pointer");
    if (N <= 0) return mvnerf::api_fail(MVNERF_E_ARG, "mvnerf_grasp_head_vjp_bwd: N=%ld", N);
    if (!al16(t_acts) || !al16(g_y) || !al16(c) || !al16(y) || !al16(q) || !al16(packed) || !al16(out_gy) || !al16(r) || !al16(m) || !al16(p))
        return mvnerf::api_fail(MVNERF_E_ALIGN, "mvnerf_grasp_head_vjp_bwd: every buffer must be 16-byte aligned");
    return hs(mvnerf::launch_grasp_head_vjp_bwd(t_acts, g_y, c, y, q, packed, N, out_gy, r, m, p, static_cast<hipStream_t>(stream)),
              "mvnerf_grasp_head_vjp_bwd");
}

}  // extern "C"
