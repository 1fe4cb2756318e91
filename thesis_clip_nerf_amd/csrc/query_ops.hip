// The trunk as a differentiable field on arbitrary query points (reference consumer:
// lmvnerf/model_v4.py:208-265, LanguageNeRF._call -> fine_embedding(...)[4:] -> GraspReadout, whose train_step
// (:277-322) differentiates the prediction w.r.t. the query poses and then that gradient w.r.t. the read-out).
//
//   field_jvp_kernel      forward-mode: tangents (t_o, t_d) of the points / directions -> tangents of the four fused
//                         activations (view mean, u1, u2, u3), primal recomputed in the same pass.  It is the
//                         transpose of the input-gradient kernel, i.e. the double-backward the reference's nested
//                         GradientTape asks of the trunk.
//   dir_seed_tangent_kernel   per-(view, point) layer-0 seed b0 + W0_dir^T PE(cam dir) and its tangent.
//
// Primal and tangent share every weight operand: a step issues 32 MFMAs (16 primal, 16 tangent) per 4 KiB of
// weights, activations of both stay in the register file (x, hid, tx, thid: 256 accumulator registers, one wave
// per SIMD).  relu' is taken from the primal pre-activation in registers - nothing is stashed in HBM.
#include <hip/hip_runtime.h>

#include <atomic>

#include "mvnerf_kernels.h"
#include "mvnerf_math.h"
#include "mvnerf_pack.h"
#include "mvnerf_field_common.h"

namespace mvnerf {

// One step = 4 k-steps x 4 output blocks for the primal (b -> acc) and the tangent (tb -> tacc) with the same A.
__device__ __forceinline__ void mfma_step2(WStream& ws, const float (&b)[4], const float (&tb)[4], f32x16 (&acc)[4],
                                           f32x16 (&tacc)[4]) {
    const f32x4 n0 = ws_load<0>(ws, ws.pos), n1 = ws_load<1024>(ws, ws.pos), n2 = ws_load<2048>(ws, ws.pos),
                n3 = ws_load<3072>(ws, ws.pos);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            acc[nb] = mfma(ws.cur[nb][e], b[e], acc[nb]);
            tacc[nb] = mfma(ws.cur[nb][e], tb[e], tacc[nb]);
        }
    }
    ws_advance(ws, n0, n1, n2, n3);
}

// acc += W^T relu(in) ; tacc += W^T (relu'(in) (.) tin)
__device__ __forceinline__ void dense128_jvp(WStream& ws, const f32x16 (&in)[4], const f32x16 (&tin)[4], f32x16 (&acc)[4],
                                             f32x16 (&tacc)[4]) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float b[4], tb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = in[kb][4 * t + e];
                b[e] = fmaxf(v, 0.0f);
                tb[e] = v > 0.0f ? tin[kb][4 * t + e] : 0.0f;
            }
            mfma_step2(ws, b, tb, acc, tacc);
        }
}

__device__ __forceinline__ void zero_acc(f32x16 (&a)[4]) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[nb][r] = 0.0f;
}

// x <- x + W2^T relu(W1^T relu(x) + b1) + b2 and its tangent (biases have no tangent)
__device__ __forceinline__ void resnet_block_jvp(WStream& ws, const float* __restrict__ bias1, int h, f32x16 (&x)[4],
                                                 f32x16 (&tx)[4], f32x16 (&hid)[4], f32x16 (&thid)[4]) {
    bias_to_acc<false>(bias1, h, hid);
    zero_acc(thid);
    dense128_jvp(ws, x, tx, hid, thid);
    bias_to_acc<true>(bias1 + 128, h, x);
    dense128_jvp(ws, hid, thid, x, tx);
}

template <bool kMultiView>
__global__ __launch_bounds__(256, 1) void field_jvp_kernel(FieldParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // 2 x 16 KiB per wave: primal | tangent stage
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* stage = lds + wave * (2 * kTile * kStageRow);
    float* tstage = stage + kTile * kStageRow;
    const long tile = (long)blockIdx.x * 4 + wave;
    if (tile >= p.n_tiles) return;

    long g = tile * kTile + j;
    const bool valid = g < p.total;
    if (!valid) g = p.total - 1;
    const int ray = (int)(g / p.S);
    const int sidx = (int)(g - (long)ray * p.S);
    (void)sidx;
    const int b = ray / p.R;
    const float dx = p.rays_d[3 * ray + 0], dy = p.rays_d[3 * ray + 1], dz = p.rays_d[3 * ray + 2];
    const float zz = p.z ? p.z[g] : 0.0f;                    // query points: z = NULL, the point is rays_o itself
    const float wx = p.rays_o[3 * ray + 0] + zz * dx, wy = p.rays_o[3 * ray + 1] + zz * dy, wz = p.rays_o[3 * ray + 2] + zz * dz;
    // tangent of the world point: t_o + z t_d
    const float twx = p.t_o[3 * ray + 0] + zz * p.t_d[3 * ray + 0], twy = p.t_o[3 * ray + 1] + zz * p.t_d[3 * ray + 1],
                twz = p.t_o[3 * ray + 2] + zz * p.t_d[3 * ray + 2];
    const float* __restrict__ net = p.net;

    f32x16 x[4], hid[4], tx[4], thid[4];
    f32x16 xsum[kMultiView ? 4 : 1], txsum[kMultiView ? 4 : 1];
    WStream ws;

    for (int v = 0; v < p.V; ++v) {
        ws_begin(ws, net, kPackTotal * 4, lane);
        const int bv = b * p.V + v;
        const float* E = p.einv + 16 * bv;
        const float* K = p.k4 + 16 * bv;
        float cam[4], tcam[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cam[r] = row_dot4(E, r, wx, wy, wz, 1.0f);
            tcam[r] = E[4 * r] * twx + E[4 * r + 1] * twy + E[4 * r + 2] * twz;
        }
        const float q0 = row_dot4(K, 0, cam[0], cam[1], cam[2], cam[3]);
        const float q1 = row_dot4(K, 1, cam[0], cam[1], cam[2], cam[3]);
        const float q2 = row_dot4(K, 2, cam[0], cam[1], cam[2], cam[3]);
        const float tq0 = row_dot4(K, 0, tcam[0], tcam[1], tcam[2], tcam[3]);
        const float tq1 = row_dot4(K, 1, tcam[0], tcam[1], tcam[2], tcam[3]);
        const float tq2 = q2 >= 1e-8f ? row_dot4(K, 2, tcam[0], tcam[1], tcam[2], tcam[3]) : 0.0f;   // max(q2, 1e-8)
        const float den = fmaxf(q2, 1e-8f);
        const float pxr = q0 / den, pyr = q1 / den;
        const float pxl = fminf(fmaxf(pxr, -1e6f), 1e6f), pyl = fminf(fmaxf(pyr, -1e6f), 1e6f);
        const Taps tp = bilinear_taps(pxl, pyl, p.H, p.W);
        const float ux = pxl - (float)tp.x0, uy = pyl - (float)tp.y0;               // unclamped lerp factors
        // clamps pass the tangent inside their closed range (tf.clip_by_value / torch.clamp), floor has none
        const float tax = (ux >= 0.0f && ux <= 1.0f && pxr >= -1e6f && pxr <= 1e6f) ? (tq0 - pxr * tq2) / den : 0.0f;
        const float tay = (uy >= 0.0f && uy <= 1.0f && pyr >= -1e6f && pyr <= 1e6f) ? (tq1 - pyr * tq2) / den : 0.0f;
        const int tl = (bv * p.H + tp.y0) * p.W + tp.x0;

        // seeds: b0 + W0_dir^T PE(cam dir) and its tangent, per (view, point)
        const long srow = (long)bv * p.R + (ray - b * p.R);
        bias_to_acc<false>(p.dir_bias + 128 * srow, h, x);
        bias_to_acc<false>(p.dir_tan + 128 * srow, h, tx);

        float pe[32], tpe[32];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int k = 0; k < kNFreq; ++k) {
                const float f = 3.14159274101257324f * (float)(1 << k);
                float sk, ck;
                sincos_f32(cam[d] * f, &sk, &ck);
                pe[d * 10 + k] = h ? ck : sk;
                tpe[d * 10 + k] = (h ? -sk : ck) * (f * tcam[d]);
            }
        }
        {
            const float* img = p.images + 3 * (long)tl;
            float rgbv[3], trgb[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float a = img[c] * 2.0f - 1.0f, bq = img[3 + c] * 2.0f - 1.0f;
                const float cq = img[3 * p.W + c] * 2.0f - 1.0f, dq = img[3 * p.W + 3 + c] * 2.0f - 1.0f;
                rgbv[c] = bilerp(a, bq, cq, dq, tp.ax, tp.ay);
                trgb[c] = tax * ((1.0f - tp.ay) * (bq - a) + tp.ay * (dq - cq)) + tay * ((cq - a) + tp.ax * ((dq - cq) - (bq - a)));
            }
            pe[30] = h ? rgbv[1] : rgbv[0];
            pe[31] = h ? 0.0f : rgbv[2];
            tpe[30] = h ? trgb[1] : trgb[0];
            tpe[31] = h ? 0.0f : trgb[2];
        }
#pragma unroll
        for (int gq = 0; gq < 8; ++gq) {
            const float bb[4] = {pe[4 * gq], pe[4 * gq + 1], pe[4 * gq + 2], pe[4 * gq + 3]};
            const float tb[4] = {tpe[4 * gq], tpe[4 * gq + 1], tpe[4 * gq + 2], tpe[4 * gq + 3]};
            mfma_step2(ws, bb, tb, x, tx);
        }
#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const f32x4* fbase = reinterpret_cast<const f32x4*>(p.features) + hf * 32 + j;
#pragma unroll 2
            for (int it = 0; it < 16; ++it) {
                const int src = 2 * it + h;
                const int tls = __shfl(tl, src);
                const float axs = __shfl(tp.ax, src), ays = __shfl(tp.ay, src);
                const float taxs = __shfl(tax, src), tays = __shfl(tay, src);
                const f32x4* f = fbase + (long)tls * 64;
                const f32x4 vtl = f[0], vtr = f[64], vbl = f[(long)p.W * 64], vbr = f[(long)p.W * 64 + 64];
                f32x4 o, ot;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float dt = vtr[c] - vtl[c], db = vbr[c] - vbl[c];
                    const float top = fmaf(axs, dt, vtl[c]);
                    const float bot = fmaf(axs, db, vbl[c]);
                    o[c] = fmaf(ays, bot - top, top);
                    ot[c] = taxs * fmaf(ays, db - dt, dt) + tays * (bot - top);
                }
                *reinterpret_cast<f32x4*>(stage + stage_offset(src, j)) = o;
                *reinterpret_cast<f32x4*>(tstage + stage_offset(src, j)) = ot;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int gg = 0; gg < 16; ++gg) {
                const f32x4 bv4 = *reinterpret_cast<const f32x4*>(stage + stage_offset(j, 2 * gg + h));
                const f32x4 tv4 = *reinterpret_cast<const f32x4*>(tstage + stage_offset(j, 2 * gg + h));
                const float bb[4] = {bv4[0], bv4[1], bv4[2], bv4[3]};
                const float tb[4] = {tv4[0], tv4[1], tv4[2], tv4[3]};
                mfma_step2(ws, bb, tb, x, tx);
            }
        }
#pragma unroll 1
        for (int bi = 0; bi < 3; ++bi) resnet_block_jvp(ws, net + kPackBHidden + 256 * bi, h, x, tx, hid, thid);
        if (kMultiView) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                xsum[nb] = (v == 0) ? x[nb] : xsum[nb] + x[nb];
                txsum[nb] = (v == 0) ? tx[nb] : txsum[nb] + tx[nb];
            }
        }
    }
    if (kMultiView) {
        const float nv = (float)p.V;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            x[nb] = xsum[nb] / nv;
            tx[nb] = txsum[nb] / nv;
        }
    }
    if (valid) {
        if (p.acts_fused) store_acc(p.acts_fused + 128 * g, h, x);
        store_acc(p.t_acts + 128 * g, h, tx);
    }
#pragma unroll 1
    for (int bi = 3; bi < 6; ++bi) {
        resnet_block_jvp(ws, net + kPackBHidden + 256 * bi, h, x, tx, hid, thid);
        if (valid) {
            if (p.acts_fused) store_acc(p.acts_fused + (bi - 2) * p.total * 128 + 128 * g, h, x);
            store_acc(p.t_acts + (bi - 2) * p.total * 128 + 128 * g, h, tx);
        }
    }
}

// One wavefront per (b*V+v, point): seed = b0 + W0[60:120]^T PE(cam dir), tangent seed = W0[60:120]^T dPE(cam dir) t_camdir
__global__ __launch_bounds__(256) void dir_seed_tangent_kernel(FieldParams p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)p.B * p.V * p.R) return;
    const int bv = (int)(row / p.R);
    const long ray = (long)(bv / p.V) * p.R + (row - (long)bv * p.R);
    const float dx = p.rays_d[3 * ray + 0], dy = p.rays_d[3 * ray + 1], dz = p.rays_d[3 * ray + 2];
    const float tdx = p.t_d[3 * ray + 0], tdy = p.t_d[3 * ray + 1], tdz = p.t_d[3 * ray + 2];
    const float* E = p.einv + 16 * bv;
    const int m = lane < 60 ? lane : 59;
    const int d = m / 20, k = (m % 20) >> 1, f = m & 1;
    const float cd = row_dot4(E, d, dx, dy, dz, 1.0f);                                   // Q3: w = 1
    const float tcd = E[4 * d] * tdx + E[4 * d + 1] * tdy + E[4 * d + 2] * tdz;
    const float freq = 3.14159274101257324f * (float)(1 << k);
    float sv, cv;
    sincos_f32(cd * freq, &sv, &cv);
    const float mine = f ? cv : sv;
    const float tmine = (f ? -sv : cv) * (freq * tcd);
    const float* wd = p.net + kPackW0Dir;
    float a0 = p.net[kPackB0Plain + lane], a1 = p.net[kPackB0Plain + 64 + lane];
    float t0 = 0.0f, t1 = 0.0f;
    for (int mm = 0; mm < 60; ++mm) {
        const float pv = __shfl(mine, mm), tv = __shfl(tmine, mm);
        const float w0 = wd[mm * 128 + lane], w1 = wd[mm * 128 + 64 + lane];
        a0 = fmaf(pv, w0, a0);
        a1 = fmaf(pv, w1, a1);
        t0 = fmaf(tv, w0, t0);
        t1 = fmaf(tv, w1, t1);
    }
    p.dir_bias[128 * row + acc_slot(lane)] = a0;
    p.dir_bias[128 * row + acc_slot(64 + lane)] = a1;
    p.dir_tan[128 * row + acc_slot(lane)] = t0;
    p.dir_tan[128 * row + acc_slot(64 + lane)] = t1;
}

hipError_t launch_field_jvp(const FieldParams& p, hipStream_t stream) {
    static std::atomic<bool> attr_done[16];      // first call per device sets the dynamic-LDS limit (idempotent)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const int lds_bytes = 4 * 2 * kTile * kStageRow * 4;           // 128 KiB
    if (dev >= 0 && dev < 16 && !attr_done[dev].load(std::memory_order_acquire)) {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&field_jvp_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&field_jvp_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess) return e;
        attr_done[dev].store(true, std::memory_order_release);
    }
    const long rows = (long)p.B * p.V * p.R;
    hipLaunchKernelGGL(dir_seed_tangent_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, p);
    const unsigned wgs = (unsigned)((p.n_tiles + 3) / 4);
    if (p.V > 1) hipLaunchKernelGGL(field_jvp_kernel<true>, dim3(wgs), dim3(256), lds_bytes, stream, p);
    else hipLaunchKernelGGL(field_jvp_kernel<false>, dim3(wgs), dim3(256), lds_bytes, stream, p);
    return hipGetLastError();
}

}  // namespace mvnerf
