// Fused radiance-field evaluation for gfx950 (MI355X):
//   project -> bilinear gather -> positional encoding -> ResNet-MLP trunk -> read-out
// for 32 samples per wavefront, activations resident in the register file, weights streamed from
// L2 in MFMA operand order (mvnerf_pack.h) one 4 KiB step ahead of use, gathered features
// transposed through wave-private LDS.
//
// Reference being replaced: model_v0.py:122-144 / :157-180 (see include/mvnerf_hip.h,
// mvnerf_field_eval).  One launch evaluates B*R*S samples; a wavefront owns 32 consecutive samples
// (half a coarse ray, a quarter of a fine ray), waves are independent (no workgroup barrier).
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>

#include "mvnerf_kernels.h"
#include "mvnerf_math.h"
#include "mvnerf_pack.h"

namespace mvnerf {


// Launch shape: measured best ("Field kernel: what was measured") is the plain one: one
// 32-sample tile per wave, 4 waves per workgroup, 2 workgroups per CU (2 waves per SIMD).  The
// persistent / ticket-queue / raised-priority forms below are kept as build switches for A/B runs.

// Tuning switches (A/B-tested on the GPU, see DESIGN.md "Field kernel: what was measured")
#ifndef MV_PERSIST
#define MV_PERSIST 0       // 1: one workgroup per CU pulling tiles from an atomic ticket; 0: one tile per wave
#endif
#ifndef MV_PRIO
#define MV_PRIO 0          // raise the priority of waves 0..3 of an 8-wave workgroup
#endif
#ifndef MV_ABL_PE
#define MV_ABL_PE 0        // timing-only ablations (wrong results): skip sin/cos
#endif
#ifndef MV_ABL_GATHER
#define MV_ABL_GATHER 0    // skip feature gather + lerp + LDS staging
#endif
#ifndef MV_ABL_BIAS
#define MV_ABL_BIAS 0      // skip bias loads
#endif
#ifndef MV_ABL_WLOAD
#define MV_ABL_WLOAD 0     // do not stream weights (re-use the first 4 chunks)
#endif
#ifndef MV_PIN_LOADS
#define MV_PIN_LOADS 1
#endif
#ifndef MV_ASM_RELU
#define MV_ASM_RELU 0
#endif
#ifndef MV_PE_RECUR
#define MV_PE_RECUR 1      // double-angle recurrence between accurate sin/cos at octaves 0 and 5
#endif
#ifndef MV_FMA_LERP
#define MV_FMA_LERP 1      // feature lerps as FMAs (6 instead of 9 VALU per channel); taps/indices unaffected
#endif
#ifndef MV_MV_OCC
#define MV_MV_OCC 2        // waves per SIMD the multi-view kernels are compiled for: 2 spills the view sum to
                           // scratch (132 B/lane) and is still 5 % faster than 1 (A/B, V=3: 745k -> 786k rays/s)
#endif
#ifndef MV_WAVES
#define MV_WAVES 4         // waves per workgroup of the single-view kernel (4 or 8)
#endif

}  // namespace mvnerf
#include "mvnerf_field_common.h"
namespace mvnerf {

// acc += W^T relu(in)   (Dense 128->128 on the pre-activated input, layers.py:285-288)
__device__ __forceinline__ void dense128(WStream& ws, const f32x16 (&in)[4], f32x16 (&acc)[4]) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float raw[4] = {in[kb][4 * t], in[kb][4 * t + 1], in[kb][4 * t + 2], in[kb][4 * t + 3]};
            float b[4];
            relu4(raw, b);
            mfma_step(ws, b, acc);
        }
    }
}

// x <- x + W2^T relu(W1^T relu(x) + b1) + b2    (ResNetMLPBlock.call, layers.py:284-298)
// Training mode: `stash` (may be null) receives the two pre-activation tensors of the block in tile layout
// (slot 0: hid = input of the second Dense, slot 1: block output = input of the next layer).
template <bool kStash>
__device__ __forceinline__ void resnet_block(WStream& ws, const float* __restrict__ bias1, int h, f32x16 (&x)[4],
                                             f32x16 (&hid)[4], float* __restrict__ stash, long slot_stride, long tile,
                                             int j, bool store_out = true) {
    bias_to_acc<false>(bias1, h, hid);
    dense128(ws, x, hid);
    if (kStash) store_tl(stash, tile, j, h, hid);
    bias_to_acc<true>(bias1 + 128, h, x);
    dense128(ws, hid, x);
    if (kStash && store_out) store_tl(stash + slot_stride, tile, j, h, x);
}

// kProj: the 256 feature rows of layer 0 come from the texel table (project_texels_kernel) instead of 128 k-steps:
// layer 0 is linear in the gathered features and the bilinear gather is linear in the texels, so
// W0f^T lerp(taps) = lerp(W0f^T taps); the lerp then runs on 128 projected channels and adds into the accumulators.
template <bool kMultiView, bool kStash, bool kProj>
__global__ __launch_bounds__(kMultiView ? 256 : 64 * MV_WAVES, kMultiView ? MV_MV_OCC : 2) void field_eval_kernel(FieldParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // 16 KiB per wave

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    float* stage = lds + wave * (kTile * kStageRow);
    if (MV_PRIO && !kMultiView && MV_WAVES == 8 && wave < 4) __builtin_amdgcn_s_setprio(1);

  for (;;) {                                             // persistent: one 32-sample tile per trip
#if MV_PERSIST
    unsigned ticket = 0;
    if (lane == 0) ticket = atomicAdd(p.tile_counter, 1u);
    const long tile = (long)__builtin_amdgcn_readfirstlane(ticket);
#else
    const long tile = (long)blockIdx.x * (blockDim.x >> 6) + wave;
#endif
    if (tile >= p.n_tiles) break;                        // whole wave leaves; there are no barriers

    long g = tile * kTile + j;
    const bool valid = g < p.total;
    if (!valid) g = p.total - 1;
    const int ray = (int)(g / p.S);                      // global ray index in [0, B*R)
    const int sidx = (int)(g - (long)ray * p.S);
    const int b = ray / p.R;

    const float ox = p.rays_o[3 * ray + 0], oy = p.rays_o[3 * ray + 1], oz = p.rays_o[3 * ray + 2];
    const float dx = p.rays_d[3 * ray + 0], dy = p.rays_d[3 * ray + 1], dz = p.rays_d[3 * ray + 2];
    const float zz = p.z[g];
    const float wx = ox + zz * dx, wy = oy + zz * dy, wz = oz + zz * dz;     // mul, then add (no FMA)

    const float* __restrict__ net = p.net;

    f32x16 x[4], hid[4];
    f32x16 xsum[kMultiView ? 4 : 1];
    WStream ws;

    for (int v = 0; v < p.V; ++v) {
        ws_begin(ws, net, kPackTotal * 4, lane);                         // layer-0 group 0 (re-read per view)
        const int bv = b * p.V + v;
        const float* E = p.einv + 16 * bv;
        const float* K = p.k4 + 16 * bv;
        float cam[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) cam[r] = row_dot4(E, r, wx, wy, wz, 1.0f);
        float pxl, pyl;
        pixel_from_cam(K, cam, &pxl, &pyl);
        const Taps tp = bilinear_taps(pxl, pyl, p.H, p.W);
        const int tl = (bv * p.H + tp.y0) * p.W + tp.x0;
        const long vrow = ((long)bv * p.R + (ray - b * p.R)) * p.S + sidx;       // row in a (B*V,R,S,..) tensor
        if (valid && h == 0) {
            if (p.tap_idx) {
                int4 t4 = make_int4(tl, tl + 1, tl + p.W, tl + p.W + 1);
                *reinterpret_cast<int4*>(p.tap_idx + 4 * vrow) = t4;
            }
            if (p.pix) {
                p.pix[2 * vrow + 0] = pxl;
                p.pix[2 * vrow + 1] = pyl;
            }
        }

        // ---- layer 0: Dense 379 -> 128 on [PE(cam xyz) | PE(cam dir) | 2*rgb-1 | features] ----
        // accumulator seed = b0 + W0_dir^T PE(cam dir) of this lane's (view, ray), from dir_bias_kernel
        bias_to_acc<false>(p.dir_bias + 128 * ((long)bv * p.R + (ray - b * p.R)), h, x);
        float pe[32];                                     // B operands of k-steps 0..31 (lower half sin, upper cos)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float a0 = cam[d] * 3.14159274101257324f;
            float sk = 0.0f, ck = 0.0f;
#pragma unroll
            for (int k = 0; k < kNFreq; ++k) {
                // fl32(x * fl32(pi*2^k)) == 2^k * fl32(x * fl32(pi)) exactly, so octave k is the k-fold double
                // angle of octave 0.  Accurate sin/cos at k = 0 and k = 5, double-angle steps in between (error
                // x16 at most: ~2e-6 absolute against the 1e-4 bar; the op-level position_encoding is exact).
                if (!MV_PE_RECUR || k == 0 || k == 5) {
#if MV_ABL_PE
                    sk = a0; ck = a0 + 1.0f;
#else
                    sincos_f32(a0 * (float)(1 << k), &sk, &ck);
#endif
                } else {
                    const float s2 = sk + sk;
                    const float cn = fmaf(-s2, sk, 1.0f);              // cos 2t = 1 - 2 sin^2 t
                    sk = s2 * ck;                                      // sin 2t = 2 sin t cos t
                    ck = cn;
                }
                pe[d * 10 + k] = h ? ck : sk;
            }
        }
        {   // rgb taps of this lane's own sample, normalised 2*img-1 before the lerp (model_v0.py:120)
            const float* img = p.images + 3 * (long)tl;
            float rgbv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float a = img[c] * 2.0f - 1.0f, bq = img[3 + c] * 2.0f - 1.0f;
                const float cq = img[3 * p.W + c] * 2.0f - 1.0f, dq = img[3 * p.W + 3 + c] * 2.0f - 1.0f;
                rgbv[c] = bilerp(a, bq, cq, dq, tp.ax, tp.ay);
            }
            pe[30] = h ? rgbv[1] : rgbv[0];
            pe[31] = h ? 0.0f : rgbv[2];
        }
#pragma unroll
        for (int gq = 0; gq < 8; ++gq) {
            const float bb[4] = {pe[4 * gq], pe[4 * gq + 1], pe[4 * gq + 2], pe[4 * gq + 3]};
            if (kProj && gq == 7 - ws_jump_lead()) ws.pos = kPackHidden * 4;   // the stream skips the 32 feature groups
            mfma_step(ws, bb, x);
        }
        if (kProj) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // earlier reads of `stage` are done
            const f32x4* tbase = reinterpret_cast<const f32x4*>(p.texel_table) + j;     // 32 float4 per texel row
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int src = 2 * it + h;                           // sample whose 4 table rows this half-wave loads
                const int tls = __shfl(tl, src);
                const float axs = __shfl(tp.ax, src), ays = __shfl(tp.ay, src);
                const f32x4* f = tbase + (long)tls * 32;
                const f32x4 vtl = f[0], vtr = f[32], vbl = f[(long)p.W * 32], vbr = f[(long)p.W * 32 + 32];
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float top = fmaf(axs, vtr[c] - vtl[c], vtl[c]);
                    const float bot = fmaf(axs, vbr[c] - vbl[c], vbl[c]);
                    o[c] = fmaf(ays, bot - top, top);
                }
                *reinterpret_cast<f32x4*>(stage + stage_offset(src, j)) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // table rows are in accumulator order [h][nb][r]: chunk 16h + 4nb + q of row j = registers 4q..4q+3 of block nb
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(stage + stage_offset(j, 16 * h + 4 * nb + q));
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[nb][4 * q + c] += t4[c];
                }
        }
#pragma unroll 1
        for (int hf = 0; hf < (kProj ? 0 : 2); ++hf) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // earlier reads of `stage` are done
            const f32x4* fbase = reinterpret_cast<const f32x4*>(p.features) + hf * 32 + j;
#pragma unroll 4
            for (int it = 0; it < (MV_ABL_GATHER ? 0 : 16); ++it) {
                const int src = 2 * it + h;                           // sample whose row this half-wave loads
                const int tls = __shfl(tl, src);
                const float axs = __shfl(tp.ax, src), ays = __shfl(tp.ay, src);
                const f32x4* f = fbase + (long)tls * 64;
                const f32x4 vtl = f[0], vtr = f[64], vbl = f[(long)p.W * 64], vbr = f[(long)p.W * 64 + 64];
                f32x4 o;
#if MV_FMA_LERP
                // same bilinear form as tfa (lerp x, then y), each lerp as one FMA after the difference
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float top = fmaf(axs, vtr[c] - vtl[c], vtl[c]);
                    const float bot = fmaf(axs, vbr[c] - vbl[c], vbl[c]);
                    o[c] = fmaf(ays, bot - top, top);
                }
#else
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = bilerp(vtl[c], vtr[c], vbl[c], vbr[c], axs, ays);
#endif
                *reinterpret_cast<f32x4*>(stage + stage_offset(src, j)) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // writes landed (same-wave DS order)
#pragma unroll
            for (int gg = 0; gg < 16; ++gg) {
                const f32x4 bv4 = *reinterpret_cast<const f32x4*>(stage + stage_offset(j, 2 * gg + h));
                const float bb[4] = {bv4[0], bv4[1], bv4[2], bv4[3]};
                mfma_step(ws, bb, x);
            }
        }

        // training mode: view tile index (all 32 samples of a tile share b because R*S % 32 == 0 when V > 1)
        const long vtile = kMultiView ? ((long)bv * (p.n_tiles / p.B) + (tile - (long)b * (p.n_tiles / p.B))) : tile;
        if (kStash) store_tl(p.stash, vtile, j, h, x);       // per-view slot 0: layer-0 output
        // ---- per-view feature blocks (layers.py:365-366); optional complete_output taps (:376-377) ----
        const long vslot = (long)p.B * p.V * p.R * p.S * 128;
        if (p.acts_view && valid) store_acc(p.acts_view + 128 * vrow, h, x);
#pragma unroll 1
        for (int bi = 0; bi < 3; ++bi) {
            resnet_block<kStash>(ws, net + kPackBHidden + 256 * bi, h, x, hid,
                                 kStash ? p.stash + (1 + 2 * bi) * p.stash_stride : nullptr, p.stash_stride, vtile, j,
                                 bi < 2);                               // per-view slot 6 = x3 is not written: nothing reads it
            if (p.acts_view && valid) store_acc(p.acts_view + (bi + 1) * vslot + 128 * vrow, h, x);
        }

        if (kMultiView) {                                             // reduce_mean over views (layers.py:368-370)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) xsum[nb] = (v == 0) ? x[nb] : xsum[nb] + x[nb];
        }
    }
    if (kMultiView) {
        const float nv = (float)p.V;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) x[nb] = xsum[nb] / nv;
    }

    // ---- fusion blocks (layers.py:373-374); the stream continues into hidden layer 6 ----
    if (kStash) store_tl(p.stash_fused, tile, j, h, x);                           // fused slot 0: the view mean
    if (p.acts_fused && valid) store_acc(p.acts_fused + 128 * g, h, x);           // the view mean
#pragma unroll 1
    for (int bi = 3; bi < 6; ++bi) {
        resnet_block<kStash>(ws, net + kPackBHidden + 256 * bi, h, x, hid,
                             kStash ? p.stash_fused + (1 + 2 * (bi - 3)) * p.stash_fused_stride : nullptr,
                             p.stash_fused_stride, tile, j);
        if (p.acts_fused && valid) store_acc(p.acts_fused + (bi - 2) * p.total * 128 + 128 * g, h, x);
    }

    if (p.embedding && valid) store_acc(p.embedding + 128 * g, h, x);   // optional: trunk output (layers.py:379)

    // ---- read-out: Dense 128 -> 4 on relu(x), sigmoid / softplus (layers.py:392-397) ----
    // on the vector ALU: 64 features per lane x 4 outputs = 256 FMAs, then one cross-half add - the MFMA form pads the
    // 4 outputs to a 32-row tile (64 MFMAs = 4096 matrix-pipe cycles per tile for 512 useful MACs per sample)
    f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
    {
        const f32x4* wr = reinterpret_cast<const f32x4*>(net + kPackWrPlain) + 4 * h;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 w4 = wr[32 * nb + 8 * q + c];
                    const float a = fmaxf(x[nb][4 * q + c], 0.0f);
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = fmaf(a, w4[k], o[k]);
                }
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (o[k] + __shfl_xor(o[k], 32)) + net[kPackBr + k];
    }
    if (valid && h == 0) {
        f32x4 out;
        out[0] = sigmoid_f32(o[0]);
        out[1] = sigmoid_f32(o[1]);
        out[2] = sigmoid_f32(o[2]);
        out[3] = softplus_f32(o[3]);
        *reinterpret_cast<f32x4*>(p.rgbs + 4 * g) = out;
    }
#if !MV_PERSIST
    break;
#endif
  }
}

// ---- per-(view, ray) layer-0 seed: b0 + W0[60:120]^T PE(cam dir)  (the direction is constant along a ray) ----
// One wavefront per (b*V+v, ray).  Output in accumulator order [h][nb][r] so the field kernel loads it like a bias.
__global__ __launch_bounds__(256) void dir_bias_kernel(FieldParams p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // over B*V*R
    if (row >= (long)p.B * p.V * p.R) return;
    const int bv = (int)(row / p.R);
    const long ray = (long)(bv / p.V) * p.R + (row - (long)bv * p.R);
    const float dx = p.rays_d[3 * ray + 0], dy = p.rays_d[3 * ray + 1], dz = p.rays_d[3 * ray + 2];
    const float* E = p.einv + 16 * bv;
    // lane m < 60 evaluates PE feature m = d*20 + 2k + f (nerf_utils.py:124 layout) of cam dir (Q3: w = 1)
    const int m = lane < 60 ? lane : 59;
    const int d = m / 20, k = (m % 20) >> 1, f = m & 1;
    const float cd = row_dot4(E, d, dx, dy, dz, 1.0f);
    float sv, cv;
    sincos_f32(cd * (3.14159274101257324f * (float)(1 << k)), &sv, &cv);
    const float mine = f ? cv : sv;
    const float* wd = p.net + kPackW0Dir;
    float a0 = p.net[kPackB0Plain + lane], a1 = p.net[kPackB0Plain + 64 + lane];
    for (int mm = 0; mm < 60; ++mm) {
        const float pv = __shfl(mine, mm);
        a0 = fmaf(pv, wd[mm * 128 + lane], a0);
        a1 = fmaf(pv, wd[mm * 128 + 64 + lane], a1);
    }
    float* out = p.dir_bias + 128 * row;
    out[acc_slot(lane)] = a0;
    out[acc_slot(64 + lane)] = a1;
}

// ---- texel table: T[b*V+v][y][x][.] = W0[123:379]^T features[b,v,y,x,:] in accumulator order [h][nb][r] ----
// One workgroup per 32 texels, one wavefront per output block nb.  The 32 feature rows (32 KiB) are staged in LDS
// with coalesced 16-byte loads (XOR-swizzled: the B operand read "lane = texel" is then conflict-free); every wave
// runs 32 steps of 4 k-steps: A = feature groups 8..39 of the packed layer-0 kernel (chunk (g, nb)), B = channels
// 8q + 4h + {0..3} of this lane's texel.  Two accumulators alternate so consecutive MFMAs do not depend on each other.
// A second net (net1 / table1, workgroups of 8 waves) shares the staged rows: coarse and fine tables from ONE read of
// the feature maps.
__global__ __launch_bounds__(512) void project_texels_kernel(const float* __restrict__ features,
                                                             const float* __restrict__ net0, const float* __restrict__ net1,
                                                             long n_texels, float* __restrict__ table0,
                                                             float* __restrict__ table1) {
    __shared__ __attribute__((aligned(16))) f32x4 srow[32 * 64];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = wv & 3;
    const float* net = wv < 4 ? net0 : net1;
    float* table = wv < 4 ? table0 : table1;
    const long t0 = (long)blockIdx.x * 32;
    const f32x4* fsrc = reinterpret_cast<const f32x4*>(features);
    const int nthreads = blockDim.x;
    for (int m = 0; m < 2048 / nthreads; ++m) {
        const int idx = tid + nthreads * m;                 // float4 index inside the 32 x 64 block
        const int row = idx >> 6, chunk = idx & 63;
        long t = t0 + row;
        if (t >= n_texels) t = n_texels - 1;
        srow[row * 64 + (chunk ^ (row & 15))] = fsrc[t * 64 + chunk];
    }
    __syncthreads();
    const f32x4* w = reinterpret_cast<const f32x4*>(net) + ((long)kL0GroupFeat * 4 + nb) * 64 + lane;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.0f;
        acc1[r] = 0.0f;
    }
#pragma unroll 4
    for (int q = 0; q < 32; q += 2) {
        const f32x4 a0 = w[(long)q * 256], a1 = w[(long)(q + 1) * 256];
        const f32x4 b0 = srow[j * 64 + ((2 * q + h) ^ (j & 15))], b1 = srow[j * 64 + ((2 * q + 2 + h) ^ (j & 15))];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = mfma(a0[e], b0[e], acc0);
            acc1 = mfma(a1[e], b1[e], acc1);
        }
    }
    const long t = t0 + j;
    if (t < n_texels) {
        f32x4* out = reinterpret_cast<f32x4*>(table + 128 * t + 64 * h + 16 * nb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc0[4 * q] + acc1[4 * q], acc0[4 * q + 1] + acc1[4 * q + 1], acc0[4 * q + 2] + acc1[4 * q + 2],
                             acc0[4 * q + 3] + acc1[4 * q + 3]};
            out[q] = v;
        }
    }
}

hipError_t launch_project_texels(const float* features, const float* packed_net, const float* packed_net1, long n_texels,
                                 float* table, float* table1, hipStream_t stream) {
    hipLaunchKernelGGL(project_texels_kernel, dim3((unsigned)((n_texels + 31) / 32)), dim3(packed_net1 ? 512 : 256), 0, stream,
                       features, packed_net, packed_net1, n_texels, table, table1);
    return hipGetLastError();
}

// ---- weight packing -------------------------------------------------------------------------
__global__ void pack_net_kernel(const float* __restrict__ src, float* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kPackTotal) return;
    float val = 0.0f;
    // position inside a 1 KiB chunk: [lane][e]
    const int e = idx % 4, lane = (idx % kChunkFloats) / 4, i = lane & 31, h = lane >> 5;
    if (idx < kPackHidden) {                                          // layer-0 kernel
        const int G = idx / kGroupFloats, nb = (idx % kGroupFloats) / kChunkFloats;
        int row = -1;
        if (G < kL0GroupFeat) {
            const int ks = 4 * G + e;
            if (ks < 30) row = (ks / 10) * 20 + 2 * (ks % 10) + h;       // sin | cos of PE(xyz)
            else if (ks == 30) row = 120 + h;                            // r | g
            else row = h ? -1 : 122;                                     // b | 0
        } else {
            const int q = G - kL0GroupFeat;
            row = 123 + 128 * (q / 16) + 8 * (q % 16) + 4 * h + e;
        }
        if (row >= 0) val = src[kKerasW0 + row * kHidden + 32 * nb + i];
    } else if (idx < kPackWr) {                                       // 12 hidden Dense kernels
        const int q = idx - kPackHidden;
        const int layer = q / kHiddenWFloats, r = q % kHiddenWFloats;
        const int wsrc = kKerasBlocks + (layer / 2) * kKerasBlockStride + (layer % 2) * (kHidden * kHidden + kHidden);
        const int grp = r / kGroupFloats, nb = (r % kGroupFloats) / kChunkFloats;
        const int kb = grp / 4, t = grp % 4;
        val = src[wsrc + (32 * kb + 8 * t + 4 * h + e) * kHidden + 32 * nb + i];
    } else if (idx < kPackB0) {                                       // read-out kernel, rows >= 4 zero
        const int chunk = (idx - kPackWr) / kChunkFloats;
        const int kb = chunk / 4, t = chunk % 4;
        if (i < 4) val = src[kKerasWr + (32 * kb + 8 * t + 4 * h + e) * 4 + i];
    } else if (idx < kPackBr) {                                       // biases, [h][nb][r] per layer
        const int q = idx - kPackB0;
        const int layer = q / 128, qq = q % 128;                      // 0 = layer 0, 1..12 = hidden
        const int feat = 32 * ((qq % 64) / 16) + acc_row(qq % 16, qq / 64);
        if (layer == 0) val = src[kKerasB0 + feat];
        else {
            const int l = layer - 1;
            val = src[kKerasBlocks + (l / 2) * kKerasBlockStride + (l % 2) * (kHidden * kHidden + kHidden) +
                      kHidden * kHidden + feat];
        }
    } else if (idx < kPackBr + 4) {
        val = src[kKerasBr + (idx - kPackBr)];
    } else if (idx >= kPackW0Dir && idx < kPackB0Plain) {             // plain W0 rows 60..119 (PE of cam dir)
        val = src[kKerasW0 + 60 * kHidden + (idx - kPackW0Dir)];
    } else if (idx >= kPackWrPlain) {
        val = src[kKerasWr + (idx - kPackWrPlain)];
    } else if (idx >= kPackB0Plain) {
        val = src[kKerasB0 + (idx - kPackB0Plain)];
    }
    dst[idx] = val;
}

hipError_t launch_pack_net(const float* net_keras, float* packed, hipStream_t stream) {
    const int threads = 256;
    hipLaunchKernelGGL(pack_net_kernel, dim3((kPackTotal + threads - 1) / threads), dim3(threads), 0, stream,
                       net_keras, packed);
    return hipGetLastError();
}

// Ticket counters for the persistent kernel: a small ring of device words so that launches in flight
// on different streams never share one; the slot is zeroed on the launch stream right before use.
constexpr int kCounterSlots = 64;
__device__ unsigned int g_tile_counters[kCounterSlots];

namespace {
struct DeviceInfo {
    int cus = 0;
    unsigned int* counters = nullptr;
    bool attr_set = false;
};
DeviceInfo g_dev[16];
#if MV_PERSIST
std::atomic<unsigned> g_launch_seq{0};
#endif
std::mutex g_dev_mutex;
}  // namespace

hipError_t launch_dir_bias(const FieldParams& p, hipStream_t stream) {
    const long rows = (long)p.B * p.V * p.R;
    hipLaunchKernelGGL(dir_bias_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_field_eval(const FieldParams& p_in, hipStream_t stream) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    DeviceInfo& di = g_dev[dev];
    {
        std::lock_guard<std::mutex> lock(g_dev_mutex);
        if (!di.attr_set) {
            hipDeviceProp_t prop;
            if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
            di.cus = prop.multiProcessorCount;
            if ((e = hipGetSymbolAddress(reinterpret_cast<void**>(&di.counters), HIP_SYMBOL(g_tile_counters))) != hipSuccess) return e;
            const int lds_sv = MV_WAVES * kTile * kStageRow * 4, lds_mv = 4 * kTile * kStageRow * 4;
            const struct { const void* fn; int bytes; } kernels[] = {
                {reinterpret_cast<const void*>(&field_eval_kernel<false, false, false>), lds_sv},
                {reinterpret_cast<const void*>(&field_eval_kernel<false, false, true>), lds_sv},
                {reinterpret_cast<const void*>(&field_eval_kernel<false, true, false>), lds_sv},
                {reinterpret_cast<const void*>(&field_eval_kernel<false, true, true>), lds_sv},
                {reinterpret_cast<const void*>(&field_eval_kernel<true, false, false>), lds_mv},
                {reinterpret_cast<const void*>(&field_eval_kernel<true, false, true>), lds_mv},
                {reinterpret_cast<const void*>(&field_eval_kernel<true, true, false>), lds_mv},
                {reinterpret_cast<const void*>(&field_eval_kernel<true, true, true>), lds_mv},
            };
            for (const auto& k : kernels)
                if ((e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, k.bytes)) != hipSuccess) return e;
            di.attr_set = true;
        }
    }
    FieldParams p = p_in;
    p.tile_counter = nullptr;
#if MV_PERSIST
    p.tile_counter = di.counters + (g_launch_seq.fetch_add(1) % kCounterSlots);
    if ((e = launch_zero(p.tile_counter, sizeof(unsigned int), stream)) != hipSuccess) return e;
#endif
    if ((e = launch_dir_bias(p, stream)) != hipSuccess) return e;
    const int waves = p.V > 1 ? 4 : MV_WAVES;
    const long want = (p.n_tiles + waves - 1) / waves;
    const long resident = (long)di.cus * (p.V > 1 ? 1 : 8 / MV_WAVES);   // workgroups that fit at once
    const unsigned wgs = (unsigned)((MV_PERSIST && want > resident) ? resident : want);
    const size_t lds_bytes = (size_t)waves * kTile * kStageRow * 4;
    if (p.V > 1) {
        if (p.stash) {
            if (((long)p.R * p.S) % 32 != 0) return hipErrorInvalidValue;     // tiles must not straddle scenes
            if (p.texel_table) hipLaunchKernelGGL((field_eval_kernel<true, true, true>), dim3(wgs), dim3(256), lds_bytes, stream, p);
            else hipLaunchKernelGGL((field_eval_kernel<true, true, false>), dim3(wgs), dim3(256), lds_bytes, stream, p);
        } else if (p.texel_table) {
            hipLaunchKernelGGL((field_eval_kernel<true, false, true>), dim3(wgs), dim3(256), lds_bytes, stream, p);
        } else {
            hipLaunchKernelGGL((field_eval_kernel<true, false, false>), dim3(wgs), dim3(256), lds_bytes, stream, p);
        }
    } else if (p.stash) {
        if (p.texel_table) hipLaunchKernelGGL((field_eval_kernel<false, true, true>), dim3(wgs), dim3(64 * MV_WAVES), lds_bytes, stream, p);
        else hipLaunchKernelGGL((field_eval_kernel<false, true, false>), dim3(wgs), dim3(64 * MV_WAVES), lds_bytes, stream, p);
    } else if (p.texel_table) {
        hipLaunchKernelGGL((field_eval_kernel<false, false, true>), dim3(wgs), dim3(64 * MV_WAVES), lds_bytes, stream, p);
    } else {
        hipLaunchKernelGGL((field_eval_kernel<false, false, false>), dim3(wgs), dim3(64 * MV_WAVES), lds_bytes, stream, p);
    }
    return hipGetLastError();
}

}  // namespace mvnerf
