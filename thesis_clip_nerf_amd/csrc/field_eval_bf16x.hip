// bf16 field kernel, round-3 form ("bf16x"): the mapping of field_eval_split16.hip (v_mfma_f32_16x16x32_bf16, lane (n, g) holds features
// 16 rb + 4g + {0..3} of samples n and 16 + n, weights by LDS-DMA through a 3-slot ring, no staging registers) with ONE bf16 product per
// block instead of six, for the texel-table form of the bf16 path (mvnerf_field_eval_bf16 with texel_table: BASELINE.json configs 3 / 5;
// same arithmetic as field_eval_bf16.hip's table variant: bf16 weights and Dense inputs, fp32 accumulation, geometry / seed / table / biases /
// residual path / read-out in fp32).
//
// What is different from field_eval_bf16_kernel, and why: that kernel spends 62 % of its time outside the matrix pipe - 30
// barrier-separated 16 KiB segments per tile, each with its own exposed relu / convert / bias phase.  Here a ring slot is a WHOLE layer (K = 128:
// 4 t-steps x 8 row blocks x 1 KiB = 32 KiB, 13 barriers per tile), and a layer runs row block by row block: all four t-steps of row block rb
// (8 MFMAs) finish that block's accumulators, so its bias row is applied just before and its relu + bf16 conversion - the next layer's B
// operand - just after, while the following row blocks' MFMAs run.  No layer boundary has vector work left outside the MFMA shadow except
// the last pair of row blocks.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>

#include "mvnerf_kernels.h"
#include "mvnerf_math.h"
#include "mvnerf_mfma.h"

namespace mvnerf {

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x8 = __attribute__((ext_vector_type(8))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// ---- weight stream: 13 positions of 32 chunks (1 KiB = [lane][8 bf16]); chunk (t, rb) of a position at index 8 t + rb ------------------
//   position 0      : layer 0, PE(cam xyz) + rgb: t = 0, 1 (slot e = 8 t + jj of lane group g: field_eval_split16.hip, s16_pe_row); t = 2, 3 zero
//   position 1 + l  : hidden layer l: input 32 t + 4 g + jj (jj < 4) | 32 t + 16 + 4 g + jj - 4
// A[i = l & 15][k = 8 g + jj] = W[input][16 rb + i]
constexpr int kXPosChunks = 32, kXPositions = 13;
constexpr int kXChunks = kXPositions * kXPosChunks;                 // 416 KiB

__host__ __device__ constexpr int x_pe_row(int g, int e) {
    return g < 3 ? 20 * g + e : (e < 12 ? 20 * (e >> 2) + 2 * (8 + ((e >> 1) & 1)) + (e & 1) : (e < 15 ? 120 + (e - 12) : -1));
}

__global__ void pack_net_bf16x_kernel(const float* __restrict__ src, __bf16* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kXChunks * 512) return;
    const int chunk = idx / 512, lane = (idx % 512) / 8, jj = idx % 8;
    const int i = lane & 15, g = lane >> 4;
    const int pos = chunk / kXPosChunks, t = (chunk % kXPosChunks) / 8, rb = chunk % 8;
    float val = 0.0f;
    if (pos == 0) {
        if (t < 2) {
            const int row = x_pe_row(g, 8 * t + jj);
            if (row >= 0) val = src[kKerasW0 + row * kHidden + 16 * rb + i];
        }
    } else {
        const int layer = pos - 1;
        const int f = 32 * t + (jj < 4 ? 4 * g + jj : 16 + 4 * g + (jj - 4));
        const int wsrc = kKerasBlocks + (layer / 2) * kKerasBlockStride + (layer % 2) * (kHidden * kHidden + kHidden);
        val = src[wsrc + f * kHidden + 16 * rb + i];
    }
    dst[idx] = (__bf16)val;
}

__device__ __forceinline__ f32x4 mfma_x(f32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct BX {
    u32x4 t[4];         // one column block's B operands for the 4 t-steps of a layer: 8 bf16 each
};

// 8 fp32 values -> bf16x8 (round to nearest even), relu on the packed bit pattern (bf16(relu(x)) == relu(bf16(x)))
template <bool kRelu>
__device__ __forceinline__ u32x4 to_bf16x8(const float (&v)[8]) {
    using i32x4 = __attribute__((ext_vector_type(4))) int;
    f32x8 t;
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = v[q];
    i32x4 r = __builtin_bit_cast(i32x4, __builtin_convertvector(t, bf16x8));
    if (kRelu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int w = r[q];
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(w) : "v"(w));
            r[q] = w;
        }
    }
    return __builtin_bit_cast(u32x4, r);
}

// ---- the slot ring: one position (32 KiB) per slot, 3 slots, filled by LDS-DMA --------------------------------------------------------------
constexpr int kXSlots = 3, kXSlotF4 = kXPosChunks * 64;                // float4 per slot

struct RingX {
    const f32x4* w;
    f32x4* base;        // LDS
    int c;              // ring slot of the current position
    int p, P;
    const int* table;   // LDS: stream position (0..12) of every ring position of one tile
    int next_pos;       // stream position the next fetch loads (read one layer ahead)
    int tid, wave;
    f32x4 a0[4];        // A chunks (t = 0..3) of row block 0 of the CURRENT position, read at the end of the previous one
};

// stream position of ring position p of a tile: per view [PE, 6 layers], then 6 fused layers
__device__ __forceinline__ int ringx_stream_pos(int p, int V) {
    if (p < 7 * V) {
        const int q = p % 7;
        return q == 0 ? 0 : q;                       // 0 = PE, 1..6 = per-view layers
    }
    return 7 + (p - 7 * V);                          // fused layers 6..11 -> stream positions 7..12
}

// 32 KiB: four wave-instructions per wave, each 1 KiB (lane l: 16 bytes at wave base + 16 l); wave w covers bytes [1024 w, +1024) of each 8 KiB
__device__ __forceinline__ void ringx_dma(const RingX& r, int stream_pos, int slot) {
    const f32x4* src = r.w + (long)stream_pos * kXSlotF4 + r.tid;
    f32x4* dst = r.base + slot * kXSlotF4 + 64 * r.wave;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 512 * i),
                                         (__attribute__((address_space(3))) void*)(dst + 512 * i), 16, 0, 0);
}

// In the layer at position p (slot c): position p + 2 goes into slot (c + 2) % 3 (the slot of position p - 1, unread since the last
// barrier); the vmcnt(0) in front of the barrier at the layer's end lets it land before it is published.
__device__ __forceinline__ void ringx_fetch(RingX& r) {
    int slot = r.c + 2;
    slot = slot >= kXSlots ? slot - kXSlots : slot;
    ringx_dma(r, r.next_pos, slot);
    int pp = r.p + 3;
    pp = pp >= r.P ? pp - r.P : pp;
    r.next_pos = r.table[pp];
}

__device__ __forceinline__ const f32x4* ringx_cur(const RingX& r) { return r.base + r.c * kXSlotF4; }
__device__ __forceinline__ const f32x4* ringx_nxt(const RingX& r) { return r.base + (r.c + 1 == kXSlots ? 0 : r.c + 1) * kXSlotF4; }

__device__ __forceinline__ void ringx_next(RingX& r) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    r.c = r.c + 1 == kXSlots ? 0 : r.c + 1;
    r.p = r.p + 1 == r.P ? 0 : r.p + 1;
}

// position of feature 16 rb + 4 g (+ 0..3) inside a 128-float vector in the 32x32 accumulator order [h][nb][r] (biases, seed, table rows)
__device__ __forceinline__ int perm_f4x(int rb, int g) {
    return ((g & 1) * 64 + (rb >> 1) * 16 + 4 * (2 * (rb & 1) + (g >> 1))) >> 2;          // in float4 units
}

// One layer (one ring position): acc[rb][cb] (+)= bias row, += sum over kT t-steps of A(t, rb)^T b[cb].t[t], row block by row block.
//   kBias 0: the accumulators already hold their start value; 1: acc = bias row (hid = b1 + ...); 2: acc += bias row (x += b2, then + ...)
//   kNext : the NEXT layer's B operands bn = bf16(relu(acc)) are produced here: t-step t' of bn needs row blocks 2 t' and 2 t' + 1, which are
//           final two groups before they are converted (their MFMAs have drained by then); the last pair is converted behind the loop.
template <int kT, int kBias, bool kNext>
__device__ __forceinline__ void layer_x(RingX& ring, int lane, int g, const BX (&b)[2], f32x4 (&acc)[8][2], const float* __restrict__ bias,
                                        BX (&bn)[2]) {
    const f32x4* cur = ringx_cur(ring) + lane;
    const f32x4* nxt = ringx_nxt(ring) + lane;
    const f32x4* bias4 = reinterpret_cast<const f32x4*>(bias);
    f32x4 a[4];
#pragma unroll
    for (int t = 0; t < kT; ++t) a[t] = ring.a0[t];
    f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
    if (kBias) bv = bias4[perm_f4x(0, g)];
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
        f32x4 an[4], bvn = bv;
        if (rb < 7) {
#pragma unroll
            for (int t = 0; t < kT; ++t) an[t] = cur[(t * 8 + rb + 1) * 64];
            if (kBias) bvn = bias4[perm_f4x(rb + 1, g)];
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) an[t] = nxt[(t * 8) * 64];           // the next position's row block 0 (published two barriers ago)
        }
        if (kBias == 1) {
            acc[rb][0] = bv;
            acc[rb][1] = bv;
        } else if (kBias == 2) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float r = acc[rb][cb][c];
                    asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(bv[c]));          // scalar adds (no v_pk_add_f32 beside MFMAs)
                    acc[rb][cb][c] = r;
                }
        }
        if (kNext && rb >= 2 && (rb & 1) == 0) {
            const int tn = rb / 2 - 1;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const float v8[8] = {acc[2 * tn][cb][0], acc[2 * tn][cb][1], acc[2 * tn][cb][2], acc[2 * tn][cb][3],
                                     acc[2 * tn + 1][cb][0], acc[2 * tn + 1][cb][1], acc[2 * tn + 1][cb][2], acc[2 * tn + 1][cb][3]};
                u32x4 r = to_bf16x8<true>(v8);
                asm volatile("" : "+v"(r));                                   // pinned in its group
                bn[cb].t[tn] = r;
            }
        }
#pragma unroll
        for (int t = 0; t < kT; ++t)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma_x(a[t], b[cb].t[t], acc[rb][cb]);
        __builtin_amdgcn_sched_barrier(0);
        if (rb == 0) {
            ringx_fetch(ring);                                                // the weights of two layers ahead
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (rb == 7 || t < kT) a[t] = an[t];
        bv = bvn;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) ring.a0[t] = a[t];
    if (kNext) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float v8[8] = {acc[6][cb][0], acc[6][cb][1], acc[6][cb][2], acc[6][cb][3], acc[7][cb][0], acc[7][cb][1], acc[7][cb][2], acc[7][cb][3]};
            bn[cb].t[3] = to_bf16x8<true>(v8);
        }
    }
}

// all four t-steps of a layer's B operands from an activation array (the exposed form: after the texel-table add, after the view mean)
__device__ __forceinline__ void operands_x(const f32x4 (&in)[8][2], BX (&b)[2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float v8[8] = {in[2 * t][cb][0], in[2 * t][cb][1], in[2 * t][cb][2], in[2 * t][cb][3],
                                 in[2 * t + 1][cb][0], in[2 * t + 1][cb][1], in[2 * t + 1][cb][2], in[2 * t + 1][cb][3]};
            b[cb].t[t] = to_bf16x8<true>(v8);
        }
}

constexpr int kXStageRowBytes = 128;        // per staged sample row: 32 fp32 channels (one of four passes over a 128-float table row)
constexpr int kXMaxPositions = 64;

struct SampleGeoX {
    int ray, sidx, b;
    float wx, wy, wz;
    bool valid;
};

// Texel-table form only (p.texel_table set).  kAux: tap_idx / embedding / acts_fused compiled in.
template <bool kMultiView, bool kAux>
__global__ __launch_bounds__(512, 2) void field_eval_bf16x_kernel(FieldParams p, const f32x4* __restrict__ wx) {
    constexpr int kW = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x[];
    constexpr int kRingBytes = kXSlots * kXSlotF4 * 16;                     // 96 KiB
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* stage = smem_x + kRingBytes + wave * (32 * kXStageRowBytes);     // 4 KiB per wave
    float* net = reinterpret_cast<float*>(smem_x + kRingBytes + kW * 32 * kXStageRowBytes) - kPackB0;
    for (int i = tid; i < kPackBr + 8 - kPackB0; i += 64 * kW) net[kPackB0 + i] = p.net[kPackB0 + i];
    int* table = reinterpret_cast<int*>(smem_x + kRingBytes + kW * 32 * kXStageRowBytes + (kPackBr + 8 - kPackB0) * 4);
    float* wr_plain = reinterpret_cast<float*>(table + kXMaxPositions);
    for (int i = tid; i < 512; i += 64 * kW) wr_plain[i] = p.net[kPackWrPlain + i];

    RingX ring;
    ring.w = wx;
    ring.base = reinterpret_cast<f32x4*>(smem_x);
    ring.c = 0;
    ring.p = 0;
    ring.P = 7 * p.V + 6;
    ring.tid = tid;
    ring.wave = wave;
    for (int i = tid; i < ring.P; i += 64 * kW) table[i] = ringx_stream_pos(i, p.V);
    ring.table = table;
    __syncthreads();
    ringx_dma(ring, table[0], 0);                                           // prologue: positions 0 and 1 into slots 0 and 1
    ringx_dma(ring, table[1], 1);
    ring.next_pos = table[2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) ring.a0[t] = ringx_cur(ring)[(t * 8) * 64 + lane];

    const long n_groups = (p.n_tiles + kW - 1) / kW;
    for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        long tile = grp * kW + wave;
        const bool tile_ok = tile < p.n_tiles;
        if (!tile_ok) tile = p.n_tiles - 1;                               // idle waves shadow the last tile, no stores
        // loop-invariant scalars and lane coordinates re-read through an empty asm (see field_eval_split16.hip: keeps the compiler from
        // carrying dozens of hoisted per-lane constants through the tile)
        int pS = p.S, pR = p.R, pH = p.H, pW = p.W, gl = g, nl = n;
        asm volatile("" : "+s"(pS), "+s"(pR), "+s"(pH), "+s"(pW));
        asm volatile("" : "+v"(gl), "+v"(nl));
        SampleGeoX sg[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            long gi = tile * 32 + 16 * cb + nl;
            sg[cb].valid = tile_ok && gi < p.total;
            if (gi >= p.total) gi = p.total - 1;
            const int ray = (int)((unsigned)gi / (unsigned)pS);
            sg[cb].ray = ray;
            sg[cb].sidx = (int)gi - ray * pS;
            sg[cb].b = (int)((unsigned)ray / (unsigned)pR);
            const float ox = p.rays_o[3 * ray + 0], oy = p.rays_o[3 * ray + 1], oz = p.rays_o[3 * ray + 2];
            const float dx = p.rays_d[3 * ray + 0], dy = p.rays_d[3 * ray + 1], dz = p.rays_d[3 * ray + 2];
            const float zz = p.z[gi];
            sg[cb].wx = ox + zz * dx;
            sg[cb].wy = oy + zz * dy;
            sg[cb].wz = oz + zz * dz;
        }

        f32x4 x[8][2], hid[8][2];
        f32x4 xsum[kMultiView ? 8 : 1][2];
        BX bop[2], bnx[2];

        for (int v = 0; v < p.V; ++v) {
            int tl[2];
            float ax[2], ay[2];
            float pe[2][16];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int bv = sg[cb].b * p.V + v;
                const float* E = p.einv + 16 * bv;
                float cam[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) cam[r] = row_dot4(E, r, sg[cb].wx, sg[cb].wy, sg[cb].wz, 1.0f);
                float pxl, pyl;
                pixel_from_cam(p.k4 + 16 * bv, cam, &pxl, &pyl);
                const Taps tp = bilinear_taps(pxl, pyl, pH, pW);
                tl[cb] = (bv * pH + tp.y0) * pW + tp.x0;
                ax[cb] = tp.ax;
                ay[cb] = tp.ay;
                if (kAux && p.tap_idx && sg[cb].valid && gl == 0) {
                    const long vrow = ((long)bv * pR + (sg[cb].ray - sg[cb].b * pR)) * pS + sg[cb].sidx;
                    int4 t4 = make_int4(tl[cb], tl[cb] + 1, tl[cb] + pW, tl[cb] + pW + 1);
                    *reinterpret_cast<int4*>(p.tap_idx + 4 * vrow) = t4;
                }
                {   // accumulator seed = b0 + W0_dir^T PE(cam dir) of this (view, ray) (dir_bias_kernel, fp32)
                    const f32x4* seed = reinterpret_cast<const f32x4*>(p.dir_bias + 128 * ((long)bv * pR + (sg[cb].ray - sg[cb].b * pR)));
#pragma unroll
                    for (int rb = 0; rb < 8; ++rb) x[rb][cb] = seed[perm_f4x(rb, gl)];
                }
                {   // this lane group's 16 of the 64 layer-0 inputs PE(cam xyz) | rgb (field_eval_split16.hip)
                    const float cd = gl == 0 ? cam[0] : (gl == 1 ? cam[1] : cam[2]);
                    const float a0 = (gl < 3 ? cd : cam[0]) * 3.14159274101257324f;
                    const float a1 = (gl < 3 ? cd : cam[1]) * 3.14159274101257324f;
                    const float a2 = cam[2] * 3.14159274101257324f;
                    float s0, c0, s1, c1, s2, c2;
                    sincos_f32(a0 * (gl < 3 ? 1.0f : 256.0f), &s0, &c0);
                    sincos_f32(a1 * (gl < 3 ? 32.0f : 256.0f), &s1, &c1);
                    sincos_f32(a2 * 256.0f, &s2, &c2);
                    auto dbl = [](float& sk, float& ck) {
                        const float t2 = sk + sk;
                        const float cn = fmaf(-t2, sk, 1.0f);
                        sk = t2 * ck;
                        ck = cn;
                    };
                    float va[16], vb[16];
                    {
                        float sk = s0, ck = c0;
                        va[0] = sk; va[1] = ck;
#pragma unroll
                        for (int k = 1; k < 5; ++k) { dbl(sk, ck); va[2 * k] = sk; va[2 * k + 1] = ck; }
                        sk = s1; ck = c1;
                        va[10] = sk; va[11] = ck;
#pragma unroll
                        for (int k = 6; k < 8; ++k) { dbl(sk, ck); va[2 * k] = sk; va[2 * k + 1] = ck; }
                    }
                    {
                        float sk = s0, ck = c0;
                        vb[0] = sk; vb[1] = ck; dbl(sk, ck); vb[2] = sk; vb[3] = ck;
                        sk = s1; ck = c1;
                        vb[4] = sk; vb[5] = ck; dbl(sk, ck); vb[6] = sk; vb[7] = ck;
                        sk = s2; ck = c2;
                        vb[8] = sk; vb[9] = ck; dbl(sk, ck); vb[10] = sk; vb[11] = ck;
                        const float* img = p.images + 3 * (long)tl[cb];
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float a = img[c] * 2.0f - 1.0f, bq = img[3 + c] * 2.0f - 1.0f;
                            const float cq = img[3 * pW + c] * 2.0f - 1.0f, dq = img[3 * pW + 3 + c] * 2.0f - 1.0f;
                            vb[12 + c] = bilerp(a, bq, cq, dq, tp.ax, tp.ay);
                        }
                        vb[15] = 0.0f;
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) pe[cb][e] = gl < 3 ? va[e] : vb[e];
                }
            }

            // ---- position 0: PE(cam xyz) + rgb rows (K = 64: two t-steps) ----
            {
                BX bq[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const float lo[8] = {pe[cb][0], pe[cb][1], pe[cb][2], pe[cb][3], pe[cb][4], pe[cb][5], pe[cb][6], pe[cb][7]};
                    const float hi[8] = {pe[cb][8], pe[cb][9], pe[cb][10], pe[cb][11], pe[cb][12], pe[cb][13], pe[cb][14], pe[cb][15]};
                    bq[cb].t[0] = to_bf16x8<false>(lo);
                    bq[cb].t[1] = to_bf16x8<false>(hi);
                    bq[cb].t[2] = bq[cb].t[0];
                    bq[cb].t[3] = bq[cb].t[0];
                }
                layer_x<2, 0, false>(ring, lane, g, bq, x, nullptr, bnx);
                ringx_next(ring);
            }

            // ---- layer 0's feature rows from the fp32 texel table: 4 passes of 32 floats through the wave-private stage; pass m = table
            // floats [16 m, 16 m + 16) and [64 + 16 m, 64 + 16 m + 16) = row blocks 2 m, 2 m + 1 in the 32x32 accumulator order ----
            asm volatile("" : "+v"(gl), "+v"(nl));
            {
                const int l8 = lane & 7, sub = lane >> 3;                  // 16-byte chunk of the pass slice, row inside a group of 8
                // 8 units (pass m, column block half), the loads of unit u + 1 in flight while unit u is lerped and staged
                f32x4 tv[2][2][4];
                float axs[2][2], ays[2][2];
                auto load_unit = [&](int u, f32x4 (&t)[2][4], float (&axu)[2], float (&ayu)[2]) {
                    const int m = u >> 1, half = u & 1;
                    const f32x4* tbase = reinterpret_cast<const f32x4*>(p.texel_table) + (l8 >> 2) * 16 + 4 * m + (l8 & 3);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int src = 8 * q + sub;                       // sample 16 half + src lives in lane src
                        const int tls = __shfl(half ? tl[1] : tl[0], src);
                        axu[q] = __shfl(half ? ax[1] : ax[0], src);
                        ayu[q] = __shfl(half ? ay[1] : ay[0], src);
                        const f32x4* f = tbase + (long)tls * 32;
                        t[q][0] = f[0];
                        t[q][1] = f[32];
                        t[q][2] = f[(long)pW * 32];
                        t[q][3] = f[(long)pW * 32 + 32];
                    }
                };
                load_unit(0, tv[0], axs[0], ays[0]);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = u >> 1, half = u & 1;
                    if (u < 7) load_unit(u + 1, tv[(u + 1) & 1], axs[(u + 1) & 1], ays[(u + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int row = 16 * half + 8 * q + sub;
                        f32x4 o;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float top = fmaf(axs[u & 1][q], tv[u & 1][q][1][c] - tv[u & 1][q][0][c], tv[u & 1][q][0][c]);
                            const float bot = fmaf(axs[u & 1][q], tv[u & 1][q][3][c] - tv[u & 1][q][2][c], tv[u & 1][q][2][c]);
                            o[c] = fmaf(ays[u & 1][q], bot - top, top);
                        }
                        *reinterpret_cast<f32x4*>(stage + row * kXStageRowBytes + ((l8 ^ (row & 7)) << 4)) = o;
                    }
                    if (half) {
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int rbl = 0; rbl < 2; ++rbl) {
                                const int row = 16 * cb + nl, chunk = 4 * (gl & 1) + 2 * rbl + (gl >> 1);
                                const f32x4 t4 = *reinterpret_cast<const f32x4*>(stage + row * kXStageRowBytes + ((chunk ^ (row & 7)) << 4));
#pragma unroll
                                for (int c = 0; c < 4; ++c) x[2 * m + rbl][cb][c] += t4[c];
                            }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }

            // ---- the three per-view ResNet blocks: 6 positions ----
            operands_x(x, bop);                                              // exposed once per (tile, view): x is new
#pragma unroll 1
            for (int bi = 0; bi < 3; ++bi) {
                const float* bias1 = net + kPackBHidden + 256 * bi;
                layer_x<4, 1, true>(ring, lane, g, bop, hid, bias1, bnx);            // hid = b1 + W1^T relu(x); bnx = bf16(relu(hid))
                ringx_next(ring);
                layer_x<4, 2, true>(ring, lane, g, bnx, x, bias1 + 128, bop);        // x += b2 + W2^T relu(hid); bop = bf16(relu(x))
                ringx_next(ring);
            }
            if (kMultiView) {
#pragma unroll
                for (int rb = 0; rb < 8; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) xsum[rb][cb] = (v == 0) ? x[rb][cb] : xsum[rb][cb] + x[rb][cb];
            }
        }
        if (kMultiView) {
            const float nvw = (float)p.V;
#pragma unroll
            for (int rb = 0; rb < 8; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) x[rb][cb] = xsum[rb][cb] / nvw;
            operands_x(x, bop);                                              // the view mean is new
        }

        // the samples' global indices, recomputed here instead of being carried through the tile
        long grow[2];
        bool gvalid[2];
        {
            int ne = n;
            asm volatile("" : "+v"(ne));
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                long gi = tile * 32 + 16 * cb + ne;
                gvalid[cb] = tile_ok && gi < p.total;
                grow[cb] = gi >= p.total ? p.total - 1 : gi;
            }
        }
        auto store_fused = [&](float* base) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                if (gvalid[cb]) {
                    float* e = base + 128 * grow[cb] + 4 * g;
#pragma unroll
                    for (int rb = 0; rb < 8; ++rb) *reinterpret_cast<f32x4*>(e + 16 * rb) = x[rb][cb];
                }
        };
        if (kAux && p.acts_fused) store_fused(p.acts_fused);                 // complete_output: the view mean
        // ---- fusion blocks: 6 positions ----
#pragma unroll 1
        for (int bi = 3; bi < 6; ++bi) {
            const float* bias1 = net + kPackBHidden + 256 * bi;
            layer_x<4, 1, true>(ring, lane, g, bop, hid, bias1, bnx);
            ringx_next(ring);
            layer_x<4, 2, true>(ring, lane, g, bnx, x, bias1 + 128, bop);
            ringx_next(ring);
            if (kAux && p.acts_fused) store_fused(p.acts_fused + (long)(bi - 2) * p.total * 128);
        }
        if (kAux && p.embedding) store_fused(p.embedding);

        // ---- read-out: Dense 128 -> 4 on relu(x), sigmoid / softplus, on the vector ALU in fp32 (layers.py:392-397) ----
        {
            float o[2][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
            const f32x4* wr = reinterpret_cast<const f32x4*>(wr_plain) + 4 * g;
#pragma unroll
            for (int rb = 0; rb < 8; ++rb)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 w4 = wr[16 * rb + c];
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const float a = fmaxf(x[rb][cb][c], 0.0f);
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[cb][k] = fmaf(a, w4[k], o[cb][k]);
                    }
                }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t = o[cb][k];
                    t = t + __shfl_xor(t, 16);
                    t = t + __shfl_xor(t, 32);
                    o[cb][k] = t + net[kPackBr + k];
                }
            const int cbs = g & 1;
            const float o0 = cbs ? o[1][0] : o[0][0], o1 = cbs ? o[1][1] : o[0][1], o2 = cbs ? o[1][2] : o[0][2], o3 = cbs ? o[1][3] : o[0][3];
            const bool ok = cbs ? gvalid[1] : gvalid[0];
            const long gi = cbs ? grow[1] : grow[0];
            if (ok && g < 2) {
                f32x4 out;
                out[0] = sigmoid_f32(o0);
                out[1] = sigmoid_f32(o1);
                out[2] = sigmoid_f32(o2);
                out[3] = softplus_f32(o3);
                *reinterpret_cast<f32x4*>(p.rgbs + 4 * gi) = out;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // DMA still in flight must land before the LDS is released
}

}  // namespace

size_t packed_net_bf16x_bytes() { return (size_t)kXChunks * 1024; }

hipError_t launch_pack_net_bf16x(const float* net_keras, void* packed16x, hipStream_t st) {
    const int nel = kXChunks * 512;
    hipLaunchKernelGGL(pack_net_bf16x_kernel, dim3((nel + 255) / 256), dim3(256), 0, st, net_keras, static_cast<__bf16*>(packed16x));
    return hipGetLastError();
}

// MVNERF_BF16_KERNEL=layers sends V > 1 here as well (tests; measured 6 % slower than the segment kernel at V = 3: the running view
// sum costs 136 B/lane of scratch on top of 256 registers)
bool field_eval_bf16x_supports(const FieldParams& p) {
    const char* e = getenv("MVNERF_BF16_KERNEL");
    const bool any_v = e && e[0] == 'l';
    return p.texel_table != nullptr && (p.V == 1 || any_v) && 7 * p.V + 6 <= kXMaxPositions && !p.pix && !p.acts_view && !p.stash;
}

hipError_t launch_field_eval_bf16x(const FieldParams& p, const void* packed16x, hipStream_t stream) {
    static std::mutex mtx;
    static bool attr_done[16] = {};
    static int cus[16] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (!field_eval_bf16x_supports(p)) return hipErrorInvalidValue;
    const int lds_bytes = kXSlots * kXSlotF4 * 16 + 8 * 32 * kXStageRowBytes + (kPackBr + 8 - kPackB0) * 4 + kXMaxPositions * 4 + 512 * 4;
    {
        std::lock_guard<std::mutex> lock(mtx);
        if (!attr_done[dev]) {
            hipDeviceProp_t prop;
            if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
            cus[dev] = prop.multiProcessorCount;
            const void* fns[4] = {reinterpret_cast<const void*>(&field_eval_bf16x_kernel<false, false>),
                                  reinterpret_cast<const void*>(&field_eval_bf16x_kernel<false, true>),
                                  reinterpret_cast<const void*>(&field_eval_bf16x_kernel<true, false>),
                                  reinterpret_cast<const void*>(&field_eval_bf16x_kernel<true, true>)};
            for (const void* fn : fns)
                if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess) return e;
            attr_done[dev] = true;
        }
    }
    if ((e = launch_dir_bias(p, stream)) != hipSuccess) return e;
    const long n_groups = (p.n_tiles + 7) / 8;
    const long resident = (long)cus[dev];
    const unsigned wgs = (unsigned)(n_groups < resident ? n_groups : resident);
    const f32x4* w = static_cast<const f32x4*>(packed16x);
    const bool aux = p.tap_idx || p.embedding || p.acts_fused;
    const dim3 grid(wgs), block(512);
    if (p.V > 1) {
        if (aux) hipLaunchKernelGGL((field_eval_bf16x_kernel<true, true>), grid, block, lds_bytes, stream, p, w);
        else hipLaunchKernelGGL((field_eval_bf16x_kernel<true, false>), grid, block, lds_bytes, stream, p, w);
    } else {
        if (aux) hipLaunchKernelGGL((field_eval_bf16x_kernel<false, true>), grid, block, lds_bytes, stream, p, w);
        else hipLaunchKernelGGL((field_eval_bf16x_kernel<false, false>), grid, block, lds_bytes, stream, p, w);
    }
    return hipGetLastError();
}

}  // namespace mvnerf
