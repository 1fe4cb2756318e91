// Backward pass of the render path (MVVNeRFRenderer.train_step, model_v0.py:186-197) for gfx950.
//
// Layer-major: the forward pass (field_eval_split16[h]_kernel<.., kStash=true>) leaves the 13 pre-activation tensors
// of the trunk in HBM in tile layout (mvnerf_mfma.h); each Dense layer's backward is then one pass
// over all 32-sample tiles, dense_bwd_split8_kernel:
//   dX = (W . G) (.) [pre > 0] (+ residual)   -- the forward's weight-stream MFMA code fed with transposed kernels
//   dW += relu(pre) . G^T, db += sum G        -- samples are the MFMA K dimension; a workgroup keeps its
//                                                128x128 partial in registers over its tiles and adds it once
//                                                with fp32 atomics
// plus per-ray / elementwise kernels: loss gradient, compositing backward (incl. depths), resample backward
// (sample_pdf + un-sort), read-out backward, layer-0 weight gradient with recomputed inputs, the gradient w.r.t.
// the sample positions, Adam with clip-by-value.  See DESIGN.md section 8.
#include <hip/hip_runtime.h>

#include <atomic>

#include "mvnerf_kernels.h"
#include "mvnerf_math.h"
#include "mvnerf_mfma.h"

namespace mvnerf {

// ---- weight images for the backward GEMMs -------------------------------------------------------------
// One Keras Dense kernel [128][128] (in, out) -> 16384 floats in hidden-layer stream order (mvnerf_pack.h) of its
// transpose M = src^T, M indexed [k_in][n_out]; rows of src that do not exist (the last slab of the 379-row layer-0
// kernel is shorter) give zeros.
// All 15 transposed streams of one MLP in one launch (blockIdx.y = stream): the 12 hidden kernels, then the three
// 128-row slabs of the 379-row layer-0 kernel.
__global__ void pack_bwd_streams_kernel(const float* __restrict__ net_keras, float* __restrict__ dst) {
    const int l = blockIdx.y;
    const float* src;
    int valid = 128;
    if (l < 12) {
        src = net_keras + kKerasBlocks + (l / 2) * kKerasBlockStride + (l % 2) * (kHidden * kHidden + kHidden);
    } else {
        const int slab = l - 12;
        src = net_keras + kKerasW0 + (long)slab * 128 * kHidden;
        valid = kIn - 128 * slab < 128 ? kIn - 128 * slab : 128;
    }
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kHiddenWFloats) return;
    const int e = idx % 4, lane = (idx % kChunkFloats) / 4, i = lane & 31, h = lane >> 5;
    const int grp = idx / kGroupFloats, nb = (idx % kGroupFloats) / kChunkFloats;
    const int k = 32 * (grp / 4) + 8 * (grp % 4) + 4 * h + e, n = 32 * nb + i;
    dst[(long)l * kHiddenWFloats + idx] = n < valid ? src[n * kHidden + k] : 0.0f;      // M = src^T, rows of src beyond `valid` are 0
}

hipError_t launch_pack_bwd_streams(const float* net_keras, float* dst, hipStream_t st) {
    hipLaunchKernelGGL(pack_bwd_streams_kernel, dim3(kHiddenWFloats / 256, 15), dim3(256), 0, st, net_keras, dst);
    return hipGetLastError();
}


// ---- where a workgroup's weight-gradient partial goes ------------------------------------------------------------------
// Every workgroup STORES its partial of the contiguous span [dW | db] at part + blockIdx.x * part_stride, and
// reduce_partials_kernel adds the partials onto the gradient in workgroup order: bit-identical gradients from run to run, and since
// round 2 also faster than fp32 atomics onto the same addresses, so api.hip always passes `part` for the weight gradients (the
// `store == false` atomic form below is still what the kernels do when a caller hands them no partial buffer).
__device__ __forceinline__ void grad_out(float* dst, float v, bool store) {
    if (store) *dst = v;
    else atomicAdd(dst, v);
}

// dst[i] += sum over the workgroups' partials, in a fixed order (bit-identical from run to run): a 64 x 16 block takes 64 elements,
// thread (x, y) adds the partials y, y + 16, y + 32, ... with four independent chains (the loads of a chain would otherwise wait for
// each other: 512 serial loads per element took 250 us), the 16 partial sums are added in y order through LDS.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ part, long part_stride, int n_parts, int count,
                                                               float* __restrict__ dst) {
    __shared__ float sred[16][64];
    const int x = threadIdx.x, y = threadIdx.y, i = blockIdx.x * 64 + x;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (i < count) {
        int w = y;
        for (; w + 48 < n_parts; w += 64) {
            s0 = s0 + part[(long)w * part_stride + i];
            s1 = s1 + part[(long)(w + 16) * part_stride + i];
            s2 = s2 + part[(long)(w + 32) * part_stride + i];
            s3 = s3 + part[(long)(w + 48) * part_stride + i];
        }
        for (; w < n_parts; w += 16) s0 = s0 + part[(long)w * part_stride + i];
    }
    sred[y][x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (y == 0 && i < count) {
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s = s + sred[q][x];
        dst[i] = dst[i] + s;
    }
}

hipError_t launch_reduce_partials(const float* part, long part_stride, int n_parts, int count, float* dst, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((count + 63) / 64), dim3(64, 16), 0, st, part, part_stride, n_parts, count, dst);
    return hipGetLastError();
}

// ---- dW[k][n] += sum_rows relu(a)[k] g[n] ; db[n] += sum_rows g[n] -----------------------------------------
// a_tl: (rows,128) pre-activations in TL (relu applied on load when relu_a); g_tl: (rows, 32*kNB) in TL.
// Wave w of a workgroup owns rows k in [32w, 32w+32) of dW and all kNB column blocks.
template <int kNB>
__global__ __launch_bounds__(256) void dw_tile_kernel(const float* __restrict__ a_tl, int relu_a, const float* __restrict__ g_tl,
                                                      long n_tiles, float* __restrict__ dW, int ldn, int n_valid,
                                                      float* __restrict__ db, long part_stride) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[kNB];
    float dbacc[kNB];
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) {
        dbacc[nb] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
    }
    // software pipeline over the workgroup's tiles: the next tile's operands are requested before the MFMAs of the current one (without
    // it the read-out's 128 x 4 gradient ran at 2.2 TB/s: one tile in flight per wave)
    f32x4 a4[4], g4[kNB][4];
    auto load_tile = [&](long tile, f32x4 (&a)[4], f32x4 (&g)[kNB][4]) {
        const f32x4* ap = reinterpret_cast<const f32x4*>(a_tl + tl_index(tile, 128, 32 * w + i, 4 * h));
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = ap[2 * t];                  // samples 8t+4h .. 8t+4h+3
#pragma unroll
        for (int nb = 0; nb < kNB; ++nb) {
            const f32x4* gp = reinterpret_cast<const f32x4*>(g_tl + tl_index(tile, 32 * kNB, 32 * nb + i, 4 * h));
#pragma unroll
            for (int t = 0; t < 4; ++t) g[nb][t] = gp[2 * t];
        }
    };
    if ((long)blockIdx.x < n_tiles) load_tile(blockIdx.x, a4, g4);
    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        f32x4 an[4], gn[kNB][4];
        const long nt = tile + gridDim.x < n_tiles ? tile + gridDim.x : tile;
        load_tile(nt, an, gn);
#pragma unroll
        for (int nb = 0; nb < kNB; ++nb) {
            float s = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float av = relu_a ? fmaxf(a4[t][e], 0.0f) : a4[t][e];
                    acc[nb] = mfma(av, g4[nb][t][e], acc[nb]);
                    s = s + g4[nb][t][e];
                }
            dbacc[nb] = dbacc[nb] + s;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a4[t] = an[t];
#pragma unroll
            for (int nb = 0; nb < kNB; ++nb) g4[nb][t] = gn[nb][t];
        }
    }
    const int col = lane & 31, hh = lane >> 5;
    const bool store = part_stride != 0;
    dW += (long)blockIdx.x * part_stride;
    if (db) db += (long)blockIdx.x * part_stride;
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) {
        if (32 * nb + col < n_valid) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                grad_out(dW + (long)(32 * w + acc_row(r, hh)) * ldn + 32 * nb + col, acc[nb][r], store);
        }
        if (db && w == 0) {
            const float s = dbacc[nb] + __shfl_xor(dbacc[nb], 32);
            if (hh == 0 && 32 * nb + col < n_valid) grad_out(db + 32 * nb + col, s, store);
        }
    }
}

// `part` (deterministic mode, else nullptr): scratch for max_wgs partials of the span [dW | db] (db directly behind dW).
hipError_t launch_dw_tile(const float* a_tl, int relu_a, const float* g_tl, int g_feats, long n_tiles, float* dW, int ldn,
                          int n_valid, float* db, int max_wgs, float* part, hipStream_t st) {
    const unsigned wgs = (unsigned)(n_tiles < max_wgs ? n_tiles : max_wgs);
    const int span = 128 * ldn + (db ? (int)(db - dW) - 128 * ldn + n_valid : 0);
    const long stride = part ? span : 0;
    float* dWk = part ? part : dW;
    float* dbk = db ? dWk + (db - dW) : nullptr;
    if (g_feats == 128)
        hipLaunchKernelGGL(dw_tile_kernel<4>, dim3(wgs), dim3(256), 0, st, a_tl, relu_a, g_tl, n_tiles, dWk, ldn, n_valid, dbk, stride);
    else if (g_feats == 32)
        hipLaunchKernelGGL(dw_tile_kernel<1>, dim3(wgs), dim3(256), 0, st, a_tl, relu_a, g_tl, n_tiles, dWk, ldn, n_valid, dbk, stride);
    else
        return hipErrorInvalidValue;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !part) return e;
    return launch_reduce_partials(part, stride, (int)wgs, span, dW, st);
}

// ---- backward of one Dense(128 -> 128) over 32-sample tiles ----------------------------------------------------------------
//   y = W^T relu(a) + b        G = dL/dy (TL),  a = pre-activation input (TL)
//   dL/da = (W . G) (.) [a > 0] (+ resid)      dW += relu(a) . G^T      db += sum G
// Two kernels: dense_dx_kernel (frozen trunk, query_vjp: dL/da only, fp32 MFMA) and dense_bwd_split8_kernel (training: dL/da, dW
// and db from one pass over the tile, both GEMMs on the bf16 matrix pipe with exactly cut operands).  Both bring the G and a tiles
// into LDS by LDS-DMA (global_load_lds_dwordx4) with the float4 chunks of a row XOR-swizzled on the SOURCE side, so that the
// "lane = sample" dword reads and the "lane = feature" 16-byte reads of the image are conflict-free.
//
// What round 2 measured on the way (scripts/bwd_probe.hip, scripts/filler_probe.hip, profiles/r02_bwd_*.log):
//  * the first form kept the next tile in registers while streaming its weight quarter from L2 inside the tile: vmcnt retires in
//    order, so the first weight wait of a tile also waited for the HBM prefetch issued just before it (340 us per 16384-tile launch);
//  * weights resident in registers + tiles by LDS-DMA: 300 us; of a tile's 16.7 k cycles per wave 11 k were the MFMA section of
//    5632 matrix cycles shared by the two waves of a SIMD, and vector instructions do not issue behind a wave's OWN
//    v_mfma_f32_32x32x2_f32 (4 independent v_add_f32 behind one: 64 -> 94 cycles; behind v_mfma_f32_32x32x16_bf16 five are free),
//    so the operand cuts of the bf16 weight-gradient GEMM were paid in full;
//  * the epilogue's stores through a pointer that had passed an inline-asm constraint were FLAT stores (they count in lgkmcnt,
//    so every LDS read behind one waited for it): 3000 -> 1200 cycles per tile once they are address_space(1) stores.
__device__ __forceinline__ int swz_f4(int f, int chunk) { return f * 8 + (chunk ^ (f & 7)); }     // float4 index

using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned int;

__device__ __forceinline__ f32x16 mfma16s(u32x4_t a, u32x4_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// exact cut of 8 fp32 values (two float4) into three bf16 pieces each (truncation; remainders are exact fp32 subtractions)
template <bool kRelu>
__device__ __forceinline__ void cut3(const f32x4& lo, const f32x4& hi, u32x4_t& p1, u32x4_t& p2, u32x4_t& p3) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v0 = q < 2 ? lo[2 * q] : hi[2 * q - 4], v1 = q < 2 ? lo[2 * q + 1] : hi[2 * q - 3];
        if (kRelu) {
            const int i0 = __builtin_bit_cast(int, v0), i1 = __builtin_bit_cast(int, v1);
            v0 = __builtin_bit_cast(float, i0 > 0 ? i0 : 0);
            v1 = __builtin_bit_cast(float, i1 > 0 ? i1 : 0);
        }
        const float r0 = v0 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v0) & 0xffff0000u);
        const float r1 = v1 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v1) & 0xffff0000u);
        const float s0 = r0 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r0) & 0xffff0000u);
        const float s1 = r1 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
        p1[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, v1), __builtin_bit_cast(unsigned, v0), 0x07060302u);
        p2[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, r1), __builtin_bit_cast(unsigned, r0), 0x07060302u);
        p3[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
    }
}

#ifndef MVT_BWD_F16
#define MVT_BWD_F16 1      // dense_bwd_split8_kernel: 1: both GEMMs as three fp16 MFMAs per product on two-piece operands (the scheme of
#endif                     //    field_eval_split16_impl.h; the gradient tile is scaled by a power of two taken from the producer's max |g|);
                           //    0: six bf16 MFMAs on exactly cut three-piece operands (round 2)
using f16x8_t = __attribute__((ext_vector_type(8))) _Float16;
using f16x2_t = __attribute__((ext_vector_type(2))) _Float16;
using f32x2_t = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ f32x16 mfma16h(u32x4_t a, u32x4_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// 8 fp32 values x (two float4) -> two fp16 pieces of t = x * s_lo: hi = rn16(t), lo = rn16(64 (t - hi)) = rn16(x * s_hi - 64 hi) with
// s_hi = 64 s_lo (both powers of two: the remainder is exact in fp32).  kRelu: relu first.  The partner of `lo` carries a factor 1/64.
template <bool kRelu>
__device__ __forceinline__ void cut2h_scaled(const f32x4& xlo, const f32x4& xhi, float s_lo, float s_hi, u32x4_t& hi, u32x4_t& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v0 = q < 2 ? xlo[2 * q] : xhi[2 * q - 4], v1 = q < 2 ? xlo[2 * q + 1] : xhi[2 * q - 3];
        if (kRelu) {
            const int i0 = __builtin_bit_cast(int, v0), i1 = __builtin_bit_cast(int, v1);
            v0 = __builtin_bit_cast(float, i0 > 0 ? i0 : 0);
            v1 = __builtin_bit_cast(float, i1 > 0 ? i1 : 0);
        }
        const f32x2_t t = {v0 * s_lo, v1 * s_lo};
        const float u0 = v0 * s_hi, u1 = v1 * s_hi;
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, f16x2_t));
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "s"(-64.0f), "v"(u0));
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "s"(-64.0f), "v"(u1));
        const f32x2_t r = {r0, r1};
        hi[q] = h;
        lo[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
    }
}

// 8 fp32 values -> the three fp16 pieces of an A operand with t = 64 x: a0 = rn16(t), a0s = a0 / 64, a1 = rn16(t - a0) (unscaled remainder)
template <bool kRelu>
__device__ __forceinline__ void cut3a(const f32x4& xlo, const f32x4& xhi, u32x4_t& a0, u32x4_t& a0s, u32x4_t& a1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v0 = q < 2 ? xlo[2 * q] : xhi[2 * q - 4], v1 = q < 2 ? xlo[2 * q + 1] : xhi[2 * q - 3];
        if (kRelu) {
            const int i0 = __builtin_bit_cast(int, v0), i1 = __builtin_bit_cast(int, v1);
            v0 = __builtin_bit_cast(float, i0 > 0 ? i0 : 0);
            v1 = __builtin_bit_cast(float, i1 > 0 ? i1 : 0);
        }
        const float t0 = v0 * 64.0f, t1 = v1 * 64.0f;
        const f32x2_t t = {t0, t1};
        const f16x2_t h = __builtin_convertvector(t, f16x2_t);
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "s"(-1.0f), "v"(t0));
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "s"(-1.0f), "v"(t1));
        const f32x2_t r = {r0, r1};
        const f16x2_t k64 = {(_Float16)0.015625f, (_Float16)0.015625f};
        a0[q] = hb;
        a0s[q] = __builtin_bit_cast(unsigned, h * k64);
        a1[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
    }
}

// max |g| of a gradient tensor travels beside it in 64 slots (the producers' workgroups spread their atomics over them): the power of two
// that brings it to [2^13, 2^14), as (2^e, 2^-e)
constexpr int kAmaxSlots = 64;
__device__ __forceinline__ void amax_scale(const float* __restrict__ amax, int lane, float* sc, float* inv) {
    float m = amax ? amax[lane] : 0.0f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    const int ex = (__builtin_bit_cast(int, m) >> 23) & 0xff;                 // biased exponent of the maximum (0: all gradients zero)
    int e = ex == 0 ? 0 : 140 - ex;                                           // m 2^e in [2^13, 2^14)
    e = e > 120 ? 120 : (e < -100 ? -100 : e);
    e = __builtin_amdgcn_readfirstlane(e);
    *sc = __builtin_bit_cast(float, (127 + e) << 23);
    *inv = __builtin_bit_cast(float, (127 - e) << 23);
}
__device__ __forceinline__ void amax_publish(float* __restrict__ amax, float m) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (amax && (threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(amax) + (blockIdx.x & (kAmaxSlots - 1)), __builtin_bit_cast(int, m));
}

#ifndef MVT_STAMP
#define MVT_STAMP 0        // scripts/bwd_probe.hip: per-wave cycle totals of the phases of a tile (dense_bwd_split8_kernel)
#endif
#if MVT_STAMP
__device__ unsigned long long g_bwd_stamp[256 * 8 * 8];
#define STAMP(k) { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[k] += t_ - st_last; st_last = t_; }
#else
#define STAMP(k)
#endif

// ---- dL/da only (frozen trunk): a wave's quarter of the transposed weight stream stays in 64 registers for the whole launch, the
// only vector-memory traffic inside the tile loop is the DMA of the NEXT tile into the other LDS buffer, the residual rows (wave-
// private, read in the epilogue) and the stores; one workgroup barrier per tile.  4 waves, wave w owns rows [32w, 32w+32).
constexpr int kBwdBufF4 = 2048;                              // one staged tile pair: G (1024 float4) then a (1024 float4)
constexpr int kBwdLdsBytes = (2 * kBwdBufF4 + 1024) * 16;    // two buffers + the residual tile: 80 KiB, two workgroups fill a CU's 160 KiB

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dense_dx_kernel(
    const float* __restrict__ g_tl, const float* __restrict__ a_tl, const float* __restrict__ wstream, const float* __restrict__ resid_tl,
    float* __restrict__ da_tl, long n_tiles) {
    extern __shared__ __attribute__((aligned(16))) f32x4 sbuf[];
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x4 wreg[16];                                          // chunk (grp, nb = w) of every 4 KiB group of the stream
    {
        const f32x4* ws = reinterpret_cast<const f32x4*>(wstream) + 64 * w + lane;
#pragma unroll
        for (int grp = 0; grp < 16; ++grp) wreg[grp] = ws[256 * grp];
    }
    // lane-dependent LDS float indices, buffer offset included (toggled by XOR at the end of a tile); everything else rides in the
    // ds_read immediates: G[row 4h+e (+8m)][sample j] and a[row 32w + 4h + e (+8m)][sample j]
    int gidx[4], eidx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        gidx[e] = swz_f4(4 * h + e, j >> 2) * 4 + (j & 3);
        eidx[e] = 4096 + swz_f4(32 * w + 4 * h + e, j >> 2) * 4 + (j & 3);
    }
    // DMA: wave-instruction m of wave w fills LDS float4 slots 256m + 64w + lane = (row 32m + 8w + lane/8, physical chunk lane%8),
    // which holds logical chunk (lane%8) ^ (row%8) = (lane%8) ^ (lane/8) of that row.  Global addresses are a uniform base pinned
    // to scalar registers plus ONE 32-bit lane offset; the other lane offsets are recomputed from it behind an empty asm (the
    // compiler would otherwise hoist 64-bit per-lane pointers into registers it has to spill).
    const unsigned src_off = (unsigned)((lane & 56) + ((lane & 7) ^ (lane >> 3))) * 16u;
    auto sbase = [](const float* p, long byte_off) {
        const char* b = reinterpret_cast<const char*>(p) + byte_off;
        asm volatile("" : "+s"(b));
        return b;
    };
    const int n_tiles32 = (int)n_tiles, stride = (int)gridDim.x;            // tile indices are 32-bit (scalar compares); byte offsets 64-bit
    auto dma_tile = [&](int t, int buf) {
        unsigned so = src_off;
        asm volatile("" : "+v"(so));
        f32x4* dst = sbuf + buf * kBwdBufF4 + 64 * w;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            __builtin_amdgcn_global_load_lds((gptr)(sbase(g_tl, (long)t * 16384 + 4096 * m + 1024 * w) + so), (lptr)(dst + 256 * m), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(sbase(a_tl, (long)t * 16384 + 4096 * m + 1024 * w) + so), (lptr)(dst + 1024 + 256 * m), 16, 0, 0);
        }
    };
    auto lane_of = [&]() {                                  // lane id back from src_off
        unsigned so = src_off;
        asm volatile("" : "+v"(so));
        const unsigned x = so >> 4;
        return (x & 56u) | ((x & 7u) ^ (x >> 3));
    };
    dma_tile((int)blockIdx.x, 0);
    if (!resid_tl) {                                        // no skip gradient: the residual image stays zero
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int m = 0; m < 4; ++m) sbuf[2 * kBwdBufF4 + 256 * w + 64 * m + lane] = z;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    int cur = 0;
    const float* sF = reinterpret_cast<const float*>(sbuf);
    for (int tile = (int)blockIdx.x; tile < n_tiles32; tile += stride) {
        // every wave has its part of this tile in LDS (waited below / above) and is done reading the other buffer
        asm volatile("s_barrier" ::: "memory");
        {
            const int nt = tile + stride < n_tiles32 ? tile + stride : tile;
            dma_tile(nt, cur ^ 1);
        }
        if (resid_tl) {
            const unsigned res_off = lane_of() * 16u;
            f32x4* rdst = sbuf + 2 * kBwdBufF4 + 256 * w;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                __builtin_amdgcn_global_load_lds((gptr)(sbase(resid_tl, (long)tile * 16384 + 4096 * w + 1024 * m) + res_off), (lptr)(rdst + 64 * m), 16, 0, 0);
        }
        // rows of block w: out[32w + i'][j] = sum_n M[n][32w + i'] G[n][j]; contraction index n = 32kb + 8t + 4h + e of group 4kb + t
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        float gcur[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) gcur[e] = sF[gidx[e]];
#pragma unroll
        for (int grp = 0; grp < 16; ++grp) {
            float gnext[4];
            const int gn = grp < 15 ? grp + 1 : grp;
#pragma unroll
            for (int e = 0; e < 4; ++e) gnext[e] = sF[gidx[e] + (32 * (gn >> 2) + 8 * (gn & 3)) * 32];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma(wreg[grp][e], gcur[e], acc);
#pragma unroll
            for (int e = 0; e < 4; ++e) gcur[e] = gnext[e];
        }
        __builtin_amdgcn_sched_barrier(0);
        // the residual rows and the next tile have landed (they had the whole MFMA section); the previous tile's stores too
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- epilogue: relu mask from the staged a tile, residual, store (address_space(1): see the header) ----
        const unsigned ln = lane_of();
        const unsigned oo = ((ln >> 5) * 128u + (ln & 31u)) * 4u;                 // ((4h) * 32 + j) floats
        const float* sR = sF + 4 * 2 * kBwdBufF4 + 32 * w * 32 + (oo >> 2);
        char* optr = const_cast<char*>(sbase(da_tl, (long)tile * 16384 + 4096 * w));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = ((r & 3) + 8 * (r >> 2)) * 32;
            const float av = sF[eidx[r & 3] + 8 * (r >> 2) * 32];
            const float v = av > 0.0f ? acc[r] : 0.0f;
            *(__attribute__((address_space(1))) float*)(optr + 4 * ro + oo) = v + sR[ro];
        }
        cur ^= 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gidx[e] ^= 4 * kBwdBufF4;
            eidx[e] ^= 4 * kBwdBufF4;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped re-load of the last tile must land before the LDS is released
}

// ---- the training form: both GEMMs of the layer backward on the bf16 matrix pipe, roles split over 8 waves --------------------
//   * dL/da and dW both run as six bf16 MFMAs per product on exactly cut operands (fp32-grade, see field_eval_split.hip):
//     2 x 48 x 32 matrix cycles per tile instead of 2 x 64 x 64 on the fp32 MFMA;
//   * G is cut ONCE per tile by the 8 waves together, into two LDS images of bf16 pieces in MFMA-operand order: packed along
//     the samples (B operand of the weight gradient, K = samples) and packed along the features (B operand of dL/da, K = n);
//   * waves 0-3 ("Z") own one 32-row block of dL/da each and keep their quarter of W as 3 x 8 x 4 piece registers for the whole
//     launch; waves 4-7 ("W") own 32 rows of dW each (64 accumulator registers); one workgroup per CU, one Z and one W wave per SIMD.
// Per tile: barrier (raw tile landed by LDS-DMA) - cut - barrier - MFMA sections - epilogue (Z) / next A cut (W).
constexpr int kBwd8Raw = 2048, kBwd8Resid = 2 * kBwd8Raw, kBwd8Prow = kBwd8Resid + 1024, kBwd8Pcol = kBwd8Prow + 1536;   // float4 units
constexpr int kBwd8LdsBytes = (kBwd8Pcol + 1536) * 16;                                                            // 128 KiB

__global__ __launch_bounds__(512, 1) void dense_bwd_split8_kernel(
    const float* __restrict__ g_tl, const float* __restrict__ a_tl, const float* __restrict__ wstream, const float* __restrict__ resid_tl,
    float* __restrict__ da_tl, long n_tiles, float* __restrict__ dW, float* __restrict__ db, long part_stride,
    const float* __restrict__ amax_in, float* __restrict__ amax_out) {
    extern __shared__ __attribute__((aligned(16))) f32x4 sbuf[];
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, i = lane & 31;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);                  // wave 0..7
#if MVT_BWD_F16
    // G is cut as g 2^e / 64 (+ remainder x 64): 2^e from the producer's max |g|, so that the pieces stay normal fp16 numbers whatever the
    // loss scale; dL/da and dW come out scaled by 2^e and are multiplied by 2^-e where they leave the registers
    float g_sc, g_inv;
    amax_scale(amax_in, lane, &g_sc, &g_inv);
    const float g_sc_lo = g_sc * 0.015625f;
    float out_max = 0.0f;
#endif
    const float* sF = reinterpret_cast<const float*>(sbuf);
    u32x4_t* sP = reinterpret_cast<u32x4_t*>(sbuf);
    const int n_tiles32 = (int)n_tiles, stride = (int)gridDim.x;
    auto sbase = [](const float* p, long byte_off) {                        // uniform address, pinned to scalar registers
        const char* b = reinterpret_cast<const char*>(p) + byte_off;
        asm volatile("" : "+s"(b));
        return b;
    };
    // sum of the 8 gradient values a lane holds (bias gradient), as single v_add_f32: left to the compiler these are packed into
    // v_pk_add_f32 behind a string of register moves
    auto sum8 = [](const f32x4& lo, const f32x4& hi, float accv) {
        float s = lo[0];
        const float t[7] = {lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int q = 0; q < 7; ++q) asm("v_add_f32_e32 %0, %1, %2" : "=v"(s) : "v"(s), "v"(t[q]));
        asm("v_add_f32_e32 %0, %1, %2" : "=v"(accv) : "v"(accv), "v"(s));
        return accv;
    };
    // raw tile by LDS-DMA: wave-instruction I = 4v + m (0..31) fills float4 slots 64 I .. 64 I + 63 of [G | a] (rows 8 I' ..., the
    // XOR swizzle of swz_f4 applied on the source side): waves 0-3 bring G, waves 4-7 bring a
    const unsigned src_off = (unsigned)((lane & 56) + ((lane & 7) ^ (lane >> 3))) * 16u;
    auto dma_raw = [&](int t, int buf) {
        unsigned so = src_off;
        asm volatile("" : "+v"(so));
        const float* src = v < 4 ? g_tl : a_tl;
        f32x4* dst = sbuf + buf * kBwd8Raw + 256 * v;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            __builtin_amdgcn_global_load_lds((gptr)(sbase(src, (long)t * 16384 + 4096 * (v & 3) + 1024 * m) + so), (lptr)(dst + 64 * m), 16, 0, 0);
    };
    // the cut of this wave's share of G (tile in raw buffer `buf`): chunk (ks = v / 4, nb = v % 4) of the sample-packed image and
    // chunk ks = v of the feature-packed image; returns the wave's bias-gradient contribution (rows 32 (v % 4) + i, samples 16 (v/4) + 8h ..)
    const int rowc = swz_f4(i, 4 * (v >> 2) + 2 * h) + 256 * (v & 3), rowc1 = swz_f4(i, 4 * (v >> 2) + 2 * h + 1) + 256 * (v & 3);
    int colf[8];                                                            // float index of G[16v + 8h + q][j]
#pragma unroll
    for (int q = 0; q < 8; ++q) colf[q] = swz_f4(16 * v + 8 * h + q, j >> 2) * 4 + (j & 3);
    auto cut_g = [&](int buf, float dbv) {
        const f32x4 lo = sbuf[buf * kBwd8Raw + rowc], hi = sbuf[buf * kBwd8Raw + rowc1];
        f32x4 clo, chi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            clo[q] = sF[buf * 4 * kBwd8Raw + colf[q]];
            chi[q] = sF[buf * 4 * kBwd8Raw + colf[4 + q]];
        }
        u32x4_t p0, p1, p2;
        u32x4_t* pr = sP + kBwd8Prow + (v * 3) * 64 + lane;                 // chunk (ks, nb) = v
        u32x4_t* pc = sP + kBwd8Pcol + (v * 3) * 64 + lane;                 // chunk ks = v
#if MVT_BWD_F16
        cut2h_scaled<false>(lo, hi, g_sc_lo, g_sc, p0, p1);                 // pieces 0 (hi) and 1 (lo x 64); slot 2 of a chunk stays unused
        pr[0] = p0;
        pr[64] = p1;
        dbv = sum8(lo, hi, dbv);
        cut2h_scaled<false>(clo, chi, g_sc_lo, g_sc, p0, p1);
        pc[0] = p0;
        pc[64] = p1;
        (void)p2;
#else
        cut3<false>(lo, hi, p0, p1, p2);
        pr[0] = p0;
        pr[64] = p1;
        pr[128] = p2;
        dbv = sum8(lo, hi, dbv);
        cut3<false>(clo, chi, p0, p1, p2);
        pc[0] = p0;
        pc[64] = p1;
        pc[128] = p2;
#endif
        return dbv;
    };
    float dbacc = 0.0f;
#if MVT_STAMP
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
    const unsigned long long st_begin = st_last;
#endif
    dma_raw((int)blockIdx.x, 0);

    if (v < 4) {
        // ================================ Z waves: rows [32v, 32v+32) of dL/da ================================
        const int w = v;
        // this wave's quarter of the transposed weight stream (fp32-MFMA order: lane (i, h) of chunk (grp, w) holds
        // W[32w + i][8 grp + 4h + e], e = 0..3), re-ordered to the bf16 operand (lane (i, h): W[32w + i][16 ks + 8h + q], q = 0..7)
        // by one exchange between the lane halves, and cut into pieces once
        u32x4_t wp[8][3];
        {
            const f32x4* ws = reinterpret_cast<const f32x4*>(wstream) + 64 * w + lane;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const f32x4 g0 = ws[256 * (2 * ks)], g1 = ws[256 * (2 * ks + 1)];       // n = 16ks + 4h + e  /  16ks + 8 + 4h + e
                f32x4 lo, hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float recv = __shfl_xor(h ? g0[e] : g1[e], 32);
                    lo[e] = h ? recv : g0[e];               // h = 0: n = 16ks + e (own g0);        h = 1: n = 16ks + 8 + e (partner's g1)
                    hi[e] = h ? g1[e] : recv;               // h = 0: n = 16ks + 4 + e (partner's g0); h = 1: n = 16ks + 12 + e (own g1)
                }
#if MVT_BWD_F16
                cut3a<false>(lo, hi, wp[ks][0], wp[ks][1], wp[ks][2]);          // (64 w, 64 w / 64, remainder)
#else
                cut3<false>(lo, hi, wp[ks][0], wp[ks][1], wp[ks][2]);
#endif
            }
        }
        int eidx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) eidx[e] = 4096 + swz_f4(32 * w + 4 * h + e, j >> 2) * 4 + (j & 3);   // a[row 32w + 4h + e (+8m)][sample j], raw buffer 0
        if (!resid_tl) {                                        // no skip gradient: the residual image stays zero
            const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < 4; ++m) sbuf[kBwd8Resid + 256 * w + 64 * m + lane] = z;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        int cur = 0;
        for (int tile = (int)blockIdx.x; tile < n_tiles32; tile += stride) {
            STAMP(5)
            asm volatile("s_barrier" ::: "memory");             // B1: raw tile complete; pieces of the previous tile consumed
            STAMP(0)
            {
                const int nt = tile + stride < n_tiles32 ? tile + stride : tile;
                dma_raw(nt, cur ^ 1);
            }
            if (resid_tl) {
                unsigned so = src_off;
                asm volatile("" : "+v"(so));
                const unsigned x = so >> 4;
                const unsigned res_off = ((x & 56u) | ((x & 7u) ^ (x >> 3))) * 16u;      // lane * 16
                f32x4* rdst = sbuf + kBwd8Resid + 256 * w;
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    __builtin_amdgcn_global_load_lds((gptr)(sbase(resid_tl, (long)tile * 16384 + 4096 * w + 1024 * m) + res_off), (lptr)(rdst + 64 * m), 16, 0, 0);
            }
            dbacc = cut_g(cur, dbacc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAMP(1)
            asm volatile("s_barrier" ::: "memory");             // B2: both piece images complete
            STAMP(2)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const u32x4_t* pc = sP + kBwd8Pcol + lane;
#if MVT_BWD_F16
            u32x4_t b[2] = {pc[0], pc[64]};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                u32x4_t bn[2];
                if (ks < 7) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) bn[q] = pc[((ks + 1) * 3 + q) * 64];
                }
                acc = mfma16h(wp[ks][1], b[1], acc);            // (64 w / 64) x (64 remainder of g)
                acc = mfma16h(wp[ks][2], b[0], acc);            // remainder of 64 w x g hi
                acc = mfma16h(wp[ks][0], b[0], acc);
                if (ks < 7) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) b[q] = bn[q];
                }
            }
#else
            u32x4_t b[3] = {pc[0], pc[64], pc[128]};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                u32x4_t bn[3];
                if (ks < 7) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) bn[q] = pc[((ks + 1) * 3 + q) * 64];
                }
                acc = mfma16s(wp[ks][2], b[0], acc);
                acc = mfma16s(wp[ks][1], b[1], acc);
                acc = mfma16s(wp[ks][0], b[2], acc);
                acc = mfma16s(wp[ks][1], b[0], acc);
                acc = mfma16s(wp[ks][0], b[1], acc);
                acc = mfma16s(wp[ks][0], b[0], acc);
                if (ks < 7) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) b[q] = bn[q];
                }
            }
#endif
            STAMP(3)
            // the residual rows and the next raw tile have landed (they had the cut and the MFMA section); the previous stores too
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(4)
            unsigned so = src_off;
            asm volatile("" : "+v"(so));
            const unsigned x = so >> 4;
            const unsigned ln = (x & 56u) | ((x & 7u) ^ (x >> 3));
            const unsigned oo = ((ln >> 5) * 128u + (ln & 31u)) * 4u;                     // ((4h) * 32 + j) floats
            const float* sR = sF + 4 * kBwd8Resid + 32 * w * 32 + (oo >> 2);
            char* optr = const_cast<char*>(sbase(da_tl, (long)tile * 16384 + 4096 * w));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = ((r & 3) + 8 * (r >> 2)) * 32;
                const float av = sF[eidx[r & 3] + 8 * (r >> 2) * 32];
#if MVT_BWD_F16
                const float val = (av > 0.0f ? acc[r] * g_inv : 0.0f) + sR[ro];
                out_max = fmaxf(out_max, fabsf(val));
                *(__attribute__((address_space(1))) float*)(optr + 4 * ro + oo) = val;
#else
                const float val = av > 0.0f ? acc[r] : 0.0f;
                *(__attribute__((address_space(1))) float*)(optr + 4 * ro + oo) = val + sR[ro];
#endif
            }
            cur ^= 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) eidx[e] ^= 4 * kBwd8Raw;
        }
#if MVT_BWD_F16
        amax_publish(amax_out, out_max);
#endif
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");        // end: all MFMA sections done, the sample-packed image is free
        // bias gradient: rows 32 (v % 4) + i; this wave's samples [8h, 8h+8) of k-step 0 and, from wave v + 4, of k-step 1
        float* xch = reinterpret_cast<float*>(sbuf + kBwd8Prow);
        asm volatile("s_barrier" ::: "memory");                              // the W waves' partials are in xch
        const float other = xch[64 * v + lane];
        float sdb = dbacc + other;
        sdb = sdb + __shfl_xor(sdb, 32);
        if (h == 0) grad_out(db + (long)blockIdx.x * part_stride + 32 * v + i, sdb, part_stride != 0);
    } else {
        // ================================ W waves: rows [32u, 32u+32) of dW ================================
        const int u = v - 4;
        f32x16 dwacc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dwacc[nb][r] = 0.0f;
        int aidx[4];                                            // float4 index of the chunk pair (4ks + 2h, +1) of row 32u + i of a, raw buffer 0
#pragma unroll
        for (int e = 0; e < 4; ++e) aidx[e] = 1024 + 256 * u + swz_f4(i, 4 * (e >> 1) + 2 * h + (e & 1));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        int cur = 0;
        for (int tile = (int)blockIdx.x; tile < n_tiles32; tile += stride) {
            STAMP(5)
            asm volatile("s_barrier" ::: "memory");             // B1
            STAMP(0)
            {
                const int nt = tile + stride < n_tiles32 ? tile + stride : tile;
                dma_raw(nt, cur ^ 1);
            }
            dbacc = cut_g(cur, dbacc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAMP(1)
            asm volatile("s_barrier" ::: "memory");             // B2
            STAMP(2)
            // this wave's own A operand (rows 32u + i of relu(a)) is cut behind the barrier: nobody waits for it
            u32x4_t ap[2][3];
            const u32x4_t* pr = sP + kBwd8Prow + lane;
#if MVT_BWD_F16
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) cut3a<true>(sbuf[aidx[2 * ks]], sbuf[aidx[2 * ks + 1]], ap[ks][0], ap[ks][1], ap[ks][2]);
            u32x4_t b[2] = {pr[0], pr[64]};
#pragma unroll
            for (int c = 0; c < 8; ++c) {                       // chunk c = (ks = c / 4, nb = c % 4)
                u32x4_t bn[2];
                if (c < 7) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) bn[q] = pr[((c + 1) * 3 + q) * 64];
                }
                const int ks = c >> 2, nb = c & 3;
                dwacc[nb] = mfma16h(ap[ks][1], b[1], dwacc[nb]);
                dwacc[nb] = mfma16h(ap[ks][2], b[0], dwacc[nb]);
                dwacc[nb] = mfma16h(ap[ks][0], b[0], dwacc[nb]);
                if (c < 7) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) b[q] = bn[q];
                }
            }
#else
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) cut3<true>(sbuf[aidx[2 * ks]], sbuf[aidx[2 * ks + 1]], ap[ks][0], ap[ks][1], ap[ks][2]);
            u32x4_t b[3] = {pr[0], pr[64], pr[128]};
#pragma unroll
            for (int c = 0; c < 8; ++c) {                       // chunk c = (ks = c / 4, nb = c % 4)
                u32x4_t bn[3];
                if (c < 7) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) bn[q] = pr[((c + 1) * 3 + q) * 64];
                }
                const int ks = c >> 2, nb = c & 3;
                dwacc[nb] = mfma16s(ap[ks][2], b[0], dwacc[nb]);
                dwacc[nb] = mfma16s(ap[ks][1], b[1], dwacc[nb]);
                dwacc[nb] = mfma16s(ap[ks][0], b[2], dwacc[nb]);
                dwacc[nb] = mfma16s(ap[ks][1], b[0], dwacc[nb]);
                dwacc[nb] = mfma16s(ap[ks][0], b[1], dwacc[nb]);
                dwacc[nb] = mfma16s(ap[ks][0], b[0], dwacc[nb]);
                if (c < 7) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) b[q] = bn[q];
                }
            }
#endif
            STAMP(3)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's part of the next raw tile has landed
            STAMP(4)
            cur ^= 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) aidx[e] ^= kBwd8Raw;
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        float* xch = reinterpret_cast<float*>(sbuf + kBwd8Prow);
        xch[64 * u + lane] = dbacc;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int col = lane & 31, hh = lane >> 5;
        const bool store = part_stride != 0;
        float* dWo = dW + (long)blockIdx.x * part_stride;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
#if MVT_BWD_F16
            for (int r = 0; r < 16; ++r) grad_out(dWo + (long)(32 * u + acc_row(r, hh)) * kHidden + 32 * nb + col, dwacc[nb][r] * g_inv, store);
#else
            for (int r = 0; r < 16; ++r) grad_out(dWo + (long)(32 * u + acc_row(r, hh)) * kHidden + 32 * nb + col, dwacc[nb][r], store);
#endif
    }
#if MVT_STAMP
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = g_bwd_stamp + ((long)blockIdx.x * 8 + v) * 8;
        for (int q = 0; q < 6; ++q) o[q] = st_acc[q];
        o[6] = __builtin_readcyclecounter() - st_begin;
        o[7] = v < 4 ? 0 : 1;
    }
#endif
}

hipError_t launch_dense_bwd_fused(const float* g_tl, const float* a_tl, const float* wstream, const float* resid_tl,
                                  float* da_tl, long n_tiles, float* dW, float* db, int max_wgs, float* part, hipStream_t st,
                                  const float* amax_in, float* amax_out) {
    // `part` != nullptr (deterministic mode): every workgroup stores its partial of the span [dW | db] (db directly behind dW) and
    // launch_reduce_partials adds them in a fixed order; otherwise fp32 atomics straight onto the gradient.
    static std::atomic<bool> attr_done[16];               // first call per device sets the dynamic-LDS limits (idempotent)
    int dev = 0;
    hipError_t ea = hipGetDevice(&dev);
    if (ea != hipSuccess) return ea;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_dx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLdsBytes);
        if (ea == hipSuccess)
            ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_bwd_split8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kBwd8LdsBytes);
        if (ea != hipSuccess) return ea;
        attr_done[dev].store(true, std::memory_order_release);
    }
    if (n_tiles <= 0 || n_tiles > 0x7fffffffL) return hipErrorInvalidValue;                // tile indices are 32-bit inside the kernels
    if (!dW) {                                             // frozen trunk (query_vjp): two 256-thread workgroups per CU
        const unsigned wgs = (unsigned)(n_tiles < max_wgs ? n_tiles : max_wgs);
        hipLaunchKernelGGL(dense_dx_kernel, dim3(wgs), dim3(256), kBwdLdsBytes, st, g_tl, a_tl, wstream, resid_tl, da_tl, n_tiles);
        return hipGetLastError();
    }
    if (part && db != dW + kHidden * kHidden) return hipErrorInvalidValue;
    if (MVT_BWD_F16 && !amax_in) return hipErrorInvalidValue;                               // the fp16 products need the gradient's scale
    const int span = kHidden * kHidden + kHidden;
    const unsigned wgs8 = (unsigned)(n_tiles < max_wgs / 2 ? n_tiles : max_wgs / 2);       // one 512-thread workgroup per CU
    hipLaunchKernelGGL(dense_bwd_split8_kernel, dim3(wgs8), dim3(512), kBwd8LdsBytes, st, g_tl, a_tl, wstream, resid_tl, da_tl, n_tiles,
                       part ? part : dW, part ? part + kHidden * kHidden : db, part ? (long)span : 0L, amax_in, amax_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !part) return e;
    return launch_reduce_partials(part, span, (int)wgs8, span, dW, st);
}

// ---- loss: d pred = 2 (pred - y) / n ; loss += sum (pred - y)^2 / n   (Keras MeanSquaredError) ----
__global__ void mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ label, long n, float inv_n,
                                float* __restrict__ d_pred, float* __restrict__ loss) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float sq = 0.0f;
    if (i < n) {
        const float d = pred[i] - label[i];
        d_pred[i] = 2.0f * d * inv_n;
        sq = d * d * inv_n;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq = sq + __shfl_xor(sq, off);
    if ((threadIdx.x & 63) == 0 && sq != 0.0f) atomicAdd(loss, sq);
}

hipError_t launch_mse_grad(const float* pred, const float* label, long n, float* d_pred, float* loss, hipStream_t st) {
    hipLaunchKernelGGL(mse_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pred, label, n, 1.0f / (float)n,
                       d_pred, loss);
    return hipGetLastError();
}

// ---- volumetric_render backward (model_v0.py:89-100) w.r.t. the per-sample (r,g,b,sigma) -----------------
// q_i = dL/dw_i = gR.c_i + gD z_i + gW_i ;  dL/dalpha_i = q_i T_i - (sum_{k>i} q_k w_k) / t_i ;
// dL/dsigma_i = dL/dalpha_i * delta_i (1 - alpha_i) [sigma_i > 0] ;  dL/dc_i = w_i gR.   (depths constant)
template <int P>
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ z, const float* __restrict__ rgbs,
                                                            const float* __restrict__ d_rgb, const float* __restrict__ d_depth,
                                                            const float* __restrict__ d_w, int n_rays,
                                                            float* __restrict__ d_rgbs, float* __restrict__ d_z) {
    constexpr int S = 64 * P;
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
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
    zl[P] = __shfl_down(zl[0], 1);
    float dist[P];
#pragma unroll
    for (int q = 0; q < P; ++q) dist[q] = zl[q + 1] - zl[q];
    if (P == 1) {
        const float prev = __shfl_up(dist[0], 1);
        if (lane == 63) dist[0] = prev;
    } else {
        if (lane == 63) dist[P - 1] = dist[P - 2];
    }
    float alpha[P], t[P], T[P], wgt[P];
    float prod = 1.0f;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        alpha[q] = sigma_to_alpha(c[q][3], dist[q]);
        t[q] = (1.0f - alpha[q]) + 1e-10f;
        prod = prod * t[q];
    }
    float incl = prod;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float up = __shfl_up(incl, off);
        if (lane >= off) incl = incl * up;
    }
    float trans = __shfl_up(incl, 1);
    if (lane == 0) trans = 1.0f;
    const float gr = d_rgb[3 * ray], gg = d_rgb[3 * ray + 1], gb = d_rgb[3 * ray + 2];
    const float gd = d_depth ? d_depth[ray] : 0.0f;
    float qw[P], qv[P];
    float lane_sum = 0.0f;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        T[q] = trans;
        wgt[q] = alpha[q] * trans;
        trans = trans * t[q];
        float qi = gr * c[q][0] + gg * c[q][1] + gb * c[q][2] + gd * zl[q];
        if (d_w) qi = qi + d_w[(long)ray * S + lane * P + q];
        qv[q] = qi;
        qw[q] = qi * wgt[q];
        lane_sum = lane_sum + qw[q];
    }
    float incl_s = lane_sum;                          // inclusive suffix sum over lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float dn = __shfl_down(incl_s, off);
        if (lane + off < 64) incl_s = incl_s + dn;
    }
    float suffix = incl_s - lane_sum;                 // sum over later lanes
    float dd[P];                                      // dL/d delta_i
#pragma unroll
    for (int q = P - 1; q >= 0; --q) {
        const float dalpha = qv[q] * T[q] - suffix / t[q];
        const float pos = c[q][3] > 0.0f ? (1.0f - alpha[q]) : 0.0f;
        const float dsigma = dalpha * dist[q] * pos;
        dd[q] = dalpha * c[q][3] * pos;
        f32x4 o = {wgt[q] * gr, wgt[q] * gg, wgt[q] * gb, dsigma};
        reinterpret_cast<f32x4*>(d_rgbs)[(long)ray * S + lane * P + q] = o;
        suffix = suffix + qw[q];
    }
    if (d_z) {
        // delta_i = z_{i+1} - z_i for i <= S-2 and delta_{S-1} = delta_{S-2} (Q6): fold the duplicate, then
        // dL/dz_k = dd_{k-1} - dd_k (+ w_k gD from the depth)
        if (P == 1) {
            const float last = __shfl_down(dd[0], 1);
            if (lane == 62) dd[0] = dd[0] + last;
            if (lane == 63) dd[0] = 0.0f;
        } else if (lane == 63) {
            dd[P - 2] = dd[P - 2] + dd[P - 1];
            dd[P - 1] = 0.0f;
        }
        float prev = __shfl_up(dd[P - 1], 1);
        if (lane == 0) prev = 0.0f;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            d_z[(long)ray * S + lane * P + q] = (prev - dd[q]) + wgt[q] * gd;
            prev = dd[q];
        }
    }
}

hipError_t launch_composite_bwd(const float* z, const float* rgbs, const float* d_rgb, const float* d_depth,
                                const float* d_w, int n_rays, int S, float* d_rgbs, float* d_z, hipStream_t st) {
    const dim3 grid((n_rays + 3) / 4), block(256);
    switch (S / 64) {
        case 1: hipLaunchKernelGGL(composite_bwd_kernel<1>, grid, block, 0, st, z, rgbs, d_rgb, d_depth, d_w, n_rays, d_rgbs, d_z); break;
        case 2: hipLaunchKernelGGL(composite_bwd_kernel<2>, grid, block, 0, st, z, rgbs, d_rgb, d_depth, d_w, n_rays, d_rgbs, d_z); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- hierarchical resampling backward (sample_pdf, nerf_utils.py:143-176; model_v0.py:150-156) ---------------
// Given dL/d(all_zs) and the rank of every importance sample inside all_zs (from the forward sort), returns
// dL/d(coarse weights).  The coarse depths are not functions of any variable, so d(bins) is not propagated.
// One wavefront per ray; pdf/cdf/above/below are recomputed exactly as in the forward kernel.
__global__ __launch_bounds__(256) void resample_bwd_kernel(const float* __restrict__ z, const float* __restrict__ weights,
                                                           const float* __restrict__ u_fine, const int32_t* __restrict__ fine_rank,
                                                           const float* __restrict__ d_z_all, int n_rays, int q7_mode,
                                                           float* __restrict__ d_weights) {
    constexpr int S = 64, NB = 63, NW = 62;
    __shared__ float lds[4][5 * 64];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wv;
    if (ray >= n_rays) return;
    float* bins = lds[wv];
    float* pdf = bins + 64;
    float* cdf = pdf + 64;
    float* dcdf = cdf + 64;
    float* dpdf = dcdf + 64;
    const long base = (long)ray * S;
    const float zi = z[base + lane];
    const float wi = weights[base + lane];
    const float znext = __shfl_down(zi, 1);
    if (lane < NB) bins[lane] = 0.5f * (znext + zi);
    if (lane >= 1 && lane <= NW) pdf[lane - 1] = wi + 1e-5f;
    dcdf[lane] = 0.0f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float wsum = 0.0f;
    for (int k = 0; k < NW; ++k) wsum = wsum + pdf[k];
    const bool unit_sum = fabsf(wsum) == 0.0f;
    if (unit_sum) wsum = 1.0f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < NW) pdf[lane] = pdf[lane] / wsum;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float run = 0.0f, mycdf = 0.0f;
    for (int k = 0; k < NW; ++k) {
        run = run + pdf[k];
        if (lane == k + 1) mycdf = run;
    }
    if (lane < NB) cdf[lane] = mycdf;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float u = u_fine[base + lane];
    int above = 0;
    for (int jj = 0; jj < NB; ++jj) above += (u >= cdf[jj]) ? 1 : 0;
    int below = above - 1;
    below = below < 0 ? 0 : (below > NB - 1 ? NB - 1 : below);
    const bool oob = above >= NB;
    const int ia = oob ? NB - 1 : above;
    const bool a_const = oob && q7_mode == 0;           // gathered value is the constant 0
    const float cdf_a = a_const ? 0.0f : cdf[ia], bins_a = a_const ? 0.0f : bins[ia];
    const float cdf_b = cdf[below], bins_b = bins[below];
    const float den_raw = cdf_a - cdf_b;
    const bool den_live = !(den_raw < 1e-5f);
    const float den = den_live ? den_raw : 1.0f;
    // forward: zf = bins_b + t (bins_a - bins_b), t = (u - cdf_b) / den
    const float dzf = d_z_all[(long)ray * 128 + fine_rank[base + lane]];
    const float dt = dzf * (bins_a - bins_b);
    float dca = 0.0f, dcb = -dt / den;
    if (den_live) {
        const float k2 = dt * (u - cdf_b) / (den * den);
        dca = -k2;
        dcb = dcb + k2;
    }
    // scatter-add into d cdf (several samples may share a bin): serialise over lanes
    for (int l = 0; l < 64; ++l) {
        if (lane == l) {
            if (!a_const) dcdf[ia] += dca;
            dcdf[below] += dcb;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // cdf_j = sum_{k<j} pdf_k  ->  d pdf_k = sum_{j>k} d cdf_j
    float mydp = 0.0f;
    if (lane < NW)
        for (int jj = lane + 1; jj < NB; ++jj) mydp = mydp + dcdf[jj];
    // pdf_k = s_k / wsum  ->  d s_k = (d pdf_k - sum_m d pdf_m pdf_m) / wsum   (wsum constant when it was forced to 1)
    float dot = lane < NW ? mydp * pdf[lane] : 0.0f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot = dot + __shfl_xor(dot, off);
    dpdf[lane] = lane < NW ? (mydp - (unit_sum ? 0.0f : dot)) / wsum : 0.0f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    d_weights[base + lane] = (lane >= 1 && lane <= NW) ? dpdf[lane - 1] : 0.0f;    // probs = weights[1:-1]
}

hipError_t launch_resample_bwd(const float* z, const float* weights, const float* u_fine, const int32_t* fine_rank,
                               const float* d_z_all, int n_rays, int q7_mode, float* d_weights, hipStream_t st) {
    hipLaunchKernelGGL(resample_bwd_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, st, z, weights, u_fine, fine_rank, d_z_all,
                       n_rays, q7_mode, d_weights);
    return hipGetLastError();
}

// ---- RenderReadout backward (layers.py:392-397) -----------------------------------------------------------
// rgbs, d_rgbs: (rows,4) row-major.  x_tl: pre-activation input of the read-out (TL, 128).
// Writes do_tl (TL, 32 features, rows 4..31 must be pre-zeroed) = dL/d(pre-activation outputs) and
// g_tl (TL, 128) = dL/dx = (Wr . do) (.) [x > 0].
__global__ __launch_bounds__(256) void readout_bwd_kernel(const float* __restrict__ x_tl, const float* __restrict__ rgbs,
                                                          const float* __restrict__ d_rgbs, const float* __restrict__ wr,
                                                          long n_rows, long n_tiles, float* __restrict__ do_tl,
                                                          float* __restrict__ g_tl, float* __restrict__ amax_out) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float out_max = 0.0f;
    if (tile < n_tiles) {
    const long row = tile * 32 + j;
    float dov[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (row < n_rows) {
        const f32x4 y = reinterpret_cast<const f32x4*>(rgbs)[row], dy = reinterpret_cast<const f32x4*>(d_rgbs)[row];
#pragma unroll
        for (int c = 0; c < 3; ++c) dov[c] = dy[c] * y[c] * (1.0f - y[c]);         // sigmoid'
        dov[3] = dy[3] * (1.0f - expf(-y[3]));                                       // softplus' = sigmoid(o) = 1 - e^-softplus
    }
    if (h == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) do_tl[tl_index(tile, 32, c, j)] = dov[c];
    }
    for (int f = 64 * h; f < 64 * h + 64; ++f) {
        const long o = tl_index(tile, 128, f, j);
        const f32x4 w4 = reinterpret_cast<const f32x4*>(wr)[f];
        float g = w4[0] * dov[0];
        g = fmaf(w4[1], dov[1], g);
        g = fmaf(w4[2], dov[2], g);
        g = fmaf(w4[3], dov[3], g);
        const float gv = x_tl[o] > 0.0f ? g : 0.0f;
        out_max = fmaxf(out_max, fabsf(gv));
        g_tl[o] = gv;
    }
    }                                                      // (a tile is a wave: waves past the last tile only take part in the reduction)
    // max |g| for the fp16 cut of the first layer launch (dense_bwd_split8_kernel): one atomic per workgroup (4 096 workgroups share 64 slots)
    __shared__ float wg_max[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) out_max = fmaxf(out_max, __shfl_xor(out_max, off));
    if (lane == 0) wg_max[threadIdx.x >> 6] = out_max;
    __syncthreads();
    if (amax_out && threadIdx.x == 0)
        atomicMax(reinterpret_cast<int*>(amax_out) + (blockIdx.x & (kAmaxSlots - 1)),
                  __builtin_bit_cast(int, fmaxf(fmaxf(wg_max[0], wg_max[1]), fmaxf(wg_max[2], wg_max[3]))));
}

__global__ void zero_kernel(unsigned* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0u;
}

hipError_t launch_zero(void* ptr, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if (bytes % 4 != 0 || reinterpret_cast<uintptr_t>(ptr) % 4 != 0) return hipErrorInvalidValue;
    const size_t n = bytes / 4;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, st, static_cast<unsigned*>(ptr), n);
    return hipGetLastError();
}

hipError_t launch_readout_bwd(const float* x_tl, const float* rgbs, const float* d_rgbs, const float* wr, long n_rows,
                              long n_tiles, float* do_tl, float* g_tl, hipStream_t st, float* amax_out) {
    hipError_t e = launch_zero(do_tl, (size_t)n_tiles * 32 * 32 * sizeof(float), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(readout_bwd_kernel, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, st, x_tl, rgbs, d_rgbs, wr,
                       n_rows, n_tiles, do_tl, g_tl, amax_out);
    return hipGetLastError();
}

// ---- (rows,128) row-major -> tile layout, optionally added to what is there (gradient injections of query_vjp) ----
__global__ __launch_bounds__(256) void rows_to_tl_kernel(const float* __restrict__ rows, long n_rows, long n_tiles, int accumulate,
                                                         float* __restrict__ out_tl) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (tile, feature, sample)
    if (i >= n_tiles * 4096) return;
    const long tile = i >> 12;
    const int f = (int)(i >> 5) & 127, j = (int)i & 31;
    const long row = tile * 32 + j;
    const float v = row < n_rows ? rows[row * 128 + f] : 0.0f;
    out_tl[i] = accumulate ? out_tl[i] + v : v;
}

hipError_t launch_rows_to_tl(const float* rows, long n_rows, long n_tiles, int accumulate, float* out_tl, hipStream_t st) {
    hipLaunchKernelGGL(rows_to_tl_kernel, dim3((unsigned)((n_tiles * 4096 + 255) / 256)), dim3(256), 0, st, rows, n_rows, n_tiles,
                       accumulate, out_tl);
    return hipGetLastError();
}

// ---- tile layout -> (rows,128) row-major: the fused activations of a query stash for the GraspReadout ----
// One workgroup per (tile, slot): the tile's 128 x 32 floats go through LDS (pitch 33: the transposed reads are conflict-free), both the
// global reads and the global writes are full 256-byte rows per wave instruction.
__global__ __launch_bounds__(256) void tl_to_rows_kernel(const float* __restrict__ tl, long slot_stride, long n_rows, float* __restrict__ rows) {
    __shared__ float t[128 * 33];
    const long tile = blockIdx.x;
    const int slot = blockIdx.y, tid = threadIdx.x;
    const float* src = tl + (size_t)slot * slot_stride + tile * 4096;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = i * 256 + tid;
        t[(e >> 5) * 33 + (e & 31)] = src[e];
    }
    __syncthreads();
    float* dst = rows + ((size_t)slot * n_rows + tile * 32) * 128;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = i * 256 + tid, r = e >> 7, k = e & 127;
        if (tile * 32 + r < n_rows) dst[(size_t)r * 128 + k] = t[k * 33 + r];
    }
}

hipError_t launch_tl_to_rows(const float* tl, long slot_stride, int n_slots, long n_rows, long n_tiles, float* rows, hipStream_t st) {
    if (n_tiles <= 0 || n_tiles > 0x7fffffffL || n_slots <= 0 || n_slots > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tl_to_rows_kernel, dim3((unsigned)n_tiles, (unsigned)n_slots), dim3(256), 0, st, tl, slot_stride, n_rows, rows);
    return hipGetLastError();
}

// Row r of the (B*V*R*S)-row per-view tensors -> batch-view index, global ray, global sample index.
struct ViewRow {
    int bv, b, ray;
    long g;
};
__device__ __forceinline__ ViewRow view_row(const FieldParams& p, long row) {
    const long rs = (long)p.R * p.S;
    const long total_view = rs * p.B * p.V;
    if (row >= total_view) row = total_view - 1;
    ViewRow vr;
    vr.bv = (int)(row / rs);
    const long in_b = row - (long)vr.bv * rs;
    vr.b = vr.bv / p.V;
    vr.ray = vr.b * p.R + (int)(in_b / p.S);
    vr.g = (long)vr.b * rs + in_b;
    return vr;
}

// g_view[(b*V+v)*tpb + k] = g_fused[b*tpb + k] / V   (backward of reduce_mean over views, layers.py:368-370)
__global__ void view_broadcast_kernel(const float* __restrict__ g_fused, int V, long tiles_per_b, long n_tiles,
                                      float* __restrict__ g_view) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // over V * n_tiles * 4096 / 4 float4s
    const long per_tile = 1024;
    if (i >= (long)V * n_tiles * per_tile) return;
    const long vt = i / per_tile, off = i % per_tile;
    const long bv = vt / tiles_per_b, k = vt % tiles_per_b;
    const f32x4 v = reinterpret_cast<const f32x4*>(g_fused)[((bv / V) * tiles_per_b + k) * per_tile + off];
    const float inv = 1.0f / (float)V;
    f32x4 o = {v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
    reinterpret_cast<f32x4*>(g_view)[i] = o;
}

hipError_t launch_view_broadcast(const float* g_fused, int V, long tiles_per_b, long n_tiles, float* g_view, hipStream_t st) {
    const long n = (long)V * n_tiles * 1024;
    hipLaunchKernelGGL(view_broadcast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g_fused, V, tiles_per_b,
                       n_tiles, g_view);
    return hipGetLastError();
}

// ---- layer-0 weight gradient: dW0[k][n] += sum_rows X0[k] g0[n], db0 += sum g0 (all view rows) ---------------
// X0 = [PE(cam xyz) 60 | PE(cam dir) 60 | 2 rgb - 1 (3) | features 256] is recomputed per tile (dw0_split8_kernel below).
struct SampleGeom {       // per sample, in LDS
    float cam[3], dir[3], ax, ay;
    int tl;
};

// ---- the GEMM on the bf16 matrix pipe (exactly cut operands), one 8-wave workgroup per CU ------------------------------------------
// The round-1 form (two launches of 4-wave workgroups, one wave per 32-row block, fp32 MFMA) spent 768 fp32 MFMAs per tile (49 k matrix
// cycles), recomputed the geometry in each of its 12 waves and, like every wave issuing v_mfma_f32_32x32x2_f32, could not hide its
// gathers / lerps behind its own MFMAs (1 343 us at 16 384 tiles).  Here, per tile of 32 samples:
//   * X0 (384 rows x 32 samples) is built ONCE by the 512 threads, cut into three bf16 pieces and laid out in LDS in A-operand order
//     (XA, 72 KiB); G is cut once into B-operand order (PB, 24 KiB) - as in dense_bwd_split8_kernel;
//   * wave v owns output block nb = v % 4 of six row blocks (6 (v / 4) ..): 72 bf16 MFMAs per tile, 96 accumulator registers;
//   * software pipeline over the workgroup's tiles: while the MFMAs of tile t run, the 64 feature taps per thread of tile t+1 are in
//     flight to registers and the PE table / rgb rows of tile t+1 and the geometry of tile t+2 are filled (spread over all threads).
struct GeomQ {            // what the gathers and lerps of a sample need, 16 bytes
    float ax, ay;
    int tl, pad;
};
constexpr int kPe8Row = 125;                                  // floats per sample of the PE + rgb table (odd: conflict-free)
constexpr int kDw8XA = 0, kDw8PB = kDw8XA + 12 * 2 * 3 * 64, kDw8Raw = kDw8PB + 2 * 4 * 3 * 64, kDw8Pe = kDw8Raw + 1024;   // float4 units
constexpr int kDw8Geom = kDw8Pe + (32 * kPe8Row + 3) / 4, kDw8GeomQ = kDw8Geom + 2 * 32 * 9 / 4;
constexpr int kDw8LdsBytes = (kDw8GeomQ + 2 * 32) * 16;

#ifndef MVT_DW0_ABL
#define MVT_DW0_ABL 0      // timing-only ablations (wrong results), bits: 1 no feature gathers, 2 no PE / rgb / geometry, 4 no MFMAs, 8 no build
#endif
__global__ __launch_bounds__(512, 1) void dw0_split8_kernel(FieldParams p, const float* __restrict__ g0_tl, float* __restrict__ dW0,
                                                            float* __restrict__ db0, long part_stride, const float* __restrict__ amax_in) {
    extern __shared__ __attribute__((aligned(16))) f32x4 sbuf[];
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32x4_t* sP = reinterpret_cast<u32x4_t*>(sbuf);
    float* pe_tab = reinterpret_cast<float*>(sbuf + kDw8Pe);
    SampleGeom* geom = reinterpret_cast<SampleGeom*>(sbuf + kDw8Geom);          // [2][32]
    GeomQ* geomq = reinterpret_cast<GeomQ*>(sbuf + kDw8GeomQ);                  // [2][32]
    const int view_tiles = (int)(p.n_tiles * p.V), stride = (int)gridDim.x;
#if MVT_BWD_F16
    // fp16 two-piece products as in dense_bwd_split8_kernel: the inputs as (rn16(x / 64), rn16(x - 64 hi)), the gradient as the three
    // pieces of 64 (g 2^e / 64) with 2^e from the tensor's max |g|; the accumulators carry a factor 2^e / 64, taken out at the store
    float g_sc, g_inv;
    amax_scale(amax_in, lane, &g_sc, &g_inv);
    const float g_pre = g_sc * 0.015625f, out_scale = g_inv * 64.0f;
#else
    const float out_scale = 1.0f;
    (void)amax_in;
#endif
    auto sum8 = [](const f32x4& lo, const f32x4& hi, float accv) {
        float s = lo[0];
        const float t[7] = {lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int q = 0; q < 7; ++q) asm("v_add_f32_e32 %0, %1, %2" : "=v"(s) : "v"(s), "v"(t[q]));
        asm("v_add_f32_e32 %0, %1, %2" : "=v"(accv) : "v"(accv), "v"(s));
        return accv;
    };
    // ---- the pieces of work of one tile ----
    auto do_geom = [&](int tile, int slot) {                  // threads 480..511 (upper half of wave 7): sample i of the view tile
        if (tid >= 480) {
            const ViewRow vr = view_row(p, (long)tile * 32 + i);
            const int ray = vr.ray, b = vr.bv;
            const float* E = p.einv + 16 * b;
            const float zz = p.z[vr.g];
            const float dx = p.rays_d[3 * ray], dy = p.rays_d[3 * ray + 1], dz = p.rays_d[3 * ray + 2];
            const float wx = p.rays_o[3 * ray] + zz * dx, wy = p.rays_o[3 * ray + 1] + zz * dy, wz = p.rays_o[3 * ray + 2] + zz * dz;
            float cam[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) cam[r] = row_dot4(E, r, wx, wy, wz, 1.0f);
            float px, py;
            pixel_from_cam(p.k4 + 16 * b, cam, &px, &py);
            const Taps tp = bilinear_taps(px, py, p.H, p.W);
            SampleGeom sg;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                sg.cam[r] = cam[r];
                sg.dir[r] = row_dot4(E, r, dx, dy, dz, 1.0f);
            }
            sg.ax = tp.ax;
            sg.ay = tp.ay;
            sg.tl = (b * p.H + tp.y0) * p.W + tp.x0;
            geom[slot * 32 + i] = sg;
            GeomQ gq;
            gq.ax = tp.ax;
            gq.ay = tp.ay;
            gq.tl = sg.tl;
            gq.pad = 0;
            geomq[slot * 32 + i] = gq;
        }
    };
    // PE rows of the table, spread over threads 0..383: thread = (sample tid % 32, source tid / 32 % 6 = {cam x,y,z, dir x,y,z},
    // half of the octaves tid / 192): one accurate sin/cos (octave 0 or 5) and the double-angle recurrence for the next four
    auto do_pe = [&](int slot) {
        if (tid < 384) {
            const int smp = tid & 31, src = (tid >> 5) % 6, half = tid / 192;
            const SampleGeom sgm = geom[slot * 32 + smp];
            const float x = src == 0 ? sgm.cam[0] : src == 1 ? sgm.cam[1] : src == 2 ? sgm.cam[2] : src == 3 ? sgm.dir[0] : src == 4 ? sgm.dir[1] : sgm.dir[2];
            const float a0 = x * 3.14159274101257324f;
            float sk, ck;
            sincos_f32(a0 * (half ? 32.0f : 1.0f), &sk, &ck);
            float* dst = pe_tab + smp * kPe8Row + 20 * src + 10 * half;       // rows 60 (src / 3) + 20 (src % 3) + 2k + {sin, cos} = 20 src + 2k
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                dst[2 * k] = sk;
                dst[2 * k + 1] = ck;
                const float s2 = sk + sk;                     // octave k + 1 is the double angle of octave k (field_eval.hip)
                const float cn = fmaf(-s2, sk, 1.0f);
                sk = s2 * ck;
                ck = cn;
            }
        }
    };
    // rgb rows 120..122: threads 384..479 = (sample, channel)
    auto do_rgb = [&](int slot) {
        if (tid >= 384 && tid < 480) {
            const int q = tid - 384, smp = q & 31, c = q >> 5;
            const GeomQ gq = geomq[slot * 32 + smp];
            const float* im = p.images + 3 * (long)gq.tl + c;
            pe_tab[smp * kPe8Row + 120 + c] = bilerp(im[0] * 2.0f - 1.0f, im[3] * 2.0f - 1.0f, im[3 * p.W] * 2.0f - 1.0f,
                                                     im[3 * p.W + 3] * 2.0f - 1.0f, gq.ax, gq.ay);
        }
    };
    // feature rows: thread = (channel c = tid % 256, sample-group parity tid / 256); pass ps covers samples 8 (2 ps + tid / 256) + q
    const int fc = tid & 255, fsg = tid >> 8;
    float tap[2][8][4];
    auto issue_gathers = [&](int slot) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int tl = __builtin_amdgcn_readfirstlane(geomq[slot * 32 + 8 * (2 * ps + fsg) + q].tl);
                const float* f = p.features + 256 * (long)tl + fc;
                tap[ps][q][0] = f[0];
                tap[ps][q][1] = f[256];
                tap[ps][q][2] = f[256 * (long)p.W];
                tap[ps][q][3] = f[256 * (long)p.W + 256];
            }
    };
    auto dma_g = [&](int tile) {                              // raw G tile, XOR-swizzled on the source side (dense_dx_kernel): 2 wave-instructions per wave
        const unsigned so = (unsigned)((lane & 56) + ((lane & 7) ^ (lane >> 3))) * 16u;
        const char* src = reinterpret_cast<const char*>(g0_tl) + (long)tile * 16384 + 2048 * v;
#pragma unroll
        for (int m = 0; m < 2; ++m)
            __builtin_amdgcn_global_load_lds((gptr)(src + 1024 * m + so), (lptr)(sbuf + kDw8Raw + 128 * v + 64 * m), 16, 0, 0);
    };
    float dbacc = 0.0f;
    auto build = [&](int slot, bool count_db) {               // everything of a tile that goes into XA / PB
        // feature rows 123 + c: the two units (k-step ps, half fsg) of this thread's channel
        const int r = 123 + fc, kb = r >> 5, li = (r & 31) + 32 * fsg;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            f32x4 lo, hi;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const GeomQ gq = geomq[slot * 32 + 8 * (2 * ps + fsg) + q];
                // (fused multiply-adds: the recomputed input does not have to round like the forward's lerp)
                const float top = fmaf(gq.ax, tap[ps][q][1] - tap[ps][q][0], tap[ps][q][0]);
                const float bot = fmaf(gq.ax, tap[ps][q][3] - tap[ps][q][2], tap[ps][q][2]);
                const float val = fmaf(gq.ay, bot - top, top);
                if (q < 4) lo[q] = val;
                else hi[q - 4] = val;
            }
            u32x4_t p0, p1, p2;
            u32x4_t* xa = sP + kDw8XA + ((kb * 2 + ps) * 3) * 64 + li;
#if MVT_BWD_F16
            cut2h_scaled<false>(lo, hi, 0.015625f, 1.0f, p0, p1);
            xa[0] = p0;
            xa[64] = p1;
            (void)p2;
#else
            cut3<false>(lo, hi, p0, p1, p2);
            xa[0] = p0;
            xa[64] = p1;
            xa[128] = p2;
#endif
        }
        // rows 0..122 from the PE + rgb table: thread = (row tid % 128, sample group tid / 128)
        {
            const int pr = tid & 127, sg = tid >> 7;
            if (pr < 123) {
                f32x4 lo, hi;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    lo[q] = pe_tab[(8 * sg + q) * kPe8Row + pr];
                    hi[q] = pe_tab[(8 * sg + 4 + q) * kPe8Row + pr];
                }
                u32x4_t p0, p1, p2;
                u32x4_t* xa = sP + kDw8XA + (((pr >> 5) * 2 + (sg >> 1)) * 3) * 64 + (pr & 31) + 32 * (sg & 1);
#if MVT_BWD_F16
                cut2h_scaled<false>(lo, hi, 0.015625f, 1.0f, p0, p1);
                xa[0] = p0;
                xa[64] = p1;
                (void)p2;
#else
                cut3<false>(lo, hi, p0, p1, p2);
                xa[0] = p0;
                xa[64] = p1;
                xa[128] = p2;
#endif
            }
        }
        // G: chunk (ks = v / 4, nb = v % 4) of the sample-packed image, and this wave's share of the bias gradient
        {
            const f32x4 lo = sbuf[kDw8Raw + swz_f4(i, 4 * (v >> 2) + 2 * h) + 256 * (v & 3)];
            const f32x4 hi = sbuf[kDw8Raw + swz_f4(i, 4 * (v >> 2) + 2 * h + 1) + 256 * (v & 3)];
            u32x4_t p0, p1, p2;
#if MVT_BWD_F16
            cut3a<false>(lo * g_pre, hi * g_pre, p0, p1, p2);             // (64 t, 64 t / 64, remainder) of t = g 2^e / 64
#else
            cut3<false>(lo, hi, p0, p1, p2);
#endif
            u32x4_t* pb = sP + kDw8PB + (v * 3) * 64 + lane;
            pb[0] = p0;
            pb[64] = p1;
            pb[128] = p2;
            if (count_db) dbacc = sum8(lo, hi, dbacc);        // (not for the clamped re-build behind the last tile)
        }
    };

    f32x16 acc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    // rows 379..383 do not exist: their XA slots (row block 11, rows 27..31) stay zero
    for (int q = tid; q < 12 * 2 * 3 * 64; q += 512) sP[kDw8XA + q] = u32x4_t{0u, 0u, 0u, 0u};
    // ---- prologue: tile t0 built, geometry of t1 ready ----
    const int t0 = (int)blockIdx.x;
    const int t1 = t0 + stride < view_tiles ? t0 + stride : t0;
    dma_g(t0);
    do_geom(t0, 0);
    do_geom(t1, 1);
    __syncthreads();
    do_pe(0);
    do_rgb(0);
    issue_gathers(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    build(0, true);
    int it = 0;
    for (int tile = t0; tile < view_tiles; tile += stride, ++it) {
        const int cur = it & 1;
        const int tn = tile + stride < view_tiles ? tile + stride : tile;          // next tile (clamped: a harmless rebuild at the end)
        const int tnn = tn + stride < view_tiles ? tn + stride : tn;
        __syncthreads();                                       // B_a: XA, PB of `tile` complete; geometry of tn in slot cur ^ 1
        dma_g(tn);
        if (!(MVT_DW0_ABL & 1)) issue_gathers(cur ^ 1);
        if (!(MVT_DW0_ABL & 2)) {
            do_pe(cur ^ 1);
            do_rgb(cur ^ 1);
            do_geom(tnn, cur);
        }
        // ---- 72 MFMAs: output block nb = v % 4 of row blocks 6 (v / 4) + {0..5} ----
        if (!(MVT_DW0_ABL & 4)) {
            const int nb = v & 3;
            const u32x4_t* pb = sP + kDw8PB + lane;
            u32x4_t b[2][3];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int q = 0; q < 3; ++q) b[ks][q] = pb[((ks * 4 + nb) * 3 + q) * 64];
            const u32x4_t* xa = sP + kDw8XA + (6 * (v >> 2)) * 2 * 3 * 64 + lane;
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#if MVT_BWD_F16
                    const u32x4_t a0 = xa[((k * 2 + ks) * 3 + 0) * 64], a1 = xa[((k * 2 + ks) * 3 + 1) * 64];
                    acc[k] = mfma16h(a1, b[ks][1], acc[k]);         // (64 remainder of x / 64) x (G0 / 64)
                    acc[k] = mfma16h(a0, b[ks][2], acc[k]);         // x hi x remainder of G
                    acc[k] = mfma16h(a0, b[ks][0], acc[k]);
#else
                    const u32x4_t a0 = xa[((k * 2 + ks) * 3 + 0) * 64], a1 = xa[((k * 2 + ks) * 3 + 1) * 64], a2 = xa[((k * 2 + ks) * 3 + 2) * 64];
                    acc[k] = mfma16s(a2, b[ks][0], acc[k]);
                    acc[k] = mfma16s(a1, b[ks][1], acc[k]);
                    acc[k] = mfma16s(a0, b[ks][2], acc[k]);
                    acc[k] = mfma16s(a1, b[ks][0], acc[k]);
                    acc[k] = mfma16s(a0, b[ks][1], acc[k]);
                    acc[k] = mfma16s(a0, b[ks][0], acc[k]);
#endif
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // taps and G of the next tile have landed
        __syncthreads();                                       // B_b: every wave is done reading XA / PB; PE table of tn complete
        if (!(MVT_DW0_ABL & 8)) build(cur ^ 1, tile + stride < view_tiles);
    }
    __syncthreads();
    // ---- outputs ----
    float* xch = reinterpret_cast<float*>(sbuf + kDw8PB);     // bias gradient: waves v and v + 4 hold the two k-steps of block v % 4
    if (v >= 4) xch[64 * (v - 4) + lane] = dbacc;
    __syncthreads();
    const bool store = part_stride != 0;
    dW0 += (long)blockIdx.x * part_stride;
    db0 += (long)blockIdx.x * part_stride;
    if (v < 4) {
        float sdb = dbacc + xch[64 * v + lane];
        sdb = sdb + __shfl_xor(sdb, 32);
        if (h == 0) grad_out(db0 + 32 * v + i, sdb, store);
    }
    const int col = lane & 31, hh = lane >> 5, nb = v & 3;
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * (6 * (v >> 2) + k) + acc_row(r, hh);
            if (row < kIn) grad_out(dW0 + (long)row * kHidden + 32 * nb + col, acc[k][r] * out_scale, store);
        }
}

hipError_t launch_dw0(const FieldParams& p, const float* g0_tl, float* dW0, float* db0, int max_wgs, float* part, hipStream_t st,
                      const float* amax_in) {
    if (MVT_BWD_F16 && !amax_in) return hipErrorInvalidValue;                  // the fp16 products need the gradient's scale
    const long view_tiles = p.n_tiles * p.V;
    if (view_tiles <= 0 || view_tiles > 0x7fffffffL) return hipErrorInvalidValue;
    static std::atomic<bool> attr_done[16];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16 && !attr_done[dev].load(std::memory_order_acquire)) {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw0_split8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kDw8LdsBytes)) != hipSuccess)
            return e;
        attr_done[dev].store(true, std::memory_order_release);
    }
    if (!part || db0 != dW0 + kIn * kHidden) return hipErrorInvalidValue;      // one span [dW0 | db0], reduced from per-workgroup partials
    const int span = kIn * kHidden + kHidden;
    const unsigned wgs = (unsigned)(view_tiles < max_wgs / 2 ? view_tiles : max_wgs / 2);      // one 512-thread workgroup per CU
    hipLaunchKernelGGL(dw0_split8_kernel, dim3(wgs), dim3(512), kDw8LdsBytes, st, p, g0_tl, part, part + kIn * kHidden, (long)span, amax_in);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_reduce_partials(part, span, (int)wgs, span, dW0, st);
}

// ---- gradient of the field w.r.t. the sample depths (single view) ----------------------------------------------
// dL/dz_sample += dir . E^-T ( dL/dcam ),  dL/dcam = PE-derivative part + K4^T dL/dq,  dL/dq from dL/dpix,
// dL/dpix = sum_channels dL/dIn_c * d(interp_c)/d(ax, ay)   (tfa bilinear form, clamps as torch.clamp).
// dL/dIn = W0 . g0 is formed per 128-row slab of W0 with the forward's weight-stream MFMA code (3 transposed
// slabs); the PE rows are reduced in "lane = sample" form, the 256 feature rows go through a wave-private LDS
// image [sample][channel] and are reduced in "lane = channel" form so that the four taps are coalesced reads.
constexpr int kDfeRow = 257;       // floats per sample row of the LDS image (odd: conflict-free scatter)

// Query mode (d_o / d_d given): the same chain ends in dL/d(ray origin) and dL/d(ray direction) per ray
// instead of dL/dz - the PE(cam dir) rows 60..119 then count too (cam dir = E^-1 [d; 1], Q3).
//
// kTable (p.texel_table given, d_features not wanted): the 256 feature rows never appear.  With T[texel] = W0[123:379]^T f[texel]
// (the forward's texel table, project_texels_kernel) the feature part of dL/d(ax, ay) is g0 . d(lerp of T rows)/d(ax, ay): four
// 128-float table rows per sample and two dot products with g0 instead of two of the three 128 x 128 slab GEMMs, the LDS image and
// 4 x 256 feature taps per sample (990 -> 260 us at 4096 x 128 samples).  A lane (sample j, half h) holds g0 in accumulator order,
// which is the order of its half of a table row.
template <bool kTable>
__global__ __launch_bounds__(256, 1) void field_dz_kernel(FieldParams p, const float* __restrict__ g0_tl,
                                                                       const float* __restrict__ w0t_streams, float* __restrict__ d_z,
                                                                       float* __restrict__ d_o, float* __restrict__ d_d,
                                                                       float* __restrict__ d_features) {
    extern __shared__ __attribute__((aligned(16))) float lds_dz[];
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile = (long)blockIdx.x * 4 + wave;          // view tile
    if (tile >= p.n_tiles * p.V) return;
    float* dfe = lds_dz + wave * (32 * kDfeRow + 64);     // [32][257] + dax[32] + day[32]
    float* dax_s = dfe + 32 * kDfeRow;
    float* day_s = dax_s + 32;

    const bool valid = tile * 32 + j < p.total * p.V;
    const ViewRow vr = view_row(p, tile * 32 + j);
    const long g = vr.g;
    const int ray = vr.ray, b = vr.bv;                      // `b` indexes the (B*V) cameras / grids below
    const float* E = p.einv + 16 * b;
    const float* K = p.k4 + 16 * b;
    const float zz = p.z ? p.z[g] : 0.0f;
    const float dx = p.rays_d[3 * ray], dy = p.rays_d[3 * ray + 1], dz = p.rays_d[3 * ray + 2];
    const float wx = p.rays_o[3 * ray] + zz * dx, wy = p.rays_o[3 * ray + 1] + zz * dy, wz = p.rays_o[3 * ray + 2] + zz * dz;
    float cam[4], cdir[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) cam[r] = row_dot4(E, r, wx, wy, wz, 1.0f);
#pragma unroll
    for (int r = 0; r < 3; ++r) cdir[r] = row_dot4(E, r, dx, dy, dz, 1.0f);
    const float q0 = row_dot4(K, 0, cam[0], cam[1], cam[2], cam[3]);
    const float q1 = row_dot4(K, 1, cam[0], cam[1], cam[2], cam[3]);
    const float q2 = row_dot4(K, 2, cam[0], cam[1], cam[2], cam[3]);
    const float den = fmaxf(q2, 1e-8f);
    const float pxr = q0 / den, pyr = q1 / den;
    const float px = fminf(fmaxf(pxr, -1e6f), 1e6f), py = fminf(fmaxf(pyr, -1e6f), 1e6f);
    const Taps tp = bilinear_taps(px, py, p.H, p.W);
    const int tl = (b * p.H + tp.y0) * p.W + tp.x0;
    const float ux = px - fminf(fmaxf(0.0f, floorf(px)), (float)(p.W - 2));      // unclamped lerp factors
    const float uy = py - fminf(fmaxf(0.0f, floorf(py)), (float)(p.H - 2));

    f32x16 bin[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) bin[kb][r] = g0_tl[tl_index(tile, 128, 32 * kb + acc_row(r, h), j)];

    float dcam[3] = {0.0f, 0.0f, 0.0f}, dcdir[3] = {0.0f, 0.0f, 0.0f};
    float dax = 0.0f, day = 0.0f;
#pragma unroll 1
    for (int slab = 0; slab < (kTable ? 1 : 3); ++slab) {
        f32x16 acc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
        WStream ws;
        ws_begin(ws, w0t_streams + (size_t)slab * kHiddenWFloats, kHiddenWFloats * 4, lane);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float bq[4] = {bin[kb][4 * t], bin[kb][4 * t + 1], bin[kb][4 * t + 2], bin[kb][4 * t + 3]};
                mfma_step(ws, bq, acc);
            }
        // acc[nb][r] = dL/dIn[row = 128*slab + 32*nb + acc_row(r,h)] of sample j
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 128 * slab + 32 * nb + acc_row(r, h);
                const float v = acc[nb][r];
                if (slab == 0 && row < 60) {
                    // PE(cam xyz) rows come as (sin, cos) pairs in adjacent registers; handle the pair at the sin row
                    if ((r & 1) == 0) {
                        const int d = row / 20, oct = (row % 20) >> 1;
                        const float f = 3.14159274101257324f * (float)(1 << oct);
                        float sv, cv;
                        sincos_f32(cam[d] * f, &sv, &cv);
                        const float contrib = f * (v * cv - acc[nb][r + 1] * sv);
                        if (d == 0) dcam[0] += contrib;
                        else if (d == 1) dcam[1] += contrib;
                        else dcam[2] += contrib;
                    }
                } else if (slab == 0 && row < 120) {
                    if (d_d && (r & 1) == 0) {                  // PE(cam dir) rows, same pairing
                        const int d = (row - 60) / 20, oct = ((row - 60) % 20) >> 1;
                        const float f = 3.14159274101257324f * (float)(1 << oct);
                        float sv, cv;
                        sincos_f32(cdir[d] * f, &sv, &cv);
                        const float contrib = f * (v * cv - acc[nb][r + 1] * sv);
                        if (d == 0) dcdir[0] += contrib;
                        else if (d == 1) dcdir[1] += contrib;
                        else dcdir[2] += contrib;
                    }
                } else if (row >= 120 && row < 123) {
                    // rgb rows: taps of this lane's own sample (scalar loads)
                    const float* im = p.images + 3 * (long)tl + (row - 120);
                    const float a = im[0] * 2.0f - 1.0f, bq = im[3] * 2.0f - 1.0f, cq = im[3 * p.W] * 2.0f - 1.0f,
                                dq = im[3 * p.W + 3] * 2.0f - 1.0f;
                    dax += v * ((1.0f - tp.ay) * (bq - a) + tp.ay * (dq - cq));
                    day += v * ((cq - a) + tp.ax * ((dq - cq) - (bq - a)));
                } else if (!kTable && row >= 123 && row < 379) {
                    dfe[j * kDfeRow + (row - 123)] = v;
                }
            }
    }
    if (kTable) {
        // this lane's half (64 h .. 64 h + 63, accumulator order: 16 nb + r <-> feature 32 nb + acc_row(r, h)) of the four table rows
        const f32x4* T = reinterpret_cast<const f32x4*>(p.texel_table) + 32 * (long)tl + 16 * h;
        const long rowstep = 32 * (long)p.W;
        const float one_m_ay = 1.0f - tp.ay;
        float pa = 0.0f, pb = 0.0f;
        asm volatile("" ::: "memory");                      // the table loads stay behind the slab GEMM (register budget: 2 waves / SIMD)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            f32x4 trow[4][4];                               // one output block at a time: 16 loads in flight
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                trow[q][0] = T[4 * nb + q];
                trow[q][1] = T[32 + 4 * nb + q];
                trow[q][2] = T[rowstep + 4 * nb + q];
                trow[q][3] = T[rowstep + 32 + 4 * nb + q];
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 ttl = trow[q][0], ttr = trow[q][1], tbl = trow[q][2], tbr = trow[q][3];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float gv = bin[nb][4 * q + c];
                    const float dt = ttr[c] - ttl[c], db = tbr[c] - tbl[c];
                    pa += gv * (one_m_ay * dt + tp.ay * db);
                    pb += gv * ((tbl[c] - ttl[c]) + tp.ax * (db - dt));
                }
            }
        }
        dax += pa;                                          // (the two halves are added below)
        day += pb;
    }
    if (!kTable && lane < 32) {
        dax_s[lane] = 0.0f;
        day_s[lane] = 0.0f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // feature channels, lane = channel (4 passes of 64), one sample at a time: coalesced tap reads
#pragma unroll 1
    for (int sidx = 0; sidx < (kTable ? 0 : 32); ++sidx) {
        const int tls = __shfl(tl, sidx);
        const float axs = __shfl(tp.ax, sidx), ays = __shfl(tp.ay, sidx);
        const float* f = p.features + 256 * (long)tls;
        const bool live = __shfl((int)valid, sidx) != 0;
        float pa = 0.0f, pb = 0.0f;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int c = 64 * pass + lane;
            const float v = dfe[sidx * kDfeRow + c];
            const float vtl = f[c], vtr = f[256 + c], vbl = f[256 * (long)p.W + c], vbr = f[256 * (long)p.W + 256 + c];
            pa += v * ((1.0f - ays) * (vtr - vtl) + ays * (vbr - vbl));
            pb += v * ((vbl - vtl) + axs * ((vbr - vbl) - (vtr - vtl)));
            if (d_features && live) {
                // gradient w.r.t. the source feature map: the sample's dL/d(lerped features) goes back to its four taps
                // with the bilinear weights (the scatter of SURVEY.md 7.7; what an upstream encoder trains on)
                float* df = d_features + 256 * (long)tls + c;
                atomicAdd(df, v * ((1.0f - axs) * (1.0f - ays)));
                atomicAdd(df + 256, v * (axs * (1.0f - ays)));
                atomicAdd(df + 256 * (long)p.W, v * ((1.0f - axs) * ays));
                atomicAdd(df + 256 * (long)p.W + 256, v * (axs * ays));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            pa += __shfl_xor(pa, off);
            pb += __shfl_xor(pb, off);
        }
        if (lane == 0) {
            dax_s[sidx] = pa;
            day_s[sidx] = pb;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // combine the two half-waves' PE / rgb parts, add the feature part of this lane's sample
#pragma unroll
    for (int d = 0; d < 3; ++d) dcam[d] += __shfl_xor(dcam[d], 32);
    dax += __shfl_xor(dax, 32);
    day += __shfl_xor(day, 32);
    if (!kTable) {
        dax += dax_s[j];
        day += day_s[j];
    }
    // clamps (torch.clamp semantics: gradient passes inside the closed range)
    const float dpx = (ux >= 0.0f && ux <= 1.0f && pxr >= -1e6f && pxr <= 1e6f) ? dax : 0.0f;
    const float dpy = (uy >= 0.0f && uy <= 1.0f && pyr >= -1e6f && pyr <= 1e6f) ? day : 0.0f;
    const float dq0 = dpx / den, dq1 = dpy / den;
    const float dq2 = q2 >= 1e-8f ? -(dpx * q0 + dpy * q1) / (den * den) : 0.0f;
    float dc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) dc[c] = K[c] * dq0 + K[4 + c] * dq1 + K[8 + c] * dq2;
#pragma unroll
    for (int c = 0; c < 3; ++c) dc[c] += dcam[c];
    float dzv = 0.0f;
    const float dirv[3] = {dx, dy, dz};
#pragma unroll
    for (int c = 0; c < 3; ++c) dcdir[c] += __shfl_xor(dcdir[c], 32);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float dworld = E[a] * dc[0] + E[4 + a] * dc[1] + E[8 + a] * dc[2] + E[12 + a] * dc[3];
        dzv += dworld * dirv[a];
        if (d_o && valid && h == 0) {                       // the samples of a ray and its V views add up
            atomicAdd(d_o + 3 * ray + a, dworld);
            atomicAdd(d_d + 3 * ray + a, zz * dworld + (E[a] * dcdir[0] + E[4 + a] * dcdir[1] + E[8 + a] * dcdir[2]));
        }
    }
    if (d_z && valid && h == 0) atomicAdd(d_z + g, dzv);    // the V views of a sample add up
}

// ---- the training form of the sample-depth gradient with the texel table: d_z only --------------------------------------------------
// Of W0 . g0 only the rows PE(cam xyz) (0..59) and rgb (120..122) are needed: the first two 32-row blocks of slab 0 go through the MFMA
// (128 instead of 256 fp32 MFMAs per tile, 32 instead of 64 accumulator registers: two waves per SIMD), the three rgb rows are 3 x 64
// fused multiply-adds per lane against W0 rows staged in LDS in accumulator order, the feature rows come from the table as in
// field_dz_kernel<true>.  358 -> see DESIGN.md section 8.
__global__ __launch_bounds__(256, 2) void field_dz_table_kernel(FieldParams p, const float* __restrict__ g0_tl,
                                                                const float* __restrict__ w0t_slab0, const float* __restrict__ w0_rgb,
                                                                float* __restrict__ d_z) {
    __shared__ __attribute__((aligned(16))) float s_rgb[3][2][64];
    for (int t = threadIdx.x; t < 384; t += 256) {
        const int c = t >> 7, pos = t & 127, hh = pos >> 6, q = pos & 63;
        s_rgb[c][hh][q] = w0_rgb[c * kHidden + 32 * (q >> 4) + acc_row(q & 15, hh)];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile = (long)blockIdx.x * 4 + wave;          // view tile
    if (tile >= p.n_tiles * p.V) return;
    const bool valid = tile * 32 + j < p.total * p.V;
    const ViewRow vr = view_row(p, tile * 32 + j);
    const long g = vr.g;
    const int ray = vr.ray, b = vr.bv;                      // `b` indexes the (B*V) cameras / grids below
    const float* E = p.einv + 16 * b;
    const float* K = p.k4 + 16 * b;
    const float zz = p.z[g];
    const float dx = p.rays_d[3 * ray], dy = p.rays_d[3 * ray + 1], dz = p.rays_d[3 * ray + 2];
    const float wx = p.rays_o[3 * ray] + zz * dx, wy = p.rays_o[3 * ray + 1] + zz * dy, wz = p.rays_o[3 * ray + 2] + zz * dz;
    float cam[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cam[r] = row_dot4(E, r, wx, wy, wz, 1.0f);
    const float q0 = row_dot4(K, 0, cam[0], cam[1], cam[2], cam[3]);
    const float q1 = row_dot4(K, 1, cam[0], cam[1], cam[2], cam[3]);
    const float q2 = row_dot4(K, 2, cam[0], cam[1], cam[2], cam[3]);
    const float den = fmaxf(q2, 1e-8f);
    const float pxr = q0 / den, pyr = q1 / den;
    const float px = fminf(fmaxf(pxr, -1e6f), 1e6f), py = fminf(fmaxf(pyr, -1e6f), 1e6f);
    const Taps tp = bilinear_taps(px, py, p.H, p.W);
    const int tl = (b * p.H + tp.y0) * p.W + tp.x0;
    const float ux = px - fminf(fmaxf(0.0f, floorf(px)), (float)(p.W - 2));      // unclamped lerp factors
    const float uy = py - fminf(fmaxf(0.0f, floorf(py)), (float)(p.H - 2));

    f32x16 bin[4];                                          // g0 of sample j in accumulator order: bin[kb][r] = g0[32 kb + acc_row(r, h)]
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) bin[kb][r] = g0_tl[tl_index(tile, 128, 32 * kb + acc_row(r, h), j)];

    // ---- rows 0..63 of W0 . g0: chunks (grp, nb = 0, 1) of the transposed slab-0 stream, requested one group ahead ----
    f32x16 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
    {
        const f32x4* ws = reinterpret_cast<const f32x4*>(w0t_slab0) + lane;
        f32x4 c0 = ws[0], c1 = ws[64];
#pragma unroll
        for (int grp = 0; grp < 16; ++grp) {
            f32x4 n0 = c0, n1 = c1;
            if (grp < 15) {
                n0 = ws[256 * (grp + 1)];
                n1 = ws[256 * (grp + 1) + 64];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bv = bin[grp >> 2][4 * (grp & 3) + e];
                acc[0] = mfma(c0[e], bv, acc[0]);
                acc[1] = mfma(c1[e], bv, acc[1]);
            }
            c0 = n0;
            c1 = n1;
        }
    }
    float dcam[3] = {0.0f, 0.0f, 0.0f};
    float dax = 0.0f, day = 0.0f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int row = 32 * nb + acc_row(r, h);        // PE(cam xyz) rows come as (sin, cos) pairs in adjacent registers
            if (row < 60) {
                const int d = row / 20, oct = (row % 20) >> 1;
                const float f = 3.14159274101257324f * (float)(1 << oct);
                float sv, cv;
                sincos_f32(cam[d] * f, &sv, &cv);
                const float contrib = f * (acc[nb][r] * cv - acc[nb][r + 1] * sv);
                if (d == 0) dcam[0] += contrib;
                else if (d == 1) dcam[1] += contrib;
                else dcam[2] += contrib;
            }
        }
    // ---- rgb rows 120..122: this lane's half of the 128-term sums, both halves added, used by the lower half-wave ----
    {
        float v3[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4* wr = reinterpret_cast<const f32x4*>(&s_rgb[c][h][0]);
            float s = 0.0f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w4 = wr[4 * kb + q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) s = fmaf(w4[e], bin[kb][4 * q + e], s);
                }
            v3[c] = s + __shfl_xor(s, 32);
        }
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* im = p.images + 3 * (long)tl + c;
                const float a = im[0] * 2.0f - 1.0f, bq = im[3] * 2.0f - 1.0f, cq = im[3 * p.W] * 2.0f - 1.0f, dq = im[3 * p.W + 3] * 2.0f - 1.0f;
                dax += v3[c] * ((1.0f - tp.ay) * (bq - a) + tp.ay * (dq - cq));
                day += v3[c] * ((cq - a) + tp.ax * ((dq - cq) - (bq - a)));
            }
        }
    }
    // ---- feature rows through the table: this lane's half (64 h .., accumulator order) of the four table rows ----
    {
        const f32x4* T = reinterpret_cast<const f32x4*>(p.texel_table) + 32 * (long)tl + 16 * h;
        const long rowstep = 32 * (long)p.W;
        const float one_m_ay = 1.0f - tp.ay;
        float pa = 0.0f, pb = 0.0f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            f32x4 trow[4][4];                               // one output block at a time: 16 loads in flight
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                trow[q][0] = T[4 * nb + q];
                trow[q][1] = T[32 + 4 * nb + q];
                trow[q][2] = T[rowstep + 4 * nb + q];
                trow[q][3] = T[rowstep + 32 + 4 * nb + q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float gv = bin[nb][4 * q + c];
                    const float dt = trow[q][1][c] - trow[q][0][c], db = trow[q][3][c] - trow[q][2][c];
                    pa += gv * (one_m_ay * dt + tp.ay * db);
                    pb += gv * ((trow[q][2][c] - trow[q][0][c]) + tp.ax * (db - dt));
                }
        }
        dax += pa;
        day += pb;
    }
    // combine the two half-waves
#pragma unroll
    for (int d = 0; d < 3; ++d) dcam[d] += __shfl_xor(dcam[d], 32);
    dax += __shfl_xor(dax, 32);
    day += __shfl_xor(day, 32);
    // clamps (torch.clamp semantics: gradient passes inside the closed range)
    const float dpx = (ux >= 0.0f && ux <= 1.0f && pxr >= -1e6f && pxr <= 1e6f) ? dax : 0.0f;
    const float dpy = (uy >= 0.0f && uy <= 1.0f && pyr >= -1e6f && pyr <= 1e6f) ? day : 0.0f;
    const float dq0 = dpx / den, dq1 = dpy / den;
    const float dq2 = q2 >= 1e-8f ? -(dpx * q0 + dpy * q1) / (den * den) : 0.0f;
    float dc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) dc[c] = K[c] * dq0 + K[4 + c] * dq1 + K[8 + c] * dq2;
#pragma unroll
    for (int c = 0; c < 3; ++c) dc[c] += dcam[c];
    float dzv = 0.0f;
    const float dirv[3] = {dx, dy, dz};
#pragma unroll
    for (int a = 0; a < 3; ++a) dzv += (E[a] * dc[0] + E[4 + a] * dc[1] + E[8 + a] * dc[2] + E[12 + a] * dc[3]) * dirv[a];
    if (valid && h == 0) atomicAdd(d_z + g, dzv);           // the V views of a sample add up
}

// ---- gradient w.r.t. the source feature maps through the texel table (mvnerf_field_backward_table with texel_grad) ---------------
// First half: dL/dT[texel][n] += (bilinear weight) x g0[sample][n] on the sample's four taps - 128 table channels instead of 256 feature
// channels per tap, and no W0 . g0 product per sample.  One wave per view tile, lane = channel pair (n, n + 64): the lane holds its two
// rows of the g0 tile (32 samples each), the samples' tap texel and weights are wave-uniform (readlane), so an atomic instruction adds 64
// consecutive floats of ONE table row (with lane = sample the 32 samples of a ray pile onto the same few addresses: 40 ms per step),
// and consecutive samples that fall into the same texel are summed in registers first.
__global__ __launch_bounds__(256) void texel_scatter_kernel(FieldParams p, const float* __restrict__ g0_tl, float* __restrict__ texel_grad) {
    const int lane = threadIdx.x & 63, j = lane & 31;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // view tile
    if (tile >= p.n_tiles * p.V) return;
    // geometry of sample j (both half-waves compute it; lanes 0..31 are read)
    const bool valid = tile * 32 + j < p.total * p.V;
    const ViewRow vr = view_row(p, tile * 32 + j);
    const int ray = vr.ray, b = vr.bv;
    const float* E = p.einv + 16 * b;
    const float zz = p.z ? p.z[vr.g] : 0.0f;
    const float dx = p.rays_d[3 * ray], dy = p.rays_d[3 * ray + 1], dz = p.rays_d[3 * ray + 2];
    const float wx = p.rays_o[3 * ray] + zz * dx, wy = p.rays_o[3 * ray + 1] + zz * dy, wz = p.rays_o[3 * ray + 2] + zz * dz;
    float cam[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cam[r] = row_dot4(E, r, wx, wy, wz, 1.0f);
    float px, py;
    pixel_from_cam(p.k4 + 16 * b, cam, &px, &py);
    const Taps tp = bilinear_taps(px, py, p.H, p.W);
    const int tl = valid ? (b * p.H + tp.y0) * p.W + tp.x0 : -1;
    // this lane's two channel rows of the tile: features lane and lane + 64, 32 samples each
    f32x4 ga[8], gb[8];
    {
        const f32x4* ra = reinterpret_cast<const f32x4*>(g0_tl + tl_index(tile, 128, lane, 0));
        const f32x4* rb = reinterpret_cast<const f32x4*>(g0_tl + tl_index(tile, 128, lane + 64, 0));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            ga[q] = ra[q];
            gb[q] = rb[q];
        }
    }
    float a0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, a1[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // taps (0,0), (0,1), (1,0), (1,1) of channels lane / lane + 64
    int cur = -1;                                            // top-left texel of the 2 x 2 window the accumulators belong to
    const long rs = 128 * (long)p.W;
    auto flush_tap = [&](int t) {                            // t = 2 * dy + dx
        float* G = texel_grad + 128 * (long)cur + lane + (t & 1) * 128 + (t >> 1) * rs;
        atomicAdd(G, a0[t]);
        atomicAdd(G + 64, a1[t]);
        a0[t] = 0.0f;
        a1[t] = 0.0f;
    };
    auto move_tap = [&](int to, int from) {
        a0[to] = a0[from];
        a1[to] = a1[from];
        a0[from] = 0.0f;
        a1[from] = 0.0f;
    };
#pragma unroll
    for (int sidx = 0; sidx < 32; ++sidx) {
        const int tls = __builtin_amdgcn_readlane(tl, sidx);
        if (tls < 0) continue;                               // padded sample (wave-uniform)
        const float axs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tp.ax), sidx));
        const float ays = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tp.ay), sidx));
        if (tls != cur) {
            // the window moved: a sample's neighbours along the ray mostly sit one texel to the side, and the two taps the windows share
            // keep their sums in registers (all branches are wave-uniform)
            const int delta = tls - cur;
            if (cur < 0) {
            } else if (delta == 1) {
                flush_tap(0);
                flush_tap(2);
                move_tap(0, 1);
                move_tap(2, 3);
            } else if (delta == -1) {
                flush_tap(1);
                flush_tap(3);
                move_tap(1, 0);
                move_tap(3, 2);
            } else if (delta == p.W) {
                flush_tap(0);
                flush_tap(1);
                move_tap(0, 2);
                move_tap(1, 3);
            } else if (delta == -p.W) {
                flush_tap(2);
                flush_tap(3);
                move_tap(2, 0);
                move_tap(3, 1);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) flush_tap(t);
            }
            cur = tls;
        }
        const float va = ga[sidx >> 2][sidx & 3], vb = gb[sidx >> 2][sidx & 3];
        const float w[4] = {(1.0f - axs) * (1.0f - ays), axs * (1.0f - ays), (1.0f - axs) * ays, axs * ays};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a0[t] = fmaf(va, w[t], a0[t]);
            a1[t] = fmaf(vb, w[t], a1[t]);
        }
    }
    if (cur >= 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) flush_tap(t);
    }
}

// Second half: dL/df[texel][c] += sum_n W0[123 + c][n] dL/dT[texel][n].  One thread per (texel, channel); W0's 128 KiB stay in cache.
__global__ __launch_bounds__(256) void texel_grad_to_features_kernel(const float* __restrict__ texel_grad, const float* __restrict__ w0_feat,
                                                                     long n_texels, float* __restrict__ d_features) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_texels * 256) return;
    const long t = idx >> 8;
    const int c = (int)(idx & 255);
    const f32x4* g = reinterpret_cast<const f32x4*>(texel_grad + 128 * t);
    const f32x4* wrow = reinterpret_cast<const f32x4*>(w0_feat + (long)c * kHidden);
    float s = 0.0f;
#pragma unroll 8
    for (int q = 0; q < 32; ++q) {
        const f32x4 gv = g[q], wv = wrow[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) s = fmaf(gv[e], wv[e], s);
    }
    d_features[idx] = d_features[idx] + s;
}

hipError_t launch_texel_scatter(const FieldParams& p, const float* g0_tl, float* texel_grad, hipStream_t st) {
    hipLaunchKernelGGL(texel_scatter_kernel, dim3((unsigned)((p.n_tiles * p.V + 3) / 4)), dim3(256), 0, st, p, g0_tl, texel_grad);
    return hipGetLastError();
}

hipError_t launch_texel_grad_to_features(const float* texel_grad, const float* w0_feat, long n_texels, float* d_features, hipStream_t st) {
    hipLaunchKernelGGL(texel_grad_to_features_kernel, dim3((unsigned)((n_texels * 256 + 255) / 256)), dim3(256), 0, st, texel_grad, w0_feat,
                       n_texels, d_features);
    return hipGetLastError();
}

hipError_t launch_field_dz(const FieldParams& p, const float* g0_tl, const float* w0t_streams, float* d_z, float* d_o,
                           float* d_d, float* d_features, hipStream_t st) {
    const unsigned wgs = (unsigned)((p.n_tiles * p.V + 3) / 4);
    if (p.texel_table && !d_features && d_z && !d_o && !d_d && p.net) {   // training: only dL/dz, rows 0..63 + rgb of W0 . g0
        hipLaunchKernelGGL(field_dz_table_kernel, dim3(wgs), dim3(256), 0, st, p, g0_tl, w0t_streams, p.net + kKerasW0 + 120 * kHidden, d_z);
        return hipGetLastError();
    }
    if (p.texel_table && !d_features) {                   // feature rows through the forward's texel table, no LDS
        hipLaunchKernelGGL(field_dz_kernel<true>, dim3(wgs), dim3(256), 0, st, p, g0_tl, w0t_streams, d_z, d_o, d_d, d_features);
        return hipGetLastError();
    }
    const size_t lds_bytes = (size_t)4 * (32 * kDfeRow + 64) * sizeof(float);
    static std::atomic<bool> attr_done[16];      // first call per device sets the dynamic-LDS limit (idempotent)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16 && !attr_done[dev].load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&field_dz_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(field_dz_kernel<false>, dim3(wgs), dim3(256), lds_bytes, st, p, g0_tl, w0t_streams, d_z, d_o, d_d, d_features);
    return hipGetLastError();
}

// ---- optimize(): clip-by-value then Adam (nerf_utils.py:8-12; tf.keras Adam, epsilon 1e-7) ----
__global__ void adam_clip_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m,
                                 float* __restrict__ v, long n, float lr_t, float beta1, float beta2, float eps, float clip,
                                 const unsigned char* __restrict__ update_mask) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (update_mask && !update_mask[i]) return;
    float g = grad[i];
    if (clip > 0.0f) g = fminf(fmaxf(g, -clip), clip);
    const float mi = beta1 * m[i] + (1.0f - beta1) * g;
    const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
    m[i] = mi;
    v[i] = vi;
    param[i] = param[i] - lr_t * mi / (sqrtf(vi) + eps);
}

hipError_t launch_adam_clip(float* param, const float* grad, float* m, float* v, long n, float lr_t, float beta1, float beta2,
                            float eps, float clip, const unsigned char* update_mask, hipStream_t st) {
    hipLaunchKernelGGL(adam_clip_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, param, grad, m, v, n, lr_t, beta1,
                       beta2, eps, clip, update_mask);
    return hipGetLastError();
}

}  // namespace mvnerf
