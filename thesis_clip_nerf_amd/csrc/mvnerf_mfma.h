// MFMA building blocks shared by the fused field kernel (field_eval.hip) and the training kernels
// (train_ops.hip): the fp32 32x32x2 MFMA wrapper, the prefetching weight stream and the 32-sample tile
// layout ("TL") used for activations that have to live in HBM.
#pragma once

#include <hip/hip_runtime.h>

#include "mvnerf_pack.h"

#ifndef MV_ABL_WLOAD
#define MV_ABL_WLOAD 0
#endif
#ifndef MV_PIN_LOADS
#define MV_PIN_LOADS 1
#endif
#ifndef MV_ASM_RELU
#define MV_ASM_RELU 0
#endif
#ifndef MV_INT_RELU
#define MV_INT_RELU 1
#endif

namespace mvnerf {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// relu of the 4 B operands of a step as four single v_max_f32 (fmaxf lowers to a canonicalising
// v_max pair + s_nop in front of every MFMA).  The trailing s_nop 1 covers the VALU-write ->
// MFMA-read wait states for the compiler-scheduled MFMAs that consume the outputs.
__device__ __forceinline__ void relu4(const float (&in)[4], float (&b)[4]) {
#if MV_ASM_RELU
    asm("v_max_f32_e32 %0, 0, %4\n\tv_max_f32_e32 %1, 0, %5\n\tv_max_f32_e32 %2, 0, %6\n\tv_max_f32_e32 %3, 0, %7\n\ts_nop 1"
        : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
        : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]));
#elif MV_INT_RELU
    // relu on the bit pattern: a signed-integer max with 0 zeroes exactly the floats with the sign bit set (-0.0 and
    // negative NaNs included) and is ONE v_max_i32 - fmaxf costs a canonicalising v_max_f32 pair
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int bits = __builtin_bit_cast(int, in[e]);
        b[e] = __builtin_bit_cast(float, bits > 0 ? bits : 0);
    }
#else
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = fmaxf(in[e], 0.0f);
#endif
}

// The weight stream: `cur` holds the 4 chunks (4 KiB per wave) of the step being consumed.  Every step first
// requests a later step (MV_WS_AHEAD steps on), then runs its 16 MFMAs (1024 cycles of matrix pipe), so an L2
// round trip is covered with margin, also across layer boundaries (the chunks of all layers are contiguous).
#ifndef MV_WS_AHEAD
#define MV_WS_AHEAD 1      // steps (of 4 KiB per wave) the weight stream is requested ahead of use: 1 or 2 (2: measured
                           // -0.6 %: the 16 extra registers push the 255-register kernels into scratch)
#endif

struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;   // buffer descriptor of the packed net (SGPRs)
    int voff;                      // lane * 16
    int pos;                       // wave-uniform byte offset of the step to REQUEST next (SGPR)
    f32x4 cur[4];
#if MV_WS_AHEAD == 2
    f32x4 nxt[4];                  // the step after `cur`, already in flight / landed
#endif
};

// buffer_load_dwordx4 v, voff, rsrc, pos offen offset:imm -- the uniform stream position rides in the
// scalar offset and the chunk index in the immediate, so a step costs no VALU address arithmetic.
template <int kImm>
__device__ __forceinline__ f32x4 ws_load(const WStream& ws, int pos) {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff + kImm, pos, 0);
    return __builtin_bit_cast(f32x4, r);
}

__device__ __forceinline__ void ws_begin(WStream& ws, const float* base, int bytes, int lane) {
    // loads past `bytes` (the prefetch of the steps after the last one) return 0 by the buffer range check
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
    ws.voff = lane * 16;
    ws.cur[0] = ws_load<0>(ws, 0);
    ws.cur[1] = ws_load<1024>(ws, 0);
    ws.cur[2] = ws_load<2048>(ws, 0);
    ws.cur[3] = ws_load<3072>(ws, 0);
#if MV_WS_AHEAD == 2
    ws.nxt[0] = ws_load<0>(ws, 4096);
    ws.nxt[1] = ws_load<1024>(ws, 4096);
    ws.nxt[2] = ws_load<2048>(ws, 4096);
    ws.nxt[3] = ws_load<3072>(ws, 4096);
    ws.pos = 8192;
#else
    ws.pos = 4096;
#endif
}

// The stream jumps: the step consumed kAfter steps from now (0 = the next mfma_step) shall come from byte offset `to`.
// With a request distance of MV_WS_AHEAD the redirect has to be announced MV_WS_AHEAD - 1 steps earlier than with 1.
__device__ __forceinline__ constexpr int ws_jump_lead() { return MV_WS_AHEAD - 1; }

// rotate the stream after a step whose next-request is n0..n3
__device__ __forceinline__ void ws_advance(WStream& ws, f32x4 n0, f32x4 n1, f32x4 n2, f32x4 n3) {
#if MV_WS_AHEAD == 2
#pragma unroll
    for (int q = 0; q < 4; ++q) ws.cur[q] = ws.nxt[q];
    ws.nxt[0] = n0;
    ws.nxt[1] = n1;
    ws.nxt[2] = n2;
    ws.nxt[3] = n3;
#else
    ws.cur[0] = n0;
    ws.cur[1] = n1;
    ws.cur[2] = n2;
    ws.cur[3] = n3;
#endif
    ws.pos += 4096;
}

// One step = 4 k-steps x 4 output blocks: acc[nb] += A(cur[nb])[e] x b[e]
__device__ __forceinline__ void mfma_step(WStream& ws, const float (&b)[4], f32x16 (&acc)[4]) {
#if MV_ABL_WLOAD
    const f32x4 n0 = ws.cur[1], n1 = ws.cur[2], n2 = ws.cur[3], n3 = ws.cur[0];
#else
    const f32x4 n0 = ws_load<0>(ws, ws.pos), n1 = ws_load<1024>(ws, ws.pos), n2 = ws_load<2048>(ws, ws.pos),
                n3 = ws_load<3072>(ws, ws.pos);
#endif
#if MV_PIN_LOADS
    __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of its use
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        acc[0] = mfma(ws.cur[0][e], b[e], acc[0]);
        acc[1] = mfma(ws.cur[1][e], b[e], acc[1]);
        acc[2] = mfma(ws.cur[2][e], b[e], acc[2]);
        acc[3] = mfma(ws.cur[3][e], b[e], acc[3]);
    }
    ws_advance(ws, n0, n1, n2, n3);
}


// ---- tile layout (TL): a (rows, F) activation matrix stored as [tile = row/32][feature][row%32], so that
// "lane = sample" accesses are 128-B contiguous per feature and "lane = feature" accesses read 4 samples as
// one float4.  Accumulator register r of block nb on lane (j, h) is feature 32*nb + acc_row(r, h) of sample j.
__device__ __forceinline__ long tl_index(long tile, int n_feat, int feat, int j) {
    return (tile * n_feat + feat) * 32 + j;
}

// Buffer stores: one address VGPR ((4h*32 + j) * 4 bytes), the tile and the 32-feature block in the scalar offset,
// the row inside the block in the 12-bit immediate - 64 stores without 64 address registers.  Offsets are 32-bit:
// callers keep tile * 16 KiB below 4 GiB (checked in api.hip for the stash entry points).
// The stash is written once and read once, a whole forward pass later: its stores carry the non-temporal hint (cache-policy bit 1), so the
// 3.5 GB a fine launch writes do not push the texel table and the weight stream out of L2 (A/B, kernel-trace means over the fine and the
// coarse launch of a training step: 1034 us plain, 944 us nt, 947 us sc0 + nt, 997 us sc1).
#ifndef MV_STASH_AUX
#define MV_STASH_AUX 2
#endif
__device__ __forceinline__ void store_tl(float* __restrict__ base, long tile, int j, int h, const f32x16 (&x)[4]) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xFFFFFFFF, 0x00020000);
    const int voff = (4 * h * 32 + j) * 4;
    const int tile_off = (int)((unsigned)tile * 16384u);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float val = x[nb][r];          // (bit_cast straight from the vector element stores element 0)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, val), rsrc,
                                                  voff + ((r & 3) + 8 * (r >> 2)) * 128, tile_off + nb * 4096, MV_STASH_AUX);
        }
}

}  // namespace mvnerf
