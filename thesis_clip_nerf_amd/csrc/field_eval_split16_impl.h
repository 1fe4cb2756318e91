// fp32-grade field kernel on the bf16 matrix pipe, 16x16x32 mapping ("split16"): same math, inputs, outputs and operand
// cut as field_eval_split.hip (x = x1 + x2 + x3 exactly, six bf16 MFMAs per product block, fp32 accumulation), issued as
// v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16.
//
// Why (round 3): the split kernel is not limited by issue stalls but by the clock the chip holds under
// bf16-MFMA load - inserting idle cycles into every k-step cost a third of their length, removing instructions returned
// nothing.  On trunk-like operands (scripts/x6_probe3.hip) the same stream of products runs 12 % faster through the
// 16x16x32 shape: it accumulates 32-deep dot products into 256 accumulators per instruction instead of 16-deep ones into
// 1024, and the chip holds 2.04 instead of 1.82 GHz under it.
//
// Mapping.  Y^T = W^T X^T as before.  One MFMA: A = W^T block (16 output features x 32 inputs: lane (i, g) = (l & 15, l >> 4)
// holds inputs 8g .. 8g + 7 of output feature i), B = activations (32 inputs x 16 samples: lane (n, g) holds inputs 8g .. 8g + 7
// of sample n), D = 16 features x 16 samples, lane (n, g) holding features 4g .. 4g + 3 of sample n.  A wave owns a tile of 32
// samples = 2 column blocks (samples n and 16 + n per lane), 8 row blocks of 16 features: x and hid are 2 x 64 registers as
// before, and lane (n, g) holds features 16 rb + 4g + {0..3} of its two samples.  The B operand of a hidden k-step t (K = 32) is
// therefore registers {x[2t][cb][0..3], x[2t + 1][cb][0..3]} - inputs 32t + 4g + jj (jj < 4) and 32t + 16 + 4g + jj - 4 -
// and the weight stream is packed with that K order: activations still never leave the register file.
// One k-step = 8 row blocks x 2 column blocks x 6 products = 96 MFMAs of 16 cycles (= two of the old k-steps); its weights are
// one 24 KiB ring slot (8 row blocks x 3 pieces x 1 KiB); 3 slots; half as many barriers per tile.
//
// Inference and, since the end of round 3, the training forward (kStash: the 13 pre-activation tensors go to the stash in tile layout).
//
// This file is the body of TWO translation units (field_eval_split16.hip: MVS16_F16 = 0, field_eval_split16h.hip: MVS16_F16 = 1):
//   MVS16_F16 = 0  operands cut EXACTLY into three bf16 pieces, six v_mfma_f32_16x16x32_bf16 per product block (as described above);
//   MVS16_F16 = 1  operands as TWO fp16 pieces, three v_mfma_f32_16x16x32_f16 per product block.  With tw = 64 w and tv = v / 64 (powers
//                  of two: exact, tw tv = w v):  A0 = rn16(tw), A1 = rn16(tw - A0), A0s = A0 / 64;  B0 = rn16(tv), B1 = rn16(64 (tv - B0))
//                  = rn16(v - 64 B0);  acc += A0s B1 + A1 B0 + A0 B0.  Both remainders are exact in fp32 before they are rounded, so each
//                  operand is represented to |error| <= 2^-23 |x| + a subnormal floor (2^-31 for a weight, 2^-25 for an activation:
//                  tests/test_f16_split_math.py); the one dropped cross term (tw - A0)(tv - B0) is <= 2^-22 |w v| (2^-25 typical).  The
//                  activation's remainder is scaled by 64 (and meets A0 / 64): its pieces keep all 11 bits for |v| >= 1/4, the weights'
//                  unscaled remainder for |w| >= 2^-8; below that they go subnormal - the MFMA honours fp16 subnormals
//                  (scripts/f16_mfma_probe.hip) - with absolute errors <= 2^-25 |w| resp. 2^-31 |v| per product.  Measured per ResNet block against float64: at or below the fp32 MFMA kernel's error
//                  (tests/test_gpu_split.py).  Same stream size (three 1 KiB weight pieces per row block), two instead of three operand
//                  pieces in registers, half the MFMAs.  Range: |w| < 1023, |v| < 4.19e6 (fp16 overflow beyond; the bf16 form has the full
//                  fp32 range).
#include <hip/hip_runtime.h>

#include <mutex>

#include "mvnerf_kernels.h"
#include "mvnerf_math.h"
#include "mvnerf_mfma.h"

#ifndef MVS16_F16
#error "include through field_eval_split16.hip / field_eval_split16h.hip"
#endif
#ifndef MVS16_MIXLO
#define MVS16_MIXLO 0      // fp16 form: 1: the cut of a value pair with v_fma_mixlo / mixhi_f16, six vector instructions instead of eight
#endif                     //    (bit-identical: scripts/mixlo_probe.hip).  Measured +0.5 % (profiles/r03_ab_f16x3_ablations.log) - and the pieces
                           //    are then written by inline asm, whose hazards in front of an MFMA the compiler does not track: left off
#if MVS16_F16
#define MVS16_BYTES packed_net_split16h_bytes
#define MVS16_PACK launch_pack_net_split16h
#define MVS16_SUPPORTS field_eval_split16h_supports
#define MVS16_LAUNCH_FN launch_field_eval_split16h
#define MVS16_KERNEL field_eval_split16h_kernel
#define MVS16_PACK_KERNEL pack_net_split16h_kernel
#else
#define MVS16_BYTES packed_net_split16_bytes
#define MVS16_PACK launch_pack_net_split16
#define MVS16_SUPPORTS field_eval_split16_supports
#define MVS16_LAUNCH_FN launch_field_eval_split16
#define MVS16_KERNEL field_eval_split16_kernel
#define MVS16_PACK_KERNEL pack_net_split16_kernel
#endif

namespace mvnerf {

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// ---- weight stream: 1 KiB chunks [lane][8 bf16]; chunk = 24 * kstep + 3 * rb + piece ------------------------------------
//   k-steps 0..1  : layer 0, PE(cam xyz) + rgb: lane group g, slot e = 8 t + jj (16 slots per group)
//                     g < 3 : W0 row 20 g + e            (dimension g, octaves 0..7, sin | cos)
//                     g = 3 : e < 12: dimension e >> 2, octave 8 + ((e >> 1) & 1), sin | cos;  e = 12..14: rgb rows 120..122;  15: zero
//   k-steps 2..9  : layer 0, the 256 feature rows: channel 32 (ks - 2) + 8 g + jj                     (direct gather only)
//   k-steps 10 + 4 l + t : hidden layer l: input 32 t + 4 g + jj (jj < 4) | 32 t + 16 + 4 g + jj - 4
constexpr int kS16SlotChunks = 24;
constexpr int kS16L0Pe = 2, kS16L0Feat = 8, kS16Hidden = 48;
constexpr int kS16Steps = kS16L0Pe + kS16L0Feat + kS16Hidden;               // 58
constexpr int kS16Chunks = kS16Steps * kS16SlotChunks;                      // 1392 KiB
constexpr int kChunkElems16 = 512;

__host__ __device__ constexpr int s16_pe_row(int g, int e) {
    return g < 3 ? 20 * g + e : (e < 12 ? 20 * (e >> 2) + 2 * (8 + ((e >> 1) & 1)) + (e & 1) : (e < 15 ? 120 + (e - 12) : -1));
}

__global__ void MVS16_PACK_KERNEL(const float* __restrict__ src, unsigned short* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kS16Chunks * kChunkElems16) return;
    const int chunk = idx / kChunkElems16, lane = (idx % kChunkElems16) / 8, jj = idx % 8;
    const int i = lane & 15, g = lane >> 4;
    const int ks = chunk / kS16SlotChunks, rb = (chunk % kS16SlotChunks) / 3, piece = chunk % 3;
    float val = 0.0f;
    if (ks < kS16L0Pe) {
        const int row = s16_pe_row(g, 8 * ks + jj);
        if (row >= 0) val = src[kKerasW0 + row * kHidden + 16 * rb + i];
    } else if (ks < kS16L0Pe + kS16L0Feat) {
        const int row = 123 + 32 * (ks - kS16L0Pe) + 8 * g + jj;
        val = src[kKerasW0 + row * kHidden + 16 * rb + i];
    } else {
        const int q = ks - kS16L0Pe - kS16L0Feat, layer = q / 4, t = q % 4;
        const int f = 32 * t + (jj < 4 ? 4 * g + jj : 16 + 4 * g + (jj - 4));
        const int wsrc = kKerasBlocks + (layer / 2) * kKerasBlockStride + (layer % 2) * (kHidden * kHidden + kHidden);
        val = src[wsrc + f * kHidden + 16 * rb + i];
    }
#if MVS16_F16
    // piece 0: A0 = rn16(64 w); piece 1: A0s = A0 / 64; piece 2: A1 = rn16(64 w - A0) (the remainder is exact in fp32)
    const float tw = val * 64.0f;
    const _Float16 a0 = (_Float16)tw;
    const _Float16 a1 = (_Float16)(tw - (float)a0);
    const _Float16 a0s = (_Float16)((float)a0 * 0.015625f);
    const _Float16 pc = piece == 0 ? a0 : (piece == 1 ? a0s : a1);
    dst[idx] = __builtin_bit_cast(unsigned short, pc);
#else
    // round-to-nearest pieces; every remainder is exact in fp32 (as pack_net_split_kernel)
    const __bf16 p1 = (__bf16)val;
    const float r1 = val - (float)p1;
    const __bf16 p2 = (__bf16)r1;
    const float r2 = r1 - (float)p2;
    const __bf16 p3 = (__bf16)r2;
    const __bf16 pc = piece == 0 ? p1 : (piece == 1 ? p2 : p3);
    dst[idx] = __builtin_bit_cast(unsigned short, pc);
#endif
}

__device__ __forceinline__ f32x4 mfma1632(u32x4 a, u32x4 b, f32x4 c) {
#if MVS16_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

struct B16 {
    u32x4 p1, p2, p3;
};

// values 2q, 2q+1 of an 8-value B operand -> dword q of the three pieces (relu first where the layer has one); truncation cut,
// every remainder exact (field_eval_split.hip)
template <bool kRelu>
__device__ __forceinline__ void cut_pair(float v0, float v1, int q, B16& b) {
    if (kRelu) {               // relu on the bit pattern: one v_max_i32
        const int i0 = __builtin_bit_cast(int, v0), i1 = __builtin_bit_cast(int, v1);
        v0 = __builtin_bit_cast(float, i0 > 0 ? i0 : 0);
        v1 = __builtin_bit_cast(float, i1 > 0 ? i1 : 0);
    }
#if MVS16_F16
    // p1 = B0 = rn16(v / 64), p3 = B1 = rn16(64 (v / 64 - B0)) = rn16(v - 64 B0): the remainder is exact in fp32, formed by one
    // mixed-precision fma per value (v_fma_mix_f32 reads B0's halves as they lie); p2 is not used
#if MVS16_MIXLO
    // four instructions per pair: v_fma_mixlo / mixhi_f16 form the product v / 64 (or the remainder v - 64 B0) in fp32 and round it ONCE
    // to the fp16 half they write (scripts/mixlo_probe.hip: bit-identical to multiply + convert / fma + convert)
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(v0), "s"(0.015625f));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(v1), "s"(0.015625f));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "s"(-64.0f), "v"(v0));
    // (the s_nop: the compiler's hazard recognizer does not see an inline-asm VALU write in front of an MFMA that reads the register - without
    // the wait states the first MFMA behind a cut reads the old B1)
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1" : "+v"(l) : "v"(h), "s"(-64.0f), "v"(v1));
    b.p1[q] = h;
    b.p3[q] = l;
#else
    const f32x2 t = {v0 * 0.015625f, v1 * 0.015625f};     // (forced into one v_pk_mul_f32 by inline asm: 2 % slower - the scheduler no longer places it)
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, f16x2));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "s"(-64.0f), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "s"(-64.0f), "v"(v1));
    const f32x2 r = {r0, r1};
    b.p1[q] = h;
    b.p3[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
#endif
#else
    const float r0 = v0 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v0) & 0xffff0000u);
    const float r1 = v1 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v1) & 0xffff0000u);
    const float s0 = r0 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r0) & 0xffff0000u);
    const float s1 = r1 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
    b.p1[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, v1), __builtin_bit_cast(unsigned, v0), 0x07060302u);
    b.p2[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, r1), __builtin_bit_cast(unsigned, r0), 0x07060302u);
    b.p3[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
#endif
}

// keeps dword q of an operand's pieces where the program order has them (an empty asm the scheduler cannot move code across)
__device__ __forceinline__ void pin_pieces(B16& b, int q) {
#if MVS16_F16
    unsigned u1 = b.p1[q], u3 = b.p3[q];
    asm volatile("" : "+v"(u1), "+v"(u3));
    b.p1[q] = u1;
    b.p3[q] = u3;
#else
    unsigned u1 = b.p1[q], u2 = b.p2[q], u3 = b.p3[q];
    asm volatile("" : "+v"(u1), "+v"(u2), "+v"(u3));
    b.p1[q] = u1;
    b.p2[q] = u2;
    b.p3[q] = u3;
#endif
}

template <bool kRelu>
__device__ __forceinline__ void cut8(const float (&v)[8], B16& b) {
#pragma unroll
    for (int q = 0; q < 4; ++q) cut_pair<kRelu>(v[2 * q], v[2 * q + 1], q, b);
}

#ifndef MVS16_ORDER
#define MVS16_ORDER 0      // order of the six products of a block (A/B experiment, see kstep16)
#endif
#ifndef MVS16_DMA_LATE
#define MVS16_DMA_LATE 0   // LDS-DMA ring: 1: the barrier at the end of k-step p waits only for position p + 1 (s_waitcnt vmcnt(3): the three
#endif                     //    requests of position p + 2, issued in k-step p, stay in flight for one more k-step); 0: vmcnt(0) - every k-step
                           //    then ends behind its own weight request's L2 round trip.  Measured on the fp16 form (profiles/r03_ab_dma_late.log): 1 is 0.5 % SLOWER -
                           //    the request has landed by the end of its k-step, and the A operands of row block 0 are better read a k-step early
#ifndef MVS16_LDSDMA
#define MVS16_LDSDMA 1     // 1: the weight stream reaches LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write);
#endif                     // 0: through registers (three dwordx4 loads per thread and k-step, stored one k-step later)

// ---- the slot ring: one k-step (24 KiB) per slot, 3 slots -------------------------------------------------------------------------
constexpr int kR16Slots = 3, kR16SlotF4 = kS16SlotChunks * 64;               // float4 per slot
constexpr int kS16PerView = kS16Hidden / 2;                                   // 24 k-steps: 3 ResNet blocks
constexpr int kS16MaxPositions = 512;

struct Ring16 {
    const f32x4* w;
    f32x4* base;        // LDS
    int c;              // ring slot of the current k-step
    int p, P;
    const int* table;   // LDS: first chunk of every position of one tile
    int start_pf;       // first chunk of the position the next fetch loads (read one k-step ahead)
    int off;            // this thread's first 16 B inside a slot (tid * 16); + 8192, + 16384
    int tid, wave;
    f32x4 stg[3];       // register-staged ring only (kDma = false): the fetched bytes on their way to LDS
    u32x4 a0[3];        // A operands (3 pieces) of row block 0 of the CURRENT k-step, read during the previous one
};

__device__ __forceinline__ int ring16_start_chunk(int p, int V, int l0_units) {
    const int per_view = l0_units + kS16PerView;
    if (p < per_view * V) {
        const int q = p % per_view;
        return (q < kS16L0Pe ? q : (q < l0_units ? q : kS16L0Pe + kS16L0Feat + (q - l0_units))) * kS16SlotChunks;
    }
    return (kS16L0Pe + kS16L0Feat + kS16PerView + (p - per_view * V)) * kS16SlotChunks;
}

// LDS-DMA of one position (24 chunks of 1 KiB starting at `start_chunk`) into ring slot `slot`: three wave-instructions per wave, each
// moving 1 KiB (lane l: 16 bytes at wave base + 16 l); wave w of the 8 covers bytes [1024 w, 1024 w + 1024) of each 8 KiB third.
__device__ __forceinline__ void ring16_dma(const Ring16& r, int start_chunk, int slot) {
    const f32x4* src = r.w + (long)start_chunk * 64 + r.tid;
    f32x4* dst = r.base + slot * kR16SlotF4 + 64 * r.wave;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 512 * i),
                                         (__attribute__((address_space(3))) void*)(dst + 512 * i), 16, 0, 0);
}

__device__ __forceinline__ void ring16_store(Ring16& r) {
    int slot = r.c + 2;
    slot = slot >= kR16Slots ? slot - kR16Slots : slot;
    char* dst = reinterpret_cast<char*>(r.base) + slot * (kR16SlotF4 * 16) + r.off;
    *reinterpret_cast<f32x4*>(dst) = r.stg[0];
    *reinterpret_cast<f32x4*>(dst + 8192) = r.stg[1];
    *reinterpret_cast<f32x4*>(dst + 16384) = r.stg[2];
}

__device__ __forceinline__ void ring16_load(Ring16& r, int start_chunk) {
    const char* src = reinterpret_cast<const char*>(r.w) + (long)start_chunk * 1024 + r.off;
    r.stg[0] = *reinterpret_cast<const f32x4*>(src);
    r.stg[1] = *reinterpret_cast<const f32x4*>(src + 8192);
    r.stg[2] = *reinterpret_cast<const f32x4*>(src + 16384);
}

// The weight fetch of the k-step at position p (slot c), issued behind its first MFMA group.
// kDma (inference): position p + 2 goes straight into slot (c + 2) % 3 by LDS-DMA - the slot of position p - 1, which nobody reads any
//   more since the last barrier; the vmcnt(0) in front of the barrier at the k-step's end (ring16_next) lets it land before it is published;
//   seven of the k-step's eight MFMA groups lie between the request and that wait.  No staging registers, no ds_write.
// !kDma (MVS16_STASH_DMA = 0; the training forward's first form): through registers - store what the previous k-step loaded (position p + 2), load position p + 3 - because vmcnt
//   retires in order: behind the stash's buffer_stores a vmcnt(0) per k-step would wait for 16 KiB of HBM writes per wave, whereas the
//   staged loads are older than the stores that follow them.
template <bool kDma>
__device__ __forceinline__ void ring16_fetch(Ring16& r) {
    if (kDma) {
        int slot = r.c + 2;
        slot = slot >= kR16Slots ? slot - kR16Slots : slot;
        ring16_dma(r, r.start_pf, slot);
    } else {
        ring16_store(r);
        ring16_load(r, r.start_pf);
    }
    int pp = r.p + (kDma ? 3 : 4);                                      // table entry the NEXT k-step's fetch needs
    pp = pp >= r.P ? pp - r.P : pp;
    r.start_pf = r.table[pp];
}

__device__ __forceinline__ const f32x4* ring16_cur(const Ring16& r) { return r.base + r.c * kR16SlotF4; }
__device__ __forceinline__ const f32x4* ring16_nxt(const Ring16& r) { return r.base + (r.c + 1 == kR16Slots ? 0 : r.c + 1) * kR16SlotF4; }

template <bool kDma>
__device__ __forceinline__ void ring16_next(Ring16& r) {
    if (kDma && MVS16_DMA_LATE) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef MVS16_ABL_BARRIER
    else if (kDma) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // timing-only ablation (races): no workgroup barrier per k-step
#else
    else if (kDma) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    r.c = r.c + 1 == kR16Slots ? 0 : r.c + 1;
    r.p = r.p + 1 == r.P ? 0 : r.p + 1;
}

// position of feature 16 rb + 4 g (+ 0..3) inside a 128-float vector in the 32x32 accumulator order [h][nb][r] that the bias
// block of the packed net, the per-(view, ray) layer-0 seed and the texel table rows use (mvnerf_pack.h acc_slot): 4 contiguous floats
__device__ __forceinline__ int perm_f4(int rb, int g) {
    return ((g & 1) * 64 + (rb >> 1) * 16 + 4 * (2 * (rb & 1) + (g >> 1))) >> 2;          // in float4 units
}

template <bool kAdd>
__device__ __forceinline__ void apply_bias_row(f32x4 (&row)[2], const f32x4& bv) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (kAdd) {
                float r = row[cb][c];
                asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(bv[c]));      // scalar adds (no v_pk_add_f32 beside MFMAs)
                row[cb][c] = r;
            } else {
                row[cb][c] = bv[c];
            }
        }
}

// One k-step (K = 32): acc[rb][cb] += A(rb)^T b[cb] for 8 row blocks x 2 column blocks, 6 MFMAs each (smallest terms first),
// the two column blocks' chains alternating.  While the 12 MFMAs of row block rb run, the three A chunks of row block rb + 1 (of
// the NEXT k-step's row block 0 at rb = 7) are read from LDS and the vector ALU prepares what comes next:
//   kMode 1: one value pair of the next k-step's B operands (nv[cb][0..7], pair rb & 3 of column block rb >> 2) is cut: 13 vector
//            instructions per 12 MFMAs;
//   kMode 2: the LAST k-step of a hidden layer: the NEXT layer's first B operands are cut from this layer's own output - row blocks
//            0 and 1 of acc, final after groups 0 and 1 (relu, then the cut) - during groups 1..4, and the layer's input array, no
//            longer needed, takes the bias work of the layer boundary (tail_bias: in[rb][cb] += bias row, or = bias row), one row
//            block per group.  With it no vector work of a layer boundary is left outside the MFMA shadow.
//   kMode 4: the last k-step of a hidden layer in the plain flow: only the bias half of kMode 2 (tail_bias onto the consumed input array).
//   kMode 0: nothing.
template <bool kRelu, int kMode, bool kTailAdd, bool kDma>
__device__ __forceinline__ void kstep16(Ring16& ring, int lane, int g, const B16 (&b)[2], const float (&nv)[2][8], B16 (&bn)[2], f32x4 (&acc)[8][2],
                                        f32x4 (&in)[8][2], const float* __restrict__ tail_bias) {
    const f32x4* cur = ring16_cur(ring) + lane;
    const f32x4* nxt = ring16_nxt(ring) + lane;
    u32x4 a[3] = {ring.a0[0], ring.a0[1], ring.a0[2]};
    if (kDma && MVS16_DMA_LATE) {
        // position p + 1 is published only by the barrier that ends k-step p: row block 0's A chunks are read here, not one k-step early
#pragma unroll
        for (int q = 0; q < 3; ++q) a[q] = __builtin_bit_cast(u32x4, cur[q * 64]);
    }
    f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
    // the lane's part of perm_f4, formed HERE (behind an empty asm): left to itself the compiler hoists the eight row addresses of every
    // layer's bias vector out of the tile loop and spills them
    int glane = (g & 1) * 16 + (g >> 1);
    if (kMode == 2 || kMode == 4) asm volatile("" : "+v"(glane));
    const f32x4* tail_rows = reinterpret_cast<const f32x4*>(tail_bias) + glane;
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
        u32x4 an[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#ifdef MVS16_ABL_A0S
            if (q == 1) continue;                                  // timing-only ablation (wrong results): the A0 / 64 piece is not read from LDS
#endif
            if (rb < 7 || !(kDma && MVS16_DMA_LATE)) an[q] = __builtin_bit_cast(u32x4, rb < 7 ? cur[((rb + 1) * 3 + q) * 64] : nxt[q * 64]);
            else an[q] = a[q];
        }
#ifdef MVS16_ABL_A0S
        an[1] = an[0];
#endif
#ifdef MVS16_ABL_CUT
        if (kMode == 1 && rb == 0) { bn[0] = b[0]; bn[1] = b[1]; }            // timing-only ablation (wrong results): no operand cut in the k-steps
#else
        if (kMode == 1) cut_pair<kRelu>(nv[rb >> 2][2 * (rb & 3)], nv[rb >> 2][2 * (rb & 3) + 1], rb & 3, bn[rb >> 2]);
#endif
        if (kMode == 2) {
            if (rb >= 1 && rb <= 4) {              // pair q = rb - 1 of both column blocks: q < 2 from acc[0], q >= 2 from acc[1]
                const int q = rb - 1;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    cut_pair<true>(acc[q >> 1][cb][2 * (q & 1)], acc[q >> 1][cb][2 * (q & 1) + 1], q, bn[cb]);
                    // pinned inside its group: this cut reads MFMA results, and left to itself the scheduler fills the group's vector
                    // slots with the bias adds and sinks the whole cut behind the k-step's last MFMA
                    pin_pieces(bn[cb], q);
                }
            }
        }
        if ((kMode == 2 || kMode == 4) && tail_bias) {
            // the bias row of row block rb is requested here and applied one group later (an LDS round trip inside a group would
            // hold this wave's MFMAs behind the wait); row 7 is applied behind the last group
            const f32x4 bv_new = tail_rows[(rb >> 1) * 4 + 2 * (rb & 1)];              // = [perm_f4(rb, g)]: the row block is an immediate offset
            if (rb > 0) apply_bias_row<kTailAdd>(in[rb - 1], bv);
            bv = bv_new;
        }
#if MVS16_F16
        // three products per block, the two small ones first: A0s B1, A1 B0, A0 B0 (a = {A0, A0s, A1}, b = {p1: B0, p3: B1})
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[1], b[cb].p3, acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[2], b[cb].p1, acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[0], b[cb].p1, acc[rb][cb]);
#elif MVS16_ORDER == 1
        // the A operand changes as rarely as possible: a2 (x2), a1 (x4), a0 (x6); per chain a2p1, a1p2, a1p1, a0p3, a0p2, a0p1
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[2], b[cb].p1, acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[1], b[cb].p2, acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[1], b[cb].p1, acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[0], b[cb].p3, acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[0], b[cb].p2, acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[0], b[cb].p1, acc[rb][cb]);
#else
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[2], b[cb].p1, acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[1], b[cb].p2, acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[0], b[cb].p3, acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[1], b[cb].p1, acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[0], b[cb].p2, acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma1632(a[0], b[cb].p1, acc[rb][cb]);
#endif
        // issue order inside the group: the first MFMA (its operands were requested one group ago), the LDS reads of the next group
        // (and the bias row), then vector instructions / MFMA alternating
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (kMode == 2 || kMode == 4) ? 4 : 3, 0);
#pragma unroll
        for (int m = 0; m < (MVS16_F16 ? 5 : 11); ++m) {
            if (kMode == 1) __builtin_amdgcn_sched_group_barrier(0x002, MVS16_F16 ? 2 : 1, 0);
            if (kMode == 2) __builtin_amdgcn_sched_group_barrier(0x002, MVS16_F16 ? 6 : 3, 0);
            if (kMode == 4) __builtin_amdgcn_sched_group_barrier(0x002, MVS16_F16 ? 2 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if (kMode == 1) __builtin_amdgcn_sched_group_barrier(0x002, MVS16_F16 ? 3 : 2, 0);
        if (kMode == 2) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (rb == 0) {   // the weight loads of two k-steps ahead
            ring16_fetch<kDma>(ring);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) a[q] = an[q];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) ring.a0[q] = a[q];
    if ((kMode == 2 || kMode == 4) && tail_bias) apply_bias_row<kTailAdd>(in[7], bv);
    if (kMode == 1 || kMode == 2) {
        // pin the pieces of the next operand HERE (otherwise the machine sinker moves the cut behind the barrier)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pin_pieces(bn[cb], q);
            }
    }
}

// the cut of relu(in[0..1]): the first B operands of a hidden layer (k-step 0 reads in[0][cb] (jj < 4) and in[1][cb] (jj >= 4))
__device__ __forceinline__ void first_operand_s16(const f32x4 (&in)[8][2], B16 (&b)[2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const float v8[8] = {in[0][cb][0], in[0][cb][1], in[0][cb][2], in[0][cb][3], in[1][cb][0], in[1][cb][1], in[1][cb][2], in[1][cb][3]};
        cut8<true>(v8, b[cb]);
    }
}

// acc += W^T relu(in) for one hidden layer: 4 k-steps; k-step t reads in[2t][cb] (jj < 4) and in[2t + 1][cb] (jj >= 4).
// b: on entry the cut of relu(in[0..1]) (first_operand_s16, or the previous layer's exit value); on exit the cut of relu(acc[0..1]),
// i.e. the next layer's entry value.  tail_bias (or nullptr): in the last k-step in[rb][cb] += / = that bias vector (32x32 accumulator
// order) - `x += b2` while the first Dense of a block writes hid, `hid = b1 of the next block` while the second one writes x.
#ifndef MVS16_TAILBIAS
#define MVS16_TAILBIAS 0   // 1: plain flow with the bias rows of the next layer's accumulator applied in the last k-step of the current layer (kMode 4).
                           //    Measured on the fp16 form (profiles/r03_ab_tailbias.log): +0.3 %, i.e. nothing - the layer boundaries' vector work is not what it waits for
#endif
#ifndef MVS16_TAIL
#define MVS16_TAIL 0       // 1: layer boundaries prepared in the previous layer's last k-step (kMode 2); 0: their vector work (first operand's
#endif                     //    cut, bias rows) stays between the layers.  Measured (profiles/r03_ab_tail*.log): 1 is 3-5 % SLOWER - see DESIGN.md 4.0

// the plain form: acc += W^T relu(in), first operand cut at the layer's head.  tail_bias (or nullptr): during the last k-step, when every
// row of `in` has been cut, in[rb][cb] += / = that bias vector (32x32 accumulator order) - `x += b2` while the first Dense of a block writes
// hid, `hid = b1 of the next block` while the second one writes x: the bias rows of the NEXT layer's accumulator cost no exposed time.
template <bool kDma, bool kTailAdd>
__device__ __forceinline__ void dense128_s16_plain(Ring16& ring, int lane, int g, f32x4 (&in)[8][2], f32x4 (&acc)[8][2], const float* __restrict__ tail_bias) {
    B16 b[2], bn[2];
    first_operand_s16(in, b);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float nv[2][8];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 8; ++q) nv[cb][q] = t < 3 ? in[2 * (t + 1) + (q >> 2)][cb][q & 3] : 0.0f;
        if (t < 3) kstep16<true, 1, false, kDma>(ring, lane, g, b, nv, bn, acc, in, nullptr);
        else kstep16<true, 4, kTailAdd, kDma>(ring, lane, g, b, nv, bn, acc, in, tail_bias);
        b[0] = bn[0];
        b[1] = bn[1];
        ring16_next<kDma>(ring);
    }
}
template <bool kAdd>
__device__ __forceinline__ void bias16(const float* __restrict__ bperm, int g, f32x4 (&acc)[8][2]);

template <bool kTailAdd>
__device__ __forceinline__ void dense128_s16(Ring16& ring, int lane, int g, f32x4 (&in)[8][2], f32x4 (&acc)[8][2], B16 (&b)[2],
                                             const float* __restrict__ tail_bias) {
    B16 bn[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float nv[2][8];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 8; ++q) nv[cb][q] = t < 3 ? in[2 * (t + 1) + (q >> 2)][cb][q & 3] : 0.0f;
        if (t < 3) kstep16<true, 1, false, true>(ring, lane, g, b, nv, bn, acc, in, nullptr);
        else kstep16<true, 2, kTailAdd, true>(ring, lane, g, b, nv, bn, acc, in, tail_bias);
        b[0] = bn[0];
        b[1] = bn[1];
        ring16_next<true>(ring);
    }
}

template <bool kAdd>
__device__ __forceinline__ void bias16(const float* __restrict__ bperm, int g, f32x4 (&acc)[8][2]) {
    const f32x4* p = reinterpret_cast<const f32x4*>(bperm);
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
        const f32x4 v = p[perm_f4(rb, g)];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (kAdd) {
                    float r = acc[rb][cb][c];
                    asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(v[c]));      // scalar adds (no v_pk_add_f32 beside MFMAs)
                    acc[rb][cb][c] = r;
                } else {
                    acc[rb][cb][c] = v[c];
                }
            }
    }
}

constexpr int kS16StageRowBytes = 256;      // per staged sample row: 64 fp32 channels

// Training forward: one activation tensor of a tile into the stash, tile layout [tile][feature][32 samples] (mvnerf_mfma.h: the layout
// the backward kernels read).  Lane (n, g) holds features 16 rb + 4g + i of samples 16 cb + n: per (rb, i, cb) a wave-instruction writes
// four 64-byte runs (one feature row half each); one address VGPR, row block in the scalar offset, (i, cb) in the immediate; non-temporal
// as store_tl (the stash is written once and read a whole pass later).
__device__ __forceinline__ void store_tl16(float* __restrict__ base, long tile, int n, int g, const f32x4 (&x)[8][2]) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xFFFFFFFF, 0x00020000);
    const int voff = (4 * g * 32 + n) * 4;
    const int tile_off = (int)((unsigned)tile * 16384u);
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const float val = x[rb][cb][i];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, val), rsrc, voff + (i * 32 + 16 * cb) * 4,
                                                      tile_off + rb * 2048, MV_STASH_AUX);
            }
}

struct SampleGeo {
    long g;           // global sample index (clamped)
    int ray, sidx, b;
    float wx, wy, wz;
    bool valid;
};

// kProj: layer 0's feature rows come from the fp32 texel table (project_texels_kernel, field_eval.hip).
// kAux: the optional outputs (tap indices, pixel coordinates, embedding, the 8 complete_output activations) are compiled in; the plain
// render variant carries none of the per-sample row indices they need through the tile (fewer spilled registers).
// kStash (training forward): the trunk's 13 pre-activation tensors also go to HBM (p.stash / p.stash_fused, mvnerf_kernels.h), exactly
// the slots field_eval_split_kernel<.., kStash> writes.
template <bool kMultiView, bool kProj, bool kAux, bool kStash>
__global__ __launch_bounds__(512, 2) void MVS16_KERNEL(FieldParams p, const f32x4* __restrict__ wsplit) {
    constexpr int kW = 8;
#ifndef MVS16_STASH_DMA
#define MVS16_STASH_DMA 1  // 1: the training forward's weight ring by LDS-DMA as well.  Its vmcnt(0) per k-step then also waits for the stash stores of
#endif                     //    the layer boundary before it - on the fp16 form that costs less than the staging registers and LDS stores of the
                           //    register-staged ring (train step 5.77 -> 5.74 ms, scripts/ab_train_libs.sh); 0: register-staged (round 3's first form)
    constexpr bool kDma = MVS16_LDSDMA && (!kStash || MVS16_STASH_DMA);     // see ring16_fetch
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s16[];
    constexpr int kRingBytes = kR16Slots * kR16SlotF4 * 16;                 // 72 KiB
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* stage = smem_s16 + kRingBytes + wave * (32 * kS16StageRowBytes);   // 8 KiB per wave
    // all biases (32x32 accumulator order) and the read-out bias live in LDS for the whole kernel
    float* net = reinterpret_cast<float*>(smem_s16 + kRingBytes + kW * 32 * kS16StageRowBytes) - kPackB0;
    for (int i = tid; i < kPackBr + 8 - kPackB0; i += 64 * kW) net[kPackB0 + i] = p.net[kPackB0 + i];
    int* table = reinterpret_cast<int*>(smem_s16 + kRingBytes + kW * 32 * kS16StageRowBytes + (kPackBr + 8 - kPackB0) * 4);
    // the read-out kernel, plain [128][4] (Keras order), for the vector-ALU read-out
    float* wr_plain = reinterpret_cast<float*>(table + kS16MaxPositions);
    for (int i = tid; i < 512; i += 64 * kW) wr_plain[i] = p.net[kPackWrPlain + i];

    Ring16 ring;
    ring.w = wsplit;
    ring.base = reinterpret_cast<f32x4*>(smem_s16);
    ring.c = 0;
    ring.p = 0;
    const int l0_units = kProj ? kS16L0Pe : kS16L0Pe + kS16L0Feat;
    ring.P = (l0_units + kS16PerView) * p.V + kS16PerView;
    for (int i = tid; i < ring.P; i += 64 * kW) table[i] = ring16_start_chunk(i, p.V, l0_units);
    ring.table = table;
    __syncthreads();                                                        // the position table is written
    ring.off = tid * 16;
    ring.tid = tid;
    ring.wave = wave;
    if (kDma) {
        ring16_dma(ring, table[0], 0);                                      // prologue: positions 0 and 1 into slots 0 and 1
        ring16_dma(ring, table[1], 1);
        ring.start_pf = table[2];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int q = 0; q < 2; ++q) {                                       // prologue: positions 0 and 1 into slots 0 and 1
            ring16_load(ring, table[q]);
            ring.c = (q + kR16Slots - 2) % kR16Slots;                       // ring16_store writes slot (c + 2) % 3
            ring16_store(ring);
        }
        ring.c = 0;
        ring16_load(ring, table[2]);
        ring.start_pf = table[3 % ring.P];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q) ring.a0[q] = __builtin_bit_cast(u32x4, ring16_cur(ring)[q * 64 + lane]);

    const long n_groups = (p.n_tiles + kW - 1) / kW;
    for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        long tile = grp * kW + wave;
        const bool tile_ok = tile < p.n_tiles;
        if (!tile_ok) tile = p.n_tiles - 1;                               // idle waves shadow the last tile, no stores
        // Loop-invariant scalars are re-read here through an empty asm: otherwise the compiler hoists everything derived from them
        // (float copies of H - 2 and W - 2, the reciprocals of the divisions by S and R, per-lane constants of the positional
        // encoding, ...) into registers that stay live across the whole tile - next to 2 x 64 accumulators that is ~60 spilled dwords
        // per lane (a first build: 260 B/lane of scratch, 330 MB of scratch traffic per fine launch by FETCH_SIZE / WRITE_SIZE).
        // The same for the lane coordinates: every swizzled LDS stage address of the gather (dozens of per-lane constants) is otherwise
        // computed once in front of the tile loop and kept.
        int pS = p.S, pR = p.R, pH = p.H, pW = p.W, gl = g, nl = n;
        asm volatile("" : "+s"(pS), "+s"(pR), "+s"(pH), "+s"(pW));
        asm volatile("" : "+v"(gl), "+v"(nl));
        SampleGeo sg[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            long gi = tile * 32 + 16 * cb + nl;
            sg[cb].valid = tile_ok && gi < p.total;
            if (gi >= p.total) gi = p.total - 1;
            sg[cb].g = gi;
            const int ray = (int)((unsigned)gi / (unsigned)pS);              // B*R*S < 2^31 (checked by the C entry points)
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
        B16 bop[2];                                                           // (MVS16_TAIL) the B operands of the next hidden layer's first k-step
        (void)bop;

        // (a sample row of 128 floats: lane (n, g) holds features 16 rb + 4g + {0..3} of samples n (cb 0) and 16 + n (cb 1))
        // the launcher sends V = 1 to the !kMultiView variants.  Telling the compiler so takes the optional-output and stash variants from 92-132
        // to 0 B/lane of scratch (hid is no longer carried around a loop it cannot see is a single trip) - and costs the plain variant 56 B,
        // which therefore keeps the runtime bound
        const int n_views = (kMultiView || !(kAux || kStash)) ? p.V : 1;
        for (int v = 0; v < n_views; ++v) {
            int tl[2];
            float ax[2], ay[2];
            long vrow[2];
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
                vrow[cb] = ((long)bv * pR + (sg[cb].ray - sg[cb].b * pR)) * pS + sg[cb].sidx;
                if (kAux && sg[cb].valid && gl == 0) {
                    if (p.tap_idx) {
                        int4 t4 = make_int4(tl[cb], tl[cb] + 1, tl[cb] + pW, tl[cb] + pW + 1);
                        *reinterpret_cast<int4*>(p.tap_idx + 4 * vrow[cb]) = t4;
                    }
                    if (p.pix) {
                        p.pix[2 * vrow[cb] + 0] = pxl;
                        p.pix[2 * vrow[cb] + 1] = pyl;
                    }
                }
                // accumulator seed = b0 + W0_dir^T PE(cam dir) of this (view, ray) (dir_bias_kernel, 32x32 accumulator order)
                {
                    const f32x4* seed = reinterpret_cast<const f32x4*>(p.dir_bias + 128 * ((long)bv * pR + (sg[cb].ray - sg[cb].b * pR)));
#pragma unroll
                    for (int rb = 0; rb < 8; ++rb) x[rb][cb] = seed[perm_f4(rb, gl)];
                }
                // this lane group's 16 of the 64 layer-0 inputs PE(cam xyz) | rgb of the sample (slot order: s16_pe_row).
                // g < 3: dimension g, octaves 0..7 - accurate sin/cos at octaves 0 and 5, double-angle steps in between
                // (fl32(x fl32(pi 2^k)) == 2^k fl32(x fl32(pi)) exactly; error x16 at most, as field_eval.hip);
                // g = 3: octaves 8 and 9 of the three dimensions - accurate at 8, one double-angle step - and the rgb taps.
                {
                    const float cd = gl == 0 ? cam[0] : (gl == 1 ? cam[1] : cam[2]);
                    const float a0 = (gl < 3 ? cd : cam[0]) * 3.14159274101257324f;
                    const float a1 = (gl < 3 ? cd : cam[1]) * 3.14159274101257324f;
                    const float a2 = cam[2] * 3.14159274101257324f;
                    float s0, c0, s1, c1, s2, c2;
#ifdef MVS16_ABL_SINCOS
                    s0 = a0; c0 = a1; s1 = a2; c1 = a0; s2 = a1; c2 = a2;        // timing-only ablation (wrong results): no accurate sin / cos
#else
                    sincos_f32(a0 * (gl < 3 ? 1.0f : 256.0f), &s0, &c0);
                    sincos_f32(a1 * (gl < 3 ? 32.0f : 256.0f), &s1, &c1);
                    sincos_f32(a2 * 256.0f, &s2, &c2);
#endif
                    auto dbl = [](float& sk, float& ck) {
                        const float t2 = sk + sk;
                        const float cn = fmaf(-t2, sk, 1.0f);              // cos 2t = 1 - 2 sin^2 t
                        sk = t2 * ck;                                      // sin 2t = 2 sin t cos t
                        ck = cn;
                    };
                    float va[16], vb[16];
                    // layout A (gl < 3): octaves 0..4 from chain 0, 5..7 from chain 1
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
                    // layout B (g = 3): (octave 8, octave 9) of dimensions 0, 1, 2, then rgb
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

            // ---- 2 k-steps: PE(cam xyz) + rgb rows ----
            {
                B16 bq[2], bqn[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const float v8[8] = {pe[cb][0], pe[cb][1], pe[cb][2], pe[cb][3], pe[cb][4], pe[cb][5], pe[cb][6], pe[cb][7]};
                    cut8<false>(v8, bq[cb]);
                }
                float nv[2][8];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int q = 0; q < 8; ++q) nv[cb][q] = pe[cb][8 + q];
                kstep16<false, 1, false, kDma>(ring, lane, g, bq, nv, bqn, x, x, nullptr);
                ring16_next<kDma>(ring);
                kstep16<false, 0, false, kDma>(ring, lane, g, bqn, nv, bq, x, x, nullptr);
                ring16_next<kDma>(ring);
            }

            // ---- layer 0's 256 feature rows through the wave-private fp32 stage ----
            // (fresh copies of the lane coordinates: the swizzled stage addresses below are computed here, not in front of the tile loop)
            asm volatile("" : "+v"(gl), "+v"(nl));
            // 16 lanes per sample row (16 B each), 4 rows per load instruction, 4 taps.  kProj: 2 passes over the 128-float table
            // rows [h][nb][16] (pass P = floats h*64 + P*32 + {0..31}), lerped rows ADD into the accumulators; direct: 4 passes
            // of 64 raw channels, the lerped rows are the B operands of 2 k-steps each.
#pragma unroll
#ifdef MVS16_ABL_GATHER
            for (int P = 0; P < 0; ++P) {                                        // timing-only ablation (wrong results): no table / feature gather
#else
            for (int P = 0; P < (kProj ? 2 : 4); ++P) {
#endif
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const f32x4* tbase = kProj ? reinterpret_cast<const f32x4*>(p.texel_table) + (nl >> 3) * 16 + (nl & 7) + P * 8
                                           : reinterpret_cast<const f32x4*>(p.features) + P * 16 + nl;
                const long row_f4 = kProj ? 32 : 64;                       // float4 per texel row
#pragma unroll
                for (int half = 0; half < 2; ++half) {                     // staged rows 16 half .. 16 half + 15 = column block `half`
                    f32x4 tv[4][4];
                    float axs[4], ays[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int src = 4 * u + gl;                         // sample 16 half + src lives in lane src (any lane group)
                        const int tls = __shfl(tl[half], src);
                        axs[u] = __shfl(ax[half], src);
                        ays[u] = __shfl(ay[half], src);
                        const f32x4* f = tbase + (long)tls * row_f4;
                        tv[u][0] = f[0];
                        tv[u][1] = f[row_f4];
                        tv[u][2] = f[(long)pW * row_f4];
                        tv[u][3] = f[(long)pW * row_f4 + row_f4];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int src = 16 * half + 4 * u + gl;
                        f32x4 o;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float top = fmaf(axs[u], tv[u][1][c] - tv[u][0][c], tv[u][0][c]);
                            const float bot = fmaf(axs[u], tv[u][3][c] - tv[u][2][c], tv[u][2][c]);
                            o[c] = fmaf(ays[u], bot - top, top);
                        }
                        *reinterpret_cast<f32x4*>(stage + src * kS16StageRowBytes + ((nl ^ (src & 15)) << 4)) = o;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (kProj) {
                    // row blocks 4P .. 4P + 3: chunk 8 (gl & 1) + 4 (rbl >> 1) + 2 (rbl & 1) + (gl >> 1) of the staged row
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int rbl = 0; rbl < 4; ++rbl) {
                            const int row = 16 * cb + nl, chunk = 8 * (gl & 1) + 4 * (rbl >> 1) + 2 * (rbl & 1) + (gl >> 1);
                            const f32x4 t4 = *reinterpret_cast<const f32x4*>(stage + row * kS16StageRowBytes + ((chunk ^ (row & 15)) << 4));
#pragma unroll
                            for (int c = 0; c < 4; ++c) x[4 * P + rbl][cb][c] += t4[c];
                        }
                } else {
                    // channels 64 P + 32 s + 8 gl + {0..7} of this lane's two samples: one k-step per s
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        B16 bq[2], bqn[2];
                        float nv[2][8];
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) {
                            const int row = 16 * cb + nl;
                            const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * kS16StageRowBytes + (((8 * s + 2 * gl) ^ (row & 15)) << 4));
                            const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * kS16StageRowBytes + (((8 * s + 2 * gl + 1) ^ (row & 15)) << 4));
                            const float b8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            cut8<false>(b8, bq[cb]);
#pragma unroll
                            for (int q = 0; q < 8; ++q) nv[cb][q] = 0.0f;
                        }
                        kstep16<false, 0, false, kDma>(ring, lane, gl, bq, nv, bqn, x, x, nullptr);
                        ring16_next<kDma>(ring);
                    }
                }
            }

            const long vslot = (long)p.B * p.V * p.R * p.S * 128;
            // training mode: view tile index (all 32 samples of a tile share b because R*S % 32 == 0 when V > 1)
            const long vtile = kMultiView ? ((long)(sg[0].b * p.V + v) * (p.n_tiles / p.B) + (tile - (long)sg[0].b * (p.n_tiles / p.B))) : tile;
            if (kStash && tile_ok) store_tl16(p.stash, vtile, n, g, x);                      // per-view slot 0: layer-0 output
            auto store_acc16 = [&](float* base, const long (&rows)[2]) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    if (sg[cb].valid) {
                        float* e = base + 128 * rows[cb] + 4 * g;
#pragma unroll
                        for (int rb = 0; rb < 8; ++rb) *reinterpret_cast<f32x4*>(e + 16 * rb) = x[rb][cb];
                    }
            };
            if (kAux && p.acts_view) store_acc16(p.acts_view, vrow);
            // ---- 24 k-steps: the three per-view ResNet blocks.  Only this entry is a layer boundary with exposed vector work (the
            // first operand's cut and the first bias row): every later boundary is prepared in the previous layer's last k-step ----
#if MVS16_TAIL
            if (v == 0)
#endif
            bias16<false>(net + kPackBHidden, g, hid);      // exposed once per (tile, view); the later layers' bias rows are applied in k-step shadows
#if MVS16_TAIL
            first_operand_s16(x, bop);
#endif
#pragma unroll 1
            for (int bi = 0; bi < 3; ++bi) {
                const float* bias1 = net + kPackBHidden + 256 * bi;
#if MVS16_TAIL
                dense128_s16<true>(ring, lane, g, x, hid, bop, bias1 + 128);     // hid = b1 + W1^T relu(x); tail: x += b2
                // x += W2^T relu(hid); tail: hid = b1 of the block that follows (the next view restarts at block 0)
                const float* next_b1 = bi < 2 ? bias1 + 256 : net + kPackBHidden + ((kMultiView && v + 1 < p.V) ? 0 : 768);
                dense128_s16<false>(ring, lane, g, hid, x, bop, next_b1);
#else
                // hid = b1 + W1^T relu(x), and in its last k-step x += b2; x += W2^T relu(hid), and in its last k-step hid = b1 of the
                // block that follows (the last view hands over to fusion block 3)
                // (a view that is followed by another one leaves hid alone: carried through the next view's sin / cos and gather it would be spilled)
#if MVS16_TAILBIAS
                const float* next_b1 = bi < 2 ? bias1 + 256 : (v + 1 == n_views ? net + kPackBHidden + 768 : nullptr);
                dense128_s16_plain<kDma, true>(ring, lane, g, x, hid, bias1 + 128);
                if (kStash && tile_ok) store_tl16(p.stash + (1 + 2 * bi) * p.stash_stride, vtile, n, g, hid);
                dense128_s16_plain<kDma, false>(ring, lane, g, hid, x, next_b1);
#else
                if (bi > 0) bias16<false>(bias1, g, hid);
                dense128_s16_plain<kDma, true>(ring, lane, g, x, hid, nullptr);
                if (kStash && tile_ok) store_tl16(p.stash + (1 + 2 * bi) * p.stash_stride, vtile, n, g, hid);
                bias16<true>(bias1 + 128, g, x);
                dense128_s16_plain<kDma, false>(ring, lane, g, hid, x, nullptr);
#endif
                // (per-view slot 6 = x3 is not written: nothing reads it, as in field_eval_split_kernel)
                if (kStash && tile_ok && bi < 2) store_tl16(p.stash + (2 + 2 * bi) * p.stash_stride, vtile, n, g, x);
#endif
                if (kAux && p.acts_view) store_acc16(p.acts_view + (bi + 1) * vslot, vrow);
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
        }

        // the samples' global indices are recomputed here instead of being carried through the whole tile (they would be spilled)
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
        auto store_fused16 = [&](float* base) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                if (gvalid[cb]) {
                    float* e = base + 128 * grow[cb] + 4 * g;
#pragma unroll
                    for (int rb = 0; rb < 8; ++rb) *reinterpret_cast<f32x4*>(e + 16 * rb) = x[rb][cb];
                }
        };
        if (kAux && p.acts_fused) store_fused16(p.acts_fused);               // complete_output: the view mean
        if (kStash && tile_ok) store_tl16(p.stash_fused, tile, n, g, x);      // fused slot 0: the view mean
        // ---- 24 k-steps: fusion blocks ----
#if MVS16_TAIL
        if (kMultiView) first_operand_s16(x, bop);                           // the view mean is new; V = 1: bop already is the cut of relu(x)
#endif
#pragma unroll 1
        for (int bi = 3; bi < 6; ++bi) {
            const float* bias1 = net + kPackBHidden + 256 * bi;
#if MVS16_TAIL
            dense128_s16<true>(ring, lane, g, x, hid, bop, bias1 + 128);
            dense128_s16<false>(ring, lane, g, hid, x, bop, bi < 5 ? bias1 + 256 : nullptr);
#else
#if MVS16_TAILBIAS
            dense128_s16_plain<kDma, true>(ring, lane, g, x, hid, bias1 + 128);
            if (kStash && tile_ok) store_tl16(p.stash_fused + (1 + 2 * (bi - 3)) * p.stash_fused_stride, tile, n, g, hid);
            dense128_s16_plain<kDma, false>(ring, lane, g, hid, x, bi < 5 ? bias1 + 256 : nullptr);
#else
            bias16<false>(bias1, g, hid);
            dense128_s16_plain<kDma, true>(ring, lane, g, x, hid, nullptr);
            if (kStash && tile_ok) store_tl16(p.stash_fused + (1 + 2 * (bi - 3)) * p.stash_fused_stride, tile, n, g, hid);
            bias16<true>(bias1 + 128, g, x);
            dense128_s16_plain<kDma, false>(ring, lane, g, hid, x, nullptr);
#endif
            if (kStash && tile_ok) store_tl16(p.stash_fused + (2 + 2 * (bi - 3)) * p.stash_fused_stride, tile, n, g, x);
#endif
            if (kAux && p.acts_fused) store_fused16(p.acts_fused + (long)(bi - 2) * p.total * 128);
        }
        if (kAux && p.embedding) store_fused16(p.embedding);

        // ---- read-out: Dense 128 -> 4 on relu(x), sigmoid / softplus (layers.py:392-397), on the vector ALU: 32 features per lane
        // x 2 samples x 4 outputs = 256 FMAs, then the sum over the four lane groups (two xor-shuffles)
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
            // lane group 0 stores column block 0's sample, lane group 1 column block 1's
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // weight requests still in flight must land before the LDS is released
}

}  // namespace

size_t MVS16_BYTES() { return (size_t)kS16Chunks * 1024; }

hipError_t MVS16_PACK(const float* net_keras, void* packed_split16, hipStream_t st) {
    const int n = kS16Chunks * kChunkElems16;
    hipLaunchKernelGGL(MVS16_PACK_KERNEL, dim3((n + 255) / 256), dim3(256), 0, st, net_keras, static_cast<unsigned short*>(packed_split16));
    return hipGetLastError();
}

bool MVS16_SUPPORTS(const FieldParams& p) {
    if (p.stash && MVS16_TAIL) return false;                                 // the stash stores are built into the plain layer flow only
    if (p.stash && (p.tap_idx || p.pix || p.embedding || p.acts_view || p.acts_fused)) return false;
    const int n_pos = ((p.texel_table ? kS16L0Pe : kS16L0Pe + kS16L0Feat) + kS16PerView) * p.V + kS16PerView;
    return n_pos <= kS16MaxPositions;
}

hipError_t MVS16_LAUNCH_FN(const FieldParams& p, const void* packed_split16, hipStream_t stream) {
    static std::mutex mtx;
    static bool attr_done[16] = {};
    static int cus[16] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (!MVS16_SUPPORTS(p)) return hipErrorInvalidValue;
    const int lds_bytes = kR16Slots * kR16SlotF4 * 16 + 8 * 32 * kS16StageRowBytes + (kPackBr + 8 - kPackB0) * 4 + kS16MaxPositions * 4 + 512 * 4;
    {
        std::lock_guard<std::mutex> lock(mtx);
        if (!attr_done[dev]) {
            hipDeviceProp_t prop;
            if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
            cus[dev] = prop.multiProcessorCount;
            const void* fns[12] = {reinterpret_cast<const void*>(&MVS16_KERNEL<false, false, false, false>),
                                   reinterpret_cast<const void*>(&MVS16_KERNEL<false, true, false, false>),
                                   reinterpret_cast<const void*>(&MVS16_KERNEL<true, false, false, false>),
                                   reinterpret_cast<const void*>(&MVS16_KERNEL<true, true, false, false>),
                                   reinterpret_cast<const void*>(&MVS16_KERNEL<false, false, true, false>),
                                   reinterpret_cast<const void*>(&MVS16_KERNEL<false, true, true, false>),
                                   reinterpret_cast<const void*>(&MVS16_KERNEL<true, false, true, false>),
                                   reinterpret_cast<const void*>(&MVS16_KERNEL<true, true, true, false>),
                                   reinterpret_cast<const void*>(&MVS16_KERNEL<false, false, false, true>),
                                   reinterpret_cast<const void*>(&MVS16_KERNEL<false, true, false, true>),
                                   reinterpret_cast<const void*>(&MVS16_KERNEL<true, false, false, true>),
                                   reinterpret_cast<const void*>(&MVS16_KERNEL<true, true, false, true>)};
            for (const void* fn : fns)
                if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess) return e;
            attr_done[dev] = true;
        }
    }
    if ((e = launch_dir_bias(p, stream)) != hipSuccess) return e;
    const bool mv = p.V > 1;
    const long n_groups = (p.n_tiles + 7) / 8;
    const long resident = (long)cus[dev];                                   // persistent: one workgroup per CU
    const unsigned wgs = (unsigned)(n_groups < resident ? n_groups : resident);
    const f32x4* w = static_cast<const f32x4*>(packed_split16);
    const dim3 grid(wgs), block(512);
#define MVS16_LAUNCH(MV, PROJ, AUX, STASH) hipLaunchKernelGGL((MVS16_KERNEL<MV, PROJ, AUX, STASH>), grid, block, lds_bytes, stream, p, w)
    const bool aux = p.tap_idx || p.pix || p.embedding || p.acts_view || p.acts_fused;
    if (p.stash && p.V > 1 && ((long)p.R * p.S) % 32 != 0) return hipErrorInvalidValue;     // tiles must not straddle scenes
    if (p.stash) {
        const int variant = (mv ? 2 : 0) + (p.texel_table ? 1 : 0);
        switch (variant) {
            case 0: MVS16_LAUNCH(false, false, false, true); break;
            case 1: MVS16_LAUNCH(false, true, false, true); break;
            case 2: MVS16_LAUNCH(true, false, false, true); break;
            default: MVS16_LAUNCH(true, true, false, true); break;
        }
        return hipGetLastError();
    }
    const int variant = (mv ? 4 : 0) + (p.texel_table ? 2 : 0) + (aux ? 1 : 0);
    switch (variant) {
        case 0: MVS16_LAUNCH(false, false, false, false); break;
        case 1: MVS16_LAUNCH(false, false, true, false); break;
        case 2: MVS16_LAUNCH(false, true, false, false); break;
        case 3: MVS16_LAUNCH(false, true, true, false); break;
        case 4: MVS16_LAUNCH(true, false, false, false); break;
        case 5: MVS16_LAUNCH(true, false, true, false); break;
        case 6: MVS16_LAUNCH(true, true, false, false); break;
        default: MVS16_LAUNCH(true, true, true, false); break;
    }
#undef MVS16_LAUNCH
    return hipGetLastError();
}

}  // namespace mvnerf
