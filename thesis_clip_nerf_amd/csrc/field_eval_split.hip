// fp32-grade field kernel on the bf16 matrix pipe ("split3"): same math, inputs and outputs as field_eval.hip.
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate on gfx950.  Every fp32 operand is therefore cut into three
// bf16 pieces, x = x1 + x2 + x3 (8 significand bits each, 24 together; the cut is exact: each remainder is formed by an
// exact fp32 subtraction), and a product a.b is issued as the six bf16 MFMAs of order >= 2^-16,
//       a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),
// accumulated in the fp32 accumulator (bf16 x bf16 products are exact in fp32).  The dropped terms a2 b3, a3 b2, a3 b3
// are <= 2^-24 relative - the size of one fp32 rounding.  Measured on one 32x32xK=128 block against float64
// (scripts/x6_probe.hip, profiles/r02_x6_probe.log): max error 6.4e-7 against 1.0e-6 for the fp32 MFMA chain - the
// result is as close to the exact product as the fp32 MFMA's - at 6/16 of its matrix-pipe time.
//
// Machine mapping (as field_eval_bf16.hip): one 512-thread workgroup per CU, persistent over groups of 8 tiles of 32
// samples; activations stay in the fp32 accumulators; the weights (split once by pack_net_split_kernel, 1.4 MiB per MLP)
// travel through a ring of LDS slots by LDS-DMA, one k-step of 16 input rows x 4 output blocks x 3 pieces = 12 KiB per
// slot; per k-step a wave cuts its 8 B values (relu first where the layer has one) with 5 vector instructions per value
// (and / sub / and / sub + byte permutes, in the shadow of the other wave's MFMAs) and issues 24 MFMAs.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <atomic>
#include <mutex>

#include "mvnerf_kernels.h"
#include "mvnerf_math.h"
#include "mvnerf_mfma.h"

namespace mvnerf {

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// ---- split weight stream: 1 KiB chunks [lane][8 bf16]; chunk index = 12 * kstep + 3 * nb + piece ------------------
//   layer 0 : k-steps 0..3  = PE(cam xyz) (lower half-wave sin rows, upper cos rows) + rgb rows
//             k-steps 4..19 = the 256 feature rows, channel 16 (ks - 4) + 8 h + jj            (direct gather only)
//   hidden l: k-steps 20 + 8 l + (2 kb + s): input feature 32 kb + 16 s + 8 (jj >> 2) + 4 h + (jj & 3)
//             (= accumulator registers 8 s .. 8 s + 7 of block kb fed as B operand)
//   read-out: 8 k-steps (2 kb + s), ONE output block (rows >= 4 zero): 3 chunks per k-step, 24 chunks = 2 slots
constexpr int kSegChunks = 12;                         // chunks per ring slot
constexpr int kSpL0Steps = 20, kSpHiddenSteps = 96;
constexpr int kSpReadoutChunk = (kSpL0Steps + kSpHiddenSteps) * kSegChunks;     // 1392
constexpr int kSpChunks = kSpReadoutChunk + 24;                                 // 1416 chunks = 1.38 MiB
constexpr int kChunkElems = 512;

__global__ void pack_net_split_kernel(const float* __restrict__ src, __bf16* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kSpChunks * kChunkElems) return;
    const int chunk = idx / kChunkElems, lane = (idx % kChunkElems) / 8, jj = idx % 8;
    const int i = lane & 31, h = lane >> 5;
    float val = 0.0f;
    int piece;
    if (chunk < kSpReadoutChunk) {
        const int ks = chunk / kSegChunks, nb = (chunk % kSegChunks) / 3;
        piece = chunk % 3;
        if (ks < kSpL0Steps) {
            int row = -1;
            if (ks < 4) {
                const int m = 8 * ks + jj;
                if (m < 30) row = (m / 10) * 20 + 2 * (m % 10) + h;
                else if (m == 30) row = 120 + h;
                else row = h ? -1 : 122;
            } else {
                row = 123 + 16 * (ks - 4) + 8 * h + jj;
            }
            if (row >= 0) val = src[kKerasW0 + row * kHidden + 32 * nb + i];
        } else {
            const int q = ks - kSpL0Steps, layer = q / 8, kbs = q % 8;
            const int f = 32 * (kbs / 2) + 16 * (kbs % 2) + 8 * (jj >> 2) + 4 * h + (jj & 3);
            const int wsrc = kKerasBlocks + (layer / 2) * kKerasBlockStride + (layer % 2) * (kHidden * kHidden + kHidden);
            val = src[wsrc + f * kHidden + 32 * nb + i];
        }
    } else {
        const int q = chunk - kSpReadoutChunk, kbs = q / 3;
        piece = q % 3;
        const int f = 32 * (kbs / 2) + 16 * (kbs % 2) + 8 * (jj >> 2) + 4 * h + (jj & 3);
        if (i < 4) val = src[kKerasWr + f * 4 + i];
    }
    // round-to-nearest pieces; every remainder is exact in fp32
    const __bf16 p1 = (__bf16)val;
    const float r1 = val - (float)p1;
    const __bf16 p2 = (__bf16)r1;
    const float r2 = r1 - (float)p2;
    const __bf16 p3 = (__bf16)r2;
    dst[idx] = piece == 0 ? p1 : (piece == 1 ? p2 : p3);
}

__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Truncation cut of 8 fp32 values into 3 x 8 bf16: piece = the upper 16 bits (v_perm_b32 packs two of them), remainder =
// x - piece (exact).  The third piece drops what is left below 24 bits (<= 2^-24 |x|).
__device__ __forceinline__ void split3(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
    float r1[8], r2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned u = __builtin_bit_cast(unsigned, x[q]);
        r1[q] = x[q] - __builtin_bit_cast(float, u & 0xffff0000u);
        const unsigned u1 = __builtin_bit_cast(unsigned, r1[q]);
        r2[q] = r1[q] - __builtin_bit_cast(float, u1 & 0xffff0000u);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        p1[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x[2 * q + 1]), __builtin_bit_cast(unsigned, x[2 * q]), 0x07060302u);
        p2[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, r1[2 * q + 1]), __builtin_bit_cast(unsigned, r1[2 * q]), 0x07060302u);
        p3[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, r2[2 * q + 1]), __builtin_bit_cast(unsigned, r2[2 * q]), 0x07060302u);
    }
}

// timing-only ablations (wrong results), scripts/ab_split.sh
#ifndef MVS_ABL_BARRIER
#define MVS_ABL_BARRIER 0  // no workgroup barrier at k-step ends
#endif
#ifndef MVS_ABL_DMA
#define MVS_ABL_DMA 0      // every weight DMA re-reads the first slot's data
#endif
#ifndef MVS_ABL_SPLIT
#define MVS_ABL_SPLIT 0    // no relu / cut of the B operands
#endif
#ifndef MVS_ABL_AREAD
#define MVS_ABL_AREAD 0    // A operands not re-read from LDS
#endif
#ifndef MVS_ABL_GATHER
#define MVS_ABL_GATHER 0   // no table / feature gather
#endif
#ifndef MVS_ABL_BIAS
#define MVS_ABL_BIAS 0     // no bias loads / adds
#endif
#ifndef MVS_ABL_NODMA
#define MVS_ABL_NODMA 0    // no weight DMA after the prologue
#endif
#ifndef MVS_DMA_AT
#define MVS_DMA_AT 1       // where a k-step issues its weight DMA: after output block MVS_DMA_AT - 1 (1..4)
#endif
#ifndef MVS_STAGE_LATE
#define MVS_STAGE_LATE 1   // 1: a k-step stores (after its first output block) what the PREVIOUS k-step loaded, then loads position p + 3:
#endif                     //    the loads have a whole k-step to land before anything waits for them; 0: load p + 2, store at the k-step's end
#ifndef MVS_MV_W4
#define MVS_MV_W4 0       // multi-view kernels as 4 waves x 512 registers (view sum in registers, no scratch) instead of 8 x 256
#endif                     // with the view sum spilled: A/B equal (0.47 vs 0.48 of peak at V = 3), 8 x 256 kept
#ifndef MVS_STAMP
#define MVS_STAMP 0        // debugging: per-wave cycle totals (k-step bodies / barrier waits / whole kernel) over the `pix` output
#endif
#ifndef MVS_DRAIN
#define MVS_DRAIN 0        // debugging: every k-step drains all vector-memory operations
#endif

// ---- the slot ring (one k-step per slot) ---------------------------------------------------------------------------
constexpr int kWgWaves = 8;
constexpr int kRing = 3, kSlotF4 = kSegChunks * 64;      // float4 per slot (12 KiB)
constexpr int kHiddenUnits = 48;                                     // k-steps of 3 ResNet blocks (6 Dense layers x 8)
constexpr int kMaxPositions = 1024;                                  // k-steps per tile the LDS position table holds

struct Ring {
    const f32x4* w;
    f32x4* base;        // LDS
    int c;              // ring slot of the current k-step
    int p, P, V;
    int l0_units;       // layer-0 k-steps per view: 4 (texel table) or 20
    int tid, wave;
    const int* table;   // LDS: first chunk of every position of one tile
    int start_pf;       // table entry (first chunk) of the position the next ring_fetch loads, read one k-step ahead
    int off_a, off_b;   // this thread's 16 + 8 bytes inside a slot: wave * 1536 + lane * 16 | wave * 1536 + 1024 + lane * 8
    f32x4 stg0;         // the fetched bytes on their way to LDS (8 waves: 16 + 8 B per thread; 4 waves: 3 x 16 B)
    f32x2 stg1;
    f32x4 stg2, stg3;
#if MVS_STAMP
    unsigned long long t_last, t_body, t_wait, t_grp[4], t_mark;
#endif
    u32x4 a0[3];        // A operands (3 pieces) of output block 0 of the CURRENT k-step, read during the previous one
};

struct BParts {
    u32x4 p1, p2, p3;
};

__device__ __forceinline__ int ring_start_chunk(int p, int V, int l0_units) {
    const int per_view = l0_units + kHiddenUnits;
    if (p < per_view * V) {
        const int q = p % per_view;
        return (q < l0_units ? q : kSpL0Steps + (q - l0_units)) * kSegChunks;
    }
    const int q = p - per_view * V;
    return q < kHiddenUnits ? (kSpL0Steps + kHiddenUnits + q) * kSegChunks : kSpReadoutChunk + (q - kHiddenUnits) * kSegChunks;
}

// The weight stream reaches LDS through registers: after its first output block a k-step at position p stores what the
// previous k-step loaded (position p + 2, 12 KiB contiguous in the stream: every thread one dwordx4 + one dwordx2 = 24 B,
// the same two instructions in every wave, no branch) into slot (c + 2) % 3 and loads position p + 3 into the same
// registers - a whole k-step of matrix work lies between a load and the store that waits for it, no counted vmcnt, three
// slots.  LDS-DMA (global_load_lds_dwordx4 into a 5-slot ring with counted waits) was the first version and measured the
// same within the box-to-box spread; 12 chunks do not divide evenly over 8 waves there, and the dwordx3
// form that would writes LDS at a 16-byte lane stride (scripts/dma3_probe.hip).
// kW = waves of the workgroup: 8 (two 256-register waves per SIMD) or 4 (multi-view kernels: one 512-register wave per SIMD,
// so that the view sum stays in registers; a thread then moves 3 x 16 B of every slot)
template <int kW>
__device__ __forceinline__ void ring_store(Ring& r) {
    if (MVS_ABL_NODMA) return;
    int slot = r.c + 2;
    slot = slot >= kRing ? slot - kRing : slot;
    char* dst = reinterpret_cast<char*>(r.base) + slot * (kSlotF4 * 16);
    if (kW == 8) {
        *reinterpret_cast<f32x4*>(dst + r.off_a) = r.stg0;
        *reinterpret_cast<f32x2*>(dst + r.off_b) = r.stg1;
    } else {
        *reinterpret_cast<f32x4*>(dst + r.off_a) = r.stg0;
        *reinterpret_cast<f32x4*>(dst + r.off_a + 4096) = r.stg2;
        *reinterpret_cast<f32x4*>(dst + r.off_a + 8192) = r.stg3;
    }
}

template <int kW>
__device__ __forceinline__ void ring_load_stage(Ring& r, const char* src) {
    if (kW == 8) {
        r.stg0 = *reinterpret_cast<const f32x4*>(src + r.off_a);
        r.stg1 = *reinterpret_cast<const f32x2*>(src + r.off_b);
    } else {
        r.stg0 = *reinterpret_cast<const f32x4*>(src + r.off_a);
        r.stg2 = *reinterpret_cast<const f32x4*>(src + r.off_a + 4096);
        r.stg3 = *reinterpret_cast<const f32x4*>(src + r.off_a + 8192);
    }
}

template <int kW>
__device__ __forceinline__ void ring_fetch(Ring& r) {
    if (MVS_ABL_NODMA) return;
    if (MVS_STAGE_LATE) ring_store<kW>(r);                              // position p + 2, loaded one k-step ago
    ring_load_stage<kW>(r, reinterpret_cast<const char*>(r.w) + (long)(MVS_ABL_DMA ? 0 : r.start_pf) * 1024);
    int pp = r.p + 3 + MVS_STAGE_LATE;                                  // table entry the NEXT k-step's fetch needs
    pp = pp >= r.P ? pp - r.P : pp;
    r.start_pf = r.table[pp];
}

__device__ __forceinline__ const f32x4* ring_cur(const Ring& r) { return r.base + r.c * kSlotF4; }
__device__ __forceinline__ const f32x4* ring_nxt(const Ring& r) { return r.base + (r.c + 1 == kRing ? 0 : r.c + 1) * kSlotF4; }

// End of the k-step at position p (slot c): the fetched position p + 2 goes into slot (c + 2) % 3 - the slot of position
// p - 1, which nobody reads any more since the previous barrier - and the barrier publishes it.  A k-step reads its own slot
// and, for the A operands of the next k-step's first output block, the next one: both were published at least one barrier
// ago.
template <int kW>
__device__ __forceinline__ void ring_next(Ring& r) {
#if MVS_STAMP
    const unsigned long long t1 = __builtin_readcyclecounter();
    r.t_body += t1 - r.t_last;
#endif
    if (!MVS_STAGE_LATE) ring_store<kW>(r);
#if MVS_ABL_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#if MVS_STAMP
    r.t_last = __builtin_readcyclecounter();
    r.t_wait += r.t_last - t1;
#endif
    r.c = r.c + 1 == kRing ? 0 : r.c + 1;
    r.p = r.p + 1 == r.P ? 0 : r.p + 1;
}

// values 2q, 2q+1 of an 8-value B operand -> dword q of the three pieces (relu first where the layer has one)
template <bool kRelu>
__device__ __forceinline__ void split_pair(float v0, float v1, int q, BParts& b) {
#if MVS_ABL_SPLIT
    b.p1[q] = __builtin_bit_cast(unsigned, v0);
    b.p2[q] = __builtin_bit_cast(unsigned, v1);
    b.p3[q] = __builtin_bit_cast(unsigned, v0);
    return;
#endif
    if (kRelu) {               // relu on the bit pattern: one v_max_i32
        const int i0 = __builtin_bit_cast(int, v0), i1 = __builtin_bit_cast(int, v1);
        v0 = __builtin_bit_cast(float, i0 > 0 ? i0 : 0);
        v1 = __builtin_bit_cast(float, i1 > 0 ? i1 : 0);
    }
    const float r0 = v0 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v0) & 0xffff0000u);
    const float r1 = v1 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v1) & 0xffff0000u);
    const float s0 = r0 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r0) & 0xffff0000u);
    const float s1 = r1 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
    b.p1[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, v1), __builtin_bit_cast(unsigned, v0), 0x07060302u);
    b.p2[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, r1), __builtin_bit_cast(unsigned, r0), 0x07060302u);
    b.p3[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
}

template <bool kRelu>
__device__ __forceinline__ void split8(const float (&v)[8], BParts& b) {
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair<kRelu>(v[2 * q], v[2 * q + 1], q, b);
}

// One k-step: acc[nb] += A(nb)^T b for the 4 output blocks, 6 MFMAs each (smallest terms first).  Software pipeline: while
// the MFMAs of output block nb run, the A operands of the next block (of the NEXT k-step's block 0 at nb = 3) are read
// from LDS and, with kNext, a quarter of the next k-step's B operand is cut on the vector ALU - two vector instructions per
// MFMA gap, which the matrix pipe hides (MI355X_MICROARCH.md: <= 5 single-issue fillers per 32x32x16 MFMA).
template <bool kRelu, bool kNext, int kW>
__device__ __forceinline__ void kstep_mfma(Ring& ring, int lane, const BParts& b, const float (&n8)[8], BParts& bn, f32x16 (&acc)[4]) {
    const f32x4* cur = ring_cur(ring) + lane;
    const f32x4* nxt = ring_nxt(ring) + lane;
    u32x4 a[3] = {ring.a0[0], ring.a0[1], ring.a0[2]};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        u32x4 an[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) an[q] = MVS_ABL_AREAD ? a[(q + 1) % 3] : __builtin_bit_cast(u32x4, nb < 3 ? cur[((nb + 1) * 3 + q) * 64] : nxt[q * 64]);
        if (kNext) split_pair<kRelu>(n8[2 * nb], n8[2 * nb + 1], nb, bn);
        acc[nb] = mfma16(a[2], b.p1, acc[nb]);
        acc[nb] = mfma16(a[1], b.p2, acc[nb]);
        acc[nb] = mfma16(a[0], b.p3, acc[nb]);
        acc[nb] = mfma16(a[1], b.p1, acc[nb]);
        acc[nb] = mfma16(a[0], b.p2, acc[nb]);
        acc[nb] = mfma16(a[0], b.p1, acc[nb]);
        // issue order inside the group: the first MFMA (its operands were requested one group ago, so the LDS wait in front
        // of it covers nothing newer), then the three LDS reads of the next group, then MFMA / 2 VALU alternating
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        if (kNext) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (kNext) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (kNext) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
#if MVS_STAMP
        {
            const unsigned long long tg = __builtin_readcyclecounter();
            ring.t_grp[nb] += tg - (nb == 0 ? ring.t_last : ring.t_mark);
            ring.t_mark = tg;
        }
#endif
        if (nb + 1 == MVS_DMA_AT) {   // the weight loads of two k-steps ahead
            ring_fetch<kW>(ring);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) a[q] = an[q];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) ring.a0[q] = a[q];
    if (kNext) {
        // pin the pieces of the next operand HERE: the k-step ends in ring_next's wave-uniform branches, and the machine sinker
        // would otherwise move the whole cut into the next k-step's block - behind the barrier, where nothing hides it
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned u1 = bn.p1[q], u2 = bn.p2[q], u3 = bn.p3[q];
            asm volatile("" : "+v"(u1), "+v"(u2), "+v"(u3));
            bn.p1[q] = u1;
            bn.p2[q] = u2;
            bn.p3[q] = u3;
        }
    }
}

// acc += W^T relu(in) for one hidden layer: 8 k-steps (input block kb = g / 2, registers 8 (g & 1) .. + 7).  The cut of
// k-step 0's operand cannot overlap anything of this wave (it needs the previous layer's last MFMA); k-steps 1..7 are cut
// during the MFMAs of their predecessors.
template <int kW>
__device__ __forceinline__ void dense128_split(Ring& ring, int lane, const f32x16 (&in)[4], f32x16 (&acc)[4]) {
    BParts b, bn;
    {
        float v8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v8[q] = in[0][q];
        split8<true>(v8, b);
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float n8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) n8[q] = g < 7 ? in[(g + 1) >> 1][8 * ((g + 1) & 1) + q] : 0.0f;
        if (g < 7) kstep_mfma<true, true, kW>(ring, lane, b, n8, bn, acc);
        else kstep_mfma<true, false, kW>(ring, lane, b, n8, bn, acc);
        b = bn;
        ring_next<kW>(ring);
    }
}

template <bool kAdd>
__device__ __forceinline__ void bias_acc(const float* __restrict__ bperm, int h, f32x16 (&acc)[4]) {
#if MVS_ABL_BIAS
    if (!kAdd) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = (float)h;
    }
    return;
#endif
    const f32x4* p = reinterpret_cast<const f32x4*>(bperm + h * 64);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = p[nb * 4 + q];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (kAdd) {
                    // scalar adds on purpose (v_pk_add_f32 beside MFMAs costs the partner wave's matrix pipe: MI355X_MICROARCH.md)
                    float r = acc[nb][4 * q + c];
                    asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(v[c]));
                    acc[nb][4 * q + c] = r;
                } else {
                    acc[nb][4 * q + c] = v[c];
                }
            }
        }
}

constexpr int kStageRowBytes = 256;      // per staged sample row: 64 fp32 channels

// kProj: layer 0's feature rows come from the fp32 texel table (project_texels_kernel, field_eval.hip).
// kStash (training forward): the trunk's pre-activations go to HBM in tile layout, as field_eval_kernel<.., kStash> does.
template <bool kMultiView, bool kProj, bool kStash>
__global__ __launch_bounds__((kMultiView && MVS_MV_W4) ? 256 : 512, (kMultiView && MVS_MV_W4) ? 1 : 2) void field_eval_split_kernel(FieldParams p, const f32x4* __restrict__ wsplit) {
    constexpr int kW = (kMultiView && MVS_MV_W4) ? 4 : 8;                                  // waves per workgroup (see ring_store)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sp[];
    constexpr int kRingBytes = kRing * kSlotF4 * 16;                        // 36 KiB
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* stage = smem_sp + kRingBytes + wave * (32 * kStageRowBytes);   // 8 KiB per wave
    // all biases (accumulator order) and the read-out bias live in LDS for the whole kernel (a global load in the middle of
    // a k-step would make the in-order vmcnt wait drain the weight prefetch)
    float* net = reinterpret_cast<float*>(smem_sp + kRingBytes + kW * 32 * kStageRowBytes) - kPackB0;
    for (int i = tid; i < kPackBr + 8 - kPackB0; i += 64 * kW) net[kPackB0 + i] = p.net[kPackB0 + i];

    Ring ring;
    ring.w = wsplit;
    ring.base = reinterpret_cast<f32x4*>(smem_sp);
    ring.c = 0;
    ring.p = 0;
    ring.V = p.V;
    ring.l0_units = kProj ? 4 : kSpL0Steps;
    ring.P = (ring.l0_units + kHiddenUnits) * p.V + kHiddenUnits + 2;
    ring.tid = tid;
    ring.wave = wave;
    {   // chunk every position of a tile starts at
        int* table = reinterpret_cast<int*>(smem_sp + kRingBytes + kW * 32 * kStageRowBytes + (kPackBr + 8 - kPackB0) * 4);
        for (int i = tid; i < ring.P; i += 64 * kW) table[i] = ring_start_chunk(i, p.V, ring.l0_units);
        ring.table = table;
    }
    __syncthreads();                                                        // the position table is written
    ring.off_a = kW == 8 ? wave * 1536 + lane * 16 : tid * 16;
    ring.off_b = wave * 1536 + 1024 + lane * 8;
    for (int q = 0; q < 2; ++q) {                                           // prologue: positions 0 and 1 into slots 0 and 1
        const char* src = reinterpret_cast<const char*>(wsplit) + (long)ring.table[q] * 1024;
        ring_load_stage<kW>(ring, src);
        ring.c = q == 0 ? 1 : 2;                                            // ring_store writes slot (c + 2) % 3
        ring_store<kW>(ring);

    }
    ring.c = 0;
    if (MVS_STAGE_LATE) ring_load_stage<kW>(ring, reinterpret_cast<const char*>(wsplit) + (long)ring.table[2] * 1024);
    ring.start_pf = ring.table[2 + MVS_STAGE_LATE];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q) ring.a0[q] = __builtin_bit_cast(u32x4, ring_cur(ring)[q * 64 + lane]);
#if MVS_STAMP
    const unsigned long long t_begin = __builtin_readcyclecounter();
    ring.t_last = t_begin;
    ring.t_body = 0;
    ring.t_wait = 0;
    ring.t_mark = 0;
    for (int q = 0; q < 4; ++q) ring.t_grp[q] = 0;
#endif

    const long n_groups = (p.n_tiles + kW - 1) / kW;
    for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        long tile = grp * kW + wave;
        const bool tile_ok = tile < p.n_tiles;
        if (!tile_ok) tile = p.n_tiles - 1;                               // idle waves shadow the last tile, no stores
        long g = tile * 32 + j;
        const bool valid = tile_ok && g < p.total;
        if (g >= p.total) g = p.total - 1;
        const int ray = (int)(g / p.S);
        const int sidx = (int)(g - (long)ray * p.S);
        const int b = ray / p.R;
        const float ox = p.rays_o[3 * ray + 0], oy = p.rays_o[3 * ray + 1], oz = p.rays_o[3 * ray + 2];
        const float dx = p.rays_d[3 * ray + 0], dy = p.rays_d[3 * ray + 1], dz = p.rays_d[3 * ray + 2];
        const float zz = p.z[g];
        const float wx = ox + zz * dx, wy = oy + zz * dy, wz = oz + zz * dz;

        f32x16 x[4], hid[4];
        f32x16 xsum[kMultiView ? 4 : 1];

        // one sample row (128 floats) from the accumulators: lane (j,h) holds features 32nb + 8q + 4h + {0..3}
        auto store_row = [&](float* row128) {
            float* e = row128 + 4 * h;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v4 = {x[nb][4 * q], x[nb][4 * q + 1], x[nb][4 * q + 2], x[nb][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(e + 32 * nb + 8 * q) = v4;
                }
        };

        for (int v = 0; v < p.V; ++v) {
            const int bv = b * p.V + v;
            const float* E = p.einv + 16 * bv;
            float cam[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) cam[r] = row_dot4(E, r, wx, wy, wz, 1.0f);
            float pxl, pyl;
            pixel_from_cam(p.k4 + 16 * bv, cam, &pxl, &pyl);
            const Taps tp = bilinear_taps(pxl, pyl, p.H, p.W);
            const int tl = (bv * p.H + tp.y0) * p.W + tp.x0;
            const long vrow = ((long)bv * p.R + (ray - b * p.R)) * p.S + sidx;
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

            // ---- 4 k-steps: PE(cam xyz) + rgb rows; accumulator seed = b0 + W0_dir^T PE(cam dir) (dir_bias_kernel) ----
            bias_acc<false>(p.dir_bias + 128 * ((long)bv * p.R + (ray - b * p.R)), h, x);
            float pe[32];
            {
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
            for (int d = 0; d < 3; ++d) {
                const float a0 = cam[d] * 3.14159274101257324f;
                float sk = 0.0f, ck = 0.0f;
#pragma unroll
                for (int k = 0; k < kNFreq; ++k) {
                    if (k == 0 || k == 5) {                 // accurate seeds, double-angle steps in between (field_eval.hip)
                        sincos_f32(a0 * (float)(1 << k), &sk, &ck);
                    } else {
                        const float s2 = sk + sk;
                        const float cn = fmaf(-s2, sk, 1.0f);
                        sk = s2 * ck;
                        ck = cn;
                    }
                    pe[d * 10 + k] = h ? ck : sk;
                }
            }
            {
                BParts bq, bqn;
                float v8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v8[q] = pe[q];
                split8<false>(v8, bq);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    float n8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) n8[q] = ks < 3 ? pe[8 * (ks + 1) + q] : 0.0f;
                    if (ks < 3) kstep_mfma<false, true, kW>(ring, lane, bq, n8, bqn, x);
                    else kstep_mfma<false, false, kW>(ring, lane, bq, n8, bqn, x);
                    bq = bqn;
                    ring_next<kW>(ring);
                }
            }

            // ---- layer 0's 256 feature rows: 4 passes of 64 channels through the wave-private fp32 stage ----
            // 16 lanes per sample row (16 B each), 4 rows per load instruction, 4 taps, batches of 4 instructions per tap.
            // kProj: 2 passes over the 128-float table rows [h][nb][16] (pass P = floats h*64 + P*32 + {0..31}), lerped
            // rows ADD into the accumulators; direct: pass = 64 raw channels, lerped rows are the B operands of 4 k-steps.
            const int l16 = lane & 15, sub = lane >> 4;
#pragma unroll
            for (int P = 0; P < (kProj ? 2 : 4); ++P) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const f32x4* tbase = kProj ? reinterpret_cast<const f32x4*>(p.texel_table) + (l16 >> 3) * 16 + (l16 & 7) + P * 8
                                           : reinterpret_cast<const f32x4*>(p.features) + P * 16 + l16;
                const long row_f4 = kProj ? 32 : 64;                       // float4 per texel row
#pragma unroll 1
                for (int it0 = 0; it0 < (MVS_ABL_GATHER ? 0 : 8); it0 += 4) {
                    f32x4 tv[4][4];
                    float axs[4], ays[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int src = 4 * (it0 + u) + sub;
                        const int tls = __shfl(tl, src);
                        axs[u] = __shfl(tp.ax, src);
                        ays[u] = __shfl(tp.ay, src);
                        const f32x4* f = tbase + (long)tls * row_f4;
                        tv[u][0] = f[0];
                        tv[u][1] = f[row_f4];
                        tv[u][2] = f[(long)p.W * row_f4];
                        tv[u][3] = f[(long)p.W * row_f4 + row_f4];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int src = 4 * (it0 + u) + sub;
                        f32x4 o;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float top = fmaf(axs[u], tv[u][1][c] - tv[u][0][c], tv[u][0][c]);
                            const float bot = fmaf(axs[u], tv[u][3][c] - tv[u][2][c], tv[u][2][c]);
                            o[c] = fmaf(ays[u], bot - top, top);
                        }
                        *reinterpret_cast<f32x4*>(stage + src * kStageRowBytes + ((l16 ^ (src & 15)) << 4)) = o;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (kProj) {
#pragma unroll
                    for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 t4 = *reinterpret_cast<const f32x4*>(stage + j * kStageRowBytes + (((8 * h + 4 * nbl + q) ^ (j & 15)) << 4));
#pragma unroll
                            for (int c = 0; c < 4; ++c) x[2 * P + nbl][4 * q + c] += t4[c];
                        }
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {                           // channels 64 P + 16 s + 8 h + {0..7}
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + j * kStageRowBytes + (((4 * s + 2 * h) ^ (j & 15)) << 4));
                        const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + j * kStageRowBytes + (((4 * s + 2 * h + 1) ^ (j & 15)) << 4));
                        const float b8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        BParts bq, bqn;
                        split8<false>(b8, bq);
                        kstep_mfma<false, false, kW>(ring, lane, bq, b8, bqn, x);
                        ring_next<kW>(ring);
                    }
                }
            }

            const long vslot = (long)p.B * p.V * p.R * p.S * 128;
            if (p.acts_view && valid) store_row(p.acts_view + 128 * vrow);
            // training mode: view tile index (all 32 samples of a tile share b because R*S % 32 == 0 when V > 1)
            const long vtile = kMultiView ? ((long)bv * (p.n_tiles / p.B) + (tile - (long)b * (p.n_tiles / p.B))) : tile;
            if (kStash && tile_ok) store_tl(p.stash, vtile, j, h, x);                       // per-view slot 0: layer-0 output
            // ---- 48 k-steps: the three per-view ResNet blocks ----
#pragma unroll 1
            for (int bi = 0; bi < 3; ++bi) {
                const float* bias1 = net + kPackBHidden + 256 * bi;
                bias_acc<false>(bias1, h, hid);
                dense128_split<kW>(ring, lane, x, hid);
                if (kStash && tile_ok) store_tl(p.stash + (1 + 2 * bi) * p.stash_stride, vtile, j, h, hid);
                bias_acc<true>(bias1 + 128, h, x);
                dense128_split<kW>(ring, lane, hid, x);
                // (per-view slot 6 = x3 is not written: nothing reads it - the backward of the view mean needs no activation, and for
                //  V = 1 it is fused slot 0)
                if (kStash && tile_ok && bi < 2) store_tl(p.stash + (2 + 2 * bi) * p.stash_stride, vtile, j, h, x);
                if (p.acts_view && valid) store_row(p.acts_view + (bi + 1) * vslot + 128 * vrow);
            }
            if (kMultiView) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) xsum[nb] = (v == 0) ? x[nb] : xsum[nb] + x[nb];
            }
        }
        if (kMultiView) {
            const float nv = (float)p.V;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) x[nb] = xsum[nb] / nv;
        }

        if (p.acts_fused && valid) store_row(p.acts_fused + 128 * g);       // complete_output: the view mean
        if (kStash && tile_ok) store_tl(p.stash_fused, tile, j, h, x);       // fused slot 0: the view mean
        // ---- 48 k-steps: fusion blocks ----
#pragma unroll 1
        for (int bi = 3; bi < 6; ++bi) {
            const float* bias1 = net + kPackBHidden + 256 * bi;
            bias_acc<false>(bias1, h, hid);
            dense128_split<kW>(ring, lane, x, hid);
            if (kStash && tile_ok) store_tl(p.stash_fused + (1 + 2 * (bi - 3)) * p.stash_fused_stride, tile, j, h, hid);
            bias_acc<true>(bias1 + 128, h, x);
            dense128_split<kW>(ring, lane, hid, x);
            if (kStash && tile_ok) store_tl(p.stash_fused + (2 + 2 * (bi - 3)) * p.stash_fused_stride, tile, j, h, x);
            if (p.acts_fused && valid) store_row(p.acts_fused + (long)(bi - 2) * p.total * 128 + 128 * g);
        }
        if (p.embedding && valid) store_row(p.embedding + 128 * g);

        // ---- 2 slots: read-out, 8 k-steps x 3 pieces of ONE output block (rows >= 4 zero) ----
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = (r < 4) ? net[kPackBr + r] : 0.0f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const f32x4* wb = ring_cur(ring) + lane;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int gq = 4 * half + s4;                               // k-step (kb = gq / 2, s = gq & 1)
                float b8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) b8[q] = x[gq >> 1][8 * (gq & 1) + q];
                BParts bq;
                split8<true>(b8, bq);
                u32x4 a[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) a[q] = s4 == 0 ? ring.a0[q] : __builtin_bit_cast(u32x4, wb[(3 * s4 + q) * 64]);
                o = mfma16(a[2], bq.p1, o);
                o = mfma16(a[1], bq.p2, o);
                o = mfma16(a[0], bq.p3, o);
                o = mfma16(a[1], bq.p1, o);
                o = mfma16(a[0], bq.p2, o);
                o = mfma16(a[0], bq.p1, o);
            }
            if (half == 1 && valid && h == 0) {
                f32x4 out;
                out[0] = sigmoid_f32(o[0]);
                out[1] = sigmoid_f32(o[1]);
                out[2] = sigmoid_f32(o[2]);
                out[3] = softplus_f32(o[3]);
                *reinterpret_cast<f32x4*>(p.rgbs + 4 * g) = out;
            }
            ring_fetch<kW>(ring);
            // the next position's first A operands (the invariant every k-step leaves behind)
            {
                const f32x4* nx = ring_nxt(ring) + lane;
#pragma unroll
                for (int q = 0; q < 3; ++q) ring.a0[q] = __builtin_bit_cast(u32x4, nx[q * 64]);
            }
            ring_next<kW>(ring);
        }
    }
#if MVS_STAMP
    if (lane == 0 && p.pix) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.pix) + 8 * (blockIdx.x * kW + wave);
        dbg[0] = ring.t_body;
        dbg[1] = ring.t_wait;
        dbg[2] = __builtin_readcyclecounter() - t_begin;
        dbg[3] = t_begin;
        for (int q = 0; q < 4; ++q) dbg[4 + q] = ring.t_grp[q];
    }
#endif
}

}  // namespace

// The packed_split buffer holds three weight streams: this kernel's (32x32x16 MFMA order, kSpChunks KiB) and, behind it, the two of the
// 16x16x32 kernels that run the inference passes (field_eval_split16.hip: three bf16 pieces; field_eval_split16h.hip: two fp16 pieces).
hipError_t launch_pack_net_split(const float* net_keras, void* packed_split, hipStream_t st) {
    const int n = kSpChunks * kChunkElems;
    hipLaunchKernelGGL(pack_net_split_kernel, dim3((n + 255) / 256), dim3(256), 0, st, net_keras, static_cast<__bf16*>(packed_split));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if ((e = launch_pack_net_split16(net_keras, static_cast<char*>(packed_split) + (size_t)kSpChunks * 1024, st)) != hipSuccess) return e;
    return launch_pack_net_split16h(net_keras, static_cast<char*>(packed_split) + (size_t)kSpChunks * 1024 + packed_net_split16_bytes(), st);
}

size_t packed_net_split_bytes() { return (size_t)kSpChunks * 1024 + packed_net_split16_bytes() + packed_net_split16h_bytes(); }

// Which kernel runs a split field pass.  MVNERF_SPLIT_MFMA (read per launch: tests flip it inside one process):
//   "32x32x16"  this file's kernel (round 2);   "bf16x6"  field_eval_split16.hip (exact three-piece bf16 cut, six products);
//   "f16x3"     field_eval_split16h.hip (two fp16 pieces, three products);   unset: mvnerf_set_split_kernel's value
//               (default f16x3).
enum SplitKernel { kSplit16F16 = 0, kSplit16Bf16 = 1, kSplit32 = 2 };            // = MVNERF_SPLIT_* of include/mvnerf_hip.h
static std::atomic<int> g_split_kernel{kSplit16F16};
int set_split_kernel(int which) {
    if (which < 0 || which > 2) return -1;
    return g_split_kernel.exchange(which);
}
static SplitKernel split_kernel_choice() {
    const char* s = getenv("MVNERF_SPLIT_MFMA");
    if (!s || !s[0]) return static_cast<SplitKernel>(g_split_kernel.load());
    return s[0] == '3' ? kSplit32 : (s[0] == 'f' ? kSplit16F16 : kSplit16Bf16);
}

hipError_t launch_field_eval_split(const FieldParams& p, const void* packed_split, hipStream_t stream) {
    const SplitKernel which = split_kernel_choice();
    const char* base16 = static_cast<const char*>(packed_split) + (size_t)kSpChunks * 1024;
    if (which == kSplit16F16 && field_eval_split16h_supports(p)) return launch_field_eval_split16h(p, base16 + packed_net_split16_bytes(), stream);
    if (which != kSplit32 && field_eval_split16_supports(p)) return launch_field_eval_split16(p, base16, stream);
    static std::mutex mtx;
    static bool attr_done[16] = {};
    static int cus[16] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    const int n_pos = ((p.texel_table ? 4 : kSpL0Steps) + kHiddenUnits) * p.V + kHiddenUnits + 2;
    if (n_pos > kMaxPositions) return hipErrorInvalidValue;                  // V <= 14 (direct) / 18 (texel table)
    const int lds_bytes = kRing * kSlotF4 * 16 + kWgWaves * 32 * kStageRowBytes + (kPackBr + 8 - kPackB0) * 4 + kMaxPositions * 4;   // sized for 8 waves
    {
        std::lock_guard<std::mutex> lock(mtx);
        if (!attr_done[dev]) {
            hipDeviceProp_t prop;
            if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
            cus[dev] = prop.multiProcessorCount;
            const void* fns[8] = {reinterpret_cast<const void*>(&field_eval_split_kernel<false, false, false>),
                                  reinterpret_cast<const void*>(&field_eval_split_kernel<false, true, false>),
                                  reinterpret_cast<const void*>(&field_eval_split_kernel<true, false, false>),
                                  reinterpret_cast<const void*>(&field_eval_split_kernel<true, true, false>),
                                  reinterpret_cast<const void*>(&field_eval_split_kernel<false, false, true>),
                                  reinterpret_cast<const void*>(&field_eval_split_kernel<false, true, true>),
                                  reinterpret_cast<const void*>(&field_eval_split_kernel<true, false, true>),
                                  reinterpret_cast<const void*>(&field_eval_split_kernel<true, true, true>)};
            for (const void* fn : fns)
                if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess) return e;
            attr_done[dev] = true;
        }
    }
    if ((e = launch_dir_bias(p, stream)) != hipSuccess) return e;
    const bool mv = p.V > 1;
    const int waves = (mv && MVS_MV_W4) ? 4 : kWgWaves;
    const long n_groups = (p.n_tiles + waves - 1) / waves;
    const long resident = (long)cus[dev];                                   // persistent: one workgroup per CU
    const unsigned wgs = (unsigned)(n_groups < resident ? n_groups : resident);
    const f32x4* w = static_cast<const f32x4*>(packed_split);
    if (p.stash && p.V > 1 && ((long)p.R * p.S) % 32 != 0) return hipErrorInvalidValue;     // tiles must not straddle scenes
    const dim3 grid(wgs), block(64 * waves);
#define MVS_LAUNCH(MV, PROJ, STASH) hipLaunchKernelGGL((field_eval_split_kernel<MV, PROJ, STASH>), grid, block, lds_bytes, stream, p, w)
    const int variant = (mv ? 4 : 0) + (p.texel_table ? 2 : 0) + (p.stash ? 1 : 0);
    switch (variant) {
        case 0: MVS_LAUNCH(false, false, false); break;
        case 1: MVS_LAUNCH(false, false, true); break;
        case 2: MVS_LAUNCH(false, true, false); break;
        case 3: MVS_LAUNCH(false, true, true); break;
        case 4: MVS_LAUNCH(true, false, false); break;
        case 5: MVS_LAUNCH(true, false, true); break;
        case 6: MVS_LAUNCH(true, true, false); break;
        default: MVS_LAUNCH(true, true, true); break;
    }
#undef MVS_LAUNCH
    return hipGetLastError();
}

}  // namespace mvnerf
