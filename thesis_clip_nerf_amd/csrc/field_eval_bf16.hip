// bf16 variant of the fused radiance-field kernel (BASELINE.json configs 3 and 5: bf16 weights and MFMA
// inputs, fp32 accumulate).  Same math and interfaces as field_eval.hip; what changes is the machine mapping:
//
//  * v_mfma_f32_32x32x16_bf16 runs 16x faster than the fp32 MFMA, so a wave can no longer stream its own
//    copy of the weights from L2 (that would need > 64 B/clk/CU of L1 bandwidth).  A workgroup of 4 waves
//    (4 tiles of 32 samples; two workgroups per CU, persistent over tile groups) shares them: the bf16 weight
//    stream is cut into 8 KiB segments (2 k-steps x 4 output blocks) that travel through a ring of 5 LDS buffers
//    by LDS-DMA (global_load_lds_dwordx4, no VGPR staging): while the waves run the MFMAs of segment i, the loads
//    of segments i+1..i+3 are in flight; one counted vmcnt + workgroup barrier per segment.  The waves of a
//    workgroup are in lock-step (they share the ring), so their gather / sin-cos phases leave the matrix pipe
//    idle; the second, independent workgroup on the CU fills those holes (MV16_WAVES=8 is the single 512-thread
//    workgroup form: half the weight traffic, no such overlap).
//  * activations stay fp32 in the accumulators (residual path, biases, read-out in fp32) and are rounded to
//    bf16 only when a register block is fed as the next MFMA's B operand (v_cvt_pk_bf16_f32); gathered
//    features are lerped in fp32, rounded once, and transposed through a wave-private bf16 LDS image.
//  * the per-(view, ray) layer-0 seed (b0 + W0_dir^T PE(dir)) is the fp32 one of dir_bias_kernel.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <mutex>

#include "mvnerf_kernels.h"
#include "mvnerf_math.h"
#include "mvnerf_mfma.h"

namespace mvnerf {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x8 = __attribute__((ext_vector_type(8))) float;

// ---- bf16 weight stream: 1 KiB chunks [lane][8 bf16], A operand of one (k-step of 16, 32-wide output block) ----
//   layer 0 : k-steps 0..3  = PE(cam xyz) (lower half-wave sin rows, upper cos rows) + rgb rows, 16 chunks
//             k-steps 4..19 = the 256 feature rows, 16 channels per k-step, 64 chunks
//   hidden l: k-step (kb, s): input feature 32kb + 16s + 8(j>>2) + 4h + (j&3) for element j of half h
//             (= accumulator registers 8s..8s+7 of block kb fed as B operand), 32 chunks per layer
//   read-out: 8 chunks, output rows >= 4 zero
constexpr int kW16ChunkElems = 512;
constexpr int kW16Hidden = 80, kW16Readout = 80 + 12 * 32;
constexpr int kW16Chunks = kW16Readout + 16;      // read-out: 8 chunks + 8 of padding so every segment is 16 chunks

__global__ void pack_net_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kW16Chunks * kW16ChunkElems) return;
    const int chunk = idx / kW16ChunkElems, lane = (idx % kW16ChunkElems) / 8, jj = idx % 8;
    const int i = lane & 31, h = lane >> 5;
    float val = 0.0f;
    if (chunk < kW16Hidden) {
        const int ks = chunk / 4, nb = chunk % 4;
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
    } else if (chunk < kW16Readout) {
        const int q = chunk - kW16Hidden, layer = q / 32, r = q % 32;
        const int kbs = r / 4, nb = r % 4;
        const int f = 32 * (kbs / 2) + 16 * (kbs % 2) + 8 * (jj >> 2) + 4 * h + (jj & 3);
        const int wsrc = kKerasBlocks + (layer / 2) * kKerasBlockStride + (layer % 2) * (kHidden * kHidden + kHidden);
        val = src[wsrc + f * kHidden + 32 * nb + i];
    } else if (chunk < kW16Readout + 8) {
        const int kbs = chunk - kW16Readout;
        const int f = 32 * (kbs / 2) + 16 * (kbs % 2) + 8 * (jj >> 2) + 4 * h + (jj & 3);
        if (i < 4) val = src[kKerasWr + f * 4 + i];
    }
    dst[idx] = (__bf16)val;
}

size_t packed_net_bf16_bytes() { return (size_t)kW16Chunks * 1024 + packed_net_bf16x_bytes(); }

hipError_t launch_pack_net_bf16(const float* net_keras, void* packed16, hipStream_t st) {
    const int n = kW16Chunks * kW16ChunkElems;
    hipLaunchKernelGGL(pack_net_bf16_kernel, dim3((n + 255) / 256), dim3(256), 0, st, net_keras, static_cast<__bf16*>(packed16));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_pack_net_bf16x(net_keras, static_cast<unsigned char*>(packed16) + (size_t)kW16Chunks * 1024, st);
}

// MVNERF_BF16_KERNEL=segments pins the round-2 segment-ring kernel for the texel-table form too (A/B runs, tests of both kernels)
static bool bf16x_enabled() {
    const char* e = getenv("MVNERF_BF16_KERNEL");                        // read per launch: a test flips it inside one process
    return !(e && e[0] == 's');
}

// ---- the segment ring ----------------------------------------------------------------------------------------
// A workgroup of kWgWaves waves shares the weight stream through a ring of kRing LDS slots of kSegChunks chunks
// (kKs k-steps x 4 output blocks), filled by LDS-DMA, 2 wave-instructions (2 KiB) per wave and segment.
// Per tile the waves consume, for every view, the layer-0 segments (PE + rgb; the feature rows unless they come from
// the texel table) and the 3 per-view blocks, then the 3 fusion blocks and the read-out.  Position p in [0, P) ->
// first chunk of the segment.  The weights are the same for every tile, so positions wrap modulo P.
#ifndef MV16_WAVES
#define MV16_WAVES 8       // 4: two independent 256-thread workgroups per CU (their gather / VALU phases overlap the
#endif                     //    other's MFMA phases); 8: one 512-thread workgroup per CU (half the weight traffic)
#ifndef MV16_PIPE_A
#define MV16_PIPE_A 1      // A operands of k-step ks+1 requested before the MFMAs of ks (0: compiler-scheduled)
#endif
#ifndef MV16_ABL_AREUSE
#define MV16_ABL_AREUSE 0  // timing-only: one LDS read per k-step instead of four (same A for all output blocks)
#endif
#ifndef MV16_ABL_CVT
#define MV16_ABL_CVT 0     // timing-only: no relu / bf16 conversion of the hidden-layer B operands
#endif
#ifndef MV16_ABL_DMA
#define MV16_ABL_DMA 0     // timing-only ablations (wrong results): no weight DMA after the prologue
#endif
#ifndef MV16_ABL_GATHER
#define MV16_ABL_GATHER 0  // no table / feature gather
#endif
#ifndef MV16_ABL_PE
#define MV16_ABL_PE 0      // no sin/cos
#endif
#ifndef MV16_ABL_BARRIER
#define MV16_ABL_BARRIER 0 // no workgroup barrier at segment ends
#endif
constexpr int kWgWaves = MV16_WAVES;
constexpr int kSegChunks = 2 * kWgWaves;                       // 16 or 8 chunks of 1 KiB
constexpr int kKs = kSegChunks / 4;                            // k-steps (of 16 rows) per segment: 4 or 2
#ifndef MV16_LDSDMA
#define MV16_LDSDMA 1      // 1: LDS-DMA ring of 5 slots (default); 0: register-staged double buffer (round 2 A/B: equal at cfg2, 14 % slower at V = 3)
#endif
#if MV16_LDSDMA
constexpr int kRing = 5, kAhead = 3, kSegF4 = kSegChunks * 64;  // float4 per slot
#else
constexpr int kRing = 2, kSegF4 = kSegChunks * 64;              // float4 per slot
#endif
constexpr int kHiddenUnits = 192 / kSegChunks;                 // segments of 3 blocks (6 Dense layers x 32 chunks)

struct Ring {
    const f32x4* w16;
    f32x4* base;        // LDS
    int c;              // ring slot of the current segment
    int p, P, V;
    int l0_units;       // layer-0 segments per view: PE + rgb only (texel table) or PE + rgb + 256 feature rows
    int tid, wave;
#if !MV16_LDSDMA
    f32x4 stg0, stg1;   // the next segment on its way to LDS (2 x 16 B per thread)
#endif
};

__device__ __forceinline__ int ring_start_chunk(int p, int V, int l0_units) {
    const int per_view = l0_units + kHiddenUnits;
    if (p < per_view * V) {
        const int q = p % per_view;
        return q < l0_units ? q * kSegChunks : kW16Hidden + (q - l0_units) * kSegChunks;
    }
    const int q = p - per_view * V;
    return q < kHiddenUnits ? kW16Hidden + 192 + q * kSegChunks : kW16Readout;
}

#if MV16_LDSDMA
// issue the LDS-DMA of position p + ahead into slot (c + ahead) % kRing: 2 x 16 B per thread, 1 KiB per wave-instruction
__device__ __forceinline__ void ring_issue(const Ring& r, int ahead) {
    int pp = r.p + ahead;
    if (pp >= r.P) pp -= r.P;
    const f32x4* src = r.w16 + (long)ring_start_chunk(pp, r.V, r.l0_units) * 64 + r.tid;
    f32x4* dst = r.base + ((r.c + ahead) % kRing) * kSegF4 + 64 * r.wave;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * kWgWaves),
                                     (__attribute__((address_space(3))) void*)(dst + 64 * kWgWaves), 16, 0, 0);
}

__device__ __forceinline__ const f32x4* ring_cur(const Ring& r) { return r.base + r.c * kSegF4; }

// end of a segment: the next segment's DMA (issued kAhead - 1 segments ago) must have landed for every wave.
// kDrain = false leaves the two younger segments (4 DMA instructions of this thread) in flight; segments that also
// issue ordinary loads or stores drain everything (their own waits are in-order with the DMA anyway).
template <bool kDrain>
__device__ __forceinline__ void ring_next(Ring& r) {
#if MV16_ABL_BARRIER
    if (kDrain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#else
    if (kDrain) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
#endif
    r.c = r.c + 1 == kRing ? 0 : r.c + 1;
    r.p = r.p + 1 == r.P ? 0 : r.p + 1;
    if (!MV16_ABL_DMA) ring_issue(r, kAhead);          // slot (c + 3) % 5 was last read two segments ago; everyone is past that barrier
}
#else
// Weight stream through registers (round 2 experiment, -DMV16_LDSDMA=0): every thread loads 2 x 16 B of the NEXT segment
// right after a barrier and stores them to the other LDS slot just before the following barrier (two slots instead of five,
// no counted vmcnt).  Measured equal to the LDS-DMA ring at cfg2 and 14 % slower at V = 3 / 480x640
// (profiles/r02_ab_bf16_staged_*.log): not the default.
__device__ __forceinline__ void ring_load(Ring& r, int pp) {                      // position pp -> staging registers
    if (pp >= r.P) pp -= r.P;
    const f32x4* src = r.w16 + (long)ring_start_chunk(pp, r.V, r.l0_units) * 64 + r.tid;
    r.stg0 = src[0];
    r.stg1 = src[64 * kWgWaves];
}

__device__ __forceinline__ const f32x4* ring_cur(const Ring& r) { return r.base + r.c * kSegF4; }

template <bool kDrain>
__device__ __forceinline__ void ring_next(Ring& r) {
    f32x4* dst = r.base + (r.c ^ 1) * kSegF4 + r.tid;                             // position p + 1, loaded during this segment
    if (!MV16_ABL_DMA) {
        dst[0] = r.stg0;
        dst[64 * kWgWaves] = r.stg1;
    }
#if MV16_ABL_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    r.c ^= 1;
    r.p = r.p + 1 == r.P ? 0 : r.p + 1;
    if (!MV16_ABL_DMA) ring_load(r, r.p + 1);
}
#endif

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- texel table on the bf16 MFMA (see field_eval.hip, project_texels_kernel, for the fp32 form) --------------
// T[texel] = W0[123:379]^T features[texel] with bf16 inputs and fp32 accumulation - the same rounding the direct bf16
// kernel applies to the gathered features, 16x less matrix time than the fp32 projection (which dominates the bf16
// path on many-texel scenes).  One workgroup per 32 texels: the rows are staged in LDS as bf16 (coalesced fp32 loads,
// XOR-swizzled 16-byte chunks), wave nb runs the 16 feature k-steps of output block nb on two alternating accumulators.
// kF16: the feature maps are stored as bf16 (NHWC, 512-byte texel rows - what encoders.FeatureProducer(out_dtype=bfloat16) emits): the rows
// go to LDS as they are (16-byte chunks of 8 channels), half the HBM bytes of the pass that bounds the bf16 path on many-texel scenes.
template <bool kF16>
__global__ __launch_bounds__(512) void project_texels_bf16_kernel(const float* __restrict__ features, const f32x4* __restrict__ w16a,
                                                                  const f32x4* __restrict__ w16b, long n_texels,
                                                                  float* __restrict__ table0, float* __restrict__ table1) {
    __shared__ __attribute__((aligned(16))) unsigned char srow[32 * 512];          // 32 texels x 256 channels x bf16
    using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = wv & 3;
    const f32x4* w16 = wv < 4 ? w16a : w16b;               // waves 4..7: the second net on the same staged rows
    float* table = wv < 4 ? table0 : table1;
    const long t0 = (long)blockIdx.x * 32;
    const f32x4* fsrc = reinterpret_cast<const f32x4*>(features);
    const int nthreads = blockDim.x;
    if (kF16) {
        for (int m = 0; m < 1024 / nthreads; ++m) {
            const int idx = tid + nthreads * m;             // 16-byte chunk index inside the 32 x 32 block of bf16 rows
            const int row = idx >> 5, chunk = idx & 31;     // channels 8 chunk .. 8 chunk + 7
            long t = t0 + row;
            if (t >= n_texels) t = n_texels - 1;
            *reinterpret_cast<f32x4*>(srow + row * 512 + ((chunk ^ (row & 15)) << 4)) = fsrc[t * 32 + chunk];
        }
    } else {
        for (int m = 0; m < 2048 / nthreads; ++m) {
            const int idx = tid + nthreads * m;             // float4 index inside the 32 x 64 block
            const int row = idx >> 6, c4 = idx & 63;        // channels 4 c4 .. 4 c4 + 3
            long t = t0 + row;
            if (t >= n_texels) t = n_texels - 1;
            const f32x4 v = fsrc[t * 64 + c4];
            const int chunk = c4 >> 1;                      // 16-byte chunk of 8 channels
            *reinterpret_cast<bf16x4*>(srow + row * 512 + ((chunk ^ (row & 15)) << 4) + ((c4 & 1) << 3)) = __builtin_convertvector(v, bf16x4);
        }
    }
    __syncthreads();
    const f32x4* w = w16 + ((long)16 + nb) * 64 + lane;    // chunk (k-step 4 + ks, nb) = 4 (4 + ks) + nb
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.0f;
        acc1[r] = 0.0f;
    }
#pragma unroll 4
    for (int ks = 0; ks < 16; ks += 2) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, w[(long)ks * 256]), a1 = __builtin_bit_cast(bf16x8, w[(long)(ks + 1) * 256]);
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(srow + j * 512 + (((2 * ks + h) ^ (j & 15)) << 4));
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(srow + j * 512 + (((2 * ks + 2 + h) ^ (j & 15)) << 4));
        acc0 = mfma16(a0, b0, acc0);
        acc1 = mfma16(a1, b1, acc1);
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

hipError_t launch_project_texels_bf16(const float* features, const void* packed16, const void* packed16b, long n_texels, float* table,
                                      float* table1, hipStream_t stream) {
    hipLaunchKernelGGL(project_texels_bf16_kernel<false>, dim3((unsigned)((n_texels + 31) / 32)), dim3(packed16b ? 512 : 256), 0, stream,
                       features, static_cast<const f32x4*>(packed16), static_cast<const f32x4*>(packed16b), n_texels, table, table1);
    return hipGetLastError();
}

hipError_t launch_project_texels_bf16maps(const void* features_bf16, const void* packed16, const void* packed16b, long n_texels, float* table,
                                          float* table1, hipStream_t stream) {
    hipLaunchKernelGGL(project_texels_bf16_kernel<true>, dim3((unsigned)((n_texels + 31) / 32)), dim3(packed16b ? 512 : 256), 0, stream,
                       static_cast<const float*>(features_bf16), static_cast<const f32x4*>(packed16), static_cast<const f32x4*>(packed16b), n_texels,
                       table, table1);
    return hipGetLastError();
}

// one k-step (16 input rows) x 4 output blocks out of the LDS-resident segment
__device__ __forceinline__ void step16(const f32x4* wbuf, int ks_local, int lane, bf16x8 b, f32x16 (&acc)[4]) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, wbuf[(ks_local * 4 + nb) * 64 + lane]);
        acc[nb] = mfma16(a, b, acc[nb]);
    }
}

__device__ __forceinline__ bf16x8 relu_to_bf16(const f32x16& v, int s) {
    // bf16(relu(x)) == relu(bf16(x)) (rounding is sign-symmetric), and on the bf16 bit pattern relu is a signed
    // 16-bit max with 0: 4 v_cvt_pk_bf16_f32 + 4 v_pk_max_i16 per 8 values instead of 16 v_max_f32 + 4 converts
    using i32x4 = __attribute__((ext_vector_type(4))) int;
    f32x8 t;
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = v[8 * s + q];
    i32x4 r = __builtin_bit_cast(i32x4, __builtin_convertvector(t, bf16x8));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int w = r[q];
        asm("v_pk_max_i16 %0, %1, 0" : "=v"(w) : "v"(w));
        r[q] = w;
    }
    return __builtin_bit_cast(bf16x8, r);
}

// acc += W^T relu(in) for one hidden layer = two 16-chunk segments (input blocks 0,1 then 2,3)
// The A operands of k-step ks+1 are requested from LDS before the MFMAs of k-step ks are issued (pinned with a
// sched_barrier: left alone the compiler keeps two A register sets and waits for an LDS round trip per MFMA pair),
// and the relu / bf16 conversion of the next B operand runs in the shadow of those MFMAs.
// bfn(ks) -> B operand (bf16x8) of k-step ks of the segment.
template <typename BFn>
__device__ __forceinline__ void segment_mfma(Ring& ring, int lane, BFn bfn, f32x16 (&acc)[4]) {
    const f32x4* wb = ring_cur(ring) + lane;
#if !MV16_PIPE_A
#pragma unroll
    for (int ks = 0; ks < kKs; ++ks) {
        const bf16x8 bq = bfn(ks);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = mfma16(__builtin_bit_cast(bf16x8, wb[(ks * 4 + nb) * 64]), bq, acc[nb]);
    }
    return;
#endif
    f32x4 a[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) a[nb] = wb[(MV16_ABL_AREUSE ? 0 : nb) * 64];
    bf16x8 b = bfn(0);
#pragma unroll
    for (int ks = 0; ks < kKs; ++ks) {
        f32x4 an[4];
        if (ks < kKs - 1) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) an[nb] = wb[((ks + 1) * 4 + (MV16_ABL_AREUSE ? 0 : nb)) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = mfma16(__builtin_bit_cast(bf16x8, a[nb]), b, acc[nb]);
        if (ks < kKs - 1) {
            b = bfn(ks + 1);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) a[nb] = an[nb];
        }
    }
}

// acc += W^T relu(in) for one hidden layer: 8 k-steps (input block kb = ks / 2, half ks % 2) in 8 / kKs segments
__device__ __forceinline__ void dense128_bf16(Ring& ring, int lane, const f32x16 (&in)[4], f32x16 (&acc)[4]) {
#pragma unroll
    for (int seg = 0; seg < 8 / kKs; ++seg) {
        segment_mfma(ring, lane, [&](int ks) {
            const int g = seg * kKs + ks;
            if (MV16_ABL_CVT) {
                const f32x4 raw = {in[g >> 1][8 * (g & 1)], in[g >> 1][8 * (g & 1) + 1], in[g >> 1][8 * (g & 1) + 2], in[g >> 1][8 * (g & 1) + 3]};
                return __builtin_bit_cast(bf16x8, raw);
            }
            return relu_to_bf16(in[g >> 1], g & 1);
        }, acc);
        ring_next<false>(ring);
    }
}

template <bool kAdd>
__device__ __forceinline__ void bias_acc(const float* __restrict__ bperm, int h, f32x16 (&acc)[4]) {
    const f32x4* p = reinterpret_cast<const f32x4*>(bperm + h * 64);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = p[nb * 4 + q];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (kAdd) {
                    // scalar adds on purpose: left to the compiler these become v_pk_add_f32, which costs the matrix pipe
                    // of the partner wave far more than two v_add_f32 (MI355X_MICROARCH.md, price of fillers beside MFMAs)
                    float r = acc[nb][4 * q + c];
                    asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(v[c]));
                    acc[nb][4 * q + c] = r;
                } else {
                    acc[nb][4 * q + c] = v[c];
                }
            }
        }
}

constexpr int kStage16Row = 256;      // bytes per staged sample row: 128 channels x bf16

// kProj: layer 0's feature rows come from the fp32 texel table (field_eval.hip, project_texels_kernel): a 128-channel
// lerp of table rows added to the accumulators replaces the four feature segments (64 MFMAs per tile) and halves the
// gather.  All gathers run in batches of 4 iterations with their 16 loads issued up front: the 8 waves of the
// workgroup share the weight ring, hence gather at the same time, and nothing else hides that latency.
// kF16 (direct gather only): p.features holds bf16 feature maps (512-byte texel rows); the taps are widened to fp32 (exact), lerped
// in fp32 and rounded once, as with fp32 maps.
template <bool kMultiView, bool kProj, bool kF16>
__global__ __launch_bounds__(64 * kWgWaves, 2) void field_eval_bf16_kernel(FieldParams p, const f32x4* __restrict__ w16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    constexpr int kRingBytes = kRing * kSegF4 * 16;                       // 40 or 80 KiB
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* stage = smem16 + kRingBytes + wave * (32 * kStage16Row);   // 8 KiB per wave
    // all biases (accumulator order, fp32) live in LDS for the whole kernel: a global bias load in the middle of a
    // segment would make the in-order vmcnt wait drain the weight prefetch issued before it
    float* net = reinterpret_cast<float*>(smem16 + kRingBytes + kWgWaves * 32 * kStage16Row) - kPackB0;   // net[kPackB0 + i] -> LDS
    for (int i = tid; i < kPackBr + 8 - kPackB0; i += 64 * kWgWaves) net[kPackB0 + i] = p.net[kPackB0 + i];

    Ring ring;
    ring.w16 = w16;
    ring.base = reinterpret_cast<f32x4*>(smem16);
    ring.c = 0;
    ring.p = 0;
    ring.V = p.V;
    ring.l0_units = (kProj ? 16 : 80) / kSegChunks;
    ring.P = (ring.l0_units + kHiddenUnits) * p.V + kHiddenUnits + 1;
    ring.tid = tid;
    ring.wave = wave;
#if MV16_LDSDMA
    ring_issue(ring, 0);
    ring_issue(ring, 1);
    ring_issue(ring, 2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ring_issue(ring, kAhead);
#else
    ring_load(ring, 0);
    ring.base[tid] = ring.stg0;                                            // position 0 -> slot 0
    ring.base[tid + 64 * kWgWaves] = ring.stg1;
    ring_load(ring, 1);
    __syncthreads();
#endif

    const long n_groups = (p.n_tiles + kWgWaves - 1) / kWgWaves;
    for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        long tile = grp * kWgWaves + wave;
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
            if (valid && h == 0 && p.tap_idx) {
                int4 t4 = make_int4(tl, tl + 1, tl + p.W, tl + p.W + 1);
                *reinterpret_cast<int4*>(p.tap_idx + 4 * vrow) = t4;
            }

            // ---- segment: PE(cam xyz) + rgb k-steps ----
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
                    if (MV16_ABL_PE) {
                        sk = a0;
                        ck = a0 + 1.0f;
                    } else if (k == 0 || k == 5) {
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
#pragma unroll
            for (int seg = 0; seg < 4 / kKs; ++seg) {
                segment_mfma(ring, lane, [&](int ks) {
                    f32x8 t;
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = pe[8 * (seg * kKs + ks) + q];
                    return __builtin_convertvector(t, bf16x8);
                }, x);
                ring_next<true>(ring);
            }

            if (kProj) {
                // ---- texel table: two passes of 64 channels (accumulator blocks nb = 2P, 2P+1 of both lane halves) ----
                // a table row is [h][nb][16]; pass P takes floats h*64 + P*32 + {0..31}: two 128-B pieces per row.
                // 16 lanes per sample row, 4 rows per load instruction, 4 taps, batches of 4 instructions per tap.
                const int l16 = lane & 15, sub = lane >> 4;
                const f32x4* tbase = reinterpret_cast<const f32x4*>(p.texel_table) + (l16 >> 3) * 16 + (l16 & 7);
#pragma unroll
                for (int P = 0; P < 2; ++P) {                       // unrolled: x[2P + nbl] must be a static register index
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
                    for (int it0 = 0; it0 < (MV16_ABL_GATHER ? 0 : 8); it0 += 4) {
                        f32x4 tv[4][4];
                        float axs[4], ays[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int src = 4 * (it0 + u) + sub;
                            const int tls = __shfl(tl, src);
                            axs[u] = __shfl(tp.ax, src);
                            ays[u] = __shfl(tp.ay, src);
                            const f32x4* f = tbase + (long)tls * 32 + P * 8;
                            tv[u][0] = f[0];
                            tv[u][1] = f[32];
                            tv[u][2] = f[(long)p.W * 32];
                            tv[u][3] = f[(long)p.W * 32 + 32];
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
                            *reinterpret_cast<f32x4*>(stage + src * kStage16Row + ((l16 ^ (src & 15)) << 4)) = o;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 t4 = *reinterpret_cast<const f32x4*>(stage + j * kStage16Row + (((8 * h + 4 * nbl + q) ^ (j & 15)) << 4));
#pragma unroll
                            for (int c = 0; c < 4; ++c) x[2 * P + nbl][4 * q + c] += t4[c];
                        }
                }
            }
            // ---- 4 segments: the two 128-channel halves of the gathered features, 4 k-steps per segment ----
#pragma unroll 1
            for (int hf = 0; hf < (kProj ? 0 : 2); ++hf) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const f32x4* fbase = reinterpret_cast<const f32x4*>(p.features) + hf * 32 + j;
                using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
                using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
                const u32x2* fbase16 = reinterpret_cast<const u32x2*>(p.features) + hf * 32 + j;      // 8 bytes = channels 4j .. 4j + 3 of the half row
                auto widen = [](u32x2 q) {                                  // four bf16 -> fp32, exact
                    f32x4 r;
                    r[0] = __builtin_bit_cast(float, q[0] << 16);
                    r[1] = __builtin_bit_cast(float, q[0] & 0xffff0000u);
                    r[2] = __builtin_bit_cast(float, q[1] << 16);
                    r[3] = __builtin_bit_cast(float, q[1] & 0xffff0000u);
                    return r;
                };
#pragma unroll 1
                for (int it0 = 0; it0 < (MV16_ABL_GATHER ? 0 : 16); it0 += 4) {
                    f32x4 tv[4][4];
                    float axs[4], ays[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int src = 2 * (it0 + u) + h;
                        const int tls = __shfl(tl, src);
                        axs[u] = __shfl(tp.ax, src);
                        ays[u] = __shfl(tp.ay, src);
                        if (kF16) {
                            const u32x2* f = fbase16 + (long)tls * 64;
                            tv[u][0] = widen(f[0]);
                            tv[u][1] = widen(f[64]);
                            tv[u][2] = widen(f[(long)p.W * 64]);
                            tv[u][3] = widen(f[(long)p.W * 64 + 64]);
                        } else {
                            const f32x4* f = fbase + (long)tls * 64;
                            tv[u][0] = f[0];
                            tv[u][1] = f[64];
                            tv[u][2] = f[(long)p.W * 64];
                            tv[u][3] = f[(long)p.W * 64 + 64];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int src = 2 * (it0 + u) + h;
                        f32x4 o;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float top = fmaf(axs[u], tv[u][1][c] - tv[u][0][c], tv[u][0][c]);
                            const float bot = fmaf(axs[u], tv[u][3][c] - tv[u][2][c], tv[u][2][c]);
                            o[c] = fmaf(ays[u], bot - top, top);
                        }
                        // row `src`, channels 4j..4j+3 (8 bytes); 16-byte chunks XOR-swizzled by the row
                        const int off = src * kStage16Row + (((j >> 1) ^ (src & 15)) << 4) + ((j & 1) << 3);
                        *reinterpret_cast<bf16x4*>(stage + off) = __builtin_convertvector(o, bf16x4);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int part = 0; part < 8 / kKs; ++part) {            // 8 k-steps of 16 channels per 128-channel half
                    segment_mfma(ring, lane, [&](int ks) {
                        const int off = j * kStage16Row + (((2 * (kKs * part + ks) + h) ^ (j & 15)) << 4);
                        return *reinterpret_cast<const bf16x8*>(stage + off);
                    }, x);
                    ring_next<true>(ring);
                }
            }

            // ---- 12 segments: the three per-view ResNet blocks ----
#pragma unroll 1
            for (int bi = 0; bi < 3; ++bi) {
                const float* bias1 = net + kPackBHidden + 256 * bi;
                bias_acc<false>(bias1, h, hid);
                dense128_bf16(ring, lane, x, hid);
                bias_acc<true>(bias1 + 128, h, x);
                dense128_bf16(ring, lane, hid, x);
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

        // one sample row (128 floats) from the accumulators: lane (j,h) holds features 32nb + 8q + 4h + {0..3}
        auto store_row = [&](float* base) {
            float* e = base + 128 * g + 4 * h;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v4 = {x[nb][4 * q], x[nb][4 * q + 1], x[nb][4 * q + 2], x[nb][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(e + 32 * nb + 8 * q) = v4;
                }
        };
        if (p.acts_fused && valid) store_row(p.acts_fused);                // complete_output: the view mean
        // ---- 12 segments: fusion blocks ----
#pragma unroll 1
        for (int bi = 3; bi < 6; ++bi) {
            const float* bias1 = net + kPackBHidden + 256 * bi;
            bias_acc<false>(bias1, h, hid);
            dense128_bf16(ring, lane, x, hid);
            bias_acc<true>(bias1 + 128, h, x);
            dense128_bf16(ring, lane, hid, x);
            if (p.acts_fused && valid) store_row(p.acts_fused + (long)(bi - 2) * p.total * 128);
        }
        if (p.embedding && valid) store_row(p.embedding);

        // ---- segment: read-out ----
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = (r < 4) ? net[kPackBr + r] : 0.0f;
        {
            const f32x4* wb = ring_cur(ring);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, wb[(kb * 2 + s) * 64 + lane]);
                    o = mfma16(a, relu_to_bf16(x[kb], s), o);
                }
        }
        if (valid && h == 0) {
            f32x4 out;
            out[0] = sigmoid_f32(o[0]);
            out[1] = sigmoid_f32(o[1]);
            out[2] = sigmoid_f32(o[2]);
            out[3] = softplus_f32(o[3]);
            *reinterpret_cast<f32x4*>(p.rgbs + 4 * g) = out;
        }
        ring_next<true>(ring);                                           // stores above: drain
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // DMA still in flight must land before the LDS is released
}

hipError_t launch_field_eval_bf16(const FieldParams& p, const void* packed16, hipStream_t stream, bool maps_bf16) {
    if (bf16x_enabled() && field_eval_bf16x_supports(p))
        return launch_field_eval_bf16x(p, static_cast<const unsigned char*>(packed16) + (size_t)kW16Chunks * 1024, stream);
    static std::mutex mtx;
    static bool attr_done[16] = {};
    static int cus[16] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    const int lds_bytes = kRing * kSegF4 * 16 + kWgWaves * 32 * kStage16Row + (kPackBr + 8 - kPackB0) * 4;
    {
        std::lock_guard<std::mutex> lock(mtx);
        if (!attr_done[dev]) {
            hipDeviceProp_t prop;
            if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
            cus[dev] = prop.multiProcessorCount;
            const void* fns[6] = {reinterpret_cast<const void*>(&field_eval_bf16_kernel<false, false, false>),
                                  reinterpret_cast<const void*>(&field_eval_bf16_kernel<false, true, false>),
                                  reinterpret_cast<const void*>(&field_eval_bf16_kernel<true, false, false>),
                                  reinterpret_cast<const void*>(&field_eval_bf16_kernel<true, true, false>),
                                  reinterpret_cast<const void*>(&field_eval_bf16_kernel<false, false, true>),
                                  reinterpret_cast<const void*>(&field_eval_bf16_kernel<true, false, true>)};
            for (const void* fn : fns)
                if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess) return e;
            attr_done[dev] = true;
        }
    }
    if ((e = launch_dir_bias(p, stream)) != hipSuccess) return e;
    const long n_groups = (p.n_tiles + kWgWaves - 1) / kWgWaves;
    const long resident = (long)cus[dev] * (8 / kWgWaves);                 // persistent: as many workgroups as fit at once
    const unsigned wgs = (unsigned)(n_groups < resident ? n_groups : resident);
    const f32x4* w16 = static_cast<const f32x4*>(packed16);
#define MV16_GO(MV, PROJ, F16) hipLaunchKernelGGL((field_eval_bf16_kernel<MV, PROJ, F16>), dim3(wgs), dim3(64 * kWgWaves), lds_bytes, stream, p, w16)
    if (p.V > 1) {
        if (p.texel_table) MV16_GO(true, true, false);                     // with the table the feature maps are not read at all
        else if (maps_bf16) MV16_GO(true, false, true);
        else MV16_GO(true, false, false);
    } else {
        if (p.texel_table) MV16_GO(false, true, false);
        else if (maps_bf16) MV16_GO(false, false, true);
        else MV16_GO(false, false, false);
    }
#undef MV16_GO
    return hipGetLastError();
}

}  // namespace mvnerf
