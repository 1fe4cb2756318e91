// bf16 variant of the fused radiance-field kernel (BASELINE.json configs 3 and 5: bf16 weights and MFMA
// inputs, fp32 accumulate).  Same math and interfaces as field_eval.hip; what changes is the machine mapping:
//
//  * v_mfma_f32_32x32x16_bf16 runs 16x faster than the fp32 MFMA, so a wave can no longer stream its own
//    copy of the weights from L2 (that would need > 64 B/clk/CU of L1 bandwidth).  A 512-thread workgroup
//    (8 waves = 8 tiles of 32 samples, one workgroup per CU, persistent over tile groups) shares them:
//    the bf16 weight stream is cut into segments of <= 32 KiB that are double-buffered in LDS; while the waves
//    run the MFMAs of segment i out of one buffer, all 512 threads have the global loads of segment i+1 in
//    flight and drop them into the other buffer, one workgroup barrier per segment (16 per tile group).
//  * activations stay fp32 in the accumulators (residual path, biases, read-out in fp32) and are rounded to
//    bf16 only when a register block is fed as the next MFMA's B operand (v_cvt_pk_bf16_f32); gathered
//    features are lerped in fp32, rounded once, and transposed through a wave-private bf16 LDS image.
//  * the per-(view, ray) layer-0 seed (b0 + W0_dir^T PE(dir)) is the fp32 one of dir_bias_kernel.
#include <hip/hip_runtime.h>

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
constexpr int kW16L0 = 0, kW16Hidden = 80, kW16Readout = 80 + 12 * 32, kW16Chunks = kW16Readout + 8;

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
    } else {
        const int kbs = chunk - kW16Readout;
        const int f = 32 * (kbs / 2) + 16 * (kbs % 2) + 8 * (jj >> 2) + 4 * h + (jj & 3);
        if (i < 4) val = src[kKerasWr + f * 4 + i];
    }
    dst[idx] = (__bf16)val;
}

hipError_t launch_pack_net_bf16(const float* net_keras, void* packed16, hipStream_t st) {
    const int n = kW16Chunks * kW16ChunkElems;
    hipLaunchKernelGGL(pack_net_bf16_kernel, dim3((n + 255) / 256), dim3(256), 0, st, net_keras, static_cast<__bf16*>(packed16));
    return hipGetLastError();
}

// ---- segments of the stream (start chunk, chunks); order per tile: per view 0..8, then 9..15 ----
__device__ __forceinline__ int seg_start(int s) { return s == 0 ? 0 : (s == 1 ? 16 : (s == 2 ? 48 : (s < 15 ? kW16Hidden + 32 * (s - 3) : kW16Readout))); }
__device__ __forceinline__ int seg_chunks(int s) { return s == 0 ? 16 : (s == 15 ? 8 : 32); }

struct SegRegs {
    f32x4 r[4];
};

__device__ __forceinline__ void seg_prefetch(SegRegs& sr, const f32x4* __restrict__ w16, int seg, int tid) {
    const f32x4* src = w16 + (long)seg_start(seg) * 64 + tid;
    const int per_thread = seg_chunks(seg) / 8;             // 1, 2 or 4 float4 per thread (512 threads)
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (m < per_thread) sr.r[m] = src[512 * m];
}

__device__ __forceinline__ void seg_commit(const SegRegs& sr, f32x4* buf, int seg, int tid) {
    const int per_thread = seg_chunks(seg) / 8;
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (m < per_thread) buf[tid + 512 * m] = sr.r[m];
    __syncthreads();                                        // segment visible; previous buffer free for the next one
}

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
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
    f32x8 t;
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = fmaxf(v[8 * s + q], 0.0f);
    return __builtin_convertvector(t, bf16x8);
}

// acc += W^T relu(in) for one hidden layer whose 32 chunks sit in wbuf
__device__ __forceinline__ void dense128_bf16(const f32x4* wbuf, int lane, const f32x16 (&in)[4], f32x16 (&acc)[4]) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s) step16(wbuf, kb * 2 + s, lane, relu_to_bf16(in[kb], s), acc);
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
            for (int c = 0; c < 4; ++c) acc[nb][4 * q + c] = kAdd ? acc[nb][4 * q + c] + v[c] : v[c];
        }
}

constexpr int kStage16Row = 256;      // bytes per staged sample row: 128 channels x bf16

template <bool kMultiView>
__global__ __launch_bounds__(512, 2) void field_eval_bf16_kernel(FieldParams p, const f32x4* __restrict__ w16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    f32x4* wbuf0 = reinterpret_cast<f32x4*>(smem16);                    // 2 x 32 KiB weight segments
    f32x4* wbuf1 = wbuf0 + 2048;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* stage = smem16 + 65536 + wave * (32 * kStage16Row);   // 8 KiB per wave
    // all biases (accumulator order, fp32) live in LDS for the whole kernel: a global bias load in the middle of a
    // segment would make the in-order vmcnt wait drain the weight prefetch issued just before it
    float* net = reinterpret_cast<float*>(smem16 + 65536 + 8 * 32 * kStage16Row) - kPackB0;   // net[kPackB0 + i] -> LDS
    for (int i = tid; i < kPackBr + 8 - kPackB0; i += 512) net[kPackB0 + i] = p.net[kPackB0 + i];

    SegRegs sr;
    seg_prefetch(sr, w16, 0, tid);
    seg_commit(sr, wbuf0, 0, tid);
    int cur = 0;                                                         // buffer holding the current segment

    const long n_groups = (p.n_tiles + 7) / 8;
    for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        long tile = grp * 8 + wave;
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

            // ---- segment 0: PE(cam xyz) + rgb k-steps ----
            const int nxt0 = 1;
            bias_acc<false>(p.dir_bias + 128 * ((long)bv * p.R + (ray - b * p.R)), h, x);   // global: before the prefetch
            float rgbv[3];
            {
                const float* img = p.images + 3 * (long)tl;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float a = img[c] * 2.0f - 1.0f, bq = img[3 + c] * 2.0f - 1.0f;
                    const float cq = img[3 * p.W + c] * 2.0f - 1.0f, dq = img[3 * p.W + 3 + c] * 2.0f - 1.0f;
                    rgbv[c] = bilerp(a, bq, cq, dq, tp.ax, tp.ay);
                }
            }
            seg_prefetch(sr, w16, nxt0, tid);
            float pe[32];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float a0 = cam[d] * 3.14159274101257324f;
                float sk = 0.0f, ck = 0.0f;
#pragma unroll
                for (int k = 0; k < kNFreq; ++k) {
                    if (k == 0 || k == 5) {
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
            pe[30] = h ? rgbv[1] : rgbv[0];
            pe[31] = h ? 0.0f : rgbv[2];
            {
                const f32x4* wb = cur ? wbuf1 : wbuf0;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    f32x8 t;
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = pe[8 * ks + q];
                    step16(wb, ks, lane, __builtin_convertvector(t, bf16x8), x);
                }
            }
            seg_commit(sr, cur ? wbuf0 : wbuf1, nxt0, tid);
            cur ^= 1;

            // ---- segments 1, 2: the two 128-channel halves of the gathered features ----
#pragma unroll 1
            for (int hf = 0; hf < 2; ++hf) {
                const int nxt = hf == 0 ? 2 : 3;
                seg_prefetch(sr, w16, nxt, tid);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const f32x4* fbase = reinterpret_cast<const f32x4*>(p.features) + hf * 32 + j;
#pragma unroll 4
                for (int it = 0; it < 16; ++it) {
                    const int src = 2 * it + h;
                    const int tls = __shfl(tl, src);
                    const float axs = __shfl(tp.ax, src), ays = __shfl(tp.ay, src);
                    const f32x4* f = fbase + (long)tls * 64;
                    const f32x4 vtl = f[0], vtr = f[64], vbl = f[(long)p.W * 64], vbr = f[(long)p.W * 64 + 64];
                    using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
                    f32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float top = fmaf(axs, vtr[c] - vtl[c], vtl[c]);
                        const float bot = fmaf(axs, vbr[c] - vbl[c], vbl[c]);
                        o[c] = fmaf(ays, bot - top, top);
                    }
                    // row `src`, channels 4j..4j+3 (8 bytes); 16-byte chunks XOR-swizzled by the row
                    const int off = src * kStage16Row + (((j >> 1) ^ (src & 15)) << 4) + ((j & 1) << 3);
                    *reinterpret_cast<bf16x4*>(stage + off) = __builtin_convertvector(o, bf16x4);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const f32x4* wb = cur ? wbuf1 : wbuf0;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int off = j * kStage16Row + (((2 * ks + h) ^ (j & 15)) << 4);
                    const bf16x8 bq = *reinterpret_cast<const bf16x8*>(stage + off);
                    step16(wb, ks, lane, bq, x);
                }
                seg_commit(sr, cur ? wbuf0 : wbuf1, nxt, tid);
                cur ^= 1;
            }

            // ---- segments 3..8: the three per-view ResNet blocks ----
#pragma unroll 1
            for (int bi = 0; bi < 3; ++bi) {
                const float* bias1 = net + kPackBHidden + 256 * bi;
                int nxt = 3 + 2 * bi + 1;
                seg_prefetch(sr, w16, nxt, tid);
                bias_acc<false>(bias1, h, hid);
                dense128_bf16(cur ? wbuf1 : wbuf0, lane, x, hid);
                seg_commit(sr, cur ? wbuf0 : wbuf1, nxt, tid);
                cur ^= 1;
                nxt = 3 + 2 * bi + 2;
                if (nxt == 9 && v + 1 < p.V) nxt = 0;                    // next view restarts at layer 0
                seg_prefetch(sr, w16, nxt, tid);
                bias_acc<true>(bias1 + 128, h, x);
                dense128_bf16(cur ? wbuf1 : wbuf0, lane, hid, x);
                seg_commit(sr, cur ? wbuf0 : wbuf1, nxt, tid);
                cur ^= 1;
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

        // ---- segments 9..14: fusion blocks ----
#pragma unroll 1
        for (int bi = 3; bi < 6; ++bi) {
            const float* bias1 = net + kPackBHidden + 256 * bi;
            int nxt = 3 + 2 * bi + 1;
            seg_prefetch(sr, w16, nxt, tid);
            bias_acc<false>(bias1, h, hid);
            dense128_bf16(cur ? wbuf1 : wbuf0, lane, x, hid);
            seg_commit(sr, cur ? wbuf0 : wbuf1, nxt, tid);
            cur ^= 1;
            nxt = 3 + 2 * bi + 2;
            seg_prefetch(sr, w16, nxt, tid);
            bias_acc<true>(bias1 + 128, h, x);
            dense128_bf16(cur ? wbuf1 : wbuf0, lane, hid, x);
            seg_commit(sr, cur ? wbuf0 : wbuf1, nxt, tid);
            cur ^= 1;
        }
        if (p.embedding && valid) {
            float* e = p.embedding + 128 * g + 4 * h;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v4 = {x[nb][4 * q], x[nb][4 * q + 1], x[nb][4 * q + 2], x[nb][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(e + 32 * nb + 8 * q) = v4;
                }
        }

        // ---- segment 15: read-out ----
        seg_prefetch(sr, w16, 0, tid);                                   // first segment of the next tile group
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = (r < 4) ? net[kPackBr + r] : 0.0f;
        {
            const f32x4* wb = cur ? wbuf1 : wbuf0;
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
        seg_commit(sr, cur ? wbuf0 : wbuf1, 0, tid);
        cur ^= 1;
    }
}

hipError_t launch_field_eval_bf16(const FieldParams& p, const void* packed16, hipStream_t stream) {
    static std::mutex mtx;
    static bool attr_done[16] = {};
    static int cus[16] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    const int lds_bytes = 65536 + 8 * 32 * kStage16Row + (kPackBr + 8 - kPackB0) * 4;
    {
        std::lock_guard<std::mutex> lock(mtx);
        if (!attr_done[dev]) {
            hipDeviceProp_t prop;
            if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
            cus[dev] = prop.multiProcessorCount;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&field_eval_bf16_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess) return e;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&field_eval_bf16_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess) return e;
            attr_done[dev] = true;
        }
    }
    if ((e = launch_dir_bias(p, stream)) != hipSuccess) return e;
    const long n_groups = (p.n_tiles + 7) / 8;
    const unsigned wgs = (unsigned)(n_groups < cus[dev] ? n_groups : cus[dev]);
    const f32x4* w16 = static_cast<const f32x4*>(packed16);
    if (p.V > 1)
        hipLaunchKernelGGL(field_eval_bf16_kernel<true>, dim3(wgs), dim3(512), lds_bytes, stream, p, w16);
    else
        hipLaunchKernelGGL(field_eval_bf16_kernel<false>, dim3(wgs), dim3(512), lds_bytes, stream, p, w16);
    return hipGetLastError();
}

}  // namespace mvnerf
