// Probe 3 (round 3): which MFMA shape / wave mapping sustains the most split-bf16 (6 products per fp32 product) work under the
// chip's power management, on RANDOM, CHANGING operands (round 2's probe 2 cycled one 12 KiB slot: constant weights run at a
// higher clock).  Every variant: one workgroup per CU, weights through a 3-slot LDS ring fed from a random
// 1.4 MB stream (register staged), one barrier per k-step, the next k-step's B operand cut (relu + 3 bf16 pieces) in the MFMA
// shadow, 6 MFMAs per (A block, B block).
//   S32 x T : v_mfma_f32_32x32x16_bf16, K = 16 per k-step, T tiles of 32 samples per wave (T = 1: 8 waves, T = 2: 4 waves x 512 regs)
//   S16 x C : v_mfma_f32_16x16x32_bf16, K = 32 per k-step, C column blocks of 16 samples per wave (C = 2: 8 waves, C = 4: 4 waves)
//   hipcc -O3 --offload-arch=gfx950 scripts/x6_probe3.hip -o build/x6_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
#define BF(v) __builtin_bit_cast(bf16x8, v)
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a), BF(b), c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a), BF(b), c, 0, 0, 0); }

struct BParts { u32x4 p1, p2, p3; };
__device__ __forceinline__ void split_pair(float v0, float v1, int q, BParts& b) {
    const int i0 = __builtin_bit_cast(int, v0), i1 = __builtin_bit_cast(int, v1);
    v0 = __builtin_bit_cast(float, i0 > 0 ? i0 : 0);
    v1 = __builtin_bit_cast(float, i1 > 0 ? i1 : 0);
    const float r0 = v0 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v0) & 0xffff0000u);
    const float r1 = v1 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v1) & 0xffff0000u);
    const float s0 = r0 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r0) & 0xffff0000u);
    const float s1 = r1 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
    b.p1[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, v1), __builtin_bit_cast(unsigned, v0), 0x07060302u);
    b.p2[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, r1), __builtin_bit_cast(unsigned, r0), 0x07060302u);
    b.p3[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
}
__device__ __forceinline__ void pin(BParts& b) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned u1 = b.p1[q], u2 = b.p2[q], u3 = b.p3[q];
        asm volatile("" : "+v"(u1), "+v"(u2), "+v"(u3));
        b.p1[q] = u1; b.p2[q] = u2; b.p3[q] = u3;
    }
}

struct Clk { unsigned long long c0, r0; };
__device__ __forceinline__ void clk_begin(Clk& k) { k.c0 = __builtin_amdgcn_s_memtime(); k.r0 = __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void clk_end(const Clk& k, unsigned long long* out) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - k.c0; out[2 * blockIdx.x + 1] = r1 - k.r0; }
}

// ------------------------------------------------------------------------------------------------------------------------
// 32x32x16: slot = 4 output blocks x 3 pieces x 1 KiB = 12 KiB per k-step of K = 16
template <int kTiles>
__global__ __launch_bounds__(kTiles == 1 ? 512 : 256, kTiles == 1 ? 2 : 1) void s32(int iters, float* __restrict__ sink, const char* __restrict__ stream,
                                                                                    int stream_slots, unsigned long long* clk, float gain, float pscale) {
    constexpr int kSlotB = 12288;
    __shared__ __attribute__((aligned(16))) char slot[3 * kSlotB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kPerThread = kSlotB / (kTiles == 1 ? 512 : 256);       // 24 or 48 bytes
    const int off = threadIdx.x * 16;                                     // first 16 B; then + nthreads * 16 ...
    constexpr int kNT = kTiles == 1 ? 512 : 256;
    for (int i = threadIdx.x; i < 2 * kSlotB / 16; i += kNT) reinterpret_cast<f32x4*>(slot)[i] = reinterpret_cast<const f32x4*>(stream)[i];
    int cslot = 0, spos = 2;
    f32x4 stg[3];
    {
        const char* src = stream + (long)spos * kSlotB;
        stg[0] = *reinterpret_cast<const f32x4*>(src + off);
        if (kTiles == 1) { if (wave < 4) stg[1] = *reinterpret_cast<const f32x4*>(src + 8192 + off); }
        else { stg[1] = *reinterpret_cast<const f32x4*>(src + 4096 + off); stg[2] = *reinterpret_cast<const f32x4*>(src + 8192 + off); }
        spos = 3;
    }
    __syncthreads();
    f32x16 acc[kTiles][4], in[kTiles][4];
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                in[t][nb][r] = (float)((lane * 7 + r * 13 + nb * 5 + t * 3 + wave) % 97) * 0.02f - 0.7f;
                acc[t][nb][r] = 0.0f;
            }
    // per-lane, per-register offsets in [-0.4, 0.4] (a hash): samples stay DIFFERENT from one another, as a trunk's are - without them
    // every sample converges to the same vector under the shared weights and the operand lanes stop toggling
    float pat[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned hsh = (unsigned)(threadIdx.x * 16 + r + 1) * 2654435761u;
        hsh ^= hsh >> 15;
        hsh *= 2246822519u;
        hsh ^= hsh >> 13;
        pat[r] = ((float)(hsh & 0xffffu) * (1.0f / 65536.0f) - 0.5f) * 0.8f * pscale;
    }
    BParts b[kTiles], bn[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) split_pair(in[t][0][2 * q], in[t][0][2 * q + 1], q, b[t]);
    u32x4 a[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) a[q] = reinterpret_cast<const u32x4*>(slot)[q * 64 + lane];
    Clk ck;
    clk_begin(ck);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const u32x4* cur = reinterpret_cast<const u32x4*>(slot + cslot * kSlotB);
            const u32x4* nxt = reinterpret_cast<const u32x4*>(slot + (cslot == 2 ? 0 : cslot + 1) * kSlotB);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                u32x4 an[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) an[q] = nb < 3 ? cur[((nb + 1) * 3 + q) * 64 + lane] : nxt[q * 64 + lane];
#pragma unroll
                for (int t = 0; t < kTiles; ++t) {
                    const int kn = (ks + 1) & 7;
                    split_pair(in[t][kn >> 1][8 * (kn & 1) + 2 * nb], in[t][kn >> 1][8 * (kn & 1) + 2 * nb + 1], nb, bn[t]);
                }
                const u32x4 am[6] = {a[2], a[1], a[0], a[1], a[0], a[0]};
#pragma unroll
                for (int m = 0; m < 6; ++m)
#pragma unroll
                    for (int t = 0; t < kTiles; ++t) {
                        const u32x4 bm = (m == 0 || m == 3 || m == 5) ? b[t].p1 : ((m == 1 || m == 4) ? b[t].p2 : b[t].p3);
                        acc[t][nb] = mfma32(am[m], bm, acc[t][nb]);
                    }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
                for (int m = 0; m < 6 * kTiles - 1; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (nb == 0) {      // store what the previous k-step loaded (position p + 2), load position p + 3
                    char* dst = slot + (cslot >= 1 ? cslot - 1 : 2) * kSlotB;
                    *reinterpret_cast<f32x4*>(dst + off) = stg[0];
                    if (kTiles == 1) { if (wave < 4) *reinterpret_cast<f32x4*>(dst + 8192 + off) = stg[1]; }
                    else { *reinterpret_cast<f32x4*>(dst + 4096 + off) = stg[1]; *reinterpret_cast<f32x4*>(dst + 8192 + off) = stg[2]; }
                    const char* src = stream + (long)spos * kSlotB;
                    stg[0] = *reinterpret_cast<const f32x4*>(src + off);
                    if (kTiles == 1) { if (wave < 4) stg[1] = *reinterpret_cast<const f32x4*>(src + 8192 + off); }
                    else { stg[1] = *reinterpret_cast<const f32x4*>(src + 4096 + off); stg[2] = *reinterpret_cast<const f32x4*>(src + 8192 + off); }
                    spos = spos + 1 == stream_slots ? 0 : spos + 1;
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) a[q] = an[q];
            }
#pragma unroll
            for (int t = 0; t < kTiles; ++t) { pin(bn[t]); b[t] = bn[t]; }
            cslot = cslot == 2 ? 0 : cslot + 1;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int t = 0; t < kTiles; ++t)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // keep the activations like a trunk's: zero-mean, about half of them positive, |x| ~ 0.5 (sigma is stationary under
                    // x <- 1.45 W^T relu(x) for these weights; the clamp and the small fixed term keep it from drifting to 0 or inf)
                    in[t][nb][r] = __builtin_amdgcn_fmed3f(fmaf(acc[t][nb][r], gain, pat[(r + 5 * nb + 3 * t) & 15]), -2.0f, 2.0f);
                    acc[t][nb][r] = 0.0f;
                }
    }
    clk_end(ck, clk);
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += in[t][nb][r];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x < 8) sink[threadIdx.x] = in[0][0][3];      // 8 lanes' value of one register (no dynamic register indexing)
}

// ------------------------------------------------------------------------------------------------------------------------
// 16x16x32: slot = 8 output blocks of 16 rows x 3 pieces x 1 KiB = 24 KiB per k-step of K = 32; kC column blocks of 16 samples
template <int kC>
__global__ __launch_bounds__(kC == 2 ? 512 : 256, kC == 2 ? 2 : 1) void s16(int iters, float* __restrict__ sink, const char* __restrict__ stream,
                                                                            int stream_slots, unsigned long long* clk, float gain, float pscale) {
    constexpr int kSlotB = 24576;
    extern __shared__ __attribute__((aligned(16))) char slot[];          // 3 x 24 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kNT = kC == 2 ? 512 : 256;
    constexpr int kLd = kSlotB / (kNT * 16);                              // 3 or 6 dwordx4 per thread per slot
    const int off = threadIdx.x * 16;
    for (int i = threadIdx.x; i < 2 * kSlotB / 16; i += kNT) reinterpret_cast<f32x4*>(slot)[i] = reinterpret_cast<const f32x4*>(stream)[i];
    int cslot = 0, spos = 2;
    f32x4 stg[kLd];
    {
        const char* src = stream + (long)spos * kSlotB;
#pragma unroll
        for (int i = 0; i < kLd; ++i) stg[i] = *reinterpret_cast<const f32x4*>(src + i * kNT * 16 + off);
        spos = 3;
    }
    __syncthreads();
    // activations: acc[rb][cb] = 16 features x 16 samples, 4 registers; in[][] the other set
    f32x4 acc[8][kC], in[8][kC];
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
        for (int cb = 0; cb < kC; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                in[rb][cb][r] = (float)((lane * 7 + r * 13 + rb * 5 + cb * 3 + wave) % 97) * 0.02f - 0.7f;
                acc[rb][cb][r] = 0.0f;
            }
    float pat[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned hsh = (unsigned)(threadIdx.x * 16 + r + 1) * 2654435761u;
        hsh ^= hsh >> 15;
        hsh *= 2246822519u;
        hsh ^= hsh >> 13;
        pat[r] = ((float)(hsh & 0xffffu) * (1.0f / 65536.0f) - 0.5f) * 0.8f * pscale;
    }
    BParts b[kC], bn[kC];
#pragma unroll
    for (int cb = 0; cb < kC; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) split_pair(in[q >> 1][cb][2 * (q & 1)], in[q >> 1][cb][2 * (q & 1) + 1], q, b[cb]);
    u32x4 a[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) a[q] = reinterpret_cast<const u32x4*>(slot)[q * 64 + lane];
    Clk ck;
    clk_begin(ck);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                                  // K = 128 = 4 k-steps of 32
            const u32x4* cur = reinterpret_cast<const u32x4*>(slot + cslot * kSlotB);
            const u32x4* nxt = reinterpret_cast<const u32x4*>(slot + (cslot == 2 ? 0 : cslot + 1) * kSlotB);
#pragma unroll
            for (int rb = 0; rb < 8; ++rb) {
                u32x4 an[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) an[q] = rb < 7 ? cur[((rb + 1) * 3 + q) * 64 + lane] : nxt[q * 64 + lane];
                // the next k-step's B operands: kC x 4 value pairs over the 8 row-block groups
                const int kn = (ks + 1) & 3;
#pragma unroll
                for (int i = 0; i < kC / 2; ++i) {
                    const int pair = rb * (kC / 2) + i, cb = pair >> 2, q = pair & 3;      // pair q of column block cb
                    split_pair(in[2 * kn + (q >> 1)][cb][2 * (q & 1)], in[2 * kn + (q >> 1)][cb][2 * (q & 1) + 1], q, bn[cb]);
                }
                const u32x4 am[6] = {a[2], a[1], a[0], a[1], a[0], a[0]};
#pragma unroll
                for (int m = 0; m < 6; ++m)
#pragma unroll
                    for (int cb = 0; cb < kC; ++cb) {
                        const u32x4 bm = (m == 0 || m == 3 || m == 5) ? b[cb].p1 : ((m == 1 || m == 4) ? b[cb].p2 : b[cb].p3);
                        acc[rb][cb] = mfma16(am[m], bm, acc[rb][cb]);
                    }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
                for (int m = 0; m < 6 * kC - 1; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (rb == 0) {
                    char* dst = slot + (cslot >= 1 ? cslot - 1 : 2) * kSlotB;
#pragma unroll
                    for (int i = 0; i < kLd; ++i) *reinterpret_cast<f32x4*>(dst + i * kNT * 16 + off) = stg[i];
                    const char* src = stream + (long)spos * kSlotB;
#pragma unroll
                    for (int i = 0; i < kLd; ++i) stg[i] = *reinterpret_cast<const f32x4*>(src + i * kNT * 16 + off);
                    spos = spos + 1 == stream_slots ? 0 : spos + 1;
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) a[q] = an[q];
            }
#pragma unroll
            for (int cb = 0; cb < kC; ++cb) { pin(bn[cb]); b[cb] = bn[cb]; }
            cslot = cslot == 2 ? 0 : cslot + 1;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int rb = 0; rb < 8; ++rb)
#pragma unroll
            for (int cb = 0; cb < kC; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    in[rb][cb][r] = __builtin_amdgcn_fmed3f(fmaf(acc[rb][cb][r], gain, pat[(r + 4 * rb + 7 * cb) & 15]), -2.0f, 2.0f);
                    acc[rb][cb][r] = 0.0f;
                }
    }
    clk_end(ck, clk);
    float s = 0.0f;
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
        for (int cb = 0; cb < kC; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += in[rb][cb][r];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x < 8) sink[threadIdx.x] = in[0][0][3];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static float* g_sink = nullptr;
template <typename F>
int run(const char* name, F launch, double flop_per_iter_per_cu, unsigned long long* dClk) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 400;
    float best = 1e9f;
    double clk_ghz = 0;
    // sustained regime: ~1.5 s of back-to-back launches first (the chip lowers its clock under MFMA load only after a while), then
    // the mean of 20 more launches
    CK(hipEventRecord(e0));
    float warm_ms = 0;
    while (warm_ms < 1500.0f) {
        for (int i = 0; i < 20; ++i) launch(iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&warm_ms, e0, e1));
    }
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) launch(iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    CK(hipEventElapsedTime(&best, e0, e1));
    best /= 20;
    {
        std::vector<unsigned long long> h(512);
        CK(hipMemcpy(h.data(), dClk, 512 * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (int i = 0; i < 256; ++i) s += (double)h[2 * i] / (double)h[2 * i + 1];
        clk_ghz = s / 256 * 0.1;                                   // s_memrealtime ticks at 100 MHz
    }
    const double flops = flop_per_iter_per_cu * iters * 256;
    {
        float hsink[16];
        CK(hipMemcpy(hsink, g_sink, sizeof(hsink), hipMemcpyDeviceToHost));
        printf("   (sample activations:");
        for (int i = 0; i < 8; ++i) printf(" %.3f", hsink[i]);
        printf(")\n");
    }
    printf("%-58s %.3f ms  %.0f TFLOP/s (%.3f of 2.5 PF)  in-kernel clock %.2f GHz\n", name, best, flops / best / 1e9, flops / best / 1e9 / 2500.0, clk_ghz);
    return 0;
}

int main(int argc, char** argv) {
    const bool constant = argc > 1 && argv[1][0] == 'c';
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> uw(-0.15f, 0.15f);
    const int slots32 = 116, slots16 = 58;                              // 1.4 MB stream either way
    std::vector<unsigned short> hs((size_t)slots32 * 12 * 64 * 8);
    for (size_t i = 0; i < hs.size(); ++i) {
        const size_t j = constant ? i % (12 * 64 * 8) : i;
        if (constant && i >= 12 * 64 * 8) { hs[i] = hs[j]; continue; }
        const float v = uw(rng) * ((j / 512) % 3 == 0 ? 1.0f : ((j / 512) % 3 == 1 ? 1.0f / 256 : 1.0f / 65536));
        hs[i] = (unsigned short)(__builtin_bit_cast(unsigned, v) >> 16);
    }
    char* dS;
    float* dO;
    unsigned long long* dClk;
    CK(hipMalloc(&dS, hs.size() * 2));
    CK(hipMalloc(&dO, 64));
    g_sink = dO;
    CK(hipMalloc(&dClk, 512 * 8));
    CK(hipMemcpy(dS, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&s16<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 24576));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&s16<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 24576));
    printf("weight stream: %s\n", constant ? "one 12 KiB slot repeated (constant operands)" : "random, 1.4 MB");
    // per iteration (one 128x128 layer) per CU: 256 samples x 128 x 128 x 2 FLOP x 6 products
    const double f = 256.0 * 128 * 128 * 2 * 6;
    for (int round = 0; round < 2; ++round) {
        const float gain = round == 0 ? 1.0f : 0.05f;
        const float pscale = round == 0 ? 1.0f : 0.0f;
        printf("activation gain %.2f (%s)\n", gain, round == 0 ? "trunk-like activations: zero-mean, |x| ~ 0.3, every sample different" : "activations decay to ~0: same instruction stream, nearly constant operands");
        if (run("32x32x16, 8 waves x 1 tile (current mapping)", [&](int it) { hipLaunchKernelGGL((s32<1>), dim3(256), dim3(512), 0, 0, it, dO, dS, slots32, dClk, gain, pscale); }, f, dClk)) return 1;
        if (run("32x32x16, 4 waves x 2 tiles (512 registers)", [&](int it) { hipLaunchKernelGGL((s32<2>), dim3(256), dim3(256), 0, 0, it, dO, dS, slots32, dClk, gain, pscale); }, f, dClk)) return 1;
        if (run("16x16x32, 8 waves x 2 column blocks", [&](int it) { hipLaunchKernelGGL((s16<2>), dim3(256), dim3(512), 3 * 24576, 0, it, dO, dS, slots16, dClk, gain, pscale); }, f, dClk)) return 1;
        if (run("16x16x32, 4 waves x 4 column blocks (512 registers)", [&](int it) { hipLaunchKernelGGL((s16<4>), dim3(256), dim3(256), 3 * 24576, 0, it, dO, dS, slots16, dClk, gain, pscale); }, f, dClk)) return 1;
    }
    return 0;
}
