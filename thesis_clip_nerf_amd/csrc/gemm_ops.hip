// fp32 "NT" GEMM for the GraspReadout's wide Dense layers (delta_ngf/layers.py:8-42 via lmvnerf/model_v4.py: block_0 maps the
// 42 x 64 = 2688 concatenated offset features of a grasp to 128 / 64 units).  LanguageNeRF's train_step evaluates that layer, its input
// gradient, its weight gradient and their derivatives on M = B x n_points <= a few thousand rows: GEMMs of 1 GFLOP whose library
// kernels run 36 workgroups (246 us) or 65 k threads of 16 x 16 tiles (589 us) on this shape (profiles/r02_language_step_trace.md).
//
//   C[M][N] = sum_k A[M][K] Bt[N][K]        (both operands K-contiguous: x @ W^T with torch's Linear weight layout)
//
// One wave per 32 x 64 block of C, v_mfma_f32_32x32x2_f32 with fp32 operands (the products and the accumulation are the library's),
// a lane's 4 k-slots of 8 consecutive k are one float4 (k = k0 + 4h + e for MFMA e), operands straight from global memory (the
// matrices are L2-resident).  Small outputs with a long K (the weight gradients: 64 x 128 outputs over K = B x n_points x 42 = 64 512
// rows) are split along K over blockIdx.y; every split stores its partial and gemm_reduce_kernel adds them in split order - no
// atomics, bit-identical from run to run.
#include <hip/hip_runtime.h>

#include "mvnerf_kernels.h"
#include "mvnerf_mfma.h"

namespace mvnerf {

__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C,
                                                          int M, int N, int K, int splits, const float* __restrict__ bias) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int tiles_n = N / 64;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= (M / 32) * tiles_n) return;
    const int tm = t / tiles_n, tn = t % tiles_n;
    const f32x4* a = reinterpret_cast<const f32x4*>(A + (long)(32 * tm + i) * K) + h;
    const f32x4* b0 = reinterpret_cast<const f32x4*>(Bt + (long)(64 * tn + i) * K) + h;
    const f32x4* b1 = reinterpret_cast<const f32x4*>(Bt + (long)(64 * tn + 32 + i) * K) + h;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.0f;
        acc1[r] = 0.0f;
    }
    const int steps = K / 8, split = (int)blockIdx.y;
    const int s_begin = (int)((long)steps * split / splits), s_end = (int)((long)steps * (split + 1) / splits);
    C += (long)split * M * N;                                // (splits > 1: C is the partial buffer)
    // the operands of step s + 1 are requested before the MFMAs of step s
    f32x4 a4 = {0.0f, 0.0f, 0.0f, 0.0f}, p4 = a4, q4 = a4;
    if (s_begin < s_end) {
        a4 = a[2 * s_begin];
        p4 = b0[2 * s_begin];
        q4 = b1[2 * s_begin];
    }
    for (int s = s_begin; s < s_end; ++s) {
        const int sn = s + 1 < s_end ? s + 1 : s;
        const f32x4 an = a[2 * sn], pn = b0[2 * sn], qn = b1[2 * sn];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = mfma(a4[e], p4[e], acc0);
            acc1 = mfma(a4[e], q4[e], acc1);
        }
        a4 = an;
        p4 = pn;
        q4 = qn;
    }
    const int col = lane & 31, hh = lane >> 5;
    // Dense bias (x W^T + b, added after the products like the separate add it replaces); with K split it is added by the reduce instead
    const float bias0 = (bias && splits == 1) ? bias[64 * tn + col] : 0.0f, bias1 = (bias && splits == 1) ? bias[64 * tn + 32 + col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float* crow = C + (long)(32 * tm + acc_row(r, hh)) * N + 64 * tn + col;
        crow[0] = acc0[r] + bias0;
        crow[32] = acc1[r] + bias1;
    }
}

// c[i] = sum over the splits' partials in a fixed order: a 64 x 16 block takes 64 float4 elements, thread (x, y) adds the partials
// y, y + 16, ... with two independent chains, the 16 sums are added in y order through LDS (a serial walk over 1024 partials
// per element took 100 us)
__global__ __launch_bounds__(1024) void gemm_reduce_kernel(const f32x4* __restrict__ part, long n4, int splits, f32x4* __restrict__ c,
                                                           const f32x4* __restrict__ bias4, int n_cols4) {
    __shared__ f32x4 sred[16][64];
    const int x = threadIdx.x, y = threadIdx.y;
    const long i = (long)blockIdx.x * 64 + x;
    f32x4 s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = s0;
    if (i < n4) {
        int q = y;
        for (; q + 16 < splits; q += 32) {
            const f32x4 u = part[(long)q * n4 + i], w = part[(long)(q + 16) * n4 + i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s0[e] = s0[e] + u[e];
                s1[e] = s1[e] + w[e];
            }
        }
        if (q < splits) {
            const f32x4 u = part[(long)q * n4 + i];
#pragma unroll
            for (int e = 0; e < 4; ++e) s0[e] = s0[e] + u[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s0[e] = s0[e] + s1[e];
    sred[y][x] = s0;
    __syncthreads();
    if (y == 0 && i < n4) {
        f32x4 s = sred[0][x];
        for (int q = 1; q < 16; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = s[e] + sred[q][x][e];
        if (bias4) {                                          // row-major (rows, 4 n_cols4) output: element i sits in columns 4 (i % n_cols4) ..
            const f32x4 b = bias4[i % n_cols4];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = s[e] + b[e];
        }
        c[i] = s;
    }
}

// how many K ranges launch_gemm_nt_f32 uses: none while the output has at least 1024 blocks (a wave per SIMD); otherwise enough for about
// two waves per SIMD, at least 8 k-steps of 8 per range, at most 512
int gemm_nt_splits(int M, int N, int K) {
    const long tiles = (long)(M / 32) * (N / 64);
    if (tiles >= 1024) return 1;
    long want = (2048 + tiles - 1) / tiles;
    const long most = K / 64 > 0 ? K / 64 : 1;
    if (want > most) want = most;
    if (want > 512) want = 512;
    return want < 1 ? 1 : (int)want;
}

hipError_t launch_gemm_nt_f32(const float* A, const float* Bt, const float* bias, float* C, int M, int N, int K, float* scratch, hipStream_t st) {
    const int tiles = (M / 32) * (N / 64), splits = gemm_nt_splits(M, N, K);
    if (splits > 1 && !scratch) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gemm_nt_f32_kernel, dim3((unsigned)((tiles + 3) / 4), (unsigned)splits), dim3(256), 0, st, A, Bt, splits > 1 ? scratch : C,
                       M, N, K, splits, bias);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || splits == 1) return e;
    const long n4 = (long)M * N / 4;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(64, 16), 0, st, reinterpret_cast<const f32x4*>(scratch), n4, splits,
                       reinterpret_cast<f32x4*>(C), reinterpret_cast<const f32x4*>(bias), N / 4);
    return hipGetLastError();
}

int gemm_tn_splits(int M, int N, int K) { return gemm_nt_splits(N, K, M); }

// ---- C[N][K] = sum_m G[m][N] A[m][K]: the weight gradient g^T . x with BOTH operands as they lie (row-major over the M rows that are
// contracted) - a transposed copy of a 64 512 x 128 activation costs 36-90 us, more than the product.  One wave per 32 (n) x 64 (k) block;
// a lane's 4 k-slots of 8 consecutive rows m are 4 dword loads per operand block (32 lanes read 128 contiguous bytes of a row).
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(const float* __restrict__ G, const float* __restrict__ A, float* __restrict__ C,
                                                          int M, int N, int K, int splits) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int tiles_k = K / 64;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= (N / 32) * tiles_k) return;
    const int tn = t / tiles_k, tk = t % tiles_k;
    const float* g = G + 32 * tn + i;
    const float* a0 = A + 64 * tk + i;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.0f;
        acc1[r] = 0.0f;
    }
    const int steps = M / 8, split = (int)blockIdx.y;
    const int s_begin = (int)((long)steps * split / splits), s_end = (int)((long)steps * (split + 1) / splits);
    C += (long)split * N * K;
    // the operands of step s + 1 are requested before the MFMAs of step s
    float gv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, p[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (s_begin < s_end) {
        const long m0 = 8L * s_begin + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gv[e] = g[(m0 + e) * N];
            p[e] = a0[(m0 + e) * K];
            q[e] = a0[(m0 + e) * K + 32];
        }
    }
    for (int s = s_begin; s < s_end; ++s) {
        const long m1 = 8L * (s + 1 < s_end ? s + 1 : s) + 4 * h;
        float gn[4], pn[4], qn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gn[e] = g[(m1 + e) * N];
            pn[e] = a0[(m1 + e) * K];
            qn[e] = a0[(m1 + e) * K + 32];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = mfma(gv[e], p[e], acc0);
            acc1 = mfma(gv[e], q[e], acc1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gv[e] = gn[e];
            p[e] = pn[e];
            q[e] = qn[e];
        }
    }
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float* crow = C + (long)(32 * tn + acc_row(r, hh)) * K + 64 * tk + col;
        crow[0] = acc0[r];
        crow[32] = acc1[r];
    }
}


// ---- the same product for a BATCH of weight gradients that share M, with strided operands, an optional second pair accumulated into the
// same output and the column sums of one of the G operands (the bias gradient) from the same pass:
//   C[b] (N,K) = G[b]^T A[b] (+ G2[b]^T A2[b]),   colsum[b][n] = sum_m Gs[b][m][n]   (Gs = G or G2)
// G[b] = G + b g_bstride with row stride ldg (a column block of a wider matrix needs no copy), A[b] = A + b a_bstride with row stride lda.
// Grid: (tiles of 32 x 64, splits over M, batch).  With splits > 1 the partial of split s goes to C + s (batch N K) and its column sums
// behind all C partials; gemm_reduce_kernel adds them in split order.
struct TnBatch {
    const float* G; const float* A; const float* G2; const float* A2;
    long g_bstride, a_bstride, g2_bstride, a2_bstride;
    int ldg, lda, ldg2, lda2;
    int colsum_of;                 // 0: none, 1: G, 2: G2
};

__global__ __launch_bounds__(256) void gemm_tn_batched_kernel(TnBatch q, float* __restrict__ C, float* __restrict__ colsum, int M, int N, int K, int splits,
                                                              int batch) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int tiles_k = K / 64;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= (N / 32) * tiles_k) return;
    const int tn = t / tiles_k, tk = t % tiles_k, b = (int)blockIdx.z, split = (int)blockIdx.y;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.0f;
        acc1[r] = 0.0f;
    }
    const int steps = M / 8;
    const int s_begin = (int)((long)steps * split / splits), s_end = (int)((long)steps * (split + 1) / splits);
    float csum = 0.0f;
    for (int pair = 0; pair < (q.G2 ? 2 : 1); ++pair) {
        const float* g = (pair ? q.G2 + (long)b * q.g2_bstride : q.G + (long)b * q.g_bstride) + 32 * tn + i;
        const float* a0 = (pair ? q.A2 + (long)b * q.a2_bstride : q.A + (long)b * q.a_bstride) + 64 * tk + i;
        const long ldg = pair ? q.ldg2 : q.ldg, lda = pair ? q.lda2 : q.lda;
        const bool want_sum = q.colsum_of == pair + 1 && tk == 0;
        float gv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, p[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (s_begin < s_end) {
            const long m0 = 8L * s_begin + 4 * h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gv[e] = g[(m0 + e) * ldg];
                p[e] = a0[(m0 + e) * lda];
                qq[e] = a0[(m0 + e) * lda + 32];
            }
        }
        for (int s = s_begin; s < s_end; ++s) {
            const long m1 = 8L * (s + 1 < s_end ? s + 1 : s) + 4 * h;
            float gn[4], pn[4], qn[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gn[e] = g[(m1 + e) * ldg];
                pn[e] = a0[(m1 + e) * lda];
                qn[e] = a0[(m1 + e) * lda + 32];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = mfma(gv[e], p[e], acc0);
                acc1 = mfma(gv[e], qq[e], acc1);
            }
            if (want_sum) csum += (gv[0] + gv[1]) + (gv[2] + gv[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gv[e] = gn[e];
                p[e] = pn[e];
                qq[e] = qn[e];
            }
        }
    }
    float* Cb = C + ((long)split * batch + b) * N * K;
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float* crow = Cb + (long)(32 * tn + acc_row(r, hh)) * K + 64 * tk + col;
        crow[0] = acc0[r];
        crow[32] = acc1[r];
    }
    if (q.colsum_of && tk == 0) {
        csum = csum + __shfl_xor(csum, 32);                    // the two lane halves hold rows 4h .. 4h + 3 of every step
        if (h == 0) colsum[((long)split * batch + b) * N + 32 * tn + i] = csum;
    }
}

size_t gemm_tn_batched_scratch_floats(int M, int N, int K, int batch, bool colsum) {
    const int splits = gemm_tn_splits(M, N, K * batch);
    if (splits == 1) return 0;
    return (size_t)splits * batch * ((size_t)N * K + (colsum ? N : 0));
}

hipError_t launch_gemm_tn_batched(const TnBatchArgs& a, float* C, float* colsum, int M, int N, int K, int batch, float* scratch, hipStream_t st) {
    TnBatch q;
    q.G = a.G; q.A = a.A; q.G2 = a.G2; q.A2 = a.A2;
    q.g_bstride = a.g_bstride; q.a_bstride = a.a_bstride; q.g2_bstride = a.g2_bstride; q.a2_bstride = a.a2_bstride;
    q.ldg = a.ldg; q.lda = a.lda; q.ldg2 = a.ldg2; q.lda2 = a.lda2;
    q.colsum_of = colsum ? a.colsum_of : 0;
    const int tiles = (N / 32) * (K / 64), splits = gemm_tn_splits(M, N, K * batch);
    if (splits > 1 && !scratch) return hipErrorInvalidValue;
    float* cpart = splits > 1 ? scratch : C;
    float* spart = splits > 1 ? scratch + (size_t)splits * batch * N * K : colsum;
    hipLaunchKernelGGL(gemm_tn_batched_kernel, dim3((unsigned)((tiles + 3) / 4), (unsigned)splits, (unsigned)batch), dim3(256), 0, st, q, cpart, spart,
                       M, N, K, splits, batch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || splits == 1) return e;
    const long n4 = (long)batch * N * K / 4;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(64, 16), 0, st, reinterpret_cast<const f32x4*>(cpart), n4, splits,
                       reinterpret_cast<f32x4*>(C), static_cast<const f32x4*>(nullptr), 1);
    if ((e = hipGetLastError()) != hipSuccess || !q.colsum_of) return e;
    const long s4 = (long)batch * N / 4;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((s4 + 63) / 64)), dim3(64, 16), 0, st, reinterpret_cast<const f32x4*>(spart), s4, splits,
                       reinterpret_cast<f32x4*>(colsum), static_cast<const f32x4*>(nullptr), 1);
    return hipGetLastError();
}

hipError_t launch_gemm_tn_f32(const float* G, const float* A, float* C, int M, int N, int K, float* scratch, hipStream_t st) {
    const int tiles = (N / 32) * (K / 64), splits = gemm_tn_splits(M, N, K);
    if (splits > 1 && !scratch) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3((unsigned)((tiles + 3) / 4), (unsigned)splits), dim3(256), 0, st, G, A, splits > 1 ? scratch : C,
                       M, N, K, splits);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || splits == 1) return e;
    const long n4 = (long)N * K / 4;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(64, 16), 0, st, reinterpret_cast<const f32x4*>(scratch), n4, splits,
                       reinterpret_cast<f32x4*>(C), static_cast<const f32x4*>(nullptr), 1);
    return hipGetLastError();
}

}  // namespace mvnerf
