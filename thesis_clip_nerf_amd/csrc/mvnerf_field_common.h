// Device helpers shared by the fused field kernels (field_eval.hip: inference / training forward; query_ops.hip:
// forward-mode tangent kernel): accumulator-order bias loads and row stores, the swizzled wave-private LDS stage.
#pragma once

#include "mvnerf_mfma.h"

#ifndef MV_ABL_BIAS
#define MV_ABL_BIAS 0
#endif

namespace mvnerf {

constexpr int kTile = 32;            // samples per wavefront (MFMA N dimension)
constexpr int kStageRow = 128;       // floats per staged sample row (half of the 256 channels)

template <bool kAdd>
__device__ __forceinline__ void bias_to_acc(const float* __restrict__ bperm, int h, f32x16 (&acc)[4]) {
#if MV_ABL_BIAS
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
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = p[nb * 4 + q];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (kAdd) acc[nb][4 * q + c] = acc[nb][4 * q + c] + v[c];
                else acc[nb][4 * q + c] = v[c];
            }
        }
    }
}

// lane (j,h) holds features 32*nb + 8*q + 4*h + {0..3} of sample j in registers 4q..4q+3 of block nb
__device__ __forceinline__ void store_acc(float* __restrict__ row128, int h, const f32x16 (&x)[4]) {
    float* e = row128 + 4 * h;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v4 = {x[nb][4 * q], x[nb][4 * q + 1], x[nb][4 * q + 2], x[nb][4 * q + 3]};
            *reinterpret_cast<f32x4*>(e + 32 * nb + 8 * q) = v4;
        }
}

__device__ __forceinline__ int stage_offset(int row, int chunk) {      // floats; XOR swizzle on 16-B chunks
    return row * kStageRow + ((chunk ^ (row & 15)) << 2);
}

}  // namespace mvnerf
