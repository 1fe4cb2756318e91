// Layout of the MFMA-ordered weight image ("packed net") streamed by the field kernel.
//
// The trunk is evaluated transposed, Y^T = W^T X^T, with v_mfma_f32_32x32x2_f32:
//   A operand = weights  : lane l supplies W[k(s,h)][32*nb + i],  i = l & 31, h = l >> 5
//   B operand = activations: lane l supplies act[k(s,h)] of sample j = l & 31
//   D (16 regs)          : lane l, reg r holds output feature 32*nb + (r&3) + 8*(r>>2) + 4*h of sample j
// so the accumulator registers of one layer ARE the B operands of the next (register s of input
// block kb feeds k-step s with k(s,h) = 32*kb + (s&3) + 8*(s>>2) + 4*h); activations never leave
// the register file.  Weights are stored so that one 16-byte load per lane yields the A operands
// of 4 consecutive k-steps and one wave-instruction reads 1 KiB contiguously:
//   chunk(g, nb)[lane][e]  with  g = "group" of 4 k-steps (s = 4t+e),  k = 32*kb + 8*t + 4*h + e
#pragma once

namespace mvnerf {

constexpr int kChunkFloats = 256;                       // 64 lanes x 4 floats = 1 KiB
// The kernel consumes the chunks strictly in storage order ("weight stream"), four chunks (one
// group x 4 output blocks, 4 KiB) per step, prefetching one step ahead across layer boundaries:
//   layer 0   : 40 groups x 4 nb.  Groups 0..7 = 32 k-steps: k-step d*10+k carries PE(cam xyz) octave k of
//               dimension d (h=0: sin row d*20+2k, h=1: cos row d*20+2k+1), k-steps 30,31 carry the rgb
//               rows (120 | 121) and (122 | -).  Groups 8..39 = the 256 feature rows.
//               The 60 PE(cam dir) rows are NOT streamed: cam dir is constant along a ray, so
//               b0 + W0_dir^T PE(dir) is computed once per (view, ray) by dir_bias_kernel from the plain
//               copy below and used as the layer-0 accumulator seed.
//   12 hidden : 16 groups x 4 nb each, group = (kb, t)
//   read-out  : 16 chunks (kb, t), output rows i >= 4 zero; consumed 4 chunks per step
// followed by the biases in accumulator order [h][nb][r], then plain copies of W0 rows 60..119, b0 and the read-out
// kernel (the fp32 field kernel evaluates the 128 -> 4 read-out on the vector ALU from that copy).
constexpr int kL0Groups = 40;
constexpr int kL0GroupFeat = 8;
constexpr int kGroupFloats = 4 * kChunkFloats;                         // 1024
constexpr int kPackW0 = 0;
constexpr int kPackW0Floats = kL0Groups * kGroupFloats;                // 40960
constexpr int kPackHidden = kPackW0 + kPackW0Floats;
constexpr int kHiddenWFloats = 16 * kGroupFloats;                      // 16384
constexpr int kNumHidden = 12;                                         // 6 blocks x 2 Dense
constexpr int kPackWr = kPackHidden + kNumHidden * kHiddenWFloats;
constexpr int kPackWrFloats = 16 * kChunkFloats;                       // 4096
constexpr int kPackB0 = kPackWr + kPackWrFloats;                       // bias perm [h][nb][r], 128
constexpr int kPackBHidden = kPackB0 + 128;                            // 12 x 128
constexpr int kPackBr = kPackBHidden + kNumHidden * 128;               // 4 (+4 pad)
constexpr int kPackW0Dir = kPackBr + 8;                                // plain [60][128]: W0 rows 60..119
constexpr int kPackB0Plain = kPackW0Dir + 60 * 128;                    // plain b0[128]
constexpr int kPackWrPlain = kPackB0Plain + 128;                       // plain read-out kernel [128][4] (Keras order)
constexpr int kPackTotal = kPackWrPlain + 512;                         // 251656 (multiple of 4)

// feature index held by accumulator register r of lane-half h inside a 32-wide block
__host__ __device__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// inverse: position of feature n inside a 128-float vector stored in accumulator order [h][nb][r]
__host__ __device__ constexpr int acc_slot(int n) {
    return (((n & 31) >> 2) & 1) * 64 + (n >> 5) * 16 + ((n & 3) + 4 * ((n & 31) >> 3));
}

}  // namespace mvnerf
