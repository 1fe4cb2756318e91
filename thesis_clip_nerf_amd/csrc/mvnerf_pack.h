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
//   layer 0   : 48 groups x 4 nb   (15 PE groups [h=0: xyz rows 0..59, h=1: dir rows 60..119],
//                                   1 rgb group, 32 feature groups)
//   12 hidden : 16 groups x 4 nb each, group = (kb, t)
//   read-out  : 16 chunks (kb, t), output rows i >= 4 zero; consumed 4 chunks per step
// followed by the biases in accumulator order [h][nb][r].
constexpr int kL0Groups = 48;
constexpr int kL0GroupPE = 0;
constexpr int kL0GroupRGB = 15;
constexpr int kL0GroupFeat = 16;
constexpr int kGroupFloats = 4 * kChunkFloats;                         // 1024
constexpr int kPackW0 = 0;
constexpr int kPackW0Floats = kL0Groups * kGroupFloats;                // 49152
constexpr int kPackHidden = kPackW0 + kPackW0Floats;                   // 49152
constexpr int kHiddenWFloats = 16 * kGroupFloats;                      // 16384
constexpr int kNumHidden = 12;                                         // 6 blocks x 2 Dense
constexpr int kPackWr = kPackHidden + kNumHidden * kHiddenWFloats;     // 245760
constexpr int kPackWrFloats = 16 * kChunkFloats;                       // 4096
constexpr int kPackB0 = kPackWr + kPackWrFloats;                       // 249856: bias perm [h][nb][r], 128
constexpr int kPackBHidden = kPackB0 + 128;                            // 12 x 128
constexpr int kPackBr = kPackBHidden + kNumHidden * 128;               // 251520
constexpr int kPackTotal = kPackBr + 8;                                // 251528 (padded to 16 B multiple)

// feature index held by accumulator register r of lane-half h inside a 32-wide block
__host__ __device__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace mvnerf
