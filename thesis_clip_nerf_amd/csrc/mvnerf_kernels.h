// Internal launch interface between api.hip and the kernel translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvnerf {

struct FieldParams {
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* images;
    const float* features;
    const float* k4;
    const float* einv;
    const float* net;      // packed (mvnerf_pack.h)
    const float* texel_table;   // optional (B*V,H,W,128): W0_features^T features per texel, accumulator order
    float* rgbs;
    float* dir_bias;       // workspace (B*V*R,128): layer-0 accumulator seed per (view, ray)
    int32_t* tap_idx;      // optional
    float* pix;            // optional
    float* embedding;      // optional (B,R,S,128): trunk output before the read-out
    float* acts_view;      // optional (4,B*V,R,S,128): layer 0 and the 3 per-view blocks (complete_output)
    float* acts_fused;     // optional (4,B,R,S,128): view mean and the 3 fusion blocks (complete_output)
    // training mode (kStash): pre-activation tensors in tile layout [slot][tile][128][32]
    float* stash;          // 7 per-view slots (x0,h1,x1,h2,x2,h3,x3), tile = view tile (b*V+v)*tiles_per_b + k
    long stash_stride;     // floats per per-view slot = V * n_tiles * 4096
    float* stash_fused;    // 7 fused slots (mean,h4,x4,h5,x5,h6,x6), tile = b*tiles_per_b + k
    long stash_fused_stride;   // floats per fused slot = n_tiles * 4096
    int B, V, R, S, H, W;
    long total;            // B*R*S samples
    long n_tiles;          // ceil(total / 32)
    unsigned int* tile_counter;   // set by launch_field_eval
    // forward-mode tangent kernel (query_ops.hip)
    const float* t_o;      // (B,R,3) tangent of rays_o
    const float* t_d;      // (B,R,3) tangent of rays_d
    float* t_acts;         // (4,B,R,S,128) tangents of the view mean and the 3 fusion blocks
    float* dir_tan;        // workspace (B*V*R,128): tangent of the layer-0 seed
};

// api.hip: records the message mvnerf_last_error() returns and hands `code` back (shared by the extern "C" translation units)
int api_fail(int code, const char* fmt, ...);

hipError_t launch_pack_net(const float* net_keras, float* packed, hipStream_t stream);
hipError_t launch_field_eval(const FieldParams& p, hipStream_t stream);
hipError_t launch_dir_bias(const FieldParams& p, hipStream_t stream);
hipError_t launch_project_texels(const float* features, const float* packed_net, const float* packed_net1, long n_texels,
                                 float* table, float* table1, hipStream_t stream);
hipError_t launch_field_jvp(const FieldParams& p, hipStream_t stream);
size_t packed_net_bf16_bytes();          // the field_eval_bf16.hip stream (480 KiB) followed by the field_eval_bf16x.hip stream
hipError_t launch_pack_net_bf16(const float* net_keras, void* packed16, hipStream_t st);   // packs both
// field_eval_bf16x.hip: the texel-table form of the bf16 field pass on v_mfma_f32_16x16x32_bf16, one ring slot per layer
size_t packed_net_bf16x_bytes();
hipError_t launch_pack_net_bf16x(const float* net_keras, void* packed16x, hipStream_t st);
bool field_eval_bf16x_supports(const FieldParams& p);
hipError_t launch_field_eval_bf16x(const FieldParams& p, const void* packed16x, hipStream_t stream);
hipError_t launch_project_texels_bf16(const float* features, const void* packed16, const void* packed16b, long n_texels, float* table,
                                      float* table1, hipStream_t stream);
hipError_t launch_field_eval_bf16(const FieldParams& p, const void* packed16, hipStream_t stream, bool maps_bf16 = false);   // maps_bf16: p.features holds bf16
hipError_t launch_project_texels_bf16maps(const void* features_bf16, const void* packed16, const void* packed16b, long n_texels, float* table,
                                          float* table1, hipStream_t stream);
size_t packed_net_split_bytes();
hipError_t launch_pack_net_split(const float* net_keras, void* packed_split, hipStream_t st);
hipError_t launch_field_eval_split(const FieldParams& p, const void* packed_split, hipStream_t stream);
// field_eval_split16.hip: the same field pass issued as v_mfma_f32_16x16x32_bf16 (inference; its weight stream follows the
// 32x32x16 kernel's inside the packed_split buffer)
size_t packed_net_split16_bytes();
hipError_t launch_pack_net_split16(const float* net_keras, void* packed_split16, hipStream_t st);
bool field_eval_split16_supports(const FieldParams& p);
hipError_t launch_field_eval_split16(const FieldParams& p, const void* packed_split16, hipStream_t stream);
int set_split_kernel(int which);          // field_eval_split.hip: which of the three split kernels runs (returns the previous value)
// field_eval_split16h.hip: the same kernel body on two fp16 pieces per operand, three v_mfma_f32_16x16x32_f16 per product block
size_t packed_net_split16h_bytes();
hipError_t launch_pack_net_split16h(const float* net_keras, void* packed_split16h, hipStream_t st);
bool field_eval_split16h_supports(const FieldParams& p);
hipError_t launch_field_eval_split16h(const FieldParams& p, const void* packed_split16h, hipStream_t stream);

// grasp_head.hip: the per-point part of GraspReadout (value, VJP, derivative of the VJP)
size_t grasp_head_packed_floats();
hipError_t launch_grasp_head_pack(const float* w4, const float* wc, float* packed, hipStream_t st);
hipError_t launch_grasp_head_fwd(const float* acts, const float* packed, const float* b4, const float* bc, long N, float* c, float* y, hipStream_t st);
hipError_t launch_grasp_head_vjp(const float* g_y, const float* c, const float* y, const float* packed, long N, float* g_v, float* q, float* g_u,
                                 float* g_acts, hipStream_t st);
hipError_t launch_grasp_head_vjp_bwd(const float* t_acts, const float* g_y, const float* c, const float* y, const float* q, const float* packed,
                                     long N, float* out_gy, float* r, float* m, float* p, hipStream_t st);

hipError_t launch_get_rays(const double* m9, const double* origin3, const float* u, const float* v, int n_rays,
                           int width, int normalize, float* rays_o, float* rays_d, double* rays_d64,
                           hipStream_t stream);
hipError_t launch_stratified(const float* u, long n, int n_samples, double near_, double far_, float* z,
                             hipStream_t stream);
hipError_t launch_composite(const float* z, const float* rgbs, int n_rays, int S, float* rgb, float* depth,
                            float* weights, hipStream_t stream);
hipError_t launch_resample(const float* z, const float* weights, const float* u_fine, int n_rays, int q7_mode,
                           float* z_all, float* z_fine, int32_t* above, int32_t* below, int32_t* fine_rank, hipStream_t stream);

hipError_t launch_sample_pdf(const float* bins, const float* weights, const float* u, int n_rays, int q7_mode,
                             float* samples, int32_t* above, int32_t* below, hipStream_t stream);

// unfused_ops.hip
hipError_t launch_points_on_rays(const float* o, const float* d, const float* z, long n, int S, float* out, hipStream_t st);
hipError_t launch_project_points(const float* world, const float* k4, const float* einv, int B, int V, long N,
                                 float* pix, float* cam, hipStream_t st);
hipError_t launch_camera_directions(const float* d, const float* einv, int B, int V, int R, float* out, hipStream_t st);
hipError_t launch_position_encoding(const float* x, long n_elems, int n_freq, float freq0, float* out, hipStream_t st);
hipError_t launch_bilinear_gather(const float* images, const float* features, const float* pix, int BV, long Q, int H,
                                  int W, float* out, int32_t* taps, hipStream_t st);
hipError_t launch_sigma_to_alpha(const float* sigma, const float* dists, long n, float* alpha, hipStream_t st);
hipError_t launch_readout(const float* emb, const float* wr, const float* br, long n, float* rgbs, hipStream_t st);
hipError_t launch_finish_view(const float* rgb, const float* depth, long n, float* minmax, uint8_t* rgb8, uint8_t* depth8,
                              hipStream_t st);

// train_ops.hip
hipError_t launch_gemm_nt_f32(const float* A, const float* Bt, const float* bias, float* C, int M, int N, int K, float* scratch,
                              hipStream_t st);   // bias: N floats added to every row, or nullptr
int gemm_nt_splits(int M, int N, int K);
hipError_t launch_gemm_tn_f32(const float* G, const float* A, float* C, int M, int N, int K, float* scratch, hipStream_t st);
// a batch of weight gradients sharing M: C[b] = G[b]^T A[b] (+ G2[b]^T A2[b]), optional column sums of G (colsum_of = 1) or G2 (= 2)
struct TnBatchArgs {
    const float* G; const float* A; const float* G2; const float* A2;
    long g_bstride, a_bstride, g2_bstride, a2_bstride;
    int ldg, lda, ldg2, lda2;
    int colsum_of;
};
size_t gemm_tn_batched_scratch_floats(int M, int N, int K, int batch, bool colsum);
hipError_t launch_gemm_tn_batched(const TnBatchArgs& a, float* C, float* colsum, int M, int N, int K, int batch, float* scratch, hipStream_t st);
int gemm_tn_splits(int M, int N, int K);
hipError_t launch_pack_bwd_streams(const float* net_keras, float* dst, hipStream_t st);
hipError_t launch_field_dz(const FieldParams& p, const float* g0_tl, const float* w0t_streams, float* d_z, float* d_o,
                           float* d_d, float* d_features, hipStream_t st);
hipError_t launch_texel_scatter(const FieldParams& p, const float* g0_tl, float* texel_grad, hipStream_t st);
hipError_t launch_texel_grad_to_features(const float* texel_grad, const float* w0_feat, long n_texels, float* d_features, hipStream_t st);
// Zero `bytes` (a multiple of 4, dword-aligned) with a kernel.  Used instead of hipMemsetAsync: a memset node captured into a HIP graph from
// the autograd thread did not run on the graph's later launches (ROCm 7.2; LanguageNeRF.compile(graph=True) replays these paths).
hipError_t launch_zero(void* ptr, size_t bytes, hipStream_t st);
// rows (n_slots, n_rows, 128) row-major <- n_slots tile-layout tensors `slot_stride` floats apart (tile = 128 features x 32 rows)
hipError_t launch_tl_to_rows(const float* tl, long slot_stride, int n_slots, long n_rows, long n_tiles, float* rows, hipStream_t st);
hipError_t launch_rows_to_tl(const float* rows, long n_rows, long n_tiles, int accumulate, float* out_tl, hipStream_t st);
hipError_t launch_dw_tile(const float* a_tl, int relu_a, const float* g_tl, int g_feats, long n_tiles, float* dW, int ldn,
                          int n_valid, float* db, int max_wgs, float* part, hipStream_t st);
hipError_t launch_dense_bwd_fused(const float* g_tl, const float* a_tl, const float* wstream, const float* resid_tl,
                                  float* da_tl, long n_tiles, float* dW, float* db, int max_wgs, float* part, hipStream_t st,
                                  const float* amax_in = nullptr, float* amax_out = nullptr);
// amax_in / amax_out (training form): 64-float slots holding max |g| of the incoming / outgoing gradient tensor (zeroed by the caller before
// the producer runs): the power-of-two scale of the fp16 cut of G (train_ops.hip, MVT_BWD_F16)
constexpr int kBwdAmaxSlots = 64;
hipError_t launch_mse_grad(const float* pred, const float* label, long n, float* d_pred, float* loss, hipStream_t st);
hipError_t launch_composite_bwd(const float* z, const float* rgbs, const float* d_rgb, const float* d_depth,
                                const float* d_w, int n_rays, int S, float* d_rgbs, float* d_z, hipStream_t st);
hipError_t launch_resample_bwd(const float* z, const float* weights, const float* u_fine, const int32_t* fine_rank,
                               const float* d_z_all, int n_rays, int q7_mode, float* d_weights, hipStream_t st);
hipError_t launch_readout_bwd(const float* x_tl, const float* rgbs, const float* d_rgbs, const float* wr, long n_rows,
                              long n_tiles, float* do_tl, float* g_tl, hipStream_t st, float* amax_out = nullptr);
hipError_t launch_dw0(const FieldParams& p, const float* g0_tl, float* dW0, float* db0, int max_wgs, float* part, hipStream_t st,
                      const float* amax_in = nullptr);
hipError_t launch_view_broadcast(const float* g_fused, int V, long tiles_per_b, long n_tiles, float* g_view, hipStream_t st);
hipError_t launch_adam_clip(float* param, const float* grad, float* m, float* v, long n, float lr_t, float beta1, float beta2,
                            float eps, float clip, const unsigned char* update_mask, hipStream_t st);

}  // namespace mvnerf
