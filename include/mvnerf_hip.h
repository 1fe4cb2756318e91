/* mvnerf_hip.h - C ABI of libmvnerf_hip.so: the MI355X (gfx950) implementation of the volumetric
 * rendering hot path of TWeber132/thesis-clip-nerf (src/lib/mvnerf).
 *
 * The reference has no FFI on this path: it is Python calling TensorFlow ops.  Each entry point
 * below therefore names the reference Python function it replaces (file:line under
 * /root/reference/src/lib); INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32/int32 unless marked [host]; tensors are dense,
 *    row-major, in the shapes the reference uses (B scenes, V source views, R rays, S samples);
 *  - the caller owns every buffer; nothing is allocated, freed or retained by the library;
 *  - `stream` is a hipStream_t (NULL = default stream); all work is stream-ordered and asynchronous;
 *  - return value 0 = success; < 0 = argument error (MVNERF_E_*); > 0 = hipError_t of a failed
 *    launch.  mvnerf_last_error() returns a thread-local description of the last failure.
 */
#ifndef MVNERF_HIP_H
#define MVNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mvnerf_stream_t; /* hipStream_t */

#define MVNERF_NET_PARAMS 247300   /* floats of one MLP in Keras order, see mvnerf_pack_net */
#define MVNERF_N_FEATURES 256
#define MVNERF_HIDDEN 128

#define MVNERF_E_ARG (-1)          /* null pointer / non-positive size */
#define MVNERF_E_SHAPE (-2)        /* unsupported shape (message says which) */
#define MVNERF_E_ALIGN (-3)        /* pointer not 16-byte aligned where required */

#define MVNERF_Q7_ZERO 0           /* sample_pdf: out-of-range gather yields 0 (TF-GPU gather_nd) */
#define MVNERF_Q7_CLAMP 1          /* sample_pdf: clamp `above` to the last bin */

int mvnerf_abi_version(void);
const char* mvnerf_last_error(void);

/* Number of floats of the MFMA-ordered weight image produced by mvnerf_pack_net. */
size_t mvnerf_packed_net_floats(void);

/* Re-lay one MLP (MVResNetMLPNeRFEmbedding + RenderReadout, layers.py:334-397) from Keras order
 *   W0[379,128] b0[128] | 6 x (W1[128,128] b1[128] W2[128,128] b2[128]) | Wr[128,4] br[4]
 * (kernel[in,out], bias[out]; 247300 floats) into the operand order the MFMA kernel streams.
 * Call once per weight update.  `packed` must be 16-byte aligned. */
int mvnerf_pack_net(const float* net_keras, float* packed, mvnerf_stream_t stream);

/* get_specific_rays / get_rays (nerf_utils.py:15-35).  d = normalize(M * [u, v, 1]) in float64,
 * rounded to fp32 on store; o = origin.  M = E[:3,:3] @ inv(K[:3,:3]) and origin = E[:3,3] are
 * [host] float64 (the 3x3 inverse stays in NumPy/LAPACK on the host, as in the reference).
 * If u and v are NULL the full image grid is generated: ray n = (v = n / width, u = n % width)
 * for n < width*height (get_rays); otherwise n_rays pixels (u[n], v[n]) (get_specific_rays).
 * rays_o, rays_d: (n_rays,3) fp32.  rays_d64 (optional, may be NULL): (n_rays,3) float64. */
int mvnerf_get_rays(const double* m3x3_host, const double* origin_host, const float* u, const float* v,
                    int n_rays, int width, int normalize, float* rays_o, float* rays_d, double* rays_d64,
                    mvnerf_stream_t stream);

/* sample_along_ray depths (nerf_utils.py:49-58): z[n,i] = fl32(near + i*step) + u[n,i]*fl32(step),
 * step = (far-near)/n_samples in float64.  u, z: (n_rays, n_samples). */
int mvnerf_stratified_depths(const float* u, int n_rays, int n_samples, double near_, double far_,
                             float* z, mvnerf_stream_t stream);

/* One evaluation pass of the radiance field (model_v0.py:122-144 coarse, :157-180 fine):
 *   p = o + z*d                              (nerf_utils.py:59-60 / model_v0.py:157-158)
 *   per view: cam = Einv*[p;1], pix = K4*cam (compute_pixel_in_image_mv, nerf_utils.py:64-81)
 *             bilinear gather of [2*img-1 | features] at pix (get_projection_features_mv,
 *             nerf_utils.py:277-285 -> tensorflow_addons interpolate_bilinear, indexing='xy')
 *             cam_dir = Einv*[d;1]           (world_to_camera_direction_vector_mv, :84-105)
 *             PE(cam xyz), PE(cam_dir)       (position_encoding, nerf_utils.py:108-126)
 *             Dense 379->128 + 3 ResNet blocks                       (layers.py:354-366)
 *   mean over views, 3 ResNet blocks         (layers.py:368-374)
 *   Dense 128->4, sigmoid / softplus         (RenderReadout, layers.py:392-397)
 * rays_o, rays_d (B,R,3); z (B,R,S); images (B,V,H,W,3) in [0,1]; features (B,V,H,W,256), 16-byte
 * aligned; intrinsics, extrinsics_inv (B,V,4,4); packed_net from mvnerf_pack_net.
 * rgbs (B,R,S,4) = (r,g,b,sigma) per sample, 16-byte aligned.
 * tap_idx (optional, may be NULL): (B,V,R,S,4) int32 linear texel indices
 * (b*V+v)*H*W + y*W + x of the tl,tr,bl,br taps (the integer contract of a6).
 * pix (optional, may be NULL): (B,V,R,S,2) fp32 pixel locations (x,y).
 * embedding (optional, may be NULL): (B,R,S,128) output of MVResNetMLPNeRFEmbedding (layers.py:379),
 * 16-byte aligned.
 * acts_per_view (optional): (4, B*V, R, S, 128) and acts_fused (optional): (4, B, R, S, 128): the eight
 * activations the trunk returns with complete_output=True (layers.py:364-377): [x0, f1, f2, f3] per view and
 * [mean, u1, u2, u3] after the view mean; this is what LanguageNeRF consumes (lmvnerf/model_v4.py:261-262).
 * Arbitrary query points (not on rays): pass the points as rays_o, their directions as rays_d, z = 0, S = 1.
 * workspace: 16-byte aligned device scratch of mvnerf_field_workspace_bytes(B,V,R) bytes. */
int mvnerf_field_eval(const float* rays_o, const float* rays_d, const float* z, const float* images,
                      const float* features, const float* intrinsics, const float* extrinsics_inv,
                      const float* packed_net, int B, int V, int R, int S, int H, int W, float* rgbs,
                      int32_t* tap_idx, float* pix, float* embedding, float* acts_per_view, float* acts_fused,
                      void* workspace, mvnerf_stream_t stream);

/* Bytes of scratch mvnerf_field_eval needs (B*V*R*128 floats: the per-(view, ray) part of layer 0). */
size_t mvnerf_field_workspace_bytes(int B, int V, int R);

/* ---- Texel table: the feature part of layer 0 hoisted from samples to texels. ----
 * The first Dense of MVResNetMLPNerfEmbedding (layers.py:357-363) is linear in its input, and the gathered feature
 * vector (model_v0.py:131-136, tfa interpolate_bilinear) is linear in the four texels it blends, so
 *     W0[123:379]^T lerp(f_tl, f_tr, f_bl, f_br) == lerp(W0f^T f_tl, W0f^T f_tr, W0f^T f_bl, W0f^T f_br)
 * up to fp32 rounding (measured: tests/test_gpu_parity.py, same 1e-4 bar as the direct form).  The table holds
 * W0f^T f for every texel of every source view: (B*V,H,W,128) floats, features in accumulator order.  It depends
 * on the feature maps and on ONE net's W0 (coarse and fine nets need a table each); it pays off when a table is
 * used for more samples than it has texels (R*S >= H*W; rendering a frame chunk by chunk re-uses it).
 * mvnerf_field_eval_table == mvnerf_field_eval with 128 of the 190 layer-0 k-steps replaced by a 128-channel
 * lerp of table rows; every output, including tap_idx / pix / the activation taps, has the same meaning. */
size_t mvnerf_texel_table_bytes(int B, int V, int H, int W);
int mvnerf_project_texels(const float* features, const float* packed_net, int B, int V, int H, int W, float* texel_table,
                          mvnerf_stream_t stream);
/* Two nets (coarse, fine) from ONE read of the feature maps; packed_net_b / texel_table_b may both be NULL. */
int mvnerf_project_texels2(const float* features, const float* packed_net, const float* packed_net_b, int B, int V, int H, int W,
                           float* texel_table, float* texel_table_b, mvnerf_stream_t stream);
int mvnerf_field_eval_table(const float* rays_o, const float* rays_d, const float* z, const float* images,
                            const float* features, const float* texel_table, const float* intrinsics,
                            const float* extrinsics_inv, const float* packed_net, int B, int V, int R, int S, int H, int W,
                            float* rgbs, int32_t* tap_idx, float* pix, float* embedding, float* acts_per_view,
                            float* acts_fused, void* workspace, mvnerf_stream_t stream);

/* ---- bf16 variant of the field pass (BASELINE.json configs 3 and 5: bf16 weights and MFMA inputs, fp32
 * accumulate, fp32 geometry / biases / read-out activations).  Not held to the 1e-4 fp32 bar; the achieved error
 * against the fp32 oracle is stated in DESIGN.md and tests/test_gpu_bf16.py. ---- */
size_t mvnerf_packed_net_bf16_bytes(void);
/* Keras-order fp32 MLP (see mvnerf_pack_net) -> bf16 MFMA operand stream.  packed16: 16-byte aligned. */
int mvnerf_pack_net_bf16(const float* net_keras, void* packed16, mvnerf_stream_t stream);
/* mvnerf_project_texels on the bf16 MFMA: features and W0's feature rows rounded to bf16, fp32 accumulation, fp32 table
 * (same layout) - for mvnerf_field_eval_bf16(texel_table = ...); 16x less matrix time than the fp32 projection. */
int mvnerf_project_texels_bf16(const float* features, const void* packed16, const void* packed16_b, int B, int V, int H, int W,
                               float* texel_table, float* texel_table_b, mvnerf_stream_t stream);   /* _b: optional second net */
/* As mvnerf_field_eval, with the Dense kernels taken from packed16 (biases and the per-ray layer-0 seed still come
 * from the fp32 image packed_net).  Optional outputs: tap_idx, embedding, acts_fused (4,B,R,S,128) = the view mean and
 * the three fusion blocks (the part of complete_output that LanguageNeRF consumes, lmvnerf/model_v4.py:261).
 * texel_table (optional, may be NULL): mvnerf_project_texels(features, packed_net) - the fp32 table of the same net;
 * the feature rows of layer 0 are then an fp32 lerp of table rows instead of bf16 MFMA products. */
int mvnerf_field_eval_bf16(const float* rays_o, const float* rays_d, const float* z, const float* images,
                           const float* features, const float* texel_table, const float* intrinsics, const float* extrinsics_inv,
                           const float* packed_net, const void* packed16, int B, int V, int R, int S, int H, int W,
                           float* rgbs, int32_t* tap_idx, float* embedding, float* acts_fused, void* workspace,
                           mvnerf_stream_t stream);

/* The two entry points above with the source feature maps STORED as bf16 (B,V,H,W,256) NHWC - BASELINE.json config 5 as SURVEY.md 8d
 * states it ("feature map + weights bf16"): 512-byte texel rows, what encoders.FeatureProducer(out_dtype=torch.bfloat16) emits.  The
 * bilinear gather of get_projection_features_mv (nerf_utils.py:277-285) widens the taps to fp32 (exact), lerps in fp32 and rounds once, as
 * with fp32 maps - so on maps whose fp32 values are bf16-representable the results are bit-identical to the fp32-map entry points;
 * the projection / gather reads half the bytes.  Source images stay fp32. */
int mvnerf_project_texels_bf16maps(const void* features_bf16, const void* packed16, const void* packed16_b, int B, int V, int H, int W,
                                   float* texel_table, float* texel_table_b, mvnerf_stream_t stream);
int mvnerf_field_eval_bf16maps(const float* rays_o, const float* rays_d, const float* z, const float* images,
                               const void* features_bf16, const float* texel_table, const float* intrinsics, const float* extrinsics_inv,
                               const float* packed_net, const void* packed16, int B, int V, int R, int S, int H, int W,
                               float* rgbs, int32_t* tap_idx, float* embedding, float* acts_fused, void* workspace,
                               mvnerf_stream_t stream);

/* ---- fp32-grade field pass on the 16-bit matrix pipe ("split"): every fp32 GEMM operand is represented by 16-bit pieces and a
 * product block is issued as a few 16-bit MFMAs with fp32 accumulation.  Same function, inputs, outputs and 1e-4 bar as
 * mvnerf_field_eval / mvnerf_field_eval_table (model_v0.py:122-144 / :157-180).  Three kernels read the same packed_split image
 * (mvnerf_set_split_kernel):
 *   MVNERF_SPLIT_F16X3 (default)  two fp16 pieces per operand (round-to-nearest twice, the remainder scaled by 64: 22-24 significant
 *                                 bits), three v_mfma_f32_16x16x32_f16 per block; per ResNet block as close to a float64 evaluation as
 *                                 the fp32 MFMA (tests/test_gpu_split.py); range |w| < 1023, |activation| < 4.19e6
 *                                 (csrc/field_eval_split16_impl.h, field_eval_split16h.hip)
 *   MVNERF_SPLIT_BF16X6           three bf16 pieces per operand, an EXACT cut, the six products of order >= 2^-16 as
 *                                 v_mfma_f32_16x16x32_bf16; dropped terms <= 2^-24 relative; full fp32 range (field_eval_split16.hip)
 *   MVNERF_SPLIT_BF16X6_32        the same products as v_mfma_f32_32x32x16_bf16 (round 2's kernel, field_eval_split.hip)
 * The training forward (mvnerf_field_eval_stash_split) follows the same choice (the 32x32x16 kernel has no fp16 form); the backward's layer
 * launches always use fp16 two-piece products with a per-tensor power-of-two scale on the gradients (csrc/train_ops.hip, MVT_BWD_F16). ---- */
#define MVNERF_SPLIT_F16X3 0
#define MVNERF_SPLIT_BF16X6 1
#define MVNERF_SPLIT_BF16X6_32 2
/* Process-wide choice of the kernel behind mvnerf_field_eval_split / mvnerf_render_fwd_split; returns the previous value (< 0 and
 * no change for an unknown value).  The environment variable MVNERF_SPLIT_MFMA ("f16x3" | "bf16x6" | "32x32x16"), read at every
 * launch, overrides it (A/B runs, tests). */
int mvnerf_set_split_kernel(int which);
size_t mvnerf_packed_net_split_bytes(void);
/* Keras-order fp32 MLP (see mvnerf_pack_net) -> the operand streams of the three kernels above.  packed_split: 16-byte aligned. */
int mvnerf_pack_net_split(const float* net_keras, void* packed_split, mvnerf_stream_t stream);
/* As mvnerf_field_eval (texel_table NULL) / mvnerf_field_eval_table, with the Dense kernels taken from packed_split
 * (biases, the per-ray layer-0 seed and the texel table still come from the fp32 image packed_net).  Optional outputs
 * as there: tap_idx, pix, embedding, acts_per_view, acts_fused. */
int mvnerf_field_eval_split(const float* rays_o, const float* rays_d, const float* z, const float* images,
                            const float* features, const float* texel_table, const float* intrinsics, const float* extrinsics_inv,
                            const float* packed_net, const void* packed_split, int B, int V, int R, int S, int H, int W,
                            float* rgbs, int32_t* tap_idx, float* pix, float* embedding, float* acts_per_view, float* acts_fused,
                            void* workspace, mvnerf_stream_t stream);

/* MVVNeRFRenderer.volumetric_render (model_v0.py:89-100) with sigma_to_alpha (nerf_utils.py:129-140).
 * z (n_rays,S); rgbs (n_rays,S,4); S in {64,128,192,256}.
 * rgb (n_rays,3); depth (n_rays); weights (optional, may be NULL) (n_rays,S). */
int mvnerf_composite(const float* z, const float* rgbs, int n_rays, int S, float* rgb, float* depth,
                     float* weights, mvnerf_stream_t stream);

/* Hierarchical resampling (model_v0.py:150-156): z_mid, probs = w[1:-1], sample_pdf
 * (nerf_utils.py:143-176) with explicit uniforms u_fine, concat, ascending sort.
 * z, weights, u_fine: (n_rays,64).  z_all: (n_rays,128).  Optional outputs (may be NULL):
 * z_fine (n_rays,64) fp32, above / below (n_rays,64) int32 (the integer contract of a13),
 * fine_rank (n_rays,64) int32: position of importance sample i inside z_all (the sort permutation, kept for
 * the backward pass). */
int mvnerf_resample(const float* z, const float* weights, const float* u_fine, int n_rays, int S,
                    int q7_mode, float* z_all, float* z_fine, int32_t* above, int32_t* below, int32_t* fine_rank,
                    mvnerf_stream_t stream);

/* ---- op-level (unfused) entry points: one per reference function, same scalar code as the fused
 * kernels; used by the op-level parity tests and by callers that want a single operator. ---- */

/* world = o + z*d (nerf_utils.py:59-60).  o,d (n_rays,3); z (n_rays,S) -> world (n_rays,S,3). */
int mvnerf_points_on_rays(const float* rays_o, const float* rays_d, const float* z, int n_rays, int S, float* world,
                          mvnerf_stream_t stream);

/* compute_pixel_in_image_mv (nerf_utils.py:64-81).  world (B,N,3); intrinsics, extrinsics_inv (B,V,4,4)
 * -> pixel_locations (B,V,N,2) [x,y], camera_points_homogeneous (B,V,N,4). */
int mvnerf_project_points(const float* world, const float* intrinsics, const float* extrinsics_inv, int B, int V,
                          int N, float* pixel_locations, float* camera_points, mvnerf_stream_t stream);

/* world_to_camera_direction_vector_mv (nerf_utils.py:84-105; homogeneous w = 1).
 * dirs (B,R,3); extrinsics_inv (B,V,4,4) -> (B,V,R,3). */
int mvnerf_camera_directions(const float* dirs, const float* extrinsics_inv, int B, int V, int R, float* out,
                             mvnerf_stream_t stream);

/* position_encoding (nerf_utils.py:108-126).  x: n_elems scalars (any (...,D) flattened);
 * out: n_elems * 2 * n_freq, layout per element [k][sin,cos] (=> (..., D*2*n_freq) as (d n f)). */
int mvnerf_position_encoding(const float* x, long n_elems, int n_freq, float pos_encoding_freq, float* out,
                             mvnerf_stream_t stream);

/* get_projection_features_mv (nerf_utils.py:277-285): tensorflow_addons interpolate_bilinear
 * (indexing='xy') of the grid [images | features].  images (BV,H,W,3) as given (the caller normalises),
 * features (BV,H,W,256), pixel_locations (BV,Q,2) -> out (BV,Q,259); tap_idx (optional) (BV,Q,4) int32. */
int mvnerf_bilinear_gather(const float* images, const float* features, const float* pixel_locations, int BV, int Q,
                           int H, int W, float* out, int32_t* tap_idx, mvnerf_stream_t stream);

/* sigma_to_alpha (nerf_utils.py:129-140), elementwise over n values. */
int mvnerf_sigma_to_alpha(const float* sigma, const float* dists, long n, float* alpha, mvnerf_stream_t stream);

/* sample_pdf (nerf_utils.py:143-176) with explicit uniforms.  bins (n_rays,63); weights (n_rays,62);
 * u (n_rays,64) -> samples (n_rays,64); above / below (optional) (n_rays,64) int32. */
int mvnerf_sample_pdf(const float* bins, const float* weights, const float* u, int n_rays, int n_bins, int n_samples,
                      int q7_mode, float* samples, int32_t* above, int32_t* below, mvnerf_stream_t stream);

/* RenderReadout (layers.py:392-397).  embedding (n,128); wr kernel[128,4], br bias[4] (Keras layout)
 * -> rgbs (n,4) = (sigmoid rgb, softplus sigma). */
int mvnerf_readout(const float* embedding, const float* wr, const float* br, long n, float* rgbs,
                   mvnerf_stream_t stream);

/* render_view epilogue (model_v0.py:275-281).  rgb (n,3), depth (n) -> rgb8 (n,3) = uint8(clip(rgb*255,0,255)),
 * depth8 (n) = uint8(255*(depth-min)/(max-min)).  minmax_scratch: 2 floats of device scratch. */
int mvnerf_finish_view(const float* rgb, const float* depth, long n, float* minmax_scratch, uint8_t* rgb8,
                       uint8_t* depth8, mvnerf_stream_t stream);

/* ---- training step (MVVNeRFRenderer.train_step, model_v0.py:186-197; optimize, nerf_utils.py:8-12) ----
 * Gradients of every MLP variable, including the path through the importance samples that the reference leaves
 * open (no stop_gradient).  V > 1 needs R*S to be a multiple of 32. */

/* Bytes of the activation stash of one mvnerf_field_eval_stash call: 7 per-view + 7 fused pre-activation tensors in tile layout
 * (per view x0 h1 x1 h2 x2 h3 x3, then mean h4 x4 h5 x5 h6 x6).  The per-view slot of x3 is part of the layout but is not written:
 * nothing reads it (the backward of the view mean needs no activation; for V = 1 it is the fused slot of the mean). */
size_t mvnerf_stash_bytes(int B, int V, int R, int S);
/* Weight-gradient reduction of mvnerf_field_backward: every workgroup stores its partial of a layer's [dW | db] span and the partials
 * are summed in workgroup order - bit-identical gradients for identical inputs.  Until the middle of round 2 this was a mode (0 =
 * fp32 atomics, 1 = stored partials); with the partials added by a parallel fixed-order reduction it is also the faster of the two
 * and the only one.  The switch is kept for its callers: it records the flag and returns the previous value, nothing else.  The
 * gradients w.r.t. sample depths (V > 1), ray origins / directions and the source feature maps still accumulate with atomics. */
int mvnerf_set_deterministic(int on);
/* Bytes of scratch mvnerf_field_backward needs (three gradient tensors in tile layout, the read-out's cotangent tile, the per-workgroup
 * weight-gradient partials and 14 x 64 floats of max |g| slots: the power-of-two scales of the layer launches' fp16 products). */
size_t mvnerf_field_backward_scratch_bytes(int B, int V, int R, int S);

/* mvnerf_field_eval in training mode: also stores the trunk's pre-activations into `stash`.
 * texel_table (optional, may be NULL): as in mvnerf_field_eval_table, for the forward value only - the backward
 * recomputes layer 0's inputs from the raw features either way. */
int mvnerf_field_eval_stash(const float* rays_o, const float* rays_d, const float* z, const float* images,
                            const float* features, const float* texel_table, const float* intrinsics,
                            const float* extrinsics_inv, const float* packed_net, int B, int V, int R, int S, int H, int W,
                            float* rgbs, float* stash, void* workspace, mvnerf_stream_t stream);
/* mvnerf_field_eval_stash with the Dense layers as split-bf16 MFMA products (mvnerf_field_eval_split): same stash layout,
 * consumed by the same mvnerf_field_backward. */
int mvnerf_field_eval_stash_split(const float* rays_o, const float* rays_d, const float* z, const float* images,
                                  const float* features, const float* texel_table, const float* intrinsics,
                                  const float* extrinsics_inv, const float* packed_net, const void* packed_split, int B, int V, int R,
                                  int S, int H, int W, float* rgbs, float* stash, void* workspace, mvnerf_stream_t stream);

/* The 12 hidden Dense kernels of one MLP and the three 128-row slabs of the layer-0 kernel, transposed, in
 * weight-stream order (15 x 16384 floats), for the dX GEMMs of the backward pass.
 * net_keras: 247300 floats (see mvnerf_pack_net). */
int mvnerf_pack_bwd_streams(const float* net_keras, float* bwd_streams, mvnerf_stream_t stream);

/* d pred = 2 (pred - label) / n and *loss += mean((pred - label)^2) (Keras MeanSquaredError, model_v0.py:193). */
int mvnerf_mse_grad(const float* pred, const float* label, long n, float* d_pred, float* loss, mvnerf_stream_t stream);

/* volumetric_render backward (model_v0.py:89-100): d_rgb (n_rays,3), d_depth (optional, n_rays), d_weights
 * (optional, n_rays x S) -> d_rgbs (n_rays,S,4) = gradient w.r.t. the per-sample (r,g,b,sigma), and d_z
 * (optional, n_rays x S) = gradient w.r.t. the depths through the intervals and the depth output.  S in {64,128}. */
int mvnerf_composite_bwd(const float* z, const float* rgbs, const float* d_rgb, const float* d_depth,
                         const float* d_weights, int n_rays, int S, float* d_rgbs, float* d_z, mvnerf_stream_t stream);

/* Backward of mvnerf_resample w.r.t. the coarse weights: d_z_all (n_rays,128), fine_rank from the forward call
 * -> d_weights (n_rays,64).  (The coarse depths depend on no variable, so no d_z is returned.) */
int mvnerf_resample_bwd(const float* z, const float* weights, const float* u_fine, const int32_t* fine_rank,
                        const float* d_z_all, int n_rays, int S, int q7_mode, float* d_weights, mvnerf_stream_t stream);

/* Backward of one mvnerf_field_eval_stash call: accumulates dL/d(net variables) into `grad` (247300 floats, Keras
 * order, caller zeroes it) given d_rgbs (B,R,S,4).  Inputs as in the forward call, plus net_keras, the
 * transposed streams and the stash.
 * d_z (optional, may be NULL): (B,R,S), INCREMENTED by the gradient w.r.t. the sample depths through the sample
 * positions (positional encoding of the camera point and the bilinear lerp factors).
 * d_features (optional, may be NULL): (B,V,H,W,256), INCREMENTED by the gradient w.r.t. the source feature maps
 * (combined_features is produced by trainable encoders in the reference, train_nerf.py:27-32; this is what flows
 * back to them). */
int mvnerf_field_backward(const float* rays_o, const float* rays_d, const float* z, const float* images,
                          const float* features, const float* intrinsics, const float* extrinsics_inv,
                          const float* net_keras, const float* bwd_streams, const float* stash, const float* rgbs,
                          const float* d_rgbs, int B, int V, int R, int S, int H, int W, void* scratch, float* grad,
                          float* d_z, float* d_features, mvnerf_stream_t stream);
/* mvnerf_field_backward with the forward's texel table of the SAME net (mvnerf_project_texels; optional, may be NULL).  With it the
 * gradient through the sample positions (d_z) takes the feature rows' part from four table rows per sample instead of recomputing
 * W0 . g0 over the 256 feature channels (same result up to fp32 rounding).  texel_grad (optional scratch, (B,V,H,W,128) floats,
 * needs texel_table): when d_features is wanted, the samples' layer-0 cotangents are scattered onto this 128-channel table gradient
 * and W0 is applied once per texel (dL/df[texel] = W0[123:379] . dL/dT[texel]) - a third of the atomics of the direct scatter and
 * no 379 x 128 product per sample; without it d_features takes the direct path.
 * Reference: the same tape of MVVNeRFRenderer.train_step (model_v0.py:186-197); the table is this library's re-association of
 * layers.py:361-366 (see mvnerf_project_texels). */
int mvnerf_field_backward_table(const float* rays_o, const float* rays_d, const float* z, const float* images,
                                const float* features, const float* texel_table, float* texel_grad, const float* intrinsics,
                                const float* extrinsics_inv, const float* net_keras, const float* bwd_streams, const float* stash,
                                const float* rgbs, const float* d_rgbs, int B, int V, int R, int S, int H, int W, void* scratch,
                                float* grad, float* d_z, float* d_features, mvnerf_stream_t stream);

/* c (M,N) = a (M,K) . bt (N,K)^T in fp32 (fp32 MFMA, one wave per 32 x 64 output block; small outputs with a long K are split along K,
 * the splits' partials go to `scratch` and are added in split order: no atomics, bit-identical from run to run).
 * For the GraspReadout's Dense layers (delta_ngf/layers.py:8-42 as used by lmvnerf/model_v4.py:290-322), their input / weight gradients
 * and the derivatives of those - skinny products (64 x 128 outputs over K = 64 512 rows, 1536 x 128 over K = 2688) on which the library
 * GEMM runs a handful of workgroups.  M % 32 == 0, N % 64 == 0, K % 8 == 0, 16-byte aligned row-major buffers;
 * scratch: mvnerf_gemm_nt_scratch_bytes(M,N,K) bytes (0 for shapes that are not split: NULL allowed). */
size_t mvnerf_gemm_nt_scratch_bytes(int M, int N, int K);
int mvnerf_gemm_nt(const float* a, const float* bt, float* c, int M, int N, int K, void* scratch, mvnerf_stream_t stream);
/* The Dense layer itself, c = a bt^T + bias (bias: N floats, added after the products - the result equals mvnerf_gemm_nt followed by a
 * broadcast add, bit for bit, in one launch less). */
int mvnerf_gemm_nt_bias(const float* a, const float* bt, const float* bias, float* c, int M, int N, int K, void* scratch,
                        mvnerf_stream_t stream);

/* c (N,K) = g (M,N)^T . a (M,K): the weight gradient of a Dense layer with both operands as they lie (rows = the M samples that are
 * contracted), same kernel family and determinism as mvnerf_gemm_nt.  M % 8 == 0, N % 32 == 0, K % 64 == 0;
 * scratch: mvnerf_gemm_tn_scratch_bytes(M,N,K) bytes. */
size_t mvnerf_gemm_tn_scratch_bytes(int M, int N, int K);
int mvnerf_gemm_tn(const float* g, const float* a, float* c, int M, int N, int K, void* scratch, mvnerf_stream_t stream);

/* A batch of such weight gradients that share M, in one launch (+ one fixed-order reduce): for b < batch
 *     c[b] (N,K) = G[b]^T A[b]  (+ G2[b]^T A2[b]),      colsum[b][n] = sum_m Gs[b][m][n]   (the bias gradient; Gs = G or G2)
 * with G[b] = g + b * g_batch_stride (floats) and row stride ldg - a column block of a wider matrix is used where it lies -, likewise
 * A, G2, A2.  This is what the weight gradients of GraspReadout's per-point layers need (delta_ngf/layers.py:8-42 under the nested tapes
 * of lmvnerf/model_v4.py:290-322): four Dense(128 -> 64) on column blocks of one cotangent, and in the second-order pass the sum of two
 * products per layer.  g2 / a2 may be NULL (one product).  colsum_of: 0 none, 1 G, 2 G2; colsum: batch * N floats.
 * M % 8 == 0, N % 32 == 0, K % 64 == 0; scratch: mvnerf_gemm_tn_batched_scratch_bytes(M,N,K,batch,colsum_of != 0) bytes. */
typedef struct mvnerf_gemm_tn_batch {
    const float* g;  const float* a;  const float* g2;  const float* a2;
    long g_batch_stride, a_batch_stride, g2_batch_stride, a2_batch_stride;
    int ldg, lda, ldg2, lda2;
    int colsum_of;
} mvnerf_gemm_tn_batch;
size_t mvnerf_gemm_tn_batched_scratch_bytes(int M, int N, int K, int batch, int with_colsum);
int mvnerf_gemm_tn_batched(const mvnerf_gemm_tn_batch* q, float* c, float* colsum, int M, int N, int K, int batch, void* scratch,
                           mvnerf_stream_t stream);

/* ---- The trunk as a differentiable field on arbitrary query points. ----
 * Reference consumer: LanguageNeRF._call (lmvnerf/model_v4.py:208-265) evaluates fine_embedding on
 * camera_points / camera_directions derived from grasp poses and keeps outputs[4:] = (view mean, u1, u2, u3)
 * (layers.py:376-377); its train_step (:277-322) takes d prediction / d pose inside a second tape and differentiates a
 * loss on that gradient w.r.t. the read-out, i.e. it needs the trunk's input-gradient (VJP) and the derivative of that
 * VJP w.r.t. its cotangent, which is the forward-mode product (JVP).  The trunk's weights are frozen there.
 * points, dirs: (B,N,3) world space (cam point = E^-1 [p;1], cam dir = (E^-1 [d;1])[:3], Q3) - the forward value is
 * mvnerf_field_eval(rays_o = points, rays_d = dirs, z = 0, S = 1, acts_fused = ...).
 *
 * mvnerf_query_jvp: tangents t_points, t_dirs (B,N,3) -> t_acts (4,B,N,128) = J [t_points; t_dirs]; acts (optional,
 *   may be NULL) receives the primal (4,B,N,128).  workspace: mvnerf_query_workspace_bytes(B,V,N).
 * mvnerf_query_vjp: cotangents g_acts (4,B,N,128) -> d_points, d_dirs (B,N,3) = J^T g_acts.  stash: from
 *   mvnerf_field_eval_stash on the same inputs (R = N, S = 1, z = zeros); bwd_streams: mvnerf_pack_bwd_streams;
 *   scratch: mvnerf_query_vjp_scratch_bytes(B,V,N), 16-byte aligned.  N % 32 == 0 when V > 1. */
size_t mvnerf_query_workspace_bytes(int B, int V, int N);
int mvnerf_query_jvp(const float* points, const float* dirs, const float* t_points, const float* t_dirs,
                     const float* images, const float* features, const float* intrinsics, const float* extrinsics_inv,
                     const float* packed_net, int B, int V, int N, int H, int W, float* acts, float* t_acts, void* workspace,
                     mvnerf_stream_t stream);
size_t mvnerf_query_vjp_scratch_bytes(int B, int V, int N);
int mvnerf_query_vjp(const float* points, const float* dirs, const float* images, const float* features,
                     const float* intrinsics, const float* extrinsics_inv, const float* bwd_streams, const float* stash,
                     const float* g_acts, int B, int V, int N, int H, int W, void* scratch, float* d_points, float* d_dirs,
                     mvnerf_stream_t stream);

/* The four fused activations (view mean, u1, u2, u3 - layers.py:376-377, what LanguageNeRF._call keeps as outputs[4:],
 * lmvnerf/model_v4.py:261-262) of a stash written by mvnerf_field_eval_stash with R = N, S = 1, as row-major acts (4, B*N, 128):
 * the stash holds them in the tile layout [tile][feature][32 points]; one transposing pass through LDS.  acts 16-byte aligned. */
int mvnerf_stash_fused_acts(const float* stash, int B, int V, int N, float* acts, mvnerf_stream_t stream);

/* optimize(): clip-by-value (clip > 0) then one Adam step with the bias-corrected rate lr_t.
 * update_mask (optional): n bytes, 0 = leave the element untouched. */
int mvnerf_adam_clip(float* param, const float* grad, float* m, float* v, long n, float lr_t, float beta1, float beta2,
                     float eps, float clip, const unsigned char* update_mask, mvnerf_stream_t stream);

/* ---- the per-point part of GraspReadout (delta_ngf/layers.py:8-42: four Dense(128 -> 64) + elu on the four fused trunk activations,
 * concatenation, Dense(256 -> 64) + elu - what LanguageNeRF._call feeds to the per-pose ResNet blocks, lmvnerf/model_v4.py:261-263) as three
 * fused passes: its value, its vector-Jacobian product and the derivative of that product, which the nested GradientTape of
 * LanguageNeRF.train_step (lmvnerf/model_v4.py:290-322) needs.  fp32, v_mfma_f32_32x32x2_f32, one wavefront per 32 points, row-major (N, F)
 * tensors, 16-byte aligned.  w4: the four Dense kernels (4, 64, 128) [out, in] (torch layout), wc: (64, 256), b4: (4, 64), bc: (64). */
size_t mvnerf_grasp_head_packed_floats(void);
int mvnerf_grasp_head_pack(const float* w4, const float* wc, float* packed, mvnerf_stream_t stream);
/* acts (4, N, 128) -> c (N, 256) = [elu(W_k a_k + b_k)] and y (N, 64) = elu(W_c c + b_c). */
int mvnerf_grasp_head_fwd(const float* acts, const float* packed, const float* b4, const float* bc, long N, float* c, float* y,
                          mvnerf_stream_t stream);
/* g_y (N, 64) = dL/dy -> g_acts (4, N, 128) = dL/d(acts), and the per-point cotangents g_v (N, 64), q (N, 256), g_u (N, 256) from which the
 * weight gradients are skinny GEMMs: dW_c = g_v^T c, db_c = sum g_v, dW_k = g_u[:, 64k:64k+64]^T a_k, db_k = sum g_u[:, 64k:64k+64]. */
int mvnerf_grasp_head_vjp(const float* g_y, const float* c, const float* y, const float* packed, long N, float* g_v, float* q, float* g_u,
                          float* g_acts, mvnerf_stream_t stream);
/* The derivative of mvnerf_grasp_head_vjp: t_acts (4, N, 128) = dL/d(g_acts) -> out_gy (N, 64) = dL/d(g_y) and r (N, 256), m (N, 64),
 * p (N, 256) with dL/dW_k = g_u_k^T t_k + p_k^T a_k, dL/db_k = sum p_k, dL/dW_c = g_v^T r + m^T c, dL/db_c = sum m. */
int mvnerf_grasp_head_vjp_bwd(const float* t_acts, const float* g_y, const float* c, const float* y, const float* q, const float* packed, long N,
                              float* out_gy, float* r, float* m, float* p, mvnerf_stream_t stream);

/* ---- the whole training step behind one call (MVVNeRFRenderer.train_step, model_v0.py:186-197: GradientTape over call(),
 * loss = MSE(y, rgb) + MSE(y, fine_rgb) :193, gradients :194, optimize() :195 = nerf_utils.py:8-12) ----
 * The matching `_bwd` of mvnerf_render_fwd: a host in any language takes a training step with these entry points and
 * a caller-provided workspace; nothing is allocated, nothing synchronises with the host. */
typedef struct mvnerf_train_call {
    /* inputs of _call (model_v0.py:113-119): device pointers, layouts as in mvnerf_render_fwd */
    const float* rays_o;          /* (B,R,3) */
    const float* rays_d;          /* (B,R,3) */
    const float* images;          /* (B,V,H,W,3) in [0,1] */
    const float* features;        /* (B,V,H,W,256) combined_features */
    const float* intrinsics;      /* (B,V,4,4) */
    const float* extrinsics_inv;  /* (B,V,4,4) */
    const float* u_coarse;        /* (B,R,64) the uniforms of sample_along_ray (nerf_utils.py:57) */
    const float* u_fine;          /* (B,R,64) the uniforms of sample_pdf (nerf_utils.py:151) */
    const float* labels;          /* (B,R,3) target colours */
    int B, V, R, S, H, W;         /* S = 64 */
    double near_, far_;
    int q7_mode;                  /* MVNERF_Q7_ZERO | MVNERF_Q7_CLAMP */
    int stop_fine_z;              /* 1: cut the gradient through the importance samples (the reference leaves it open) */
    int use_texel_tables;         /* 1: forward (and the position / feature-map gradients) through per-texel layer-0 tables built inside the
                                   * workspace (mvnerf_project_texels2); pays when R*S >= 2*H*W */
    /* the two MLPs: Keras-order variables (247300 floats each) and the images the kernels read, all caller-owned */
    const float* net_coarse;      /* updated in place by mvnerf_apply_gradients / mvnerf_train_step */
    const float* net_fine;
    const float* packed_coarse;   /* mvnerf_pack_net */
    const float* packed_fine;
    const void* split_coarse;     /* mvnerf_pack_net_split, or both NULL: forward on the fp32 MFMA */
    const void* split_fine;
    const float* bwd_streams_coarse;   /* mvnerf_pack_bwd_streams */
    const float* bwd_streams_fine;
    /* outputs */
    float* loss;                  /* 1 float: overwritten */
    float* grad;                  /* 2 x 247300 floats [coarse | fine], Keras order: overwritten */
    float* rgb;                   /* (B,R,3) */
    float* depth;                 /* (B,R) */
    float* fine_rgb;              /* (B,R,3) */
    float* fine_depth;            /* (B,R) */
    float* d_features;            /* optional (B,V,H,W,256): overwritten with dL/d(combined_features) - what flows back to the reference's
                                   * trainable encoders (train_nerf.py:27-32); NULL = not wanted */
    void* workspace;              /* 256-byte aligned, >= mvnerf_train_workspace_bytes(...) */
    size_t workspace_bytes;
    void* fine_grad_event;        /* optional hipEvent_t (NULL = none): recorded on the stream as soon as the FINE net's half of `grad` is
                                   * final, i.e. before the coarse net's backward is launched - a data-parallel host starts that half's
                                   * all-reduce on a second stream behind this event and overlaps it with the coarse backward */
} mvnerf_train_call;

typedef struct mvnerf_adam_state {
    float* m;                     /* 2 x 247300 first moments  [coarse | fine] */
    float* v;                     /* 2 x 247300 second moments */
    float lr_t;                   /* bias-corrected rate lr * sqrt(1 - beta2^t) / (1 - beta1^t) (tf.keras Adam) */
    float beta1, beta2, eps;      /* Keras defaults 0.9, 0.999, 1e-7 */
    float clip;                   /* clip-by-value bound of optimize() (nerf_utils.py:9), <= 0: none */
    const unsigned char* update_mask;   /* optional 2 x 247300 bytes, 0 = variable not in the optimizer's list */
    int repack;                   /* 1: rebuild packed_* / split_* / bwd_streams_* of the call from the updated variables */
} mvnerf_adam_state;

size_t mvnerf_train_workspace_bytes(int B, int V, int R, int S, int H, int W, int use_texel_tables, int want_d_features);
/* Forward with stash + loss + full backward: fills loss, grad, the four rendered outputs (and d_features).  model_v0.py:190-194. */
int mvnerf_loss_and_grads(const mvnerf_train_call* call, mvnerf_stream_t stream);
/* optimize() on `grad` (a data-parallel host all-reduces it first): clip-by-value, Adam, optional re-pack.  nerf_utils.py:8-12. */
int mvnerf_apply_gradients(const mvnerf_train_call* call, const mvnerf_adam_state* adam, mvnerf_stream_t stream);
/* mvnerf_loss_and_grads followed by mvnerf_apply_gradients: MVVNeRFRenderer.train_step on one device. */
int mvnerf_train_step(const mvnerf_train_call* call, const mvnerf_adam_state* adam, mvnerf_stream_t stream);

/* Bytes of scratch mvnerf_render_fwd needs for (B,V,R,S). */
size_t mvnerf_render_workspace_bytes(int B, int V, int R, int S);

/* MVVNeRFRenderer._call (model_v0.py:113-184): stratified depths -> coarse field -> composite ->
 * resample -> fine field -> composite, all on `stream`, no host synchronisation.
 * u_coarse, u_fine (B,R,S) are the uniforms the reference draws inside the graph
 * (nerf_utils.py:57,151), here explicit.  S must be 64.
 * Outputs: rgb, fine_rgb (B,R,3); depth, fine_depth (B,R)  (the reference's 4-tuple, :184).
 * workspace: 16-byte aligned, mvnerf_render_workspace_bytes(B,V,R,S) bytes.
 * texel_tables (optional, may be NULL = gather raw features): 2 x mvnerf_texel_table_bytes(B,V,H,W) bytes, coarse
 * net's table then fine net's.  tables_ready == 0: both are (re)built by this call; != 0: used as they are (same
 * feature maps and nets as the call that built them, e.g. the next ray chunk of the same frame). */
int mvnerf_render_fwd(const float* rays_o, const float* rays_d, const float* images, const float* features,
                      const float* intrinsics, const float* extrinsics_inv, const float* packed_coarse,
                      const float* packed_fine, const float* u_coarse, const float* u_fine, int B, int V,
                      int R, int S, int H, int W, double near_, double far_, int q7_mode, float* rgb,
                      float* depth, float* fine_rgb, float* fine_depth, void* workspace, float* texel_tables,
                      int tables_ready, mvnerf_stream_t stream);

/* mvnerf_render_fwd with both field passes on the split-bf16 kernel (mvnerf_field_eval_split): split_coarse / split_fine =
 * mvnerf_pack_net_split of the two nets; packed_coarse / packed_fine still supply biases, the per-ray layer-0 seed and the
 * texel tables.  Same outputs, same 1e-4 bar. */
int mvnerf_render_fwd_split(const float* rays_o, const float* rays_d, const float* images, const float* features,
                            const float* intrinsics, const float* extrinsics_inv, const float* packed_coarse,
                            const float* packed_fine, const void* split_coarse, const void* split_fine, const float* u_coarse,
                            const float* u_fine, int B, int V, int R, int S, int H, int W, double near_, double far_, int q7_mode,
                            float* rgb, float* depth, float* fine_rgb, float* fine_depth, void* workspace, float* texel_tables,
                            int tables_ready, mvnerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVNERF_HIP_H */
