// extern "C" training step: MVVNeRFRenderer.train_step (model_v0.py:186-197) + optimize (nerf_utils.py:8-12) behind ONE call
// each, so that a host in any language can take a training step without re-implementing the sequencing that
// thesis_clip_nerf_amd/model.py used to do in Python:
//   mvnerf_loss_and_grads  = forward with stash (coarse, fine) -> MSE + MSE and their gradients -> composite_bwd -> field
//                            backward (fine) -> resample_bwd -> composite_bwd -> field backward (coarse); the `_bwd` of
//                            mvnerf_render_fwd named in SURVEY.md 8b
//   mvnerf_apply_gradients = clip-by-value + Adam on both MLPs (+ re-packing the weight images the next forward needs)
//   mvnerf_train_step      = the two in sequence (single device; a data-parallel host all-reduces `grad` in between)
// No allocation: every intermediate lives in the caller's workspace (mvnerf_train_workspace_bytes).  Stream-ordered, no host
// synchronisation, no global state.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mvnerf_hip.h"
#include "mvnerf_kernels.h"
#include "mvnerf_math.h"

namespace {

constexpr size_t kAlign = 256;
size_t up(size_t n) { return (n + kAlign - 1) / kAlign * kAlign; }

struct TrainWs {
    float *z, *weights, *z_all, *rgbs_c, *rgbs_f, *d_rgb, *d_fine, *d_rgbs_c, *d_rgbs_f, *d_z_all, *d_w, *field_ws, *tables, *texel_grad;
    int32_t* rank;
    float *stash_c, *stash_f;
    void* bwd_scratch;
    size_t bytes;
};

TrainWs carve_train(void* base, int B, int V, int R, int S, int H, int W, int use_tables, int want_d_features) {
    TrainWs w;
    char* p = static_cast<char*>(base);
    const size_t n = (size_t)B * R * S, rays = (size_t)B * R;
    auto take = [&](size_t bytes) {
        char* q = p;
        p += up(bytes);
        return q;
    };
    w.z = reinterpret_cast<float*>(take(n * 4));
    w.weights = reinterpret_cast<float*>(take(n * 4));
    w.z_all = reinterpret_cast<float*>(take(2 * n * 4));
    w.rank = reinterpret_cast<int32_t*>(take(n * 4));
    w.rgbs_c = reinterpret_cast<float*>(take(4 * n * 4));
    w.rgbs_f = reinterpret_cast<float*>(take(8 * n * 4));
    w.d_rgb = reinterpret_cast<float*>(take(rays * 3 * 4));
    w.d_fine = reinterpret_cast<float*>(take(rays * 3 * 4));
    w.d_rgbs_c = reinterpret_cast<float*>(take(4 * n * 4));
    w.d_rgbs_f = reinterpret_cast<float*>(take(8 * n * 4));
    w.d_z_all = reinterpret_cast<float*>(take(2 * n * 4));
    w.d_w = reinterpret_cast<float*>(take(n * 4));
    w.field_ws = reinterpret_cast<float*>(take(mvnerf_field_workspace_bytes(B, V, R)));
    w.stash_c = reinterpret_cast<float*>(take(mvnerf_stash_bytes(B, V, R, S)));
    w.stash_f = reinterpret_cast<float*>(take(mvnerf_stash_bytes(B, V, R, 2 * S)));
    w.bwd_scratch = take(mvnerf_field_backward_scratch_bytes(B, V, R, 2 * S));
    const size_t tb = mvnerf_texel_table_bytes(B, V, H, W);
    w.tables = use_tables ? reinterpret_cast<float*>(take(2 * tb)) : nullptr;
    w.texel_grad = (use_tables && want_d_features) ? reinterpret_cast<float*>(take(tb)) : nullptr;
    w.bytes = (size_t)(p - static_cast<char*>(base));
    return w;
}

}  // namespace

extern "C" {

size_t mvnerf_train_workspace_bytes(int B, int V, int R, int S, int H, int W, int use_texel_tables, int want_d_features) {
    if (B <= 0 || V <= 0 || R <= 0 || S <= 0 || H < 2 || W < 2) return 0;
    return carve_train(nullptr, B, V, R, S, H, W, use_texel_tables, want_d_features).bytes;
}

#define MV_RC(x)                 \
    do {                         \
        int rc_ = (x);           \
        if (rc_ != 0) return rc_; \
    } while (0)

int mvnerf_loss_and_grads(const mvnerf_train_call* c, mvnerf_stream_t stream) {
    using mvnerf::api_fail;
    if (!c) return api_fail(MVNERF_E_ARG, "mvnerf_loss_and_grads: null call");
    if (!c->rays_o || !c->rays_d || !c->images || !c->features || !c->intrinsics || !c->extrinsics_inv || !c->u_coarse || !c->u_fine ||
        !c->labels || !c->net_coarse || !c->net_fine || !c->packed_coarse || !c->packed_fine || !c->bwd_streams_coarse ||
        !c->bwd_streams_fine || !c->loss || !c->grad || !c->rgb || !c->depth || !c->fine_rgb || !c->fine_depth || !c->workspace)
        return api_fail(MVNERF_E_ARG, "mvnerf_loss_and_grads: null pointer (only split_*, d_features may be NULL)");
    if ((c->split_coarse == nullptr) != (c->split_fine == nullptr))
        return api_fail(MVNERF_E_ARG, "mvnerf_loss_and_grads: split_coarse and split_fine must both be given or both NULL");
    const int B = c->B, V = c->V, R = c->R, S = c->S, H = c->H, W = c->W;
    if (B <= 0 || V <= 0 || R <= 0 || H < 2 || W < 2) return api_fail(MVNERF_E_ARG, "mvnerf_loss_and_grads: B=%d V=%d R=%d H=%d W=%d", B, V, R, H, W);
    if (S != 64) return api_fail(MVNERF_E_SHAPE, "mvnerf_loss_and_grads: S=%d, only the reference's n_samples=64 is built", S);
    if ((reinterpret_cast<uintptr_t>(c->workspace) & 255u) != 0)
        return api_fail(MVNERF_E_ALIGN, "mvnerf_loss_and_grads: workspace must be 256-byte aligned");
    const int want_df = c->d_features != nullptr;
    const TrainWs w = carve_train(c->workspace, B, V, R, S, H, W, c->use_texel_tables, want_df);
    if (c->workspace_bytes < w.bytes)
        return api_fail(MVNERF_E_SHAPE, "mvnerf_loss_and_grads: workspace %zu bytes, need %zu (mvnerf_train_workspace_bytes)", c->workspace_bytes, w.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int n_rays = B * R;
    const float *tab_c = nullptr, *tab_f = nullptr;
    if (w.tables) {                                                         // forward value (and d_z / d_features) through the texel tables
        float* tf = w.tables + mvnerf_texel_table_bytes(B, V, H, W) / sizeof(float);
        MV_RC(mvnerf_project_texels2(c->features, c->packed_coarse, c->packed_fine, B, V, H, W, w.tables, tf, stream));
        tab_c = w.tables;
        tab_f = tf;
    }
    // ---- forward, keeping the trunk pre-activations ----
    MV_RC(mvnerf_stratified_depths(c->u_coarse, n_rays, S, c->near_, c->far_, w.z, stream));
    if (c->split_coarse)
        MV_RC(mvnerf_field_eval_stash_split(c->rays_o, c->rays_d, w.z, c->images, c->features, tab_c, c->intrinsics, c->extrinsics_inv,
                                            c->packed_coarse, c->split_coarse, B, V, R, S, H, W, w.rgbs_c, w.stash_c, w.field_ws, stream));
    else
        MV_RC(mvnerf_field_eval_stash(c->rays_o, c->rays_d, w.z, c->images, c->features, tab_c, c->intrinsics, c->extrinsics_inv,
                                      c->packed_coarse, B, V, R, S, H, W, w.rgbs_c, w.stash_c, w.field_ws, stream));
    MV_RC(mvnerf_composite(w.z, w.rgbs_c, n_rays, S, c->rgb, c->depth, w.weights, stream));
    MV_RC(mvnerf_resample(w.z, w.weights, c->u_fine, n_rays, S, c->q7_mode, w.z_all, nullptr, nullptr, nullptr, w.rank, stream));
    if (c->split_fine)
        MV_RC(mvnerf_field_eval_stash_split(c->rays_o, c->rays_d, w.z_all, c->images, c->features, tab_f, c->intrinsics, c->extrinsics_inv,
                                            c->packed_fine, c->split_fine, B, V, R, 2 * S, H, W, w.rgbs_f, w.stash_f, w.field_ws, stream));
    else
        MV_RC(mvnerf_field_eval_stash(c->rays_o, c->rays_d, w.z_all, c->images, c->features, tab_f, c->intrinsics, c->extrinsics_inv,
                                      c->packed_fine, B, V, R, 2 * S, H, W, w.rgbs_f, w.stash_f, w.field_ws, stream));
    MV_RC(mvnerf_composite(w.z_all, w.rgbs_f, n_rays, 2 * S, c->fine_rgb, c->fine_depth, nullptr, stream));
    // ---- loss = MSE(y, rgb) + MSE(y, fine_rgb) (model_v0.py:193) and its gradient w.r.t. the two images ----
    if (mvnerf::launch_zero(c->loss, sizeof(float), st) != hipSuccess) return api_fail(1, "mvnerf_loss_and_grads: launch_zero failed");
    MV_RC(mvnerf_mse_grad(c->rgb, c->labels, (long)n_rays * 3, w.d_rgb, c->loss, stream));
    MV_RC(mvnerf_mse_grad(c->fine_rgb, c->labels, (long)n_rays * 3, w.d_fine, c->loss, stream));
    // ---- backward ----
    if (mvnerf::launch_zero(c->grad, 2 * (size_t)mvnerf::kNetParams * sizeof(float), st) != hipSuccess) return 1;
    if (want_df && mvnerf::launch_zero(c->d_features, (size_t)B * V * H * W * 256 * sizeof(float), st) != hipSuccess) return 1;
    float* gc = c->grad;
    float* gf = c->grad + mvnerf::kNetParams;
    MV_RC(mvnerf_composite_bwd(w.z_all, w.rgbs_f, w.d_fine, nullptr, nullptr, n_rays, 2 * S, w.d_rgbs_f, c->stop_fine_z ? nullptr : w.d_z_all,
                               stream));
    MV_RC(mvnerf_field_backward_table(c->rays_o, c->rays_d, w.z_all, c->images, c->features, tab_f, w.texel_grad, c->intrinsics, c->extrinsics_inv,
                                      c->net_fine, c->bwd_streams_fine, w.stash_f, w.rgbs_f, w.d_rgbs_f, B, V, R, 2 * S, H, W, w.bwd_scratch, gf,
                                      c->stop_fine_z ? nullptr : w.d_z_all, c->d_features, stream));
    if (c->fine_grad_event && hipEventRecord(static_cast<hipEvent_t>(c->fine_grad_event), st) != hipSuccess)
        return api_fail(1, "mvnerf_loss_and_grads: hipEventRecord(fine_grad_event) failed");
    const float* d_w = nullptr;
    if (!c->stop_fine_z) {                                                  // fine loss -> fine sample depths -> sample_pdf -> coarse weights (F12)
        MV_RC(mvnerf_resample_bwd(w.z, w.weights, c->u_fine, w.rank, w.d_z_all, n_rays, S, c->q7_mode, w.d_w, stream));
        d_w = w.d_w;
    }
    MV_RC(mvnerf_composite_bwd(w.z, w.rgbs_c, w.d_rgb, nullptr, d_w, n_rays, S, w.d_rgbs_c, nullptr, stream));
    MV_RC(mvnerf_field_backward_table(c->rays_o, c->rays_d, w.z, c->images, c->features, tab_c, w.texel_grad, c->intrinsics, c->extrinsics_inv,
                                      c->net_coarse, c->bwd_streams_coarse, w.stash_c, w.rgbs_c, w.d_rgbs_c, B, V, R, S, H, W, w.bwd_scratch, gc,
                                      nullptr, c->d_features, stream));
    return 0;
}

int mvnerf_apply_gradients(const mvnerf_train_call* c, const mvnerf_adam_state* a, mvnerf_stream_t stream) {
    if (!c || !a || !c->net_coarse || !c->net_fine || !c->grad || !a->m || !a->v)
        return mvnerf::api_fail(MVNERF_E_ARG, "mvnerf_apply_gradients: null pointer");
    const long n = mvnerf::kNetParams;
    float* nets[2] = {const_cast<float*>(c->net_coarse), const_cast<float*>(c->net_fine)};
    for (int k = 0; k < 2; ++k)
        MV_RC(mvnerf_adam_clip(nets[k], c->grad + k * n, a->m + k * n, a->v + k * n, n, a->lr_t, a->beta1, a->beta2, a->eps, a->clip,
                               a->update_mask ? a->update_mask + k * n : nullptr, stream));
    if (a->repack) {                                                        // the weight images the next forward / backward reads
        float* packed[2] = {const_cast<float*>(c->packed_coarse), const_cast<float*>(c->packed_fine)};
        void* split[2] = {const_cast<void*>(c->split_coarse), const_cast<void*>(c->split_fine)};
        float* bwd[2] = {const_cast<float*>(c->bwd_streams_coarse), const_cast<float*>(c->bwd_streams_fine)};
        for (int k = 0; k < 2; ++k) {
            if (packed[k]) MV_RC(mvnerf_pack_net(nets[k], packed[k], stream));
            if (split[k]) MV_RC(mvnerf_pack_net_split(nets[k], split[k], stream));
            if (bwd[k]) MV_RC(mvnerf_pack_bwd_streams(nets[k], bwd[k], stream));
        }
    }
    return 0;
}

int mvnerf_train_step(const mvnerf_train_call* c, const mvnerf_adam_state* a, mvnerf_stream_t stream) {
    MV_RC(mvnerf_loss_and_grads(c, stream));
    return mvnerf_apply_gradients(c, a, stream);
}

}  // extern "C"
