// extern "C" surface of libmvnerf_hip.so (include/mvnerf_hip.h): argument validation, launch
// orchestration, error reporting.  No allocation, no host synchronisation, no global state besides
// the thread-local error string; safe to call from one process per GPU on any stream.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <atomic>

#include "../../include/mvnerf_hip.h"
#include "mvnerf_kernels.h"
#include "mvnerf_math.h"
#include "mvnerf_pack.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    return fail((int)e, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace

namespace mvnerf {
int api_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace mvnerf

namespace {

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct Workspace {          // carve-up of the caller's scratch for mvnerf_render_fwd (floats)
    float *z, *rgbs_c, *weights, *z_all, *rgbs_f, *dir_bias;
    size_t bytes;
};

Workspace carve(void* base, long n_rays, int V, int S) {
    Workspace w;
    float* p = static_cast<float*>(base);
    const size_t n = (size_t)n_rays * S;
    w.z = p;            p += n;
    w.weights = p;      p += n;
    w.z_all = p;        p += 2 * n;
    w.rgbs_c = p;       p += 4 * n;
    w.rgbs_f = p;       p += 8 * n;
    w.dir_bias = p;     p += (size_t)n_rays * V * 128;
    w.bytes = (size_t)(p - static_cast<float*>(base)) * sizeof(float);
    return w;
}

}  // namespace

extern "C" {

int mvnerf_abi_version(void) { return 1; }

const char* mvnerf_last_error(void) { return g_err; }

size_t mvnerf_packed_net_floats(void) { return (size_t)mvnerf::kPackTotal; }

int mvnerf_pack_net(const float* net_keras, float* packed, mvnerf_stream_t stream) {
    if (!net_keras || !packed) return fail(MVNERF_E_ARG, "mvnerf_pack_net: null pointer");
    if (!aligned16(packed)) return fail(MVNERF_E_ALIGN, "mvnerf_pack_net: packed must be 16-byte aligned");
    return hip_status(mvnerf::launch_pack_net(net_keras, packed, static_cast<hipStream_t>(stream)), "mvnerf_pack_net");
}

int mvnerf_get_rays(const double* m3x3_host, const double* origin_host, const float* u, const float* v, int n_rays,
                    int width, int normalize, float* rays_o, float* rays_d, double* rays_d64,
                    mvnerf_stream_t stream) {
    if (!m3x3_host || !origin_host || !rays_o || !rays_d) return fail(MVNERF_E_ARG, "mvnerf_get_rays: null pointer");
    if ((u == nullptr) != (v == nullptr)) return fail(MVNERF_E_ARG, "mvnerf_get_rays: u and v must both be given or both NULL");
    if (n_rays <= 0 || (!u && width <= 0)) return fail(MVNERF_E_ARG, "mvnerf_get_rays: n_rays=%d width=%d", n_rays, width);
    return hip_status(mvnerf::launch_get_rays(m3x3_host, origin_host, u, v, n_rays, width, normalize, rays_o, rays_d,
                                              rays_d64, static_cast<hipStream_t>(stream)),
                      "mvnerf_get_rays");
}

int mvnerf_stratified_depths(const float* u, int n_rays, int n_samples, double near_, double far_, float* z,
                             mvnerf_stream_t stream) {
    if (!u || !z) return fail(MVNERF_E_ARG, "mvnerf_stratified_depths: null pointer");
    if (n_rays <= 0 || n_samples <= 0) return fail(MVNERF_E_ARG, "mvnerf_stratified_depths: n_rays=%d n_samples=%d", n_rays, n_samples);
    return hip_status(mvnerf::launch_stratified(u, (long)n_rays * n_samples, n_samples, near_, far_, z,
                                                static_cast<hipStream_t>(stream)),
                      "mvnerf_stratified_depths");
}

static int field_eval_impl(const float* rays_o, const float* rays_d, const float* z, const float* images,
                           const float* features, const float* texel_table, const float* intrinsics,
                           const float* extrinsics_inv, const float* packed_net, int B, int V, int R, int S, int H, int W,
                           float* rgbs, int32_t* tap_idx, float* pix, float* embedding, float* acts_per_view,
                           float* acts_fused, void* workspace, mvnerf_stream_t stream) {
    if (texel_table && !aligned16(texel_table)) return fail(MVNERF_E_ALIGN, "mvnerf_field_eval_table: texel_table must be 16-byte aligned");
    if (!rays_o || !rays_d || !z || !images || !features || !intrinsics || !extrinsics_inv || !packed_net || !rgbs || !workspace)
        return fail(MVNERF_E_ARG, "mvnerf_field_eval: null pointer");
    if (B <= 0 || V <= 0 || R <= 0 || S <= 0) return fail(MVNERF_E_ARG, "mvnerf_field_eval: B=%d V=%d R=%d S=%d", B, V, R, S);
    if (H < 2 || W < 2) return fail(MVNERF_E_SHAPE, "mvnerf_field_eval: source image %dx%d, need H,W >= 2 (bilinear taps)", H, W);
    const long total = (long)B * R * S;
    if (total >= (1L << 31) || (long)B * V * H * W >= (1L << 31))
        return fail(MVNERF_E_SHAPE, "mvnerf_field_eval: B*R*S=%ld or B*V*H*W too large for int32 indices", total);
    if (!aligned16(features) || !aligned16(packed_net) || !aligned16(rgbs) || (tap_idx && !aligned16(tap_idx)) ||
        (embedding && !aligned16(embedding)) || (acts_per_view && !aligned16(acts_per_view)) ||
        (acts_fused && !aligned16(acts_fused)) || !aligned16(workspace))
        return fail(MVNERF_E_ALIGN, "mvnerf_field_eval: features, packed_net, rgbs, tap_idx, embedding must be 16-byte aligned");
    mvnerf::FieldParams p = {};
    p.texel_table = texel_table;
    p.rays_o = rays_o; p.rays_d = rays_d; p.z = z; p.images = images; p.features = features;
    p.k4 = intrinsics; p.einv = extrinsics_inv; p.net = packed_net; p.rgbs = rgbs; p.tap_idx = tap_idx; p.pix = pix; p.embedding = embedding; p.acts_view = acts_per_view; p.acts_fused = acts_fused; p.dir_bias = static_cast<float*>(workspace); p.stash = nullptr; p.stash_stride = 0;
    p.B = B; p.V = V; p.R = R; p.S = S; p.H = H; p.W = W;
    p.total = total;
    p.n_tiles = (total + 31) / 32;
    return hip_status(mvnerf::launch_field_eval(p, static_cast<hipStream_t>(stream)), "mvnerf_field_eval");
}

int mvnerf_field_eval(const float* rays_o, const float* rays_d, const float* z, const float* images,
                      const float* features, const float* intrinsics, const float* extrinsics_inv,
                      const float* packed_net, int B, int V, int R, int S, int H, int W, float* rgbs,
                      int32_t* tap_idx, float* pix, float* embedding, float* acts_per_view, float* acts_fused,
                      void* workspace, mvnerf_stream_t stream) {
    return field_eval_impl(rays_o, rays_d, z, images, features, nullptr, intrinsics, extrinsics_inv, packed_net, B, V, R, S,
                           H, W, rgbs, tap_idx, pix, embedding, acts_per_view, acts_fused, workspace, stream);
}

size_t mvnerf_texel_table_bytes(int B, int V, int H, int W) {
    if (B <= 0 || V <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * V * H * W * 128 * sizeof(float);
}

int mvnerf_project_texels2(const float* features, const float* packed_net, const float* packed_net_b, int B, int V, int H, int W,
                           float* texel_table, float* texel_table_b, mvnerf_stream_t stream) {
    if (!features || !packed_net || !texel_table) return fail(MVNERF_E_ARG, "mvnerf_project_texels: null pointer");
    if ((packed_net_b == nullptr) != (texel_table_b == nullptr))
        return fail(MVNERF_E_ARG, "mvnerf_project_texels2: the second net and the second table go together");
    if ((packed_net_b && !aligned16(packed_net_b)) || (texel_table_b && !aligned16(texel_table_b)))
        return fail(MVNERF_E_ALIGN, "mvnerf_project_texels2: packed_net_b, texel_table_b must be 16-byte aligned");
    if (B <= 0 || V <= 0 || H < 2 || W < 2) return fail(MVNERF_E_ARG, "mvnerf_project_texels: B=%d V=%d H=%d W=%d", B, V, H, W);
    if ((long)B * V * H * W >= (1L << 31)) return fail(MVNERF_E_SHAPE, "mvnerf_project_texels: B*V*H*W too large for int32 indices");
    if (!aligned16(features) || !aligned16(packed_net) || !aligned16(texel_table))
        return fail(MVNERF_E_ALIGN, "mvnerf_project_texels: features, packed_net, texel_table must be 16-byte aligned");
    return hip_status(mvnerf::launch_project_texels(features, packed_net, packed_net_b, (long)B * V * H * W, texel_table,
                                                    texel_table_b, static_cast<hipStream_t>(stream)),
                      "mvnerf_project_texels");
}

int mvnerf_project_texels(const float* features, const float* packed_net, int B, int V, int H, int W, float* texel_table,
                          mvnerf_stream_t stream) {
    return mvnerf_project_texels2(features, packed_net, nullptr, B, V, H, W, texel_table, nullptr, stream);
}

int mvnerf_field_eval_table(const float* rays_o, const float* rays_d, const float* z, const float* images,
                            const float* features, const float* texel_table, const float* intrinsics,
                            const float* extrinsics_inv, const float* packed_net, int B, int V, int R, int S, int H, int W,
                            float* rgbs, int32_t* tap_idx, float* pix, float* embedding, float* acts_per_view,
                            float* acts_fused, void* workspace, mvnerf_stream_t stream) {
    if (!texel_table) return fail(MVNERF_E_ARG, "mvnerf_field_eval_table: null texel_table");
    return field_eval_impl(rays_o, rays_d, z, images, features, texel_table, intrinsics, extrinsics_inv, packed_net, B, V, R,
                           S, H, W, rgbs, tap_idx, pix, embedding, acts_per_view, acts_fused, workspace, stream);
}

size_t mvnerf_packed_net_bf16_bytes(void) { return mvnerf::packed_net_bf16_bytes(); }

int mvnerf_pack_net_bf16(const float* net_keras, void* packed16, mvnerf_stream_t stream) {
    if (!net_keras || !packed16) return fail(MVNERF_E_ARG, "mvnerf_pack_net_bf16: null pointer");
    if (!aligned16(packed16)) return fail(MVNERF_E_ALIGN, "mvnerf_pack_net_bf16: packed16 must be 16-byte aligned");
    return hip_status(mvnerf::launch_pack_net_bf16(net_keras, packed16, static_cast<hipStream_t>(stream)), "mvnerf_pack_net_bf16");
}

int mvnerf_project_texels_bf16(const float* features, const void* packed16, const void* packed16_b, int B, int V, int H, int W,
                               float* texel_table, float* texel_table_b, mvnerf_stream_t stream) {
    if (!features || !packed16 || !texel_table) return fail(MVNERF_E_ARG, "mvnerf_project_texels_bf16: null pointer");
    if ((packed16_b == nullptr) != (texel_table_b == nullptr))
        return fail(MVNERF_E_ARG, "mvnerf_project_texels_bf16: the second net and the second table go together");
    if ((packed16_b && !aligned16(packed16_b)) || (texel_table_b && !aligned16(texel_table_b)))
        return fail(MVNERF_E_ALIGN, "mvnerf_project_texels_bf16: packed16_b, texel_table_b must be 16-byte aligned");
    if (B <= 0 || V <= 0 || H < 2 || W < 2) return fail(MVNERF_E_ARG, "mvnerf_project_texels_bf16: B=%d V=%d H=%d W=%d", B, V, H, W);
    if ((long)B * V * H * W >= (1L << 31)) return fail(MVNERF_E_SHAPE, "mvnerf_project_texels_bf16: B*V*H*W too large for int32 indices");
    if (!aligned16(features) || !aligned16(packed16) || !aligned16(texel_table))
        return fail(MVNERF_E_ALIGN, "mvnerf_project_texels_bf16: features, packed16, texel_table must be 16-byte aligned");
    return hip_status(mvnerf::launch_project_texels_bf16(features, packed16, packed16_b, (long)B * V * H * W, texel_table, texel_table_b,
                                                         static_cast<hipStream_t>(stream)),
                      "mvnerf_project_texels_bf16");
}

}  // extern "C"

static int field_eval_bf16_impl(const char* who, bool maps_bf16, const float* rays_o, const float* rays_d, const float* z, const float* images,
                           const float* features, const float* texel_table, const float* intrinsics, const float* extrinsics_inv,
                           const float* packed_net, const void* packed16, int B, int V, int R, int S, int H, int W,
                           float* rgbs, int32_t* tap_idx, float* embedding, float* acts_fused, void* workspace,
                           mvnerf_stream_t stream) {
    if (acts_fused && !aligned16(acts_fused)) return fail(MVNERF_E_ALIGN, "%s: acts_fused must be 16-byte aligned", who);
    if (texel_table && !aligned16(texel_table)) return fail(MVNERF_E_ALIGN, "%s: texel_table must be 16-byte aligned", who);
    if (!rays_o || !rays_d || !z || !images || !features || !intrinsics || !extrinsics_inv || !packed_net || !packed16 || !rgbs || !workspace)
        return fail(MVNERF_E_ARG, "%s: null pointer", who);
    if (B <= 0 || V <= 0 || R <= 0 || S <= 0) return fail(MVNERF_E_ARG, "%s: B=%d V=%d R=%d S=%d", who, B, V, R, S);
    if (H < 2 || W < 2) return fail(MVNERF_E_SHAPE, "%s: source image %dx%d, need H,W >= 2", who, H, W);
    const long total = (long)B * R * S;
    if (total >= (1L << 31) || (long)B * V * H * W >= (1L << 31)) return fail(MVNERF_E_SHAPE, "%s: sizes too large for int32 indices", who);
    if (!aligned16(features) || !aligned16(packed_net) || !aligned16(packed16) || !aligned16(rgbs) || (tap_idx && !aligned16(tap_idx)) ||
        (embedding && !aligned16(embedding)) || !aligned16(workspace))
        return fail(MVNERF_E_ALIGN, "%s: features, packed nets, rgbs, tap_idx, embedding, workspace must be 16-byte aligned", who);
    mvnerf::FieldParams p = {};
    p.rays_o = rays_o; p.rays_d = rays_d; p.z = z; p.images = images; p.features = features;
    p.k4 = intrinsics; p.einv = extrinsics_inv; p.net = packed_net; p.rgbs = rgbs; p.tap_idx = tap_idx; p.embedding = embedding;
    p.acts_fused = acts_fused;
    p.texel_table = texel_table;
    p.dir_bias = static_cast<float*>(workspace);
    p.B = B; p.V = V; p.R = R; p.S = S; p.H = H; p.W = W;
    p.total = total;
    p.n_tiles = (total + 31) / 32;
    return hip_status(mvnerf::launch_field_eval_bf16(p, packed16, static_cast<hipStream_t>(stream), maps_bf16), who);
}

extern "C" {

int mvnerf_project_texels_bf16maps(const void* features_bf16, const void* packed16, const void* packed16_b, int B, int V, int H, int W,
                                   float* texel_table, float* texel_table_b, mvnerf_stream_t stream) {
    if (!features_bf16 || !packed16 || !texel_table) return fail(MVNERF_E_ARG, "mvnerf_project_texels_bf16maps: null pointer");
    if ((packed16_b == nullptr) != (texel_table_b == nullptr))
        return fail(MVNERF_E_ARG, "mvnerf_project_texels_bf16maps: the second net and the second table go together");
    if ((packed16_b && !aligned16(packed16_b)) || (texel_table_b && !aligned16(texel_table_b)))
        return fail(MVNERF_E_ALIGN, "mvnerf_project_texels_bf16maps: packed16_b, texel_table_b must be 16-byte aligned");
    if (B <= 0 || V <= 0 || H < 2 || W < 2) return fail(MVNERF_E_ARG, "mvnerf_project_texels_bf16maps: B=%d V=%d H=%d W=%d", B, V, H, W);
    if ((long)B * V * H * W >= (1L << 31)) return fail(MVNERF_E_SHAPE, "mvnerf_project_texels_bf16maps: B*V*H*W too large for int32 indices");
    if (!aligned16(features_bf16) || !aligned16(packed16) || !aligned16(texel_table))
        return fail(MVNERF_E_ALIGN, "mvnerf_project_texels_bf16maps: features_bf16, packed16, texel_table must be 16-byte aligned");
    return hip_status(mvnerf::launch_project_texels_bf16maps(features_bf16, packed16, packed16_b, (long)B * V * H * W, texel_table, texel_table_b,
                                                             static_cast<hipStream_t>(stream)),
                      "mvnerf_project_texels_bf16maps");
}

int mvnerf_field_eval_bf16(const float* rays_o, const float* rays_d, const float* z, const float* images,
                           const float* features, const float* texel_table, const float* intrinsics, const float* extrinsics_inv,
                           const float* packed_net, const void* packed16, int B, int V, int R, int S, int H, int W,
                           float* rgbs, int32_t* tap_idx, float* embedding, float* acts_fused, void* workspace,
                           mvnerf_stream_t stream) {
    return field_eval_bf16_impl("mvnerf_field_eval_bf16", false, rays_o, rays_d, z, images, features, texel_table, intrinsics, extrinsics_inv,
                                packed_net, packed16, B, V, R, S, H, W, rgbs, tap_idx, embedding, acts_fused, workspace, stream);
}

int mvnerf_field_eval_bf16maps(const float* rays_o, const float* rays_d, const float* z, const float* images,
                               const void* features_bf16, const float* texel_table, const float* intrinsics, const float* extrinsics_inv,
                               const float* packed_net, const void* packed16, int B, int V, int R, int S, int H, int W,
                               float* rgbs, int32_t* tap_idx, float* embedding, float* acts_fused, void* workspace,
                               mvnerf_stream_t stream) {
    return field_eval_bf16_impl("mvnerf_field_eval_bf16maps", true, rays_o, rays_d, z, images, static_cast<const float*>(features_bf16), texel_table,
                                intrinsics, extrinsics_inv, packed_net, packed16, B, V, R, S, H, W, rgbs, tap_idx, embedding, acts_fused,
                                workspace, stream);
}

size_t mvnerf_packed_net_split_bytes(void) { return mvnerf::packed_net_split_bytes(); }

int mvnerf_pack_net_split(const float* net_keras, void* packed_split, mvnerf_stream_t stream) {
    if (!net_keras || !packed_split) return fail(MVNERF_E_ARG, "mvnerf_pack_net_split: null pointer");
    if (!aligned16(packed_split)) return fail(MVNERF_E_ALIGN, "mvnerf_pack_net_split: packed_split must be 16-byte aligned");
    return hip_status(mvnerf::launch_pack_net_split(net_keras, packed_split, static_cast<hipStream_t>(stream)), "mvnerf_pack_net_split");
}

int mvnerf_field_eval_split(const float* rays_o, const float* rays_d, const float* z, const float* images,
                            const float* features, const float* texel_table, const float* intrinsics, const float* extrinsics_inv,
                            const float* packed_net, const void* packed_split, int B, int V, int R, int S, int H, int W,
                            float* rgbs, int32_t* tap_idx, float* pix, float* embedding, float* acts_per_view, float* acts_fused,
                            void* workspace, mvnerf_stream_t stream) {
    if (!rays_o || !rays_d || !z || !images || !features || !intrinsics || !extrinsics_inv || !packed_net || !packed_split || !rgbs || !workspace)
        return fail(MVNERF_E_ARG, "mvnerf_field_eval_split: null pointer");
    if (B <= 0 || V <= 0 || R <= 0 || S <= 0) return fail(MVNERF_E_ARG, "mvnerf_field_eval_split: B=%d V=%d R=%d S=%d", B, V, R, S);
    if (H < 2 || W < 2) return fail(MVNERF_E_SHAPE, "mvnerf_field_eval_split: source image %dx%d, need H,W >= 2 (bilinear taps)", H, W);
    const long total = (long)B * R * S;
    if (total >= (1L << 31) || (long)B * V * H * W >= (1L << 31))
        return fail(MVNERF_E_SHAPE, "mvnerf_field_eval_split: B*R*S=%ld or B*V*H*W too large for int32 indices", total);
    if (!aligned16(features) || !aligned16(packed_net) || !aligned16(packed_split) || !aligned16(rgbs) || !aligned16(workspace) ||
        (texel_table && !aligned16(texel_table)) || (tap_idx && !aligned16(tap_idx)) || (embedding && !aligned16(embedding)) ||
        (acts_per_view && !aligned16(acts_per_view)) || (acts_fused && !aligned16(acts_fused)))
        return fail(MVNERF_E_ALIGN, "mvnerf_field_eval_split: features, packed nets, texel_table, rgbs, tap_idx, embedding, acts, workspace must be 16-byte aligned");
    mvnerf::FieldParams p = {};
    p.rays_o = rays_o; p.rays_d = rays_d; p.z = z; p.images = images; p.features = features;
    p.k4 = intrinsics; p.einv = extrinsics_inv; p.net = packed_net; p.rgbs = rgbs; p.tap_idx = tap_idx; p.pix = pix;
    p.embedding = embedding; p.acts_view = acts_per_view; p.acts_fused = acts_fused;
    p.texel_table = texel_table;
    p.dir_bias = static_cast<float*>(workspace);
    p.B = B; p.V = V; p.R = R; p.S = S; p.H = H; p.W = W;
    p.total = total;
    p.n_tiles = (total + 31) / 32;
    return hip_status(mvnerf::launch_field_eval_split(p, packed_split, static_cast<hipStream_t>(stream)), "mvnerf_field_eval_split");
}

int mvnerf_composite(const float* z, const float* rgbs, int n_rays, int S, float* rgb, float* depth, float* weights,
                     mvnerf_stream_t stream) {
    if (!z || !rgbs || !rgb || !depth) return fail(MVNERF_E_ARG, "mvnerf_composite: null pointer");
    if (n_rays <= 0) return fail(MVNERF_E_ARG, "mvnerf_composite: n_rays=%d", n_rays);
    if (S % 64 != 0 || S < 64 || S > 256) return fail(MVNERF_E_SHAPE, "mvnerf_composite: S=%d, supported: 64,128,192,256", S);
    if (!aligned16(rgbs)) return fail(MVNERF_E_ALIGN, "mvnerf_composite: rgbs must be 16-byte aligned");
    return hip_status(mvnerf::launch_composite(z, rgbs, n_rays, S, rgb, depth, weights, static_cast<hipStream_t>(stream)),
                      "mvnerf_composite");
}

int mvnerf_resample(const float* z, const float* weights, const float* u_fine, int n_rays, int S, int q7_mode,
                    float* z_all, float* z_fine, int32_t* above, int32_t* below, int32_t* fine_rank,
                    mvnerf_stream_t stream) {
    if (!z || !weights || !u_fine || !z_all) return fail(MVNERF_E_ARG, "mvnerf_resample: null pointer");
    if (n_rays <= 0) return fail(MVNERF_E_ARG, "mvnerf_resample: n_rays=%d", n_rays);
    if (S != 64) return fail(MVNERF_E_SHAPE, "mvnerf_resample: S=%d, only the reference's n_samples=64 is built", S);
    if (q7_mode != MVNERF_Q7_ZERO && q7_mode != MVNERF_Q7_CLAMP) return fail(MVNERF_E_ARG, "mvnerf_resample: q7_mode=%d", q7_mode);
    return hip_status(mvnerf::launch_resample(z, weights, u_fine, n_rays, q7_mode, z_all, z_fine, above, below, fine_rank,
                                              static_cast<hipStream_t>(stream)),
                      "mvnerf_resample");
}

int mvnerf_points_on_rays(const float* rays_o, const float* rays_d, const float* z, int n_rays, int S, float* world,
                          mvnerf_stream_t stream) {
    if (!rays_o || !rays_d || !z || !world) return fail(MVNERF_E_ARG, "mvnerf_points_on_rays: null pointer");
    if (n_rays <= 0 || S <= 0) return fail(MVNERF_E_ARG, "mvnerf_points_on_rays: n_rays=%d S=%d", n_rays, S);
    return hip_status(mvnerf::launch_points_on_rays(rays_o, rays_d, z, (long)n_rays * S, S, world, static_cast<hipStream_t>(stream)),
                      "mvnerf_points_on_rays");
}

int mvnerf_project_points(const float* world, const float* intrinsics, const float* extrinsics_inv, int B, int V, int N,
                          float* pixel_locations, float* camera_points, mvnerf_stream_t stream) {
    if (!world || !intrinsics || !extrinsics_inv || !pixel_locations || !camera_points)
        return fail(MVNERF_E_ARG, "mvnerf_project_points: null pointer");
    if (B <= 0 || V <= 0 || N <= 0) return fail(MVNERF_E_ARG, "mvnerf_project_points: B=%d V=%d N=%d", B, V, N);
    return hip_status(mvnerf::launch_project_points(world, intrinsics, extrinsics_inv, B, V, N, pixel_locations, camera_points,
                                                    static_cast<hipStream_t>(stream)),
                      "mvnerf_project_points");
}

int mvnerf_camera_directions(const float* dirs, const float* extrinsics_inv, int B, int V, int R, float* out,
                             mvnerf_stream_t stream) {
    if (!dirs || !extrinsics_inv || !out) return fail(MVNERF_E_ARG, "mvnerf_camera_directions: null pointer");
    if (B <= 0 || V <= 0 || R <= 0) return fail(MVNERF_E_ARG, "mvnerf_camera_directions: B=%d V=%d R=%d", B, V, R);
    return hip_status(mvnerf::launch_camera_directions(dirs, extrinsics_inv, B, V, R, out, static_cast<hipStream_t>(stream)),
                      "mvnerf_camera_directions");
}

int mvnerf_position_encoding(const float* x, long n_elems, int n_freq, float pos_encoding_freq, float* out,
                             mvnerf_stream_t stream) {
    if (!x || !out) return fail(MVNERF_E_ARG, "mvnerf_position_encoding: null pointer");
    if (n_elems <= 0 || n_freq <= 0 || n_freq > 32) return fail(MVNERF_E_ARG, "mvnerf_position_encoding: n_elems=%ld n_freq=%d", n_elems, n_freq);
    return hip_status(mvnerf::launch_position_encoding(x, n_elems, n_freq, pos_encoding_freq, out, static_cast<hipStream_t>(stream)),
                      "mvnerf_position_encoding");
}

int mvnerf_bilinear_gather(const float* images, const float* features, const float* pixel_locations, int BV, int Q, int H,
                           int W, float* out, int32_t* tap_idx, mvnerf_stream_t stream) {
    if (!images || !features || !pixel_locations || !out) return fail(MVNERF_E_ARG, "mvnerf_bilinear_gather: null pointer");
    if (BV <= 0 || Q <= 0) return fail(MVNERF_E_ARG, "mvnerf_bilinear_gather: BV=%d Q=%d", BV, Q);
    if (H < 2 || W < 2) return fail(MVNERF_E_SHAPE, "mvnerf_bilinear_gather: grid %dx%d, need H,W >= 2", H, W);
    if ((long)BV * H * W >= (1L << 31)) return fail(MVNERF_E_SHAPE, "mvnerf_bilinear_gather: BV*H*W too large for int32 indices");
    return hip_status(mvnerf::launch_bilinear_gather(images, features, pixel_locations, BV, Q, H, W, out, tap_idx,
                                                     static_cast<hipStream_t>(stream)),
                      "mvnerf_bilinear_gather");
}

int mvnerf_sigma_to_alpha(const float* sigma, const float* dists, long n, float* alpha, mvnerf_stream_t stream) {
    if (!sigma || !dists || !alpha) return fail(MVNERF_E_ARG, "mvnerf_sigma_to_alpha: null pointer");
    if (n <= 0) return fail(MVNERF_E_ARG, "mvnerf_sigma_to_alpha: n=%ld", n);
    return hip_status(mvnerf::launch_sigma_to_alpha(sigma, dists, n, alpha, static_cast<hipStream_t>(stream)), "mvnerf_sigma_to_alpha");
}

int mvnerf_sample_pdf(const float* bins, const float* weights, const float* u, int n_rays, int n_bins, int n_samples,
                      int q7_mode, float* samples, int32_t* above, int32_t* below, mvnerf_stream_t stream) {
    if (!bins || !weights || !u || !samples) return fail(MVNERF_E_ARG, "mvnerf_sample_pdf: null pointer");
    if (n_rays <= 0) return fail(MVNERF_E_ARG, "mvnerf_sample_pdf: n_rays=%d", n_rays);
    if (n_bins != 63 || n_samples != 64)
        return fail(MVNERF_E_SHAPE, "mvnerf_sample_pdf: n_bins=%d n_samples=%d, only the reference's 63 bins / 64 samples is built", n_bins, n_samples);
    if (q7_mode != MVNERF_Q7_ZERO && q7_mode != MVNERF_Q7_CLAMP) return fail(MVNERF_E_ARG, "mvnerf_sample_pdf: q7_mode=%d", q7_mode);
    return hip_status(mvnerf::launch_sample_pdf(bins, weights, u, n_rays, q7_mode, samples, above, below, static_cast<hipStream_t>(stream)),
                      "mvnerf_sample_pdf");
}

int mvnerf_readout(const float* embedding, const float* wr, const float* br, long n, float* rgbs, mvnerf_stream_t stream) {
    if (!embedding || !wr || !br || !rgbs) return fail(MVNERF_E_ARG, "mvnerf_readout: null pointer");
    if (n <= 0) return fail(MVNERF_E_ARG, "mvnerf_readout: n=%ld", n);
    return hip_status(mvnerf::launch_readout(embedding, wr, br, n, rgbs, static_cast<hipStream_t>(stream)), "mvnerf_readout");
}

int mvnerf_finish_view(const float* rgb, const float* depth, long n, float* minmax_scratch, uint8_t* rgb8, uint8_t* depth8,
                       mvnerf_stream_t stream) {
    if (!rgb || !depth || !minmax_scratch || !rgb8 || !depth8) return fail(MVNERF_E_ARG, "mvnerf_finish_view: null pointer");
    if (n <= 0) return fail(MVNERF_E_ARG, "mvnerf_finish_view: n=%ld", n);
    return hip_status(mvnerf::launch_finish_view(rgb, depth, n, minmax_scratch, rgb8, depth8, static_cast<hipStream_t>(stream)),
                      "mvnerf_finish_view");
}

// ---- training --------------------------------------------------------------------------------------------
namespace {
long tiles_for(int B, int R, int S) { return ((long)B * R * S + 31) / 32; }
constexpr int kBwdMaxWGs = 512;      // resident workgroups of the dW kernels (2 per CU)
constexpr int kFusedBwdWGs = 512;    // fused dX+dW kernel: 2 waves/SIMD by registers -> 2 resident workgroups per CU
static_assert(kBwdMaxWGs == kFusedBwdWGs, "partial_floats() assumes one workgroup budget for every weight-gradient kernel");
}  // namespace

size_t mvnerf_stash_bytes(int B, int V, int R, int S) {
    if (B <= 0 || V <= 0 || R <= 0 || S <= 0) return 0;
    return (size_t)(7 * V + 7) * tiles_for(B, R, S) * 128 * 32 * sizeof(float);     // 7 per-view + 7 fused slots
}

// Per-workgroup partials of a weight-gradient span, summed in workgroup order by reduce_partials_kernel.  The three users:
// dw0_split8_kernel (layer 0: 379 x 128 + 128 floats per workgroup, kBwdMaxWGs / 2 = 256 workgroups at most, never more than there are
// view tiles), dense_bwd_split8_kernel (128 x 128 + 128, kFusedBwdWGs / 2) and the read-out's dw_tile_kernel (516 floats, kBwdMaxWGs).
// Behind them: 14 x 64 floats, max |g| of the gradient tensor each layer launch of the training backward reads (the power-of-two scale
// of its fp16 cut; train_ops.hip, MVT_BWD_F16).
constexpr int kAmaxTensors = 14;
static size_t partial_only_floats(long n_tiles, int V) {
    const long wg0 = n_tiles * V < kBwdMaxWGs / 2 ? n_tiles * V : kBwdMaxWGs / 2;
    const long wgr = n_tiles < kBwdMaxWGs ? n_tiles : kBwdMaxWGs;
    const size_t a = (size_t)wg0 * (mvnerf::kIn * mvnerf::kHidden + mvnerf::kHidden), b = (size_t)wgr * 516;
    return ((a > b ? a : b) + 3) & ~(size_t)3;
}
static size_t partial_floats(long n_tiles, int V) { return partial_only_floats(n_tiles, V) + (size_t)kAmaxTensors * mvnerf::kBwdAmaxSlots; }
std::atomic<int> g_deterministic{0};

size_t mvnerf_field_backward_scratch_bytes(int B, int V, int R, int S) {
    if (B <= 0 || V <= 0 || R <= 0 || S <= 0) return 0;
    return ((size_t)tiles_for(B, R, S) * ((size_t)3 * V * 128 + 32) * 32 + partial_floats(tiles_for(B, R, S), V)) * sizeof(float);
}

int mvnerf_set_deterministic(int on) { return g_deterministic.exchange(on ? 1 : 0); }

int mvnerf_set_split_kernel(int which) {
    const int prev = mvnerf::set_split_kernel(which);
    if (prev < 0) return fail(MVNERF_E_ARG, "mvnerf_set_split_kernel: which=%d (0 f16x3, 1 bf16x6, 2 bf16x6 as 32x32x16)", which);
    return prev;
}

int mvnerf_field_eval_stash(const float* rays_o, const float* rays_d, const float* z, const float* images,
                            const float* features, const float* texel_table, const float* intrinsics,
                            const float* extrinsics_inv, const float* packed_net, int B, int V, int R, int S, int H, int W,
                            float* rgbs, float* stash, void* workspace, mvnerf_stream_t stream) {
    if (texel_table && !aligned16(texel_table)) return fail(MVNERF_E_ALIGN, "mvnerf_field_eval_stash: texel_table must be 16-byte aligned");
    if (!rays_o || !rays_d || !z || !images || !features || !intrinsics || !extrinsics_inv || !packed_net || !rgbs || !stash || !workspace)
        return fail(MVNERF_E_ARG, "mvnerf_field_eval_stash: null pointer");
    if (B <= 0 || V <= 0 || R <= 0 || S <= 0 || H < 2 || W < 2) return fail(MVNERF_E_ARG, "mvnerf_field_eval_stash: B=%d V=%d R=%d S=%d H=%d W=%d", B, V, R, S, H, W);
    if (V > 1 && ((long)R * S) % 32 != 0) return fail(MVNERF_E_SHAPE, "mvnerf_field_eval_stash: R*S=%ld must be a multiple of 32 when V > 1", (long)R * S);
    const long total = (long)B * R * S;
    if (total >= (1L << 31)) return fail(MVNERF_E_SHAPE, "mvnerf_field_eval_stash: B*R*S too large");
    if ((long)V * ((total + 31) / 32) >= (1L << 18))      // stash slots are addressed with 32-bit byte offsets (16 KiB per tile)
        return fail(MVNERF_E_SHAPE, "mvnerf_field_eval_stash: V*B*R*S/32 = %ld tiles per stash slot, at most 262143", (long)V * ((total + 31) / 32));
    if (!aligned16(features) || !aligned16(packed_net) || !aligned16(rgbs) || !aligned16(stash) || !aligned16(workspace))
        return fail(MVNERF_E_ALIGN, "mvnerf_field_eval_stash: features, packed_net, rgbs, stash, workspace must be 16-byte aligned");
    mvnerf::FieldParams p = {};
    p.rays_o = rays_o; p.rays_d = rays_d; p.z = z; p.images = images; p.features = features;
    p.k4 = intrinsics; p.einv = extrinsics_inv; p.net = packed_net; p.rgbs = rgbs;
    p.texel_table = texel_table;
    p.dir_bias = static_cast<float*>(workspace);
    p.B = B; p.V = V; p.R = R; p.S = S; p.H = H; p.W = W;
    p.total = total;
    p.n_tiles = (total + 31) / 32;
    p.stash = stash;
    p.stash_stride = (long)V * p.n_tiles * 4096;
    p.stash_fused = stash + 7 * p.stash_stride;
    p.stash_fused_stride = p.n_tiles * 4096;
    return hip_status(mvnerf::launch_field_eval(p, static_cast<hipStream_t>(stream)), "mvnerf_field_eval_stash");
}

int mvnerf_field_eval_stash_split(const float* rays_o, const float* rays_d, const float* z, const float* images,
                                  const float* features, const float* texel_table, const float* intrinsics,
                                  const float* extrinsics_inv, const float* packed_net, const void* packed_split, int B, int V, int R,
                                  int S, int H, int W, float* rgbs, float* stash, void* workspace, mvnerf_stream_t stream) {
    if (texel_table && !aligned16(texel_table)) return fail(MVNERF_E_ALIGN, "mvnerf_field_eval_stash_split: texel_table must be 16-byte aligned");
    if (!rays_o || !rays_d || !z || !images || !features || !intrinsics || !extrinsics_inv || !packed_net || !packed_split || !rgbs || !stash || !workspace)
        return fail(MVNERF_E_ARG, "mvnerf_field_eval_stash_split: null pointer");
    if (B <= 0 || V <= 0 || R <= 0 || S <= 0 || H < 2 || W < 2) return fail(MVNERF_E_ARG, "mvnerf_field_eval_stash_split: B=%d V=%d R=%d S=%d H=%d W=%d", B, V, R, S, H, W);
    if (V > 1 && ((long)R * S) % 32 != 0) return fail(MVNERF_E_SHAPE, "mvnerf_field_eval_stash_split: R*S=%ld must be a multiple of 32 when V > 1", (long)R * S);
    const long total = (long)B * R * S;
    if (total >= (1L << 31)) return fail(MVNERF_E_SHAPE, "mvnerf_field_eval_stash_split: B*R*S too large");
    if ((long)V * ((total + 31) / 32) >= (1L << 18))      // stash slots are addressed with 32-bit byte offsets (16 KiB per tile)
        return fail(MVNERF_E_SHAPE, "mvnerf_field_eval_stash_split: V*B*R*S/32 = %ld tiles per stash slot, at most 262143", (long)V * ((total + 31) / 32));
    if (!aligned16(features) || !aligned16(packed_net) || !aligned16(packed_split) || !aligned16(rgbs) || !aligned16(stash) || !aligned16(workspace))
        return fail(MVNERF_E_ALIGN, "mvnerf_field_eval_stash_split: features, packed nets, rgbs, stash, workspace must be 16-byte aligned");
    mvnerf::FieldParams p = {};
    p.rays_o = rays_o; p.rays_d = rays_d; p.z = z; p.images = images; p.features = features;
    p.k4 = intrinsics; p.einv = extrinsics_inv; p.net = packed_net; p.rgbs = rgbs;
    p.texel_table = texel_table;
    p.dir_bias = static_cast<float*>(workspace);
    p.B = B; p.V = V; p.R = R; p.S = S; p.H = H; p.W = W;
    p.total = total;
    p.n_tiles = (total + 31) / 32;
    p.stash = stash;
    p.stash_stride = (long)V * p.n_tiles * 4096;
    p.stash_fused = stash + 7 * p.stash_stride;
    p.stash_fused_stride = p.n_tiles * 4096;
    return hip_status(mvnerf::launch_field_eval_split(p, packed_split, static_cast<hipStream_t>(stream)), "mvnerf_field_eval_stash_split");
}

int mvnerf_pack_bwd_streams(const float* net_keras, float* bwd_streams, mvnerf_stream_t stream) {
    if (!net_keras || !bwd_streams) return fail(MVNERF_E_ARG, "mvnerf_pack_bwd_streams: null pointer");
    if (!aligned16(bwd_streams)) return fail(MVNERF_E_ALIGN, "mvnerf_pack_bwd_streams: bwd_streams must be 16-byte aligned");
    return hip_status(mvnerf::launch_pack_bwd_streams(net_keras, bwd_streams, static_cast<hipStream_t>(stream)), "mvnerf_pack_bwd_streams");
}

int mvnerf_mse_grad(const float* pred, const float* label, long n, float* d_pred, float* loss, mvnerf_stream_t stream) {
    if (!pred || !label || !d_pred || !loss) return fail(MVNERF_E_ARG, "mvnerf_mse_grad: null pointer");
    if (n <= 0) return fail(MVNERF_E_ARG, "mvnerf_mse_grad: n=%ld", n);
    return hip_status(mvnerf::launch_mse_grad(pred, label, n, d_pred, loss, static_cast<hipStream_t>(stream)), "mvnerf_mse_grad");
}

int mvnerf_composite_bwd(const float* z, const float* rgbs, const float* d_rgb, const float* d_depth,
                         const float* d_weights, int n_rays, int S, float* d_rgbs, float* d_z, mvnerf_stream_t stream) {
    if (!z || !rgbs || !d_rgb || !d_rgbs) return fail(MVNERF_E_ARG, "mvnerf_composite_bwd: null pointer");
    if (n_rays <= 0) return fail(MVNERF_E_ARG, "mvnerf_composite_bwd: n_rays=%d", n_rays);
    if (S != 64 && S != 128) return fail(MVNERF_E_SHAPE, "mvnerf_composite_bwd: S=%d, supported: 64, 128", S);
    if (!aligned16(rgbs) || !aligned16(d_rgbs)) return fail(MVNERF_E_ALIGN, "mvnerf_composite_bwd: rgbs, d_rgbs must be 16-byte aligned");
    return hip_status(mvnerf::launch_composite_bwd(z, rgbs, d_rgb, d_depth, d_weights, n_rays, S, d_rgbs, d_z,
                                                   static_cast<hipStream_t>(stream)),
                      "mvnerf_composite_bwd");
}

int mvnerf_resample_bwd(const float* z, const float* weights, const float* u_fine, const int32_t* fine_rank,
                        const float* d_z_all, int n_rays, int S, int q7_mode, float* d_weights, mvnerf_stream_t stream) {
    if (!z || !weights || !u_fine || !fine_rank || !d_z_all || !d_weights) return fail(MVNERF_E_ARG, "mvnerf_resample_bwd: null pointer");
    if (n_rays <= 0) return fail(MVNERF_E_ARG, "mvnerf_resample_bwd: n_rays=%d", n_rays);
    if (S != 64) return fail(MVNERF_E_SHAPE, "mvnerf_resample_bwd: S=%d, only the reference's n_samples=64 is built", S);
    return hip_status(mvnerf::launch_resample_bwd(z, weights, u_fine, fine_rank, d_z_all, n_rays, q7_mode, d_weights,
                                                  static_cast<hipStream_t>(stream)),
                      "mvnerf_resample_bwd");
}

int mvnerf_field_backward(const float* rays_o, const float* rays_d, const float* z, const float* images,
                          const float* features, const float* intrinsics, const float* extrinsics_inv,
                          const float* net_keras, const float* bwd_streams, const float* stash, const float* rgbs,
                          const float* d_rgbs, int B, int V, int R, int S, int H, int W, void* scratch, float* grad,
                          float* d_z, float* d_features, mvnerf_stream_t stream) {
    return mvnerf_field_backward_table(rays_o, rays_d, z, images, features, nullptr, nullptr, intrinsics, extrinsics_inv, net_keras,
                                       bwd_streams, stash, rgbs, d_rgbs, B, V, R, S, H, W, scratch, grad, d_z, d_features, stream);
}

int mvnerf_field_backward_table(const float* rays_o, const float* rays_d, const float* z, const float* images,
                                const float* features, const float* texel_table, float* texel_grad, const float* intrinsics,
                                const float* extrinsics_inv, const float* net_keras, const float* bwd_streams, const float* stash,
                                const float* rgbs, const float* d_rgbs, int B, int V, int R, int S, int H, int W, void* scratch,
                                float* grad, float* d_z, float* d_features, mvnerf_stream_t stream) {
    using namespace mvnerf;
    if ((texel_table && !aligned16(texel_table)) || (texel_grad && !aligned16(texel_grad)))
        return fail(MVNERF_E_ALIGN, "mvnerf_field_backward: texel_table, texel_grad must be 16-byte aligned");
    if (texel_grad && !texel_table) return fail(MVNERF_E_ARG, "mvnerf_field_backward: texel_grad needs texel_table");
    if (!rays_o || !rays_d || !z || !images || !features || !intrinsics || !extrinsics_inv || !net_keras || !bwd_streams ||
        !stash || !rgbs || !d_rgbs || !scratch || !grad)
        return fail(MVNERF_E_ARG, "mvnerf_field_backward: null pointer");
    if (B <= 0 || V <= 0 || R <= 0 || S <= 0 || H < 2 || W < 2) return fail(MVNERF_E_ARG, "mvnerf_field_backward: bad sizes");
    if (V > 1 && ((long)R * S) % 32 != 0) return fail(MVNERF_E_SHAPE, "mvnerf_field_backward: R*S must be a multiple of 32 when V > 1");
    if (!aligned16(net_keras) || !aligned16(bwd_streams) || !aligned16(stash) || !aligned16(rgbs) || !aligned16(d_rgbs) || !aligned16(scratch))
        return fail(MVNERF_E_ALIGN, "mvnerf_field_backward: net_keras, bwd_streams, stash, rgbs, d_rgbs, scratch must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long total = (long)B * R * S, n_tiles = (total + 31) / 32, view_tiles = n_tiles * V;
    const size_t vslot = (size_t)view_tiles * 4096, fslot = (size_t)n_tiles * 4096;
    float* buf[3] = {static_cast<float*>(scratch), static_cast<float*>(scratch) + vslot, static_cast<float*>(scratch) + 2 * vslot};
    float* do_tl = static_cast<float*>(scratch) + 3 * vslot;
    // per-workgroup partials of every weight-gradient span, added in workgroup order by reduce_partials_kernel: bit-identical from run
    // to run AND faster than fp32 atomics onto the same addresses (7.54 vs 7.60 ms per step at cfg2), so it is the only mode since
    // round 2; mvnerf_set_deterministic is kept for its callers and has no effect on the weight gradients any more
    float* part = do_tl + (size_t)n_tiles * 32 * 32;
    auto view_slot = [&](int k) { return stash + (size_t)k * vslot; };                   // x0,h1,x1,h2,x2,h3,x3
    auto fused_slot = [&](int m) { return stash + 7 * vslot + (size_t)m * fslot; };      // mean,h4,x4,h5,x5,h6,x6
    hipError_t e;
#define MV_TRY(call) if ((e = (call)) != hipSuccess) return hip_status(e, "mvnerf_field_backward")
    // max |g| slots of the 13 gradient tensors of the chain (zeroed here, filled by each tensor's producer)
    float* amax = part + partial_only_floats(n_tiles, V);
    MV_TRY(mvnerf::launch_zero(amax, (size_t)kAmaxTensors * mvnerf::kBwdAmaxSlots * sizeof(float), st));
    int am = 0;                                        // amax + 64 am belongs to buf[g]
    auto amax_of = [&](int k) { return amax + (size_t)k * mvnerf::kBwdAmaxSlots; };
    // read-out
    MV_TRY(launch_readout_bwd(fused_slot(6), rgbs, d_rgbs, net_keras + kKerasWr, total, n_tiles, do_tl, buf[0], st, amax_of(0)));
    MV_TRY(launch_dw_tile(fused_slot(6), 1, do_tl, 32, n_tiles, grad + kKerasWr, 4, 4, grad + kKerasBr, kBwdMaxWGs, part, st));
    int g = 0;                                         // buf[g] holds dL/d(block output)
    for (int bi = 5; bi >= 0; --bi) {
        if (bi == 2 && V > 1) {                        // reduce_mean over views (layers.py:368-370)
            const int gn = (g + 1) % 3;
            MV_TRY(launch_view_broadcast(buf[g], V, n_tiles / B, n_tiles, buf[gn], st));
            g = gn;
        }
        const bool fused = bi >= 3;
        const long nt = fused ? n_tiles : view_tiles;
        const float* pre_in = fused ? fused_slot(2 * (bi - 3)) : view_slot(2 * bi);
        const float* pre_hid = fused ? fused_slot(2 * (bi - 3) + 1) : view_slot(2 * bi + 1);
        float* gb = grad + kKerasBlocks + bi * kKerasBlockStride;
        const int dh = (g + 1) % 3, gn = (g + 2) % 3;
        // second Dense of the block: out = x_in + W2^T relu(hid) + b2      (dX and dW in one pass over the tiles)
        MV_TRY(launch_dense_bwd_fused(buf[g], pre_hid, bwd_streams + (size_t)(2 * bi + 1) * kHiddenWFloats, nullptr, buf[dh], nt,
                                      gb + kHidden * kHidden + kHidden, gb + 2 * kHidden * kHidden + kHidden, kFusedBwdWGs, part, st,
                                      amax_of(am), amax_of(am + 1)));
        // first Dense: hid = W1^T relu(x_in) + b1 ; the identity branch adds dL/d(out) back
        MV_TRY(launch_dense_bwd_fused(buf[dh], pre_in, bwd_streams + (size_t)(2 * bi) * kHiddenWFloats, buf[g], buf[gn], nt, gb,
                                      gb + kHidden * kHidden, kFusedBwdWGs, part, st, amax_of(am + 1), amax_of(am + 2)));
        am += 2;                                       // (the view broadcast scales by 1 / V: its output keeps its input's bound)
        g = gn;
    }
    // layer 0 (inputs recomputed)
    FieldParams p = {};
    p.rays_o = rays_o; p.rays_d = rays_d; p.z = z; p.images = images; p.features = features;
    p.k4 = intrinsics; p.einv = extrinsics_inv;
    p.B = B; p.V = V; p.R = R; p.S = S; p.H = H; p.W = W; p.total = total; p.n_tiles = n_tiles;
    p.texel_table = texel_table;                           // only the sample-position gradient uses it (launch_field_dz)
    p.net = net_keras;                                     // (Keras layout here: field_dz_table_kernel reads W0's rgb rows from it)
    MV_TRY(launch_dw0(p, buf[g], grad + kKerasW0, grad + kKerasB0, kBwdMaxWGs, part, st, amax_of(am)));
    // d_features through the texel table (texel_grad given): the samples' g0 is scattered onto the 128-channel table gradient and W0 is
    // applied once per texel afterwards; the sample-position gradient then takes the table path as well
    const bool via_table = d_features && texel_table && texel_grad;
    if (via_table) {
        const long n_texels = (long)B * V * H * W;
        MV_TRY(mvnerf::launch_zero(texel_grad, (size_t)n_texels * 128 * sizeof(float), st));
        MV_TRY(launch_texel_scatter(p, buf[g], texel_grad, st));
        MV_TRY(launch_texel_grad_to_features(texel_grad, net_keras + kKerasW0 + 123 * kHidden, n_texels, d_features, st));
    }
    if (d_z || (d_features && !via_table))
        MV_TRY(launch_field_dz(p, buf[g], bwd_streams + (size_t)12 * kHiddenWFloats, d_z, nullptr, nullptr, via_table ? nullptr : d_features, st));
#undef MV_TRY
    return 0;
}

static bool gemm_nt_shape_ok(int M, int N, int K) { return M > 0 && N > 0 && K > 0 && M % 32 == 0 && N % 64 == 0 && K % 8 == 0; }

size_t mvnerf_gemm_nt_scratch_bytes(int M, int N, int K) {
    if (!gemm_nt_shape_ok(M, N, K)) return 0;
    const int splits = mvnerf::gemm_nt_splits(M, N, K);
    return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

static int gemm_nt_impl(const char* who, const float* a, const float* bt, const float* bias, float* c, int M, int N, int K, void* scratch,
                        mvnerf_stream_t stream) {
    if (!a || !bt || !c) return fail(MVNERF_E_ARG, "%s: null pointer", who);
    if (!gemm_nt_shape_ok(M, N, K))
        return fail(MVNERF_E_SHAPE, "%s: M=%d N=%d K=%d, needs M %% 32 == 0, N %% 64 == 0, K %% 8 == 0", who, M, N, K);
    if (!aligned16(a) || !aligned16(bt) || !aligned16(c) || (scratch && !aligned16(scratch)) || (bias && !aligned16(bias)))
        return fail(MVNERF_E_ALIGN, "%s: a, bt, bias, c, scratch must be 16-byte aligned", who);
    if (mvnerf_gemm_nt_scratch_bytes(M, N, K) && !scratch) return fail(MVNERF_E_ARG, "%s: this shape needs scratch", who);
    return hip_status(mvnerf::launch_gemm_nt_f32(a, bt, bias, c, M, N, K, static_cast<float*>(scratch), static_cast<hipStream_t>(stream)), who);
}

int mvnerf_gemm_nt(const float* a, const float* bt, float* c, int M, int N, int K, void* scratch, mvnerf_stream_t stream) {
    return gemm_nt_impl("mvnerf_gemm_nt", a, bt, nullptr, c, M, N, K, scratch, stream);
}

int mvnerf_gemm_nt_bias(const float* a, const float* bt, const float* bias, float* c, int M, int N, int K, void* scratch,
                        mvnerf_stream_t stream) {
    if (!bias) return fail(MVNERF_E_ARG, "mvnerf_gemm_nt_bias: null pointer");
    return gemm_nt_impl("mvnerf_gemm_nt_bias", a, bt, bias, c, M, N, K, scratch, stream);
}

static bool gemm_tn_shape_ok(int M, int N, int K) { return M > 0 && N > 0 && K > 0 && M % 8 == 0 && N % 32 == 0 && K % 64 == 0; }

size_t mvnerf_gemm_tn_scratch_bytes(int M, int N, int K) {
    if (!gemm_tn_shape_ok(M, N, K)) return 0;
    const int splits = mvnerf::gemm_tn_splits(M, N, K);
    return splits > 1 ? (size_t)splits * N * K * sizeof(float) : 0;
}

int mvnerf_gemm_tn(const float* g, const float* a, float* c, int M, int N, int K, void* scratch, mvnerf_stream_t stream) {
    if (!g || !a || !c) return fail(MVNERF_E_ARG, "mvnerf_gemm_tn: null pointer");
    if (!gemm_tn_shape_ok(M, N, K))
        return fail(MVNERF_E_SHAPE, "mvnerf_gemm_tn: M=%d N=%d K=%d, needs M %% 8 == 0, N %% 32 == 0, K %% 64 == 0", M, N, K);
    if (!aligned16(g) || !aligned16(a) || !aligned16(c) || (scratch && !aligned16(scratch)))
        return fail(MVNERF_E_ALIGN, "mvnerf_gemm_tn: g, a, c, scratch must be 16-byte aligned");
    if (mvnerf_gemm_tn_scratch_bytes(M, N, K) && !scratch) return fail(MVNERF_E_ARG, "mvnerf_gemm_tn: this shape needs scratch");
    return hip_status(mvnerf::launch_gemm_tn_f32(g, a, c, M, N, K, static_cast<float*>(scratch), static_cast<hipStream_t>(stream)),
                      "mvnerf_gemm_tn");
}

size_t mvnerf_gemm_tn_batched_scratch_bytes(int M, int N, int K, int batch, int with_colsum) {
    if (!gemm_tn_shape_ok(M, N, K) || batch <= 0) return 0;
    return mvnerf::gemm_tn_batched_scratch_floats(M, N, K, batch, with_colsum != 0) * sizeof(float);
}

int mvnerf_gemm_tn_batched(const mvnerf_gemm_tn_batch* q, float* c, float* colsum, int M, int N, int K, int batch, void* scratch,
                           mvnerf_stream_t stream) {
    if (!q || !q->g || !q->a || !c) return fail(MVNERF_E_ARG, "mvnerf_gemm_tn_batched: null pointer");
    if ((q->g2 == nullptr) != (q->a2 == nullptr)) return fail(MVNERF_E_ARG, "mvnerf_gemm_tn_batched: g2 and a2 come together");
    if (!gemm_tn_shape_ok(M, N, K) || batch <= 0)
        return fail(MVNERF_E_SHAPE, "mvnerf_gemm_tn_batched: M=%d N=%d K=%d batch=%d, needs M %% 8 == 0, N %% 32 == 0, K %% 64 == 0", M, N, K, batch);
    if (q->ldg < N || q->lda < K || (q->g2 && (q->ldg2 < N || q->lda2 < K)))
        return fail(MVNERF_E_SHAPE, "mvnerf_gemm_tn_batched: row strides ldg=%d lda=%d ldg2=%d lda2=%d below N=%d / K=%d", q->ldg, q->lda, q->ldg2, q->lda2, N, K);
    if (q->colsum_of < 0 || q->colsum_of > 2 || (q->colsum_of == 2 && !q->g2) || (q->colsum_of != 0 && !colsum))
        return fail(MVNERF_E_ARG, "mvnerf_gemm_tn_batched: colsum_of=%d without its operand or output", q->colsum_of);
    if (!aligned16(c) || (colsum && !aligned16(colsum)) || (scratch && !aligned16(scratch)))
        return fail(MVNERF_E_ALIGN, "mvnerf_gemm_tn_batched: c, colsum, scratch must be 16-byte aligned");
    if (mvnerf_gemm_tn_batched_scratch_bytes(M, N, K, batch, q->colsum_of != 0) && !scratch)
        return fail(MVNERF_E_ARG, "mvnerf_gemm_tn_batched: this shape needs scratch");
    mvnerf::TnBatchArgs a;
    a.G = q->g; a.A = q->a; a.G2 = q->g2; a.A2 = q->a2;
    a.g_bstride = q->g_batch_stride; a.a_bstride = q->a_batch_stride; a.g2_bstride = q->g2_batch_stride; a.a2_bstride = q->a2_batch_stride;
    a.ldg = q->ldg; a.lda = q->lda; a.ldg2 = q->ldg2; a.lda2 = q->lda2;
    a.colsum_of = q->colsum_of;
    return hip_status(mvnerf::launch_gemm_tn_batched(a, c, colsum, M, N, K, batch, static_cast<float*>(scratch), static_cast<hipStream_t>(stream)),
                      "mvnerf_gemm_tn_batched");
}

// ---- the trunk as a differentiable field on query points --------------------------------------
size_t mvnerf_query_workspace_bytes(int B, int V, int N) {
    if (B <= 0 || V <= 0 || N <= 0) return 0;
    return (size_t)2 * B * V * N * 128 * sizeof(float);        // layer-0 seed and its tangent per (view, point)
}

int mvnerf_query_jvp(const float* points, const float* dirs, const float* t_points, const float* t_dirs,
                     const float* images, const float* features, const float* intrinsics, const float* extrinsics_inv,
                     const float* packed_net, int B, int V, int N, int H, int W, float* acts, float* t_acts, void* workspace,
                     mvnerf_stream_t stream) {
    if (!points || !dirs || !t_points || !t_dirs || !images || !features || !intrinsics || !extrinsics_inv || !packed_net ||
        !t_acts || !workspace)
        return fail(MVNERF_E_ARG, "mvnerf_query_jvp: null pointer");
    if (B <= 0 || V <= 0 || N <= 0) return fail(MVNERF_E_ARG, "mvnerf_query_jvp: B=%d V=%d N=%d", B, V, N);
    if (H < 2 || W < 2) return fail(MVNERF_E_SHAPE, "mvnerf_query_jvp: source image %dx%d, need H,W >= 2", H, W);
    if ((long)B * N >= (1L << 31) || (long)B * V * H * W >= (1L << 31)) return fail(MVNERF_E_SHAPE, "mvnerf_query_jvp: sizes too large for int32 indices");
    if (!aligned16(features) || !aligned16(packed_net) || !aligned16(t_acts) || (acts && !aligned16(acts)) || !aligned16(workspace))
        return fail(MVNERF_E_ALIGN, "mvnerf_query_jvp: features, packed_net, acts, t_acts, workspace must be 16-byte aligned");
    mvnerf::FieldParams p = {};
    p.rays_o = points; p.rays_d = dirs; p.z = nullptr; p.t_o = t_points; p.t_d = t_dirs;
    p.images = images; p.features = features; p.k4 = intrinsics; p.einv = extrinsics_inv; p.net = packed_net;
    p.acts_fused = acts; p.t_acts = t_acts;
    p.dir_bias = static_cast<float*>(workspace);
    p.dir_tan = p.dir_bias + (size_t)B * V * N * 128;
    p.B = B; p.V = V; p.R = N; p.S = 1; p.H = H; p.W = W;
    p.total = (long)B * N;
    p.n_tiles = (p.total + 31) / 32;
    return hip_status(mvnerf::launch_field_jvp(p, static_cast<hipStream_t>(stream)), "mvnerf_query_jvp");
}

size_t mvnerf_query_vjp_scratch_bytes(int B, int V, int N) {
    if (B <= 0 || V <= 0 || N <= 0) return 0;
    const size_t tiles = ((size_t)B * N + 31) / 32;
    return (size_t)3 * V * tiles * 4096 * sizeof(float);
}

int mvnerf_query_vjp(const float* points, const float* dirs, const float* images, const float* features,
                     const float* intrinsics, const float* extrinsics_inv, const float* bwd_streams, const float* stash,
                     const float* g_acts, int B, int V, int N, int H, int W, void* scratch, float* d_points, float* d_dirs,
                     mvnerf_stream_t stream) {
    using namespace mvnerf;
    if (!points || !dirs || !images || !features || !intrinsics || !extrinsics_inv || !bwd_streams || !stash || !g_acts ||
        !scratch || !d_points || !d_dirs)
        return fail(MVNERF_E_ARG, "mvnerf_query_vjp: null pointer");
    if (B <= 0 || V <= 0 || N <= 0 || H < 2 || W < 2) return fail(MVNERF_E_ARG, "mvnerf_query_vjp: bad sizes");
    if (V > 1 && N % 32 != 0) return fail(MVNERF_E_SHAPE, "mvnerf_query_vjp: N=%d must be a multiple of 32 when V > 1", N);
    if (!aligned16(features) || !aligned16(bwd_streams) || !aligned16(stash) || !aligned16(g_acts) || !aligned16(scratch))
        return fail(MVNERF_E_ALIGN, "mvnerf_query_vjp: features, bwd_streams, stash, g_acts, scratch must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long total = (long)B * N, n_tiles = (total + 31) / 32, view_tiles = n_tiles * V;
    const size_t vslot = (size_t)view_tiles * 4096, fslot = (size_t)n_tiles * 4096;
    float* buf[3] = {static_cast<float*>(scratch), static_cast<float*>(scratch) + vslot, static_cast<float*>(scratch) + 2 * vslot};
    auto view_slot = [&](int k) { return stash + (size_t)k * vslot; };
    auto fused_slot = [&](int m) { return stash + 7 * vslot + (size_t)m * fslot; };
    hipError_t e;
#define MV_TRY(call) if ((e = (call)) != hipSuccess) return hip_status(e, "mvnerf_query_vjp")
    MV_TRY(launch_zero(d_points, (size_t)total * 3 * sizeof(float), st));
    MV_TRY(launch_zero(d_dirs, (size_t)total * 3 * sizeof(float), st));
    // buf[g] holds dL/d(block output); the cotangents of u3, u2, u1 and the view mean enter where those tensors are produced
    MV_TRY(launch_rows_to_tl(g_acts + (size_t)3 * total * 128, total, n_tiles, 0, buf[0], st));
    int g = 0;
    for (int bi = 5; bi >= 0; --bi) {
        if (bi == 2 && V > 1) {
            const int gn = (g + 1) % 3;
            MV_TRY(launch_view_broadcast(buf[g], V, n_tiles / B, n_tiles, buf[gn], st));
            g = gn;
        }
        const bool fused = bi >= 3;
        const long nt = fused ? n_tiles : view_tiles;
        const float* pre_in = fused ? fused_slot(2 * (bi - 3)) : view_slot(2 * bi);
        const float* pre_hid = fused ? fused_slot(2 * (bi - 3) + 1) : view_slot(2 * bi + 1);
        const int dh = (g + 1) % 3, gn = (g + 2) % 3;
        MV_TRY(launch_dense_bwd_fused(buf[g], pre_hid, bwd_streams + (size_t)(2 * bi + 1) * kHiddenWFloats, nullptr, buf[dh], nt,
                                      nullptr, nullptr, kFusedBwdWGs, nullptr, st));
        MV_TRY(launch_dense_bwd_fused(buf[dh], pre_in, bwd_streams + (size_t)(2 * bi) * kHiddenWFloats, buf[g], buf[gn], nt,
                                      nullptr, nullptr, kFusedBwdWGs, nullptr, st));
        g = gn;
        if (fused) MV_TRY(launch_rows_to_tl(g_acts + (size_t)(bi - 3) * total * 128, total, n_tiles, 1, buf[g], st));
    }
    FieldParams p = {};
    p.rays_o = points; p.rays_d = dirs; p.z = nullptr; p.images = images; p.features = features;
    p.k4 = intrinsics; p.einv = extrinsics_inv;
    p.B = B; p.V = V; p.R = N; p.S = 1; p.H = H; p.W = W; p.total = total; p.n_tiles = n_tiles;
    MV_TRY(launch_field_dz(p, buf[g], bwd_streams + (size_t)12 * kHiddenWFloats, nullptr, d_points, d_dirs, nullptr, st));
#undef MV_TRY
    return 0;
}

int mvnerf_stash_fused_acts(const float* stash, int B, int V, int N, float* acts, mvnerf_stream_t stream) {
    if (!stash || !acts) return fail(MVNERF_E_ARG, "mvnerf_stash_fused_acts: null pointer");
    if (B <= 0 || V <= 0 || N <= 0) return fail(MVNERF_E_ARG, "mvnerf_stash_fused_acts: B=%d V=%d N=%d", B, V, N);
    if (!aligned16(stash) || !aligned16(acts)) return fail(MVNERF_E_ALIGN, "mvnerf_stash_fused_acts: stash, acts must be 16-byte aligned");
    const long total = (long)B * N, n_tiles = (total + 31) / 32;
    const size_t vslot = (size_t)n_tiles * V * 4096, fslot = (size_t)n_tiles * 4096;
    // fused slots 0, 2, 4, 6 = view mean, u1, u2, u3 (the odd ones are the blocks' hidden pre-activations)
    return hip_status(mvnerf::launch_tl_to_rows(stash + 7 * vslot, (long)(2 * fslot), 4, total, n_tiles, acts, static_cast<hipStream_t>(stream)),
                      "mvnerf_stash_fused_acts");
}

int mvnerf_adam_clip(float* param, const float* grad, float* m, float* v, long n, float lr_t, float beta1, float beta2,
                     float eps, float clip, const unsigned char* update_mask, mvnerf_stream_t stream) {
    if (!param || !grad || !m || !v) return fail(MVNERF_E_ARG, "mvnerf_adam_clip: null pointer");
    if (n <= 0) return fail(MVNERF_E_ARG, "mvnerf_adam_clip: n=%ld", n);
    return hip_status(mvnerf::launch_adam_clip(param, grad, m, v, n, lr_t, beta1, beta2, eps, clip, update_mask,
                                               static_cast<hipStream_t>(stream)),
                      "mvnerf_adam_clip");
}

size_t mvnerf_field_workspace_bytes(int B, int V, int R) {
    if (B <= 0 || V <= 0 || R <= 0) return 0;
    return (size_t)B * V * R * 128 * sizeof(float);
}

size_t mvnerf_render_workspace_bytes(int B, int V, int R, int S) {
    if (B <= 0 || V <= 0 || R <= 0 || S <= 0) return 0;
    return carve(nullptr, (long)B * R, V, S).bytes;
}

int mvnerf_render_fwd(const float* rays_o, const float* rays_d, const float* images, const float* features,
                      const float* intrinsics, const float* extrinsics_inv, const float* packed_coarse,
                      const float* packed_fine, const float* u_coarse, const float* u_fine, int B, int V, int R,
                      int S, int H, int W, double near_, double far_, int q7_mode, float* rgb, float* depth,
                      float* fine_rgb, float* fine_depth, void* workspace, float* texel_tables, int tables_ready,
                      mvnerf_stream_t stream) {
    if (!u_coarse || !u_fine || !rgb || !depth || !fine_rgb || !fine_depth || !workspace || !packed_fine)
        return fail(MVNERF_E_ARG, "mvnerf_render_fwd: null pointer");
    if (S != 64) return fail(MVNERF_E_SHAPE, "mvnerf_render_fwd: S=%d, only the reference's n_samples=64 is built", S);
    if (B <= 0 || R <= 0) return fail(MVNERF_E_ARG, "mvnerf_render_fwd: B=%d R=%d", B, R);
    if (!aligned16(workspace)) return fail(MVNERF_E_ALIGN, "mvnerf_render_fwd: workspace must be 16-byte aligned");
    const long n_rays = (long)B * R;
    if (n_rays * 2 * S >= (1L << 31)) return fail(MVNERF_E_SHAPE, "mvnerf_render_fwd: B*R*2S too large");
    const Workspace w = carve(workspace, n_rays, V, S);
    int rc;
    const float *table_c = nullptr, *table_f = nullptr;
    if (texel_tables) {                                       // [coarse net | fine net], mvnerf_texel_table_bytes each
        float* tf = texel_tables + mvnerf_texel_table_bytes(B, V, H, W) / sizeof(float);
        if (!tables_ready) {
            if ((rc = mvnerf_project_texels2(features, packed_coarse, packed_fine, B, V, H, W, texel_tables, tf, stream))) return rc;
        }
        table_c = texel_tables;
        table_f = tf;
    }
    if ((rc = mvnerf_stratified_depths(u_coarse, (int)n_rays, S, near_, far_, w.z, stream))) return rc;
    if ((rc = field_eval_impl(rays_o, rays_d, w.z, images, features, table_c, intrinsics, extrinsics_inv, packed_coarse, B, V,
                              R, S, H, W, w.rgbs_c, nullptr, nullptr, nullptr, nullptr, nullptr, w.dir_bias, stream)))
        return rc;
    if ((rc = mvnerf_composite(w.z, w.rgbs_c, (int)n_rays, S, rgb, depth, w.weights, stream))) return rc;
    if ((rc = mvnerf_resample(w.z, w.weights, u_fine, (int)n_rays, S, q7_mode, w.z_all, nullptr, nullptr, nullptr, nullptr, stream)))
        return rc;
    if ((rc = field_eval_impl(rays_o, rays_d, w.z_all, images, features, table_f, intrinsics, extrinsics_inv, packed_fine, B,
                              V, R, 2 * S, H, W, w.rgbs_f, nullptr, nullptr, nullptr, nullptr, nullptr, w.dir_bias, stream)))
        return rc;
    return mvnerf_composite(w.z_all, w.rgbs_f, (int)n_rays, 2 * S, fine_rgb, fine_depth, nullptr, stream);
}

int mvnerf_render_fwd_split(const float* rays_o, const float* rays_d, const float* images, const float* features,
                            const float* intrinsics, const float* extrinsics_inv, const float* packed_coarse,
                            const float* packed_fine, const void* split_coarse, const void* split_fine, const float* u_coarse,
                            const float* u_fine, int B, int V, int R, int S, int H, int W, double near_, double far_, int q7_mode,
                            float* rgb, float* depth, float* fine_rgb, float* fine_depth, void* workspace, float* texel_tables,
                            int tables_ready, mvnerf_stream_t stream) {
    if (!u_coarse || !u_fine || !rgb || !depth || !fine_rgb || !fine_depth || !workspace || !packed_fine || !split_coarse || !split_fine)
        return fail(MVNERF_E_ARG, "mvnerf_render_fwd_split: null pointer");
    if (S != 64) return fail(MVNERF_E_SHAPE, "mvnerf_render_fwd_split: S=%d, only the reference's n_samples=64 is built", S);
    if (B <= 0 || R <= 0) return fail(MVNERF_E_ARG, "mvnerf_render_fwd_split: B=%d R=%d", B, R);
    if (!aligned16(workspace)) return fail(MVNERF_E_ALIGN, "mvnerf_render_fwd_split: workspace must be 16-byte aligned");
    const long n_rays = (long)B * R;
    if (n_rays * 2 * S >= (1L << 31)) return fail(MVNERF_E_SHAPE, "mvnerf_render_fwd_split: B*R*2S too large");
    const Workspace w = carve(workspace, n_rays, V, S);
    int rc;
    const float *table_c = nullptr, *table_f = nullptr;
    if (texel_tables) {                                       // [coarse net | fine net], mvnerf_texel_table_bytes each
        float* tf = texel_tables + mvnerf_texel_table_bytes(B, V, H, W) / sizeof(float);
        if (!tables_ready) {
            if ((rc = mvnerf_project_texels2(features, packed_coarse, packed_fine, B, V, H, W, texel_tables, tf, stream))) return rc;
        }
        table_c = texel_tables;
        table_f = tf;
    }
    if ((rc = mvnerf_stratified_depths(u_coarse, (int)n_rays, S, near_, far_, w.z, stream))) return rc;
    if ((rc = mvnerf_field_eval_split(rays_o, rays_d, w.z, images, features, table_c, intrinsics, extrinsics_inv, packed_coarse, split_coarse,
                                      B, V, R, S, H, W, w.rgbs_c, nullptr, nullptr, nullptr, nullptr, nullptr, w.dir_bias, stream)))
        return rc;
    if ((rc = mvnerf_composite(w.z, w.rgbs_c, (int)n_rays, S, rgb, depth, w.weights, stream))) return rc;
    if ((rc = mvnerf_resample(w.z, w.weights, u_fine, (int)n_rays, S, q7_mode, w.z_all, nullptr, nullptr, nullptr, nullptr, stream)))
        return rc;
    if ((rc = mvnerf_field_eval_split(rays_o, rays_d, w.z_all, images, features, table_f, intrinsics, extrinsics_inv, packed_fine, split_fine,
                                      B, V, R, 2 * S, H, W, w.rgbs_f, nullptr, nullptr, nullptr, nullptr, nullptr, w.dir_bias, stream)))
        return rc;
    return mvnerf_composite(w.z_all, w.rgbs_f, (int)n_rays, 2 * S, fine_rgb, fine_depth, nullptr, stream);
}

}  // extern "C"
