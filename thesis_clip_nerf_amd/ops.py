"""Torch-tensor front end of the C ABI (include/mvnerf_hip.h).

PyTorch is used only for device memory and streams: every function checks device / dtype /
contiguity, allocates outputs with ``torch.empty`` and passes raw pointers plus the current HIP
stream to libmvnerf_hip.so.  Shape errors raise ``ValueError`` (like the reference's TensorSpec
mismatches), HIP failures ``RuntimeError``.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import NET_PARAMS, Q7_CLAMP, Q7_ZERO  # noqa: F401


def _chk(t, name, dtype=torch.float32, shape=None):
    if not isinstance(t, torch.Tensor):
        raise ValueError(f'{name}: expected a torch.Tensor, got {type(t).__name__}')
    if not t.is_cuda:
        raise ValueError(f'{name}: must live on a HIP device (got {t.device}); there is no CPU path')
    if t.dtype != dtype:
        raise ValueError(f'{name}: dtype {t.dtype}, expected {dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name}: must be contiguous')
    if shape is not None:
        if len(shape) != t.dim() or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise ValueError(f'{name}: shape {tuple(t.shape)}, expected {tuple(shape)}')
    return t


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def packed_net_floats():
    return int(_lib.lib().mvnerf_packed_net_floats())


def pack_net(net_keras):
    """Keras-order flat MLP (247 300 floats) -> MFMA operand image (mvnerf_pack_net)."""
    _chk(net_keras, 'net_keras', shape=(NET_PARAMS,))
    out = torch.empty(packed_net_floats(), dtype=torch.float32, device=net_keras.device)
    with torch.cuda.device(net_keras.device):
        _lib.check(_lib.lib().mvnerf_pack_net(_p(net_keras), _p(out), _stream(net_keras)), 'pack_net')
    return out


def get_rays_device(m3x3, origin, device, u=None, v=None, width=0, height=0, normalize=True, return_f64=False):
    """mvnerf_get_rays.  m3x3 = E[:3,:3] @ inv(K[:3,:3]) and origin = E[:3,3] are host float64."""
    m = np.ascontiguousarray(m3x3, dtype=np.float64).reshape(9)
    o = np.ascontiguousarray(origin, dtype=np.float64).reshape(3)
    if u is None:
        n = int(width) * int(height)
    else:
        _chk(u, 'u')
        _chk(v, 'v', shape=tuple(u.shape))
        n = u.numel()
    device = torch.device(device)
    rays_o = torch.empty((n, 3), dtype=torch.float32, device=device)
    rays_d = torch.empty((n, 3), dtype=torch.float32, device=device)
    d64 = torch.empty((n, 3), dtype=torch.float64, device=device) if return_f64 else None
    with torch.cuda.device(device):
        rc = _lib.lib().mvnerf_get_rays(m.ctypes.data_as(ctypes.c_void_p), o.ctypes.data_as(ctypes.c_void_p), _p(u),
                                        _p(v), n, int(width), int(bool(normalize)), _p(rays_o), _p(rays_d), _p(d64),
                                        _stream(rays_o))
    _lib.check(rc, 'get_rays')
    return (rays_o, rays_d, d64) if return_f64 else (rays_o, rays_d)


def stratified_depths(u, near, far):
    """u (..., S) uniforms -> z (..., S) (mvnerf_stratified_depths)."""
    _chk(u, 'u')
    s = u.shape[-1]
    z = torch.empty_like(u)
    with torch.cuda.device(u.device):
        rc = _lib.lib().mvnerf_stratified_depths(_p(u), u.numel() // s, s, float(near), float(far), _p(z), _stream(u))
    _lib.check(rc, 'stratified_depths')
    return z


def texel_table_pays(n_rays_per_scene, n_samples, h, w):
    """Host-side rule for building a texel table for ONE field pass: the table costs H*W rows of the 256->128
    product per view, the direct form R*S rows; the factor 2 covers the table's own launch and traffic."""
    return n_rays_per_scene * n_samples >= 2 * h * w


def project_texels(features, packed_net, out=None):
    """mvnerf_project_texels: features (B,V,H,W,256), one net's packed image -> table (B,V,H,W,128)
    (W0[123:379]^T features per texel, accumulator order) for field_eval(..., texel_table=table)."""
    _chk(features, 'features', shape=(None, None, None, None, 256))
    b, v, h, w, _ = features.shape
    _chk(packed_net, 'packed_net', shape=(packed_net_floats(),))
    if out is None:
        out = torch.empty((b, v, h, w, 128), dtype=torch.float32, device=features.device)
    else:
        _chk(out, 'texel_table', shape=(b, v, h, w, 128))
    with torch.cuda.device(features.device):
        rc = _lib.lib().mvnerf_project_texels(_p(features), _p(packed_net), b, v, h, w, _p(out), _stream(features))
    _lib.check(rc, 'project_texels')
    return out


def project_texels2(features, packed_a, packed_b, out=None):
    """mvnerf_project_texels2: both nets' tables (2,B,V,H,W,128) [a | b] from one read of the feature maps."""
    _chk(features, 'features', shape=(None, None, None, None, 256))
    b, v, h, w, _ = features.shape
    _chk(packed_a, 'packed_a', shape=(packed_net_floats(),))
    _chk(packed_b, 'packed_b', shape=(packed_net_floats(),))
    if out is None:
        out = torch.empty((2, b, v, h, w, 128), dtype=torch.float32, device=features.device)
    else:
        _chk(out, 'texel_tables', shape=(2, b, v, h, w, 128))
    with torch.cuda.device(features.device):
        rc = _lib.lib().mvnerf_project_texels2(_p(features), _p(packed_a), _p(packed_b), b, v, h, w, _p(out[0]), _p(out[1]),
                                               _stream(features))
    _lib.check(rc, 'project_texels2')
    return out


def field_eval(rays_o, rays_d, z, images, features, intrinsics, extrinsics_inv, packed_net, return_taps=False,
               return_pix=False, return_embedding=False, complete_output=False, texel_table=None):
    """mvnerf_field_eval: -> rgbs (B,R,S,4) [+ tap_idx (B,V,R,S,4) int32] [+ pix (B,V,R,S,2)] [+ embedding (B,R,S,128)].
    texel_table: project_texels(features, packed_net) of the SAME net -> mvnerf_field_eval_table."""
    _chk(rays_o, 'rays_o', shape=(None, None, 3))
    b, r, _ = rays_o.shape
    _chk(rays_d, 'rays_d', shape=(b, r, 3))
    _chk(z, 'z', shape=(b, r, None))
    s = z.shape[2]
    _chk(images, 'images', shape=(b, None, None, None, 3))
    _, v, h, w, _ = images.shape
    _chk(features, 'features', shape=(b, v, h, w, 256))
    _chk(intrinsics, 'intrinsics', shape=(b, v, 4, 4))
    _chk(extrinsics_inv, 'extrinsics_inv', shape=(b, v, 4, 4))
    _chk(packed_net, 'packed_net', shape=(packed_net_floats(),))
    dev = rays_o.device
    rgbs = torch.empty((b, r, s, 4), dtype=torch.float32, device=dev)
    taps = torch.empty((b, v, r, s, 4), dtype=torch.int32, device=dev) if return_taps else None
    pix = torch.empty((b, v, r, s, 2), dtype=torch.float32, device=dev) if return_pix else None
    emb = torch.empty((b, r, s, 128), dtype=torch.float32, device=dev) if return_embedding else None
    acts_v = torch.empty((4, b * v, r, s, 128), dtype=torch.float32, device=dev) if complete_output else None
    acts_f = torch.empty((4, b, r, s, 128), dtype=torch.float32, device=dev) if complete_output else None
    ws = torch.empty(int(_lib.lib().mvnerf_field_workspace_bytes(b, v, r)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        if texel_table is None:
            rc = _lib.lib().mvnerf_field_eval(_p(rays_o), _p(rays_d), _p(z), _p(images), _p(features), _p(intrinsics),
                                              _p(extrinsics_inv), _p(packed_net), b, v, r, s, h, w, _p(rgbs), _p(taps),
                                              _p(pix), _p(emb), _p(acts_v), _p(acts_f), _p(ws), _stream(rays_o))
        else:
            _chk(texel_table, 'texel_table', shape=(b, v, h, w, 128))
            rc = _lib.lib().mvnerf_field_eval_table(_p(rays_o), _p(rays_d), _p(z), _p(images), _p(features),
                                                    _p(texel_table), _p(intrinsics), _p(extrinsics_inv), _p(packed_net),
                                                    b, v, r, s, h, w, _p(rgbs), _p(taps), _p(pix), _p(emb), _p(acts_v),
                                                    _p(acts_f), _p(ws), _stream(rays_o))
    _lib.check(rc, 'field_eval')
    out = (rgbs,)
    if return_taps:
        out += (taps,)
    if return_pix:
        out += (pix,)
    if return_embedding:
        out += (emb,)
    if complete_output:               # the reference's `outputs` list (layers.py:364-377): 4 per-view + 4 fused
        out += (list(acts_v.unbind(0)) + list(acts_f.unbind(0)),)
    return out if len(out) > 1 else rgbs


def query_field(points, dirs, images, features, intrinsics, extrinsics_inv, packed_net, complete_output=False):
    """Trunk as a field on arbitrary points (lmvnerf/model_v4.py:217-262 use of fine_embedding):
    points, dirs (B,N,3) -> rgbs (B,N,4) and either the embedding (B,N,128) or, with complete_output, the
    list of 8 activations [x0,f1,f2,f3 (B*V,N,128) | mean,u1,u2,u3 (B,N,128)]."""
    _chk(points, 'points', shape=(None, None, 3))
    z = torch.zeros(tuple(points.shape[:2]) + (1,), dtype=torch.float32, device=points.device)   # p = o + 0*d = o
    res = field_eval(points, dirs, z, images, features, intrinsics, extrinsics_inv, packed_net,
                     return_embedding=not complete_output, complete_output=complete_output)
    rgbs, extra = res[0], res[1]
    if complete_output:
        return rgbs[:, :, 0], [a[:, :, 0] for a in extra]
    return rgbs[:, :, 0], extra[:, :, 0]


def composite(z, rgbs, return_weights=True):
    """mvnerf_composite: z (...,S), rgbs (...,S,4) -> rgb (...,3), depth (...), weights (...,S)."""
    _chk(z, 'z')
    s = z.shape[-1]
    _chk(rgbs, 'rgbs', shape=tuple(z.shape) + (4,))
    lead = tuple(z.shape[:-1])
    n = z.numel() // s
    rgb = torch.empty(lead + (3,), dtype=torch.float32, device=z.device)
    depth = torch.empty(lead, dtype=torch.float32, device=z.device)
    weights = torch.empty_like(z) if return_weights else None
    with torch.cuda.device(z.device):
        rc = _lib.lib().mvnerf_composite(_p(z), _p(rgbs), n, s, _p(rgb), _p(depth), _p(weights), _stream(z))
    _lib.check(rc, 'composite')
    return rgb, depth, weights


def resample(z, weights, u_fine, q7_mode=Q7_ZERO, return_aux=False, return_rank=False):
    """mvnerf_resample: -> z_all (...,2S) [+ z_fine, above, below] [+ fine_rank]."""
    _chk(z, 'z')
    s = z.shape[-1]
    _chk(weights, 'weights', shape=tuple(z.shape))
    _chk(u_fine, 'u_fine', shape=tuple(z.shape))
    n = z.numel() // s
    z_all = torch.empty(tuple(z.shape[:-1]) + (2 * s,), dtype=torch.float32, device=z.device)
    z_fine = torch.empty_like(z) if return_aux else None
    above = torch.empty(z.shape, dtype=torch.int32, device=z.device) if return_aux else None
    below = torch.empty(z.shape, dtype=torch.int32, device=z.device) if return_aux else None
    rank = torch.empty(z.shape, dtype=torch.int32, device=z.device) if return_rank else None
    with torch.cuda.device(z.device):
        rc = _lib.lib().mvnerf_resample(_p(z), _p(weights), _p(u_fine), n, s, int(q7_mode), _p(z_all), _p(z_fine),
                                        _p(above), _p(below), _p(rank), _stream(z))
    _lib.check(rc, 'resample')
    out = (z_all, z_fine, above, below) if return_aux else (z_all,)
    if return_rank:
        out += (rank,)
    return out if len(out) > 1 else z_all


def resample_bwd(z, weights, u_fine, fine_rank, d_z_all, q7_mode=Q7_ZERO):
    """Backward of resample w.r.t. the coarse weights: -> d_weights (..., S)."""
    _chk(z, 'z')
    s = z.shape[-1]
    _chk(weights, 'weights', shape=tuple(z.shape))
    _chk(u_fine, 'u_fine', shape=tuple(z.shape))
    _chk(fine_rank, 'fine_rank', dtype=torch.int32, shape=tuple(z.shape))
    _chk(d_z_all, 'd_z_all', shape=tuple(z.shape[:-1]) + (2 * s,))
    out = torch.empty_like(z)
    with torch.cuda.device(z.device):
        rc = _lib.lib().mvnerf_resample_bwd(_p(z), _p(weights), _p(u_fine), _p(fine_rank), _p(d_z_all), z.numel() // s, s,
                                            int(q7_mode), _p(out), _stream(z))
    _lib.check(rc, 'resample_bwd')
    return out


def render_workspace_bytes(b, v, r, s):
    return int(_lib.lib().mvnerf_render_workspace_bytes(int(b), int(v), int(r), int(s)))


def render_fwd(rays_o, rays_d, images, features, intrinsics, extrinsics_inv, packed_coarse, packed_fine, u_coarse,
               u_fine, near, far, q7_mode=Q7_ZERO, workspace=None, out=None, texel_tables=None, tables_ready=False, split=None):
    """mvnerf_render_fwd = MVVNeRFRenderer._call (model_v0.py:113-184) -> (rgb, depth, fine_rgb, fine_depth).
    split: (pack_net_split(coarse), pack_net_split(fine)) -> mvnerf_render_fwd_split (fp32-grade Dense layers on the bf16 MFMA).
    texel_tables: None = gather raw features; 'auto' = allocate and build when texel_table_pays(); or a float32
    tensor (2,B,V,H,W,128) [coarse net | fine net], built by this call unless tables_ready."""
    _chk(rays_o, 'rays_o', shape=(None, None, 3))
    b, r, _ = rays_o.shape
    _chk(rays_d, 'rays_d', shape=(b, r, 3))
    _chk(images, 'images', shape=(b, None, None, None, 3))
    _, v, h, w, _ = images.shape
    _chk(features, 'features', shape=(b, v, h, w, 256))
    _chk(intrinsics, 'intrinsics', shape=(b, v, 4, 4))
    _chk(extrinsics_inv, 'extrinsics_inv', shape=(b, v, 4, 4))
    _chk(u_coarse, 'u_coarse', shape=(b, r, None))
    s = u_coarse.shape[2]
    _chk(u_fine, 'u_fine', shape=(b, r, s))
    _chk(packed_coarse, 'packed_coarse', shape=(packed_net_floats(),))
    _chk(packed_fine, 'packed_fine', shape=(packed_net_floats(),))
    dev = rays_o.device
    need = render_workspace_bytes(b, v, r, s)
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    else:
        _chk(workspace, 'workspace', dtype=torch.uint8)
        if workspace.numel() < need:
            raise ValueError(f'workspace: {workspace.numel()} bytes, need {need}')
    if out is None:
        out = (torch.empty((b, r, 3), dtype=torch.float32, device=dev), torch.empty((b, r), dtype=torch.float32, device=dev),
               torch.empty((b, r, 3), dtype=torch.float32, device=dev), torch.empty((b, r), dtype=torch.float32, device=dev))
    rgb, depth, fine_rgb, fine_depth = out
    if isinstance(texel_tables, str):
        if texel_tables != 'auto':
            raise ValueError(f"texel_tables: {texel_tables!r}, expected None, 'auto' or a tensor")
        texel_tables = (torch.empty((2, b, v, h, w, 128), dtype=torch.float32, device=dev)
                        if texel_table_pays(r, s, h, w) else None)
        tables_ready = False
    if texel_tables is not None:
        _chk(texel_tables, 'texel_tables', shape=(2, b, v, h, w, 128))
    with torch.cuda.device(dev):
        if split is None:
            rc = _lib.lib().mvnerf_render_fwd(_p(rays_o), _p(rays_d), _p(images), _p(features), _p(intrinsics),
                                              _p(extrinsics_inv), _p(packed_coarse), _p(packed_fine), _p(u_coarse),
                                              _p(u_fine), b, v, r, s, h, w, float(near), float(far), int(q7_mode), _p(rgb),
                                              _p(depth), _p(fine_rgb), _p(fine_depth), _p(workspace), _p(texel_tables),
                                              int(bool(tables_ready)), _stream(rays_o))
        else:
            nbytes = int(_lib.lib().mvnerf_packed_net_split_bytes())
            _chk(split[0], 'split_coarse', dtype=torch.uint8, shape=(nbytes,))
            _chk(split[1], 'split_fine', dtype=torch.uint8, shape=(nbytes,))
            rc = _lib.lib().mvnerf_render_fwd_split(_p(rays_o), _p(rays_d), _p(images), _p(features), _p(intrinsics),
                                                    _p(extrinsics_inv), _p(packed_coarse), _p(packed_fine), _p(split[0]), _p(split[1]),
                                                    _p(u_coarse), _p(u_fine), b, v, r, s, h, w, float(near), float(far),
                                                    int(q7_mode), _p(rgb), _p(depth), _p(fine_rgb), _p(fine_depth), _p(workspace),
                                                    _p(texel_tables), int(bool(tables_ready)), _stream(rays_o))
    _lib.check(rc, 'render_fwd')
    return rgb, depth, fine_rgb, fine_depth


# ---- op-level (unfused) operators: one per reference function ---------------------------------------
def points_on_rays(rays_o, rays_d, z):
    """o + z*d: rays (...,3), z (...,S) -> (...,S,3)."""
    _chk(rays_o, 'rays_o')
    _chk(rays_d, 'rays_d', shape=tuple(rays_o.shape))
    _chk(z, 'z', shape=tuple(rays_o.shape[:-1]) + (None,))
    s = z.shape[-1]
    out = torch.empty(tuple(z.shape) + (3,), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        rc = _lib.lib().mvnerf_points_on_rays(_p(rays_o), _p(rays_d), _p(z), z.numel() // s, s, _p(out), _stream(z))
    _lib.check(rc, 'points_on_rays')
    return out


def project_points(world, intrinsics, extrinsics_inv):
    """compute_pixel_in_image_mv: world (B,...,3) -> pix (B,V,...,2), cam (B,V,...,4)."""
    _chk(world, 'world')
    b = world.shape[0]
    _chk(intrinsics, 'intrinsics', shape=(b, None, 4, 4))
    v = intrinsics.shape[1]
    _chk(extrinsics_inv, 'extrinsics_inv', shape=(b, v, 4, 4))
    mid = tuple(world.shape[1:-1])
    n = world.numel() // (3 * b)
    pix = torch.empty((b, v) + mid + (2,), dtype=torch.float32, device=world.device)
    cam = torch.empty((b, v) + mid + (4,), dtype=torch.float32, device=world.device)
    with torch.cuda.device(world.device):
        rc = _lib.lib().mvnerf_project_points(_p(world), _p(intrinsics), _p(extrinsics_inv), b, v, n, _p(pix), _p(cam),
                                              _stream(world))
    _lib.check(rc, 'project_points')
    return pix, cam


def camera_directions(dirs, extrinsics_inv):
    """world_to_camera_direction_vector_mv: dirs (B,R,3) -> (B,V,R,3)."""
    _chk(dirs, 'dirs', shape=(None, None, 3))
    b, r, _ = dirs.shape
    _chk(extrinsics_inv, 'extrinsics_inv', shape=(b, None, 4, 4))
    v = extrinsics_inv.shape[1]
    out = torch.empty((b, v, r, 3), dtype=torch.float32, device=dirs.device)
    with torch.cuda.device(dirs.device):
        rc = _lib.lib().mvnerf_camera_directions(_p(dirs), _p(extrinsics_inv), b, v, r, _p(out), _stream(dirs))
    _lib.check(rc, 'camera_directions')
    return out


def position_encoding(position, n_freq=10, pos_encoding_freq=np.pi):
    """position (...,D) -> (..., D*2*n_freq), layout (d, k, {sin,cos})."""
    _chk(position, 'position')
    out = torch.empty(tuple(position.shape[:-1]) + (position.shape[-1] * 2 * n_freq,), dtype=torch.float32,
                      device=position.device)
    with torch.cuda.device(position.device):
        rc = _lib.lib().mvnerf_position_encoding(_p(position), position.numel(), int(n_freq),
                                                 float(np.float32(pos_encoding_freq)), _p(out), _stream(position))
    _lib.check(rc, 'position_encoding')
    return out


def bilinear_gather(images, features, pixel_locations, return_taps=False):
    """images (N,H,W,3), features (N,H,W,256), pixel_locations (N,Q,2) -> (N,Q,259) [+ taps (N,Q,4)]."""
    _chk(images, 'images', shape=(None, None, None, 3))
    n, h, w, _ = images.shape
    _chk(features, 'features', shape=(n, h, w, 256))
    _chk(pixel_locations, 'pixel_locations', shape=(n, None, 2))
    q = pixel_locations.shape[1]
    out = torch.empty((n, q, 259), dtype=torch.float32, device=images.device)
    taps = torch.empty((n, q, 4), dtype=torch.int32, device=images.device) if return_taps else None
    with torch.cuda.device(images.device):
        rc = _lib.lib().mvnerf_bilinear_gather(_p(images), _p(features), _p(pixel_locations), n, q, h, w, _p(out),
                                               _p(taps), _stream(images))
    _lib.check(rc, 'bilinear_gather')
    return (out, taps) if return_taps else out


def sigma_to_alpha(sigma, dists):
    _chk(sigma, 'sigma')
    _chk(dists, 'dists', shape=tuple(sigma.shape))
    out = torch.empty_like(sigma)
    with torch.cuda.device(sigma.device):
        rc = _lib.lib().mvnerf_sigma_to_alpha(_p(sigma), _p(dists), sigma.numel(), _p(out), _stream(sigma))
    _lib.check(rc, 'sigma_to_alpha')
    return out


def sample_pdf(bins, weights, u, q7_mode=Q7_ZERO, return_indices=False):
    """bins (...,63), weights (...,62), u (...,64) -> samples (...,64) [+ above, below int32]."""
    _chk(bins, 'bins')
    lead = tuple(bins.shape[:-1])
    _chk(weights, 'weights', shape=lead + (None,))
    _chk(u, 'u', shape=lead + (None,))
    n = bins.numel() // bins.shape[-1]
    samples = torch.empty_like(u)
    above = torch.empty(u.shape, dtype=torch.int32, device=u.device) if return_indices else None
    below = torch.empty(u.shape, dtype=torch.int32, device=u.device) if return_indices else None
    if weights.shape[-1] != bins.shape[-1] - 1:
        raise ValueError(f'weights: last dim {weights.shape[-1]}, expected {bins.shape[-1] - 1}')
    with torch.cuda.device(u.device):
        rc = _lib.lib().mvnerf_sample_pdf(_p(bins), _p(weights), _p(u), n, bins.shape[-1], u.shape[-1], int(q7_mode),
                                          _p(samples), _p(above), _p(below), _stream(u))
    _lib.check(rc, 'sample_pdf')
    return (samples, above, below) if return_indices else samples


def readout(embedding, wr, br):
    """RenderReadout: embedding (...,128), wr (128,4), br (4,) -> rgbs (...,4)."""
    _chk(embedding, 'embedding', shape=tuple(embedding.shape[:-1]) + (128,))
    _chk(wr, 'wr', shape=(128, 4))
    _chk(br, 'br', shape=(4,))
    out = torch.empty(tuple(embedding.shape[:-1]) + (4,), dtype=torch.float32, device=embedding.device)
    with torch.cuda.device(embedding.device):
        rc = _lib.lib().mvnerf_readout(_p(embedding), _p(wr), _p(br), embedding.numel() // 128, _p(out), _stream(embedding))
    _lib.check(rc, 'readout')
    return out


def finish_view(rgb, depth):
    """render_view epilogue: rgb (n,3), depth (n) -> (rgb8 (n,3) uint8, depth8 (n) uint8)."""
    _chk(depth, 'depth')
    n = depth.numel()
    _chk(rgb, 'rgb', shape=tuple(depth.shape) + (3,))
    rgb8 = torch.empty(rgb.shape, dtype=torch.uint8, device=rgb.device)
    depth8 = torch.empty(depth.shape, dtype=torch.uint8, device=rgb.device)
    scratch = torch.empty(2, dtype=torch.float32, device=rgb.device)
    with torch.cuda.device(rgb.device):
        rc = _lib.lib().mvnerf_finish_view(_p(rgb), _p(depth), n, _p(scratch), _p(rgb8), _p(depth8), _stream(rgb))
    _lib.check(rc, 'finish_view')
    return rgb8, depth8


# ---- training step (model_v0.py:186-197): forward with stash, loss gradient, backward, Adam ---------------
def stash_bytes(b, v, r, s):
    return int(_lib.lib().mvnerf_stash_bytes(int(b), int(v), int(r), int(s)))


def field_eval_stash(rays_o, rays_d, z, images, features, intrinsics, extrinsics_inv, packed_net, stash=None, texel_table=None,
                     packed_split=None):
    """Training-mode field pass: -> (rgbs (B,R,S,4), stash uint8 tensor with the trunk pre-activations).
    packed_split: pack_net_split(net) -> the split-bf16 kernel (mvnerf_field_eval_stash_split), same stash."""
    _chk(rays_o, 'rays_o', shape=(None, None, 3))
    b, r, _ = rays_o.shape
    _chk(rays_d, 'rays_d', shape=(b, r, 3))
    _chk(z, 'z', shape=(b, r, None))
    s = z.shape[2]
    _chk(images, 'images', shape=(b, None, None, None, 3))
    _, v, h, w, _ = images.shape
    _chk(features, 'features', shape=(b, v, h, w, 256))
    _chk(intrinsics, 'intrinsics', shape=(b, v, 4, 4))
    _chk(extrinsics_inv, 'extrinsics_inv', shape=(b, v, 4, 4))
    _chk(packed_net, 'packed_net', shape=(packed_net_floats(),))
    dev = rays_o.device
    need = stash_bytes(b, v, r, s)
    if stash is None or stash.numel() < need:
        stash = torch.empty(need, dtype=torch.uint8, device=dev)
    rgbs = torch.empty((b, r, s, 4), dtype=torch.float32, device=dev)
    ws = torch.empty(int(_lib.lib().mvnerf_field_workspace_bytes(b, v, r)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        if texel_table is not None:
            _chk(texel_table, 'texel_table', shape=(b, v, h, w, 128))
        if packed_split is None:
            rc = _lib.lib().mvnerf_field_eval_stash(_p(rays_o), _p(rays_d), _p(z), _p(images), _p(features), _p(texel_table),
                                                    _p(intrinsics), _p(extrinsics_inv), _p(packed_net), b, v, r, s, h, w, _p(rgbs),
                                                    _p(stash), _p(ws), _stream(rays_o))
        else:
            _chk(packed_split, 'packed_split', dtype=torch.uint8, shape=(int(_lib.lib().mvnerf_packed_net_split_bytes()),))
            rc = _lib.lib().mvnerf_field_eval_stash_split(_p(rays_o), _p(rays_d), _p(z), _p(images), _p(features), _p(texel_table),
                                                          _p(intrinsics), _p(extrinsics_inv), _p(packed_net), _p(packed_split), b, v, r, s,
                                                          h, w, _p(rgbs), _p(stash), _p(ws), _stream(rays_o))
    _lib.check(rc, 'field_eval_stash')
    return rgbs, stash


def pack_bwd_streams(net_keras):
    _chk(net_keras, 'net_keras', shape=(NET_PARAMS,))
    out = torch.empty(15 * 16384, dtype=torch.float32, device=net_keras.device)
    with torch.cuda.device(net_keras.device):
        _lib.check(_lib.lib().mvnerf_pack_bwd_streams(_p(net_keras), _p(out), _stream(net_keras)), 'pack_bwd_streams')
    return out


def mse_grad(pred, label, loss):
    """d pred of Keras MeanSquaredError; adds the loss value into the 1-element device tensor `loss`."""
    _chk(pred, 'pred')
    _chk(label, 'label', shape=tuple(pred.shape))
    _chk(loss, 'loss', shape=(1,))
    d = torch.empty_like(pred)
    with torch.cuda.device(pred.device):
        _lib.check(_lib.lib().mvnerf_mse_grad(_p(pred), _p(label), pred.numel(), _p(d), _p(loss), _stream(pred)), 'mse_grad')
    return d


def composite_bwd(z, rgbs, d_rgb, d_depth=None, d_weights=None, return_dz=False):
    """volumetric_render backward w.r.t. the per-sample (r,g,b,sigma): -> d_rgbs (...,S,4) [+ d_z (...,S)]."""
    _chk(z, 'z')
    s = z.shape[-1]
    lead = tuple(z.shape[:-1])
    _chk(rgbs, 'rgbs', shape=tuple(z.shape) + (4,))
    _chk(d_rgb, 'd_rgb', shape=lead + (3,))
    if d_depth is not None:
        _chk(d_depth, 'd_depth', shape=lead)
    if d_weights is not None:
        _chk(d_weights, 'd_weights', shape=tuple(z.shape))
    out = torch.empty_like(rgbs)
    d_z = torch.empty_like(z) if return_dz else None
    with torch.cuda.device(z.device):
        rc = _lib.lib().mvnerf_composite_bwd(_p(z), _p(rgbs), _p(d_rgb), _p(d_depth), _p(d_weights), z.numel() // s, s,
                                             _p(out), _p(d_z), _stream(z))
    _lib.check(rc, 'composite_bwd')
    return (out, d_z) if return_dz else out


def field_backward(rays_o, rays_d, z, images, features, intrinsics, extrinsics_inv, net_keras, bwd_streams, stash, rgbs,
                   d_rgbs, grad, scratch=None, d_z=None, d_features=None, texel_table=None, texel_grad=None):
    """Accumulate dL/d(net variables) of one field_eval_stash call into `grad` (247300 floats, Keras order);
    d_z (optional, (B,R,S)) is incremented by the gradient through the sample positions, d_features (optional,
    (B,V,H,W,256)) by the gradient w.r.t. the source feature maps.  texel_table (optional): project_texels of the same
    net; used for the feature rows' part of d_z, and - with the scratch texel_grad (B,V,H,W,128) - for d_features (the cotangents are
    scattered onto the 128-channel table gradient and W0 is applied once per texel; mvnerf_field_backward_table)."""
    b, r, s = z.shape
    _, v, h, w, _ = images.shape
    _chk(net_keras, 'net_keras', shape=(NET_PARAMS,))
    _chk(bwd_streams, 'bwd_streams', shape=(15 * 16384,))
    if d_z is not None:
        _chk(d_z, 'd_z', shape=(b, r, s))
    _chk(rgbs, 'rgbs', shape=(b, r, s, 4))
    _chk(d_rgbs, 'd_rgbs', shape=(b, r, s, 4))
    _chk(grad, 'grad', shape=(NET_PARAMS,))
    if d_features is not None:
        _chk(d_features, 'd_features', shape=(b, v, h, w, 256))
    need = int(_lib.lib().mvnerf_field_backward_scratch_bytes(b, v, r, s))
    if scratch is None or scratch.numel() < need:
        scratch = torch.empty(need, dtype=torch.uint8, device=z.device)
    if texel_table is not None:
        _chk(texel_table, 'texel_table', shape=(b, v, h, w, 128))
    if texel_grad is not None:
        _chk(texel_grad, 'texel_grad', shape=(b, v, h, w, 128))
    with torch.cuda.device(z.device):
        rc = _lib.lib().mvnerf_field_backward_table(_p(rays_o), _p(rays_d), _p(z), _p(images), _p(features), _p(texel_table),
                                                    _p(texel_grad), _p(intrinsics), _p(extrinsics_inv), _p(net_keras), _p(bwd_streams), _p(stash),
                                                    _p(rgbs), _p(d_rgbs), b, v, r, s, h, w, _p(scratch), _p(grad), _p(d_z),
                                                    _p(d_features), _stream(z))
    _lib.check(rc, 'field_backward')
    return scratch


# ---- the per-point part of GraspReadout as fused passes (csrc/grasp_head.hip; delta_ngf/layers.py:8-42, lmvnerf/model_v4.py:261-322) ----
def grasp_head_pack(w4, wc):
    """The four Dense(128 -> 64) kernels w4 (4,64,128) and the Dense(256 -> 64) kernel wc (64,256), torch [out, in] layout -> MFMA operand image."""
    _chk(w4, 'w4', shape=(4, 64, 128))
    _chk(wc, 'wc', shape=(64, 256))
    out = torch.empty(int(_lib.lib().mvnerf_grasp_head_packed_floats()), dtype=torch.float32, device=w4.device)
    with torch.cuda.device(w4.device):
        _lib.check(_lib.lib().mvnerf_grasp_head_pack(_p(w4), _p(wc), _p(out), _stream(w4)), 'grasp_head_pack')
    return out


def grasp_head_fwd(acts, packed, b4, bc):
    """acts (4,N,128) -> c (N,256) = [elu(W_k a_k + b_k)], y (N,64) = elu(W_c c + b_c)."""
    _chk(acts, 'acts', shape=(4, None, 128))
    n = acts.shape[1]
    _chk(packed, 'packed', shape=(int(_lib.lib().mvnerf_grasp_head_packed_floats()),))
    _chk(b4, 'b4', shape=(4, 64))
    _chk(bc, 'bc', shape=(64,))
    c = torch.empty((n, 256), dtype=torch.float32, device=acts.device)
    y = torch.empty((n, 64), dtype=torch.float32, device=acts.device)
    with torch.cuda.device(acts.device):
        _lib.check(_lib.lib().mvnerf_grasp_head_fwd(_p(acts), _p(packed), _p(b4), _p(bc), n, _p(c), _p(y), _stream(acts)), 'grasp_head_fwd')
    return c, y


def grasp_head_vjp(g_y, c, y, packed):
    """g_y (N,64) -> g_v (N,64), q (N,256), g_u (N,256), g_acts (4,N,128)."""
    _chk(g_y, 'g_y', shape=(None, 64))
    n = g_y.shape[0]
    _chk(c, 'c', shape=(n, 256))
    _chk(y, 'y', shape=(n, 64))
    _chk(packed, 'packed', shape=(int(_lib.lib().mvnerf_grasp_head_packed_floats()),))
    dev = g_y.device
    g_v = torch.empty((n, 64), dtype=torch.float32, device=dev)
    q = torch.empty((n, 256), dtype=torch.float32, device=dev)
    g_u = torch.empty((n, 256), dtype=torch.float32, device=dev)
    g_acts = torch.empty((4, n, 128), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().mvnerf_grasp_head_vjp(_p(g_y), _p(c), _p(y), _p(packed), n, _p(g_v), _p(q), _p(g_u), _p(g_acts), _stream(g_y)),
                   'grasp_head_vjp')
    return g_v, q, g_u, g_acts


def grasp_head_vjp_bwd(t_acts, g_y, c, y, q, packed):
    """t_acts (4,N,128) = dL/d(g_acts) -> out_gy (N,64) = dL/d(g_y), r (N,256), m (N,64), p (N,256) (see include/mvnerf_hip.h)."""
    _chk(t_acts, 't_acts', shape=(4, None, 128))
    n = t_acts.shape[1]
    _chk(g_y, 'g_y', shape=(n, 64))
    _chk(c, 'c', shape=(n, 256))
    _chk(y, 'y', shape=(n, 64))
    _chk(q, 'q', shape=(n, 256))
    _chk(packed, 'packed', shape=(int(_lib.lib().mvnerf_grasp_head_packed_floats()),))
    dev = t_acts.device
    out_gy = torch.empty((n, 64), dtype=torch.float32, device=dev)
    r = torch.empty((n, 256), dtype=torch.float32, device=dev)
    m = torch.empty((n, 64), dtype=torch.float32, device=dev)
    p_ = torch.empty((n, 256), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().mvnerf_grasp_head_vjp_bwd(_p(t_acts), _p(g_y), _p(c), _p(y), _p(q), _p(packed), n, _p(out_gy), _p(r), _p(m),
                                                        _p(p_), _stream(t_acts)), 'grasp_head_vjp_bwd')
    return out_gy, r, m, p_


def train_workspace_bytes(b, v, r, s, h, w, use_tables, want_d_features):
    return int(_lib.lib().mvnerf_train_workspace_bytes(int(b), int(v), int(r), int(s), int(h), int(w), int(bool(use_tables)),
                                                       int(bool(want_d_features))))


def train_call(rays_o, rays_d, images, features, intrinsics, extrinsics_inv, u_coarse, u_fine, labels, near, far, nets, packed, split,
               bwd_streams, loss, grad, outputs, workspace, q7_mode=Q7_ZERO, stop_fine_z=False, use_tables=False, d_features=None):
    """Fill a mvnerf_train_call (include/mvnerf_hip.h) from device tensors after checking shapes; the caller keeps the tensors alive.
    nets / packed / split / bwd_streams: (coarse, fine) pairs (split may be None: fp32-MFMA forward); outputs: (rgb, depth, fine_rgb,
    fine_depth); grad: (2 x 247300,) [coarse | fine]; workspace: uint8 tensor of train_workspace_bytes(...)."""
    _chk(rays_o, 'rays_o', shape=(None, None, 3))
    b, r, _ = rays_o.shape
    _chk(rays_d, 'rays_d', shape=(b, r, 3))
    _chk(images, 'images', shape=(b, None, None, None, 3))
    _, v, h, w, _ = images.shape
    _chk(features, 'features', shape=(b, v, h, w, 256))
    _chk(intrinsics, 'intrinsics', shape=(b, v, 4, 4))
    _chk(extrinsics_inv, 'extrinsics_inv', shape=(b, v, 4, 4))
    _chk(u_coarse, 'u_coarse', shape=(b, r, None))
    s = u_coarse.shape[2]
    _chk(u_fine, 'u_fine', shape=(b, r, s))
    _chk(labels, 'labels', shape=(b, r, 3))
    for k in range(2):
        _chk(nets[k], 'net', shape=(NET_PARAMS,))
        _chk(packed[k], 'packed', shape=(packed_net_floats(),))
        _chk(bwd_streams[k], 'bwd_streams', shape=(15 * 16384,))
        if split is not None:
            _chk(split[k], 'split', dtype=torch.uint8, shape=(int(_lib.lib().mvnerf_packed_net_split_bytes()),))
    _chk(loss, 'loss', shape=(1,))
    _chk(grad, 'grad', shape=(2 * NET_PARAMS,))
    rgb, depth, fine_rgb, fine_depth = outputs
    _chk(rgb, 'rgb', shape=(b, r, 3))
    _chk(depth, 'depth', shape=(b, r))
    _chk(fine_rgb, 'fine_rgb', shape=(b, r, 3))
    _chk(fine_depth, 'fine_depth', shape=(b, r))
    if d_features is not None:
        _chk(d_features, 'd_features', shape=(b, v, h, w, 256))
    _chk(workspace, 'workspace', dtype=torch.uint8)
    c = _lib.TrainCall()
    for name, t in (('rays_o', rays_o), ('rays_d', rays_d), ('images', images), ('features', features), ('intrinsics', intrinsics),
                    ('extrinsics_inv', extrinsics_inv), ('u_coarse', u_coarse), ('u_fine', u_fine), ('labels', labels),
                    ('net_coarse', nets[0]), ('net_fine', nets[1]), ('packed_coarse', packed[0]), ('packed_fine', packed[1]),
                    ('split_coarse', None if split is None else split[0]), ('split_fine', None if split is None else split[1]),
                    ('bwd_streams_coarse', bwd_streams[0]), ('bwd_streams_fine', bwd_streams[1]), ('loss', loss), ('grad', grad),
                    ('rgb', rgb), ('depth', depth), ('fine_rgb', fine_rgb), ('fine_depth', fine_depth), ('d_features', d_features),
                    ('workspace', workspace)):
        setattr(c, name, None if t is None else t.data_ptr())
    c.B, c.V, c.R, c.S, c.H, c.W = b, v, r, s, h, w
    c.near_, c.far_ = float(near), float(far)
    c.q7_mode, c.stop_fine_z, c.use_texel_tables = int(q7_mode), int(bool(stop_fine_z)), int(bool(use_tables))
    c.workspace_bytes = workspace.numel()
    return c


def adam_state(m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-7, clip=1.0, update_mask=None, repack=True):
    _chk(m, 'm', shape=(2 * NET_PARAMS,))
    _chk(v, 'v', shape=(2 * NET_PARAMS,))
    if update_mask is not None:
        _chk(update_mask, 'update_mask', dtype=torch.uint8, shape=(2 * NET_PARAMS,))
    a = _lib.AdamState()
    a.m, a.v, a.update_mask = m.data_ptr(), v.data_ptr(), None if update_mask is None else update_mask.data_ptr()
    a.lr_t, a.beta1, a.beta2, a.eps, a.clip = float(lr_t), float(beta1), float(beta2), float(eps), float(clip)
    a.repack = int(bool(repack))
    return a


def loss_and_grads(call, stream_of):
    """mvnerf_loss_and_grads: MVVNeRFRenderer.train_step's tape (model_v0.py:190-194) on a filled train_call."""
    with torch.cuda.device(stream_of.device):
        _lib.check(_lib.lib().mvnerf_loss_and_grads(ctypes.byref(call), _stream(stream_of)), 'loss_and_grads')


def apply_gradients(call, adam, stream_of):
    """mvnerf_apply_gradients: optimize() (nerf_utils.py:8-12) - clip-by-value, Adam, re-pack."""
    with torch.cuda.device(stream_of.device):
        _lib.check(_lib.lib().mvnerf_apply_gradients(ctypes.byref(call), ctypes.byref(adam), _stream(stream_of)), 'apply_gradients')


def train_step_c(call, adam, stream_of):
    """mvnerf_train_step: the two above in one C call."""
    with torch.cuda.device(stream_of.device):
        _lib.check(_lib.lib().mvnerf_train_step(ctypes.byref(call), ctypes.byref(adam), _stream(stream_of)), 'train_step')


def gemm_nt_ok(m, n, k):
    """Shapes mvnerf_gemm_nt takes."""
    return m > 0 and n > 0 and k > 0 and m % 32 == 0 and n % 64 == 0 and k % 8 == 0


def gemm_nt(a, bt, bias=None):
    """mvnerf_gemm_nt / mvnerf_gemm_nt_bias: a (M,K) @ bt (N,K)^T [+ bias (N,)] -> (M,N), fp32, deterministic (K split into ranges whose
    partials are added in order)."""
    m, k = a.shape
    n = bt.shape[0]
    _chk(a, 'a', shape=(m, k))
    _chk(bt, 'bt', shape=(n, k))
    if bias is not None:
        _chk(bias, 'bias', shape=(n,))
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    need = int(_lib.lib().mvnerf_gemm_nt_scratch_bytes(m, n, k))
    scratch = torch.empty(need, dtype=torch.uint8, device=a.device) if need else None
    with torch.cuda.device(a.device):
        if bias is None:
            rc = _lib.lib().mvnerf_gemm_nt(_p(a), _p(bt), _p(out), m, n, k, _p(scratch), _stream(a))
        else:
            rc = _lib.lib().mvnerf_gemm_nt_bias(_p(a), _p(bt), _p(bias), _p(out), m, n, k, _p(scratch), _stream(a))
    _lib.check(rc, 'gemm_nt')
    return out


def gemm_tn_ok(m, n, k):
    """Shapes mvnerf_gemm_tn takes."""
    return m > 0 and n > 0 and k > 0 and m % 8 == 0 and n % 32 == 0 and k % 64 == 0


def gemm_tn(g, a):
    """mvnerf_gemm_tn: g (M,N)^T @ a (M,K) -> (N,K), fp32, deterministic; no transposed copies."""
    m, n = g.shape
    k = a.shape[1]
    _chk(g, 'g', shape=(m, n))
    _chk(a, 'a', shape=(m, k))
    out = torch.empty((n, k), dtype=torch.float32, device=a.device)
    need = int(_lib.lib().mvnerf_gemm_tn_scratch_bytes(m, n, k))
    scratch = torch.empty(need, dtype=torch.uint8, device=a.device) if need else None
    with torch.cuda.device(a.device):
        rc = _lib.lib().mvnerf_gemm_tn(_p(g), _p(a), _p(out), m, n, k, _p(scratch), _stream(a))
    _lib.check(rc, 'gemm_tn')
    return out


def gemm_tn_batched(g, a, g2=None, a2=None, colsum_of=0):
    """mvnerf_gemm_tn_batched: out[b] = g[b]^T @ a[b] (+ g2[b]^T @ a2[b]) for a batch of weight gradients that share their M rows, and the
    column sums of g (colsum_of=1) or g2 (=2) from the same pass.  g, g2: (batch, M, N) views whose last dimension is contiguous - e.g.
    `x.view(M, batch, N).permute(1, 0, 2)` for the column blocks of one (M, batch * N) matrix: no copies are made; a, a2: (batch, M, K).
    -> (batch, N, K) [, (batch, N)]."""
    def strides(t, name):
        if t.dim() != 3 or t.stride(2) != 1 or t.dtype != torch.float32 or not t.is_cuda:
            raise ValueError(f'{name}: needs a float32 device tensor (batch, M, cols) with contiguous rows')
        return t.stride(0), t.stride(1)
    batch, m, n = g.shape
    k = a.shape[2]
    if tuple(a.shape[:2]) != (batch, m):
        raise ValueError(f'a: {tuple(a.shape)} does not match g {tuple(g.shape)}')
    q = _lib.GemmTnBatch()
    q.g, q.a = _p(g), _p(a)
    (q.g_batch_stride, q.ldg), (q.a_batch_stride, q.lda) = strides(g, 'g'), strides(a, 'a')
    if g2 is not None:
        if tuple(g2.shape) != tuple(g.shape) or tuple(a2.shape) != tuple(a.shape):
            raise ValueError('g2 / a2 must have the shapes of g / a')
        q.g2, q.a2 = _p(g2), _p(a2)
        (q.g2_batch_stride, q.ldg2), (q.a2_batch_stride, q.lda2) = strides(g2, 'g2'), strides(a2, 'a2')
    q.colsum_of = int(colsum_of)
    out = torch.empty((batch, n, k), dtype=torch.float32, device=a.device)
    colsum = torch.empty((batch, n), dtype=torch.float32, device=a.device) if colsum_of else None
    need = int(_lib.lib().mvnerf_gemm_tn_batched_scratch_bytes(m, n, k, batch, int(bool(colsum_of))))
    scratch = torch.empty(need, dtype=torch.uint8, device=a.device) if need else None
    with torch.cuda.device(a.device):
        rc = _lib.lib().mvnerf_gemm_tn_batched(ctypes.byref(q), _p(out), _p(colsum), m, n, k, batch, _p(scratch), _stream(a))
    _lib.check(rc, 'gemm_tn_batched')
    return (out, colsum) if colsum_of else out


SPLIT_KERNELS = {'split_f16': 0, 'split_bf16': 1, 'split_bf16_32x32x16': 2}


def set_split_kernel(name):
    """mvnerf_set_split_kernel: which kernel runs the split field passes of this process - 'split_f16' (default: two fp16 pieces per
    operand, three MFMAs per product block), 'split_bf16' (exact three-piece bf16 cut, six MFMAs) or 'split_bf16_32x32x16' (round 2's
    kernel).  Returns the previous name.  The environment variable MVNERF_SPLIT_MFMA overrides it at every launch."""
    if name not in SPLIT_KERNELS:
        raise ValueError(f'split kernel must be one of {sorted(SPLIT_KERNELS)}, got {name!r}')
    prev = int(_lib.lib().mvnerf_set_split_kernel(SPLIT_KERNELS[name]))
    return {v: k for k, v in SPLIT_KERNELS.items()}[prev]


def set_deterministic(on=True):
    """mvnerf_set_deterministic.  Weight gradients are ALWAYS summed in a fixed order since round 2 (stored per-workgroup partials +
    a parallel fixed-order reduction turned out faster than fp32 atomics); the call is kept for its callers, records the flag and
    returns the previous value."""
    return bool(_lib.lib().mvnerf_set_deterministic(int(bool(on))))


def adam_clip(param, grad, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-7, clip=1.0, update_mask=None):
    """optimize() (nerf_utils.py:8-12): clip-by-value then Adam, in place on `param`, `m`, `v`."""
    _chk(param, 'param')
    for t, n in ((grad, 'grad'), (m, 'm'), (v, 'v')):
        _chk(t, n, shape=tuple(param.shape))
    if update_mask is not None:
        _chk(update_mask, 'update_mask', dtype=torch.uint8, shape=tuple(param.shape))
    with torch.cuda.device(param.device):
        rc = _lib.lib().mvnerf_adam_clip(_p(param), _p(grad), _p(m), _p(v), param.numel(), float(lr_t), float(beta1),
                                         float(beta2), float(eps), float(clip), _p(update_mask), _stream(param))
    _lib.check(rc, 'adam_clip')


# ---- bf16 variant of the field pass (configs 3 / 5) ---------------------------------------------------------
def pack_net_bf16(net_keras):
    """Keras-order fp32 MLP -> bf16 MFMA operand stream (uint8 tensor of mvnerf_packed_net_bf16_bytes())."""
    _chk(net_keras, 'net_keras', shape=(NET_PARAMS,))
    out = torch.empty(int(_lib.lib().mvnerf_packed_net_bf16_bytes()), dtype=torch.uint8, device=net_keras.device)
    with torch.cuda.device(net_keras.device):
        _lib.check(_lib.lib().mvnerf_pack_net_bf16(_p(net_keras), _p(out), _stream(net_keras)), 'pack_net_bf16')
    return out


def project_texels_bf16(features, packed16, out=None, packed16_b=None):
    """mvnerf_project_texels_bf16: the texel table of one net on the bf16 MFMA (fp32 table, same layout as project_texels);
    with packed16_b both nets' tables (2,B,V,H,W,128) from one read of the feature maps.
    features: fp32, or bfloat16 (B,V,H,W,256) -> mvnerf_project_texels_bf16maps (half the bytes of the pass)."""
    maps16 = isinstance(features, torch.Tensor) and features.dtype == torch.bfloat16
    _chk(features, 'features', dtype=torch.bfloat16 if maps16 else torch.float32, shape=(None, None, None, None, 256))
    b, v, h, w, _ = features.shape
    nbytes = int(_lib.lib().mvnerf_packed_net_bf16_bytes())
    _chk(packed16, 'packed16', dtype=torch.uint8, shape=(nbytes,))
    shape = (b, v, h, w, 128) if packed16_b is None else (2, b, v, h, w, 128)
    if packed16_b is not None:
        _chk(packed16_b, 'packed16_b', dtype=torch.uint8, shape=(nbytes,))
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=features.device)
    else:
        _chk(out, 'texel_table', shape=shape)
    t0, t1 = (out, None) if packed16_b is None else (out[0], out[1])
    with torch.cuda.device(features.device):
        fn = _lib.lib().mvnerf_project_texels_bf16maps if maps16 else _lib.lib().mvnerf_project_texels_bf16
        rc = fn(_p(features), _p(packed16), _p(packed16_b), b, v, h, w, _p(t0), _p(t1), _stream(features))
    _lib.check(rc, 'project_texels_bf16')
    return out


def field_eval_bf16(rays_o, rays_d, z, images, features, intrinsics, extrinsics_inv, packed_net, packed16, return_taps=False,
                    return_embedding=False, return_fused_acts=False, texel_table=None):
    """mvnerf_field_eval_bf16: as field_eval with the Dense layers on the bf16 MFMA path.
    return_fused_acts: + (4,B,R,S,128) = view mean and the three fusion blocks (complete_output[4:]).
    features: fp32, or bfloat16 (B,V,H,W,256) -> mvnerf_field_eval_bf16maps (the gather reads bf16 texel rows)."""
    maps16 = isinstance(features, torch.Tensor) and features.dtype == torch.bfloat16
    _chk(rays_o, 'rays_o', shape=(None, None, 3))
    b, r, _ = rays_o.shape
    _chk(rays_d, 'rays_d', shape=(b, r, 3))
    _chk(z, 'z', shape=(b, r, None))
    s = z.shape[2]
    _chk(images, 'images', shape=(b, None, None, None, 3))
    _, v, h, w, _ = images.shape
    _chk(features, 'features', dtype=torch.bfloat16 if maps16 else torch.float32, shape=(b, v, h, w, 256))
    _chk(intrinsics, 'intrinsics', shape=(b, v, 4, 4))
    _chk(extrinsics_inv, 'extrinsics_inv', shape=(b, v, 4, 4))
    _chk(packed_net, 'packed_net', shape=(packed_net_floats(),))
    _chk(packed16, 'packed16', dtype=torch.uint8, shape=(int(_lib.lib().mvnerf_packed_net_bf16_bytes()),))
    dev = rays_o.device
    rgbs = torch.empty((b, r, s, 4), dtype=torch.float32, device=dev)
    taps = torch.empty((b, v, r, s, 4), dtype=torch.int32, device=dev) if return_taps else None
    emb = torch.empty((b, r, s, 128), dtype=torch.float32, device=dev) if return_embedding else None
    fused = torch.empty((4, b, r, s, 128), dtype=torch.float32, device=dev) if return_fused_acts else None
    if texel_table is not None:
        _chk(texel_table, 'texel_table', shape=(b, v, h, w, 128))
    ws = torch.empty(int(_lib.lib().mvnerf_field_workspace_bytes(b, v, r)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        fn = _lib.lib().mvnerf_field_eval_bf16maps if maps16 else _lib.lib().mvnerf_field_eval_bf16
        rc = fn(_p(rays_o), _p(rays_d), _p(z), _p(images), _p(features), _p(texel_table), _p(intrinsics), _p(extrinsics_inv), _p(packed_net),
                _p(packed16), b, v, r, s, h, w, _p(rgbs), _p(taps), _p(emb), _p(fused), _p(ws), _stream(rays_o))
    _lib.check(rc, 'field_eval_bf16')
    out = (rgbs,) + ((taps,) if return_taps else ()) + ((emb,) if return_embedding else ()) + ((fused,) if return_fused_acts else ())
    return out if len(out) > 1 else rgbs


def render_fwd_bf16(rays_o, rays_d, images, features, intrinsics, extrinsics_inv, packed_coarse, packed_fine, packed16_coarse,
                    packed16_fine, u_coarse, u_fine, near, far, q7_mode=Q7_ZERO, texel_tables='auto'):
    """`_call` with both field passes on the bf16 path (sampling, compositing and resampling stay fp32).
    texel_tables: 'auto' (build the two fp32 tables when texel_table_pays), None, or a (2,B,V,H,W,128) tensor to fill."""
    geo = (images, features, intrinsics, extrinsics_inv)
    tab_c = tab_f = None
    if isinstance(texel_tables, str):
        b, r, s = u_coarse.shape
        h, w_ = images.shape[2:4]
        texel_tables = (torch.empty((2,) + tuple(features.shape[:4]) + (128,), dtype=torch.float32, device=features.device)
                        if texel_table_pays(r, s, h, w_) else None)
    if texel_tables is not None:
        tab_c, tab_f = project_texels_bf16(features, packed16_coarse, out=texel_tables, packed16_b=packed16_fine).unbind(0)
    z = stratified_depths(u_coarse, near, far)
    rgbs_c = field_eval_bf16(rays_o, rays_d, z, *geo, packed_coarse, packed16_coarse, texel_table=tab_c)
    rgb, depth, w = composite(z, rgbs_c)
    z_all = resample(z, w, u_fine, q7_mode)
    rgbs_f = field_eval_bf16(rays_o, rays_d, z_all, *geo, packed_fine, packed16_fine, texel_table=tab_f)
    fine_rgb, fine_depth, _ = composite(z_all, rgbs_f, return_weights=False)
    return rgb, depth, fine_rgb, fine_depth


# ---- fp32-grade field pass on the bf16 matrix pipe (three-piece operand split, csrc/field_eval_split.hip) ------------------
def pack_net_split(net_keras):
    """Keras-order fp32 MLP -> three-piece bf16 operand stream (uint8 tensor of mvnerf_packed_net_split_bytes())."""
    _chk(net_keras, 'net_keras', shape=(NET_PARAMS,))
    out = torch.empty(int(_lib.lib().mvnerf_packed_net_split_bytes()), dtype=torch.uint8, device=net_keras.device)
    with torch.cuda.device(net_keras.device):
        _lib.check(_lib.lib().mvnerf_pack_net_split(_p(net_keras), _p(out), _stream(net_keras)), 'pack_net_split')
    return out


def field_eval_split(rays_o, rays_d, z, images, features, intrinsics, extrinsics_inv, packed_net, packed_split, return_taps=False,
                     return_pix=False, return_embedding=False, complete_output=False, texel_table=None):
    """mvnerf_field_eval_split: field_eval (same outputs, same fp32 bar) with the Dense layers as split-bf16 MFMA products.
    texel_table: project_texels(features, packed_net) of the same net (fp32)."""
    _chk(rays_o, 'rays_o', shape=(None, None, 3))
    b, r, _ = rays_o.shape
    _chk(rays_d, 'rays_d', shape=(b, r, 3))
    _chk(z, 'z', shape=(b, r, None))
    s = z.shape[2]
    _chk(images, 'images', shape=(b, None, None, None, 3))
    _, v, h, w, _ = images.shape
    _chk(features, 'features', shape=(b, v, h, w, 256))
    _chk(intrinsics, 'intrinsics', shape=(b, v, 4, 4))
    _chk(extrinsics_inv, 'extrinsics_inv', shape=(b, v, 4, 4))
    _chk(packed_net, 'packed_net', shape=(packed_net_floats(),))
    _chk(packed_split, 'packed_split', dtype=torch.uint8, shape=(int(_lib.lib().mvnerf_packed_net_split_bytes()),))
    dev = rays_o.device
    rgbs = torch.empty((b, r, s, 4), dtype=torch.float32, device=dev)
    taps = torch.empty((b, v, r, s, 4), dtype=torch.int32, device=dev) if return_taps else None
    pix = torch.empty((b, v, r, s, 2), dtype=torch.float32, device=dev) if return_pix else None
    emb = torch.empty((b, r, s, 128), dtype=torch.float32, device=dev) if return_embedding else None
    acts_v = torch.empty((4, b * v, r, s, 128), dtype=torch.float32, device=dev) if complete_output else None
    acts_f = torch.empty((4, b, r, s, 128), dtype=torch.float32, device=dev) if complete_output else None
    if texel_table is not None:
        _chk(texel_table, 'texel_table', shape=(b, v, h, w, 128))
    ws = torch.empty(int(_lib.lib().mvnerf_field_workspace_bytes(b, v, r)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().mvnerf_field_eval_split(_p(rays_o), _p(rays_d), _p(z), _p(images), _p(features), _p(texel_table),
                                                _p(intrinsics), _p(extrinsics_inv), _p(packed_net), _p(packed_split), b, v, r, s, h, w,
                                                _p(rgbs), _p(taps), _p(pix), _p(emb), _p(acts_v), _p(acts_f), _p(ws), _stream(rays_o))
    _lib.check(rc, 'field_eval_split')
    out = (rgbs,)
    if return_taps:
        out += (taps,)
    if return_pix:
        out += (pix,)
    if return_embedding:
        out += (emb,)
    if complete_output:
        out += (list(acts_v.unbind(0)) + list(acts_f.unbind(0)),)
    return out if len(out) > 1 else rgbs


def render_fwd_split(rays_o, rays_d, images, features, intrinsics, extrinsics_inv, packed_coarse, packed_fine, split_coarse,
                     split_fine, u_coarse, u_fine, near, far, q7_mode=Q7_ZERO, texel_tables='auto'):
    """`_call` (model_v0.py:113-184) with both field passes on the split-bf16 kernel (fp32-grade products, 1e-4 bar).
    texel_tables: 'auto' (build the two fp32 tables when texel_table_pays), None, or a (2,B,V,H,W,128) tensor to fill."""
    geo = (images, features, intrinsics, extrinsics_inv)
    tab_c = tab_f = None
    if isinstance(texel_tables, str):
        b, r, s = u_coarse.shape
        h, w_ = images.shape[2:4]
        texel_tables = (torch.empty((2,) + tuple(features.shape[:4]) + (128,), dtype=torch.float32, device=features.device)
                        if texel_table_pays(r, s, h, w_) else None)
    if texel_tables is not None:
        tab_c, tab_f = project_texels2(features, packed_coarse, packed_fine, out=texel_tables).unbind(0)
    z = stratified_depths(u_coarse, near, far)
    rgbs_c = field_eval_split(rays_o, rays_d, z, *geo, packed_coarse, split_coarse, texel_table=tab_c)
    rgb, depth, w = composite(z, rgbs_c)
    z_all = resample(z, w, u_fine, q7_mode)
    rgbs_f = field_eval_split(rays_o, rays_d, z_all, *geo, packed_fine, split_fine, texel_table=tab_f)
    fine_rgb, fine_depth, _ = composite(z_all, rgbs_f, return_weights=False)
    return rgb, depth, fine_rgb, fine_depth


# ---- the trunk as a differentiable field on query points (lmvnerf/model_v4.py:208-265) -------------
def _query_shapes(points, dirs, images, features, intrinsics, extrinsics_inv):
    _chk(points, 'points', shape=(None, None, 3))
    b, n, _ = points.shape
    _chk(dirs, 'dirs', shape=(b, n, 3))
    _chk(images, 'images', shape=(b, None, None, None, 3))
    _, v, h, w, _ = images.shape
    _chk(features, 'features', shape=(b, v, h, w, 256))
    _chk(intrinsics, 'intrinsics', shape=(b, v, 4, 4))
    _chk(extrinsics_inv, 'extrinsics_inv', shape=(b, v, 4, 4))
    return b, v, n, h, w


def query_jvp(points, dirs, t_points, t_dirs, images, features, intrinsics, extrinsics_inv, packed_net, return_primal=False):
    """mvnerf_query_jvp: tangents (B,N,3) of the query points / directions -> tangents (4,B,N,128) of the four fused
    activations (view mean, u1, u2, u3) [+ the activations themselves]."""
    b, v, n, h, w = _query_shapes(points, dirs, images, features, intrinsics, extrinsics_inv)
    _chk(t_points, 't_points', shape=(b, n, 3))
    _chk(t_dirs, 't_dirs', shape=(b, n, 3))
    _chk(packed_net, 'packed_net', shape=(packed_net_floats(),))
    dev = points.device
    t_acts = torch.empty((4, b, n, 128), dtype=torch.float32, device=dev)
    acts = torch.empty((4, b, n, 128), dtype=torch.float32, device=dev) if return_primal else None
    ws = torch.empty(int(_lib.lib().mvnerf_query_workspace_bytes(b, v, n)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().mvnerf_query_jvp(_p(points), _p(dirs), _p(t_points), _p(t_dirs), _p(images), _p(features), _p(intrinsics),
                                         _p(extrinsics_inv), _p(packed_net), b, v, n, h, w, _p(acts), _p(t_acts), _p(ws),
                                         _stream(points))
    _lib.check(rc, 'query_jvp')
    return (t_acts, acts) if return_primal else t_acts


def query_stash(points, dirs, images, features, intrinsics, extrinsics_inv, packed_net, stash=None, packed_split=None):
    """Forward of the trunk on query points with the pre-activations kept for query_vjp (field_eval_stash, S = 1, z = 0);
    packed_split: run it on the split-bf16 kernel."""
    z = torch.zeros(tuple(points.shape[:2]) + (1,), dtype=torch.float32, device=points.device)
    return field_eval_stash(points, dirs, z, images, features, intrinsics, extrinsics_inv, packed_net, stash, packed_split=packed_split)[1]


def stash_fused_acts(stash, b, v, n):
    """mvnerf_stash_fused_acts: the four fused activations of a query stash (query_stash on (b, n) points, v views) as (4, b, n, 128)."""
    _chk(stash, 'stash', dtype=torch.uint8)
    if stash.numel() < stash_bytes(b, v, n, 1):
        raise ValueError(f'stash: {stash.numel()} bytes, need {stash_bytes(b, v, n, 1)}')
    acts = torch.empty((4, b, n, 128), dtype=torch.float32, device=stash.device)
    with torch.cuda.device(stash.device):
        rc = _lib.lib().mvnerf_stash_fused_acts(_p(stash), b, v, n, _p(acts), _stream(stash))
    _lib.check(rc, 'stash_fused_acts')
    return acts


def query_vjp(points, dirs, images, features, intrinsics, extrinsics_inv, bwd_streams, stash, g_acts):
    """mvnerf_query_vjp: cotangents (4,B,N,128) of the four fused activations -> (d_points, d_dirs), each (B,N,3)."""
    b, v, n, h, w = _query_shapes(points, dirs, images, features, intrinsics, extrinsics_inv)
    _chk(g_acts, 'g_acts', shape=(4, b, n, 128))
    _chk(bwd_streams, 'bwd_streams', shape=(15 * 16384,))
    _chk(stash, 'stash', dtype=torch.uint8)
    if stash.numel() < stash_bytes(b, v, n, 1):
        raise ValueError(f'stash: {stash.numel()} bytes, need {stash_bytes(b, v, n, 1)}')
    dev = points.device
    d_points = torch.empty((b, n, 3), dtype=torch.float32, device=dev)
    d_dirs = torch.empty((b, n, 3), dtype=torch.float32, device=dev)
    scratch = torch.empty(int(_lib.lib().mvnerf_query_vjp_scratch_bytes(b, v, n)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().mvnerf_query_vjp(_p(points), _p(dirs), _p(images), _p(features), _p(intrinsics), _p(extrinsics_inv),
                                         _p(bwd_streams), _p(stash), _p(g_acts), b, v, n, h, w, _p(scratch), _p(d_points),
                                         _p(d_dirs), _stream(points))
    _lib.check(rc, 'query_vjp')
    return d_points, d_dirs
