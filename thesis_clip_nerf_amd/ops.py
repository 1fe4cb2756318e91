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


def field_eval(rays_o, rays_d, z, images, features, intrinsics, extrinsics_inv, packed_net, return_taps=False,
               return_pix=False):
    """mvnerf_field_eval: -> rgbs (B,R,S,4) [+ tap_idx (B,V,R,S,4) int32] [+ pix (B,V,R,S,2)]."""
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
    with torch.cuda.device(dev):
        rc = _lib.lib().mvnerf_field_eval(_p(rays_o), _p(rays_d), _p(z), _p(images), _p(features), _p(intrinsics),
                                          _p(extrinsics_inv), _p(packed_net), b, v, r, s, h, w, _p(rgbs), _p(taps),
                                          _p(pix), _stream(rays_o))
    _lib.check(rc, 'field_eval')
    out = (rgbs,)
    if return_taps:
        out += (taps,)
    if return_pix:
        out += (pix,)
    return out if len(out) > 1 else rgbs


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


def resample(z, weights, u_fine, q7_mode=Q7_ZERO, return_aux=False):
    """mvnerf_resample: -> z_all (...,2S) [+ z_fine, above, below]."""
    _chk(z, 'z')
    s = z.shape[-1]
    _chk(weights, 'weights', shape=tuple(z.shape))
    _chk(u_fine, 'u_fine', shape=tuple(z.shape))
    n = z.numel() // s
    z_all = torch.empty(tuple(z.shape[:-1]) + (2 * s,), dtype=torch.float32, device=z.device)
    z_fine = torch.empty_like(z) if return_aux else None
    above = torch.empty(z.shape, dtype=torch.int32, device=z.device) if return_aux else None
    below = torch.empty(z.shape, dtype=torch.int32, device=z.device) if return_aux else None
    with torch.cuda.device(z.device):
        rc = _lib.lib().mvnerf_resample(_p(z), _p(weights), _p(u_fine), n, s, int(q7_mode), _p(z_all), _p(z_fine),
                                        _p(above), _p(below), _stream(z))
    _lib.check(rc, 'resample')
    return (z_all, z_fine, above, below) if return_aux else z_all


def render_workspace_bytes(b, r, s):
    return int(_lib.lib().mvnerf_render_workspace_bytes(int(b), int(r), int(s)))


def render_fwd(rays_o, rays_d, images, features, intrinsics, extrinsics_inv, packed_coarse, packed_fine, u_coarse,
               u_fine, near, far, q7_mode=Q7_ZERO, workspace=None, out=None):
    """mvnerf_render_fwd = MVVNeRFRenderer._call (model_v0.py:113-184) -> (rgb, depth, fine_rgb, fine_depth)."""
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
    need = render_workspace_bytes(b, r, s)
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
    with torch.cuda.device(dev):
        rc = _lib.lib().mvnerf_render_fwd(_p(rays_o), _p(rays_d), _p(images), _p(features), _p(intrinsics),
                                          _p(extrinsics_inv), _p(packed_coarse), _p(packed_fine), _p(u_coarse),
                                          _p(u_fine), b, v, r, s, h, w, float(near), float(far), int(q7_mode), _p(rgb),
                                          _p(depth), _p(fine_rgb), _p(fine_depth), _p(workspace), _stream(rays_o))
    _lib.check(rc, 'render_fwd')
    return rgb, depth, fine_rgb, fine_depth
