"""Synthetic scenes, cameras and network weights for tests and benchmarks.

There is no dataset in this image (the reference's `src/lib/dataset` submodule is absent), so every
test and benchmark input is drawn from ``numpy.random.default_rng(seed)``:

* cameras on a look-at ring (centre (0,0,0.8)... see :func:`look_at_pose`), OpenCV convention
  (x right, y down, z forward) which is what ``get_specific_rays`` (nerf_utils.py:27-35) and
  ``compute_pixel_in_image_mv`` (nerf_utils.py:64-81) assume;
* images U[0,1), feature maps N(0, 0.5^2), both fp32, NHWC;
* Glorot-uniform kernels (the Keras default used by layers.py:263-270), zero bias unless asked.
"""
from __future__ import annotations

import numpy as np

N_IN = 379
N_HIDDEN = 128
N_BLOCKS = 6
NET_PARAMS = 247300


def look_at_pose(position, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """Camera-to-world 4x4 (float64) with +z looking at `target`, y pointing down."""
    c = np.asarray(position, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - c
    f /= np.linalg.norm(f)
    x = np.cross(f, np.asarray(up, dtype=np.float64))
    x /= np.linalg.norm(x)
    y = np.cross(f, x)
    pose = np.eye(4)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = x, y, f, c
    return pose


def ring_pose(azimuth, centre=(0.0, 0.0, 0.0), radius=0.8, elevation=np.pi / 4):
    centre = np.asarray(centre, dtype=np.float64)
    pos = centre + radius * np.array([np.cos(elevation) * np.cos(azimuth),
                                      np.cos(elevation) * np.sin(azimuth),
                                      np.sin(elevation)])
    return look_at_pose(pos, centre)


def pinhole(width, height, focal_scale=0.9):
    f = focal_scale * width
    return np.array([[f, 0, width / 2], [0, f, height / 2], [0, 0, 1]], dtype=np.float32)


def pad_intrinsics(k3):
    """3x3 -> 4x4 as data_generator/util.py:4-10 does."""
    k4 = np.eye(4, dtype=np.float64)
    k4[:3, :3] = k3
    return k4


def glorot_net(rng, bias_scale=0.0):
    """One MLP (trunk + read-out) as the flat 247 300-float Keras-order buffer.

    Order: W0[379,128] b0 | 6 x (W1[128,128] b1 W2[128,128] b2) | Wr[128,4] br.
    """
    parts = []

    def dense(n_in, n_out):
        lim = np.sqrt(6.0 / (n_in + n_out))
        parts.append(rng.uniform(-lim, lim, size=(n_in, n_out)).astype(np.float32).reshape(-1))
        parts.append((bias_scale * rng.standard_normal(n_out)).astype(np.float32))

    dense(N_IN, N_HIDDEN)
    for _ in range(N_BLOCKS):
        dense(N_HIDDEN, N_HIDDEN)
        dense(N_HIDDEN, N_HIDDEN)
    dense(N_HIDDEN, 4)
    flat = np.concatenate(parts)
    assert flat.size == NET_PARAMS
    return flat


def _draw_features(rng, shape, sigma, as_f32):
    if as_f32:
        return np.float32(sigma) * rng.standard_normal(shape, dtype=np.float32)
    return (sigma * rng.standard_normal(shape)).astype(np.float32)


def make_scene(seed=0, batch=1, n_views=1, height=64, width=64, n_rays=None, n_samples=64,
               bias_scale=0.0, feature_sigma=0.5, features32=False, with_features=True):
    """All inputs of one `_call` (model_v0.py:113) as NumPy fp32 arrays.

    n_rays=None -> every pixel of a `height x width` target view (row-major), i.e. H*W rays.
    Returns a dict: rays_o, rays_d (B,R,3); images (B,V,H,W,3); features (B,V,H,W,256);
    intrinsics, extrinsics_inv (B,V,4,4); u_coarse (B,R,S); u_fine (B,R,S); coarse, fine (flat
    nets); near, far; tgt_pose (B,4,4 f64); tgt_intrinsics (3,3 f32).
    features32: draw the feature maps directly as float32 (480x640 maps: no float64 temporary; a different random
    stream than the default, so values differ from features32=False).  with_features=False: `features` is None (the
    caller fills large maps on the device).
    """
    rng = np.random.default_rng(seed)
    k3 = pinhole(width, height)
    k4 = pad_intrinsics(k3)
    rays_o, rays_d, tgt_poses, einv, kk = [], [], [], [], []
    for _ in range(batch):
        tgt = ring_pose(rng.uniform(0, 2 * np.pi))
        tgt_poses.append(tgt)
        m = tgt[:3, :3] @ np.linalg.inv(k3)
        if n_rays is None:
            uu, vv = np.meshgrid(np.arange(width, dtype=np.float32), np.arange(height, dtype=np.float32),
                                 indexing='xy')
            uu, vv = uu.reshape(-1), vv.reshape(-1)
        else:
            vv = rng.integers(0, height, size=n_rays).astype(np.float32)
            uu = rng.integers(0, width, size=n_rays).astype(np.float32)
        d = (m @ np.stack([uu, vv, np.ones_like(uu)], 0)).T
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays_d.append(d.astype(np.float32))
        rays_o.append(np.broadcast_to(tgt[:3, 3], d.shape).astype(np.float32))
        ev, kv = [], []
        for _ in range(n_views):
            src = ring_pose(rng.uniform(0, 2 * np.pi))
            ev.append(np.linalg.inv(src))
            kv.append(k4)
        einv.append(ev)
        kk.append(kv)
    r = rays_o[0].shape[0]
    scene = dict(
        rays_o=np.stack(rays_o), rays_d=np.stack(rays_d),
        images=rng.random((batch, n_views, height, width, 3), dtype=np.float32),
        features=_draw_features(rng, (batch, n_views, height, width, 256), feature_sigma, features32) if with_features else None,
        intrinsics=np.asarray(kk, dtype=np.float32), extrinsics_inv=np.asarray(einv, dtype=np.float32),
        u_coarse=rng.random((batch, r, n_samples), dtype=np.float32),
        u_fine=rng.random((batch, r, n_samples), dtype=np.float32),
        coarse=glorot_net(rng, bias_scale), fine=glorot_net(rng, bias_scale),
        near=0.3, far=1.3, n_samples=n_samples,
        tgt_pose=np.stack(tgt_poses), tgt_intrinsics=k3,
    )
    return scene
