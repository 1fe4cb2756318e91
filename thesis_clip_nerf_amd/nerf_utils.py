"""Drop-in names for the hot-path functions of the reference's `src/lib/mvnerf/nerf_utils.py`, each
backed by a HIP kernel of libmvnerf_hip.so (torch device tensors in, torch device tensors out).

Signatures keep the reference's argument order and meaning; size arguments the reference needed
only for TensorFlow reshapes (`batch_size`, `n_rays`, `n_samples`) are accepted and checked.  The
two random draws the reference makes inside the graph are explicit keyword arguments (`u=`) and
default to `torch.rand` on the device.  Host-side NumPy functions (`bbox_biased_sample`,
`camera_parameters`, the 3x3 inverse of `get_specific_rays`) stay NumPy, as in the reference.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import Q7_ZERO


def optimize(optimizer, variables, gradients, gradients_clip=0.0):
    """nerf_utils.py:8-12 on torch: clip-by-value then one optimizer step."""
    for var, grad in zip(variables, gradients):
        var.grad = grad.clamp(-gradients_clip, gradients_clip) if gradients_clip > 0 else grad
    optimizer.step()


def _ray_matrix(extrinsics, intrinsics):
    extrinsics = np.asarray(extrinsics)
    intrinsics = np.asarray(intrinsics)
    return extrinsics[:3, :3] @ np.linalg.inv(intrinsics[:3, :3]), extrinsics[:3, -1]     # nerf_utils.py:30


def get_rays(image_width, image_height, extrinsics, intrinsics, norm_direction_vector=True, device='cuda',
             dtype=torch.float32):
    """nerf_utils.py:15-24 -> (rays_o, rays_d), each (H,W,3).  float64 math on the device (Q2);
    dtype=torch.float64 returns the unrounded directions like the reference does."""
    m, origin = _ray_matrix(extrinsics, intrinsics)
    o, d, d64 = ops.get_rays_device(m, origin, device, width=image_width, height=image_height,
                                    normalize=norm_direction_vector, return_f64=True)
    shape = (image_height, image_width, 3)
    if dtype == torch.float64:
        o64 = torch.from_numpy(np.broadcast_to(np.asarray(origin, np.float64), shape).copy()).to(d64.device)
        return o64, d64.reshape(shape)
    return o.reshape(shape), d.reshape(shape)


def get_specific_rays(u, v, extrinsics, intrinsics, norm_direction_vector=True, device='cuda'):
    """nerf_utils.py:27-35 for pixel lists u (cols), v (rows) -> (rays_o, rays_d) (N,3) fp32."""
    m, origin = _ray_matrix(extrinsics, intrinsics)
    u = torch.as_tensor(np.asarray(u), dtype=torch.float32).to(device).contiguous()
    v = torch.as_tensor(np.asarray(v), dtype=torch.float32).to(device).contiguous()
    return ops.get_rays_device(m, origin, device, u=u, v=v, normalize=norm_direction_vector)


def bbox_biased_sample(n_sample, bboxes, image_height, image_width, in_box_p=0.8):
    """nerf_utils.py:38-46 (host NumPy; consumes the global np.random state exactly like the reference)."""
    n_inside = int(n_sample * in_box_p)
    inside = np.random.randint(bboxes[:2], bboxes[2:], (n_inside, 2))
    anywhere = np.random.randint((image_height, image_width), size=(n_sample - n_inside, 2))
    return np.concatenate([inside, anywhere], axis=0)


def sample_along_ray(rays_origin, rays_direction, near, far, batch_size, n_rays, n_samples, u=None, generator=None):
    """nerf_utils.py:49-61 -> (world_points (B,R,S,3), points_along_ray (B,R,S))."""
    if tuple(rays_origin.shape) != (batch_size, n_rays, 3):
        raise ValueError(f'rays_origin: shape {tuple(rays_origin.shape)}, expected ({batch_size}, {n_rays}, 3)')
    if u is None:
        u = torch.rand((batch_size, n_rays, n_samples), dtype=torch.float32, device=rays_origin.device, generator=generator)
    z = ops.stratified_depths(u, near, far)
    return ops.points_on_rays(rays_origin, rays_direction, z), z


def compute_pixel_in_image_mv(world_points, src_intrinsics, src_extrinsics_inv):
    """nerf_utils.py:64-81 -> (pixel_locations (B,V,R,S,2), camera_points_homogeneous (B,V,R,S,4))."""
    return ops.project_points(world_points, src_intrinsics, src_extrinsics_inv)


def world_to_camera_direction_vector_mv(world_direction_vectors, extrinsics_inverse, n_views):
    """nerf_utils.py:84-105 -> (B,V,R,3)."""
    if extrinsics_inverse.shape[1] != n_views:
        raise ValueError(f'extrinsics_inverse has {extrinsics_inverse.shape[1]} views, n_views={n_views}')
    return ops.camera_directions(world_direction_vectors, extrinsics_inverse)


def position_encoding(position, n_freq, pos_encoding_freq):
    """nerf_utils.py:108-126."""
    return ops.position_encoding(position, n_freq, pos_encoding_freq)


def sigma_to_alpha(sigma, dists):
    """nerf_utils.py:129-140."""
    return ops.sigma_to_alpha(sigma, dists)


def sample_pdf(bins, weights, n_samples, u=None, generator=None, q7_mode=Q7_ZERO):
    """nerf_utils.py:143-176 -> samples (B,R,n_samples)."""
    if u is None:
        u = torch.rand(tuple(bins.shape[:-1]) + (n_samples,), dtype=torch.float32, device=bins.device, generator=generator)
    elif u.shape[-1] != n_samples:
        raise ValueError(f'u: last dim {u.shape[-1]}, n_samples={n_samples}')
    return ops.sample_pdf(bins.contiguous(), weights.contiguous(), u, q7_mode)


def get_projection_features_mv(inputs, features, pixel_locations, n_rays, n_samples, batch_size):
    """nerf_utils.py:277-285: inputs = normalised images (B,V,H,W,3), features (B,V,H,W,256),
    pixel_locations (B,V,R,S,2) -> (B,V,R,S,259)."""
    b, v, h, w, _ = inputs.shape
    if (b, pixel_locations.shape[2], pixel_locations.shape[3]) != (batch_size, n_rays, n_samples):
        raise ValueError('pixel_locations does not match (batch_size, n_rays, n_samples)')
    out = ops.bilinear_gather(inputs.reshape(b * v, h, w, 3), features.reshape(b * v, h, w, 256),
                              pixel_locations.reshape(b * v, n_rays * n_samples, 2).contiguous())
    return out.reshape(b, v, n_rays, n_samples, 259)


class WarmupScheduler:
    """nerf_utils.py:288-300: linear warm-up -> constant -> x0.1 after `scale_down_after` steps.  float32 arithmetic as the
    reference (`tf.cast(step, tf.float32)`; `step / warmup * target` evaluated left to right); called with the optimizer's
    iteration count BEFORE the update (0 on the first step), as Keras does."""

    def __init__(self, target_learning_rate, warmup_steps, scale_down_after=400000):
        self.target_learning_rate = np.float32(target_learning_rate)
        self.warmup_steps = max(np.float32(1.0), np.float32(warmup_steps))
        self.scale_down_after = np.float32(scale_down_after)

    def __call__(self, step):
        step = np.float32(step)
        if step <= self.warmup_steps:
            return float(np.float32(step / self.warmup_steps) * self.target_learning_rate)
        if step <= self.scale_down_after:
            return float(self.target_learning_rate)
        return float(np.float32(0.1) * self.target_learning_rate)
