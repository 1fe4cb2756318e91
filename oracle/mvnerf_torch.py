"""Differentiable twin of oracle/mvnerf_oracle.py in PyTorch (CPU).  TEST INFRASTRUCTURE ONLY.

Purpose: reference gradients for the training step (model_v0.py:186-197: loss = MSE(y, rgb) +
MSE(y, fine_rgb), gradients w.r.t. every MLP variable) via autograd, in fp32 or fp64.  The forward
math is the same op sequence as the NumPy oracle (checked in tests/test_oracle_torch.py); gradients
flow exactly where TensorFlow's would: through the bilinear lerp factors, the positional encoding,
`sample_pdf` (cdf/bin gathers and the `t` interpolation) and the sort - there is NO stop_gradient on
the importance samples in the reference.  `stop_fine_z=True` detaches the fine depths
instead, which is the variant a first backward implementation may target; both are available so a
test can state which one it checks.

Parity status: as for the NumPy oracle - unpinned by the reference (no tests, TensorFlow absent).
"""
from __future__ import annotations

import math

import numpy as np
import torch

N_FREQ = 10
N_HIDDEN = 128
N_IN = 379
N_BLOCKS = 6
NET_PARAMS = 247300


def unflatten_net(flat):
    """Views into a flat Keras-order parameter tensor (gradients accumulate into `flat.grad`)."""
    assert flat.numel() == NET_PARAMS
    pos = 0

    def take(*shape):
        nonlocal pos
        n = int(np.prod(shape))
        out = flat[pos:pos + n].reshape(shape)
        pos += n
        return out

    net = {'W0': take(N_IN, N_HIDDEN), 'b0': take(N_HIDDEN), 'blocks': []}
    for _ in range(N_BLOCKS):
        net['blocks'].append((take(N_HIDDEN, N_HIDDEN), take(N_HIDDEN), take(N_HIDDEN, N_HIDDEN), take(N_HIDDEN)))
    net['Wr'] = take(N_HIDDEN, 4)
    net['br'] = take(4)
    return net


def points_on_rays(o, d, z):
    return o[:, :, None, :] + z[..., None] * d[:, :, None, :]


def sample_along_ray(o, d, near, far, n_samples, u):
    step = (far - near) / n_samples
    lower = torch.tensor([near + i * step for i in range(n_samples)], dtype=u.dtype)
    z = lower + u * torch.tensor(step, dtype=u.dtype)
    return points_on_rays(o, d, z), z


def _matvec4(m, x, y, z, w):
    return [((m[..., r, 0] * x + m[..., r, 1] * y) + m[..., r, 2] * z) + m[..., r, 3] * w for r in range(4)]


def compute_pixel_in_image_mv(world, k4, einv):
    p = world[:, None]
    e = einv[:, :, None, None]
    k = k4[:, :, None, None]
    one = torch.ones((), dtype=world.dtype)
    c = _matvec4(e, p[..., 0], p[..., 1], p[..., 2], one)
    q = _matvec4(k, c[0], c[1], c[2], c[3])
    den = torch.clamp(q[2], min=1e-8)
    px = torch.clamp(q[0] / den, -1e6, 1e6)
    py = torch.clamp(q[1] / den, -1e6, 1e6)
    return torch.stack([px, py], -1), torch.stack(c, -1)


def world_to_camera_direction_vector_mv(dirs, einv):
    d = dirs[:, None]
    e = einv[:, :, None]
    c = _matvec4(e, d[..., 0], d[..., 1], d[..., 2], torch.ones((), dtype=dirs.dtype))
    return torch.stack(c[:3], -1)


def interpolate_bilinear_xy(grid, query):
    """tfa.image.interpolate_bilinear(indexing='xy'): differentiable w.r.t. grid and query (via alpha)."""
    n, h, w, c = grid.shape
    x, y = query[..., 0], query[..., 1]
    fx = torch.clamp(torch.floor(x), 0, w - 2)
    fy = torch.clamp(torch.floor(y), 0, h - 2)
    ax = torch.clamp(x - fx, 0, 1)[..., None]
    ay = torch.clamp(y - fy, 0, 1)[..., None]
    x0, y0 = fx.long(), fy.long()
    flat = grid.reshape(n * h * w, c)
    base = torch.arange(n)[:, None] * (h * w) + y0 * w + x0
    tl, tr, bl, br = flat[base], flat[base + 1], flat[base + w], flat[base + w + 1]
    top = ax * (tr - tl) + tl
    bot = ax * (br - bl) + bl
    return ay * (bot - top) + top


def position_encoding(pos, n_freq=N_FREQ, freq0=math.pi):
    f0 = torch.tensor(np.float32(freq0).item(), dtype=pos.dtype)
    freq = f0 * (2.0 ** torch.arange(n_freq, dtype=pos.dtype))
    arg = pos[..., None] * freq
    enc = torch.stack([torch.sin(arg), torch.cos(arg)], -1)
    return enc.reshape(*pos.shape[:-1], -1)


def resnet_block(x, w1, b1, w2, b2):
    r = torch.relu(x) @ w1 + b1
    r = torch.relu(r) @ w2 + b2
    return x + r


def mv_embedding(net, cam_xyz, cam_dir, feat, n_views, complete_output=False):
    x = torch.cat([position_encoding(cam_xyz), position_encoding(cam_dir), feat], -1)
    x = x @ net['W0'] + net['b0']
    outs = [x]
    for blk in net['blocks'][:3]:
        outs.append(resnet_block(outs[-1], *blk))
    pre = outs[-1].reshape(outs[-1].shape[0] // n_views, n_views, *outs[-1].shape[1:])
    outs.append(pre.sum(1) / n_views)
    for blk in net['blocks'][3:]:
        outs.append(resnet_block(outs[-1], *blk))
    return outs if complete_output else outs[-1]


def render_readout(net, emb):
    o = torch.relu(emb) @ net['Wr'] + net['br']
    return torch.sigmoid(o[..., :3]), torch.nn.functional.softplus(o[..., 3])


def volumetric_render(zs, density, chroma):
    dists = zs[..., 1:] - zs[..., :-1]
    dists = torch.cat([dists, dists[..., -1:]], -1)
    alpha = 1.0 - torch.exp(-dists * torch.relu(density))
    t = 1.0 - alpha + 1e-10
    trans = torch.cumprod(t, -1)
    trans = torch.cat([torch.ones_like(trans[..., :1]), trans[..., :-1]], -1)
    w = alpha * trans
    return (w[..., None] * chroma).sum(-2), (w * zs).sum(-1), w


def sample_pdf(bins, weights, u, q7_zero=True, unfused=False):
    """nerf_utils.py:143-176.  unfused=True keeps the reference's op granularity for the CPU-baseline timing:
    `above` by the 63-step compare-accumulate of the tf.scan (:153-160) instead of one broadcast compare, cdf / bins
    materialised n_samples times (`repeat`, :163,167) before the four gathers.  Same values either way."""
    stable = weights + 1e-5
    w_sum = stable.sum(-1, keepdim=True)
    w_sum = torch.where(w_sum.abs() == 0, torch.ones_like(w_sum), w_sum)
    pdf = stable / w_sum
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    nb = bins.shape[-1]
    if unfused:
        above = torch.zeros(u.shape, dtype=torch.int32)
        for j in range(cdf.shape[-1]):
            above = above + (u >= cdf[..., j:j + 1]).to(torch.int32)
        above = above.to(torch.int64)
    else:
        above = (u[..., None] >= cdf[..., None, :]).sum(-1)
    below = torch.clamp(above - 1, 0, nb - 1)

    if unfused:
        n = u.shape[-1]
        cdf_rep = cdf[..., None, :].repeat(1, 1, n, 1)
        bins_rep = bins[..., None, :].repeat(1, 1, n, 1)

        def gather(tab, idx):
            rep = cdf_rep if tab is cdf else bins_rep
            val = torch.gather(rep, -1, torch.clamp(idx, max=nb - 1)[..., None])[..., 0]
            return torch.where(idx >= nb, torch.zeros_like(val), val) if q7_zero else val
    else:
        def gather(tab, idx):
            val = torch.gather(tab, -1, torch.clamp(idx, max=nb - 1))
            return torch.where(idx >= nb, torch.zeros_like(val), val) if q7_zero else val

    cdf_a, cdf_b = gather(cdf, above), gather(cdf, below)
    bins_a, bins_b = gather(bins, above), gather(bins, below)
    den = cdf_a - cdf_b
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    t = (u - cdf_b) / den
    return bins_b + t * (bins_a - bins_b)


def field_eval(net, o, d, zs, images, features, k4, einv):
    b, v, h, w, _ = images.shape
    r, s = zs.shape[1:3]
    world = points_on_rays(o, d, zs)
    pix, cam = compute_pixel_in_image_mv(world, k4, einv)
    grid = torch.cat([images * 2.0 - 1.0, features], -1).reshape(b * v, h, w, -1)
    feat = interpolate_bilinear_xy(grid, pix.reshape(b * v, r * s, 2)).reshape(b * v, r, s, -1)
    cdir = world_to_camera_direction_vector_mv(d, einv)
    cdir = cdir[:, :, :, None, :].expand(b, v, r, s, 3)
    emb = mv_embedding(net, cam[..., :3].reshape(b * v, r, s, 3), cdir.reshape(b * v, r, s, 3), feat, v)
    return render_readout(net, emb)


def query_acts(net, points, dirs, images, features, k4, einv):
    """The trunk on arbitrary query points with `complete_output` (lmvnerf/model_v4.py:216-262: camera points and
    directions from poses, interpolate_bilinear, fine_embedding(...)[4:]): points, dirs (B,N,3) world space ->
    list of the 4 fused activations [view mean, u1, u2, u3], each (B,N,128).  Differentiable w.r.t. points / dirs
    (through PE, the lerp factors) to any order autograd supports."""
    b, v, h, w, _ = images.shape
    n = points.shape[1]
    world = points[:, :, None, :]                                        # (B,N,1,3): one "sample" per point
    pix, cam = compute_pixel_in_image_mv(world, k4, einv)
    grid = torch.cat([images * 2.0 - 1.0, features], -1).reshape(b * v, h, w, -1)
    feat = interpolate_bilinear_xy(grid, pix.reshape(b * v, n, 2)).reshape(b * v, n, 1, -1)
    cdir = world_to_camera_direction_vector_mv(dirs, einv)             # (B,V,N,3)
    outs = mv_embedding(net, cam[..., :3].reshape(b * v, n, 1, 3), cdir.reshape(b * v, n, 1, 3), feat, v, complete_output=True)
    return [o[:, :, 0] for o in outs[4:]]


def render_call(coarse_flat, fine_flat, o, d, images, k4, einv, features, near, far, n_samples, u_coarse, u_fine,
                stop_fine_z=False, q7_zero=True, unfused=False):
    """model_v0.py:113-184 -> (rgb, depth, fine_rgb, fine_depth); differentiable w.r.t. the flat nets.
    unfused: sample_pdf at the reference's op granularity (bench.py cpu_baseline)."""
    cn, fn = unflatten_net(coarse_flat), unflatten_net(fine_flat)
    _, z = sample_along_ray(o, d, near, far, n_samples, u_coarse)
    c_rgb, c_sigma = field_eval(cn, o, d, z, images, features, k4, einv)
    rgb, depth, weights = volumetric_render(z, c_sigma, c_rgb)
    z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
    z_fine = sample_pdf(z_mid, weights[..., 1:-1], u_fine, q7_zero, unfused)
    all_zs = torch.sort(torch.cat([z, z_fine], -1), -1).values
    if stop_fine_z:
        all_zs = all_zs.detach()
    f_rgb, f_sigma = field_eval(fn, o, d, all_zs, images, features, k4, einv)
    fine_rgb, fine_depth, _ = volumetric_render(all_zs, f_sigma, f_rgb)
    return rgb, depth, fine_rgb, fine_depth


def train_loss_and_grads(coarse_flat, fine_flat, labels, scene, dtype=torch.float64, stop_fine_z=False, feature_grad=False):
    """model_v0.py:190-194: loss = MSE(labels, rgb) + MSE(labels, fine_rgb); returns loss and the two flat
    gradient vectors (Keras MeanSquaredError: mean over every element); with feature_grad also dL/d(features)."""
    def t(a):
        return torch.as_tensor(np.asarray(a)).to(dtype)
    cf = t(coarse_flat).clone().requires_grad_(True)
    ff = t(fine_flat).clone().requires_grad_(True)
    feats = t(scene['features']).clone().requires_grad_(feature_grad)
    out = render_call(cf, ff, t(scene['rays_o']), t(scene['rays_d']), t(scene['images']), t(scene['intrinsics']),
                      t(scene['extrinsics_inv']), feats, scene['near'], scene['far'], scene['n_samples'],
                      t(scene['u_coarse']), t(scene['u_fine']), stop_fine_z=stop_fine_z)
    y = t(labels)
    loss = ((y - out[0]) ** 2).mean() + ((y - out[2]) ** 2).mean()
    loss.backward()
    res = (loss.item(), cf.grad.numpy(), ff.grad.numpy(), [o.detach().numpy() for o in out])
    return res + (feats.grad.numpy(),) if feature_grad else res
