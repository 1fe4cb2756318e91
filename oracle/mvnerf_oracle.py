"""CPU oracle for the MVNeRF volumetric-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This module is a NumPy fp32 restatement of the reference algorithm
(TWeber132/thesis-clip-nerf, TensorFlow 2.11).  It is the *checker* for the HIP path in
``thesis_clip_nerf_amd`` and the ``cpu_baseline`` leg of ``bench.py``; nothing in the product
package may import it (``tests/test_layout.py`` enforces that).

Parity pin status
-----------------
* ``get_rays`` / ``get_specific_rays`` / ``bbox_biased_sample`` are pinned by golden vectors
  produced by the reference's own NumPy functions (``tests/golden/make_golden.py``).
* Everything that is TensorFlow in the reference (``nerf_utils.py:49-176,277-285``,
  ``layers.py:262-397``, ``model_v0.py:89-184``) cannot run in this image (no TensorFlow, no
  tensorflow_addons, no network) and the reference ships no tests or fixtures:
  **parity unpinned** for those functions beyond the closed-form known-answer tests in
  ``tests/test_oracle_kat.py``.
* Third-party arithmetic restated from its published algorithm:
  ``tensorflow_addons.image.interpolate_bilinear`` (tensorflow_addons 0.19/0.20, unpinned in
  ``dev.Dockerfile:30``) -> :func:`interpolate_bilinear_xy`.

Arithmetic contract (what "bit-exact integer indices" is measured against)
-------------------------------------------------------------------------
All floating point is IEEE fp32, one rounding per written operation, **no fused multiply-add**
on the geometry chain (sample depth -> world point -> camera point -> pixel -> floor/alpha) and on
the CDF chain of ``sample_pdf``; 4x4 mat-vec products are evaluated left to right
``((m0*x + m1*y) + m2*z) + m3*w``; sums that feed integer results (``w_sum``, ``cdf``) are
strictly sequential.  TensorFlow leaves these orders to cuBLAS/Eigen, so this is a choice, made
once here and mirrored by the HIP kernels.  Dense layers use ``numpy.matmul`` (BLAS order); the
HIP path uses k-ordered fp32 MFMA chains; they agree to ~1e-6 relative, inside the 1e-4 bar.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
N_FREQ = 10
N_HIDDEN = 128
N_FEAT = 256
N_IN = 2 * 3 * 2 * N_FREQ + 3 + N_FEAT  # 60 + 60 + 3 + 256 = 379
N_BLOCKS = 6
NET_PARAMS = N_IN * N_HIDDEN + N_HIDDEN + N_BLOCKS * 2 * (N_HIDDEN * N_HIDDEN + N_HIDDEN) + N_HIDDEN * 4 + 4

Q7_ZERO = 0   # out-of-range gather returns 0 (TensorFlow-GPU gather_nd behaviour; the reference ran on GPU)
Q7_CLAMP = 1  # clamp `above` to the last bin


# --------------------------------------------------------------------------------------
# a1-a3: host-side ray generation (NumPy in the reference as well)
# --------------------------------------------------------------------------------------
def get_specific_rays(u, v, extrinsics, intrinsics, norm_direction_vector=True):
    """Rays through pixel (u=col, v=row).  Reference: nerf_utils.py:27-35.

    Q1: no +0.5 pixel-centre offset.  Q2: float64 math (extrinsics are float64, the float32
    intrinsics are inverted in float32 by LAPACK and then promoted).
    Returns (rays_o, rays_d) as float64 (N,3).
    """
    u = np.asarray(u)
    v = np.asarray(v)
    pix = np.stack((u, v, np.ones_like(u)), axis=0)                      # (3,N)
    m = extrinsics[:3, :3] @ np.linalg.inv(intrinsics[:3, :3])           # (3,3)
    d = (m @ pix).T                                                      # (N,3)
    if norm_direction_vector:
        d = d / np.linalg.norm(d, axis=1, keepdims=True)
    o = np.broadcast_to(extrinsics[:3, -1], d.shape)
    return o, d


def get_rays(image_width, image_height, extrinsics, intrinsics, norm_direction_vector=True):
    """Full-image ray grid, row-major (H,W,3).  Reference: nerf_utils.py:15-24."""
    u, v = np.meshgrid(np.arange(image_width, dtype=np.float32),
                       np.arange(image_height, dtype=np.float32), indexing='xy')
    o, d = get_specific_rays(u.reshape(-1), v.reshape(-1), extrinsics, intrinsics, norm_direction_vector)
    return (o.reshape(image_height, image_width, 3), d.reshape(image_height, image_width, 3))


def bbox_biased_sample(n_sample, bboxes, image_height, image_width, in_box_p=0.8):
    """Training pixel choice, returns (n,2) int64 (row, col).  Reference: nerf_utils.py:38-46.

    Consumes the *global* ``np.random`` state exactly like the reference (two ``randint`` calls).
    """
    n_inside = int(n_sample * in_box_p)
    n_random = n_sample - n_inside
    inside = np.random.randint(bboxes[:2], bboxes[2:], (n_inside, 2))
    anywhere = np.random.randint((image_height, image_width), size=(n_random, 2))
    return np.concatenate([inside, anywhere], axis=0)


def camera_parameters(pose, intrinsics3x3):
    """(E^-1, K padded to 4x4).  Reference: data_generator/util.py:4-10."""
    k = np.reshape(intrinsics3x3, (3, 3))
    k4 = np.concatenate((k, np.zeros((3, 1))), axis=1)
    k4 = np.concatenate((k4, np.array([[0, 0, 0, 1]])), axis=0)
    return np.linalg.inv(pose), k4


# --------------------------------------------------------------------------------------
# a4: stratified sampling
# --------------------------------------------------------------------------------------
def stratified_lower_bounds(near, far, n_samples):
    """nerf_utils.py:50-52: python-float (f64) edges, cast to fp32."""
    step = (far - near) / n_samples
    edges = np.array([near + i * step for i in range(n_samples + 1)], dtype=F32)
    return edges[:-1], F32(step)


def sample_along_ray(rays_o, rays_d, near, far, n_samples, u):
    """z = lower + u*step ; p = o + z*d.  Reference: nerf_utils.py:49-61 (F11: u is explicit).

    rays_o, rays_d: (B,R,3) f32 ; u: (B,R,S) f32 in [0,1).  Returns (world (B,R,S,3), z (B,R,S)).
    """
    lower, step = stratified_lower_bounds(near, far, n_samples)
    z = (lower[None, None, :] + (u.astype(F32) * step)).astype(F32)
    world = points_on_rays(rays_o, rays_d, z)
    return world, z


def points_on_rays(rays_o, rays_d, z):
    """o + z*d with separate mul and add roundings (model_v0.py:157-158, nerf_utils.py:59-60)."""
    return (rays_o[:, :, None, :] + (z[..., None] * rays_d[:, :, None, :]).astype(F32)).astype(F32)


# --------------------------------------------------------------------------------------
# a5, a7: projection into the source views
# --------------------------------------------------------------------------------------
def _matvec4(m, x, y, z, w):
    """Rows of a (..,4,4) matrix times (x,y,z,w); left-to-right, one rounding per op."""
    out = []
    for r in range(4):
        acc = (m[..., r, 0] * x).astype(F32)
        acc = (acc + (m[..., r, 1] * y).astype(F32)).astype(F32)
        acc = (acc + (m[..., r, 2] * z).astype(F32)).astype(F32)
        acc = (acc + (m[..., r, 3] * w).astype(F32)).astype(F32)
        out.append(acc)
    return out


def compute_pixel_in_image_mv(world_points, src_intrinsics, src_extrinsics_inv):
    """Reference: nerf_utils.py:64-81.

    world_points (B,R,S,3); K4, Einv (B,V,4,4).
    Returns pixel_locations (B,V,R,S,2) [x,y] and camera_points_homogeneous (B,V,R,S,4).
    Q5: divide by max(q_z, 1e-8), clip to +-1e6.
    """
    p = world_points.astype(F32)[:, None]                      # (B,1,R,S,3)
    e = src_extrinsics_inv.astype(F32)[:, :, None, None]       # (B,V,1,1,4,4)
    k = src_intrinsics.astype(F32)[:, :, None, None]
    one = F32(1.0)
    c = _matvec4(e, p[..., 0], p[..., 1], p[..., 2], one)
    q = _matvec4(k, c[0], c[1], c[2], c[3])
    den = np.maximum(q[2], F32(1e-8))
    px = np.clip((q[0] / den).astype(F32), F32(-1e6), F32(1e6))
    py = np.clip((q[1] / den).astype(F32), F32(-1e6), F32(1e6))
    return np.stack([px, py], axis=-1), np.stack(c, axis=-1)


def world_to_camera_direction_vector_mv(world_dirs, extrinsics_inverse):
    """Reference: nerf_utils.py:84-105.  Q3: homogeneous w = 1, so the translation is added.

    world_dirs (B,R,3); Einv (B,V,4,4) -> (B,V,R,3).
    """
    d = world_dirs.astype(F32)[:, None]                        # (B,1,R,3)
    e = extrinsics_inverse.astype(F32)[:, :, None]             # (B,V,1,4,4)
    c = _matvec4(e, d[..., 0], d[..., 1], d[..., 2], F32(1.0))
    return np.stack(c[:3], axis=-1)


# --------------------------------------------------------------------------------------
# a6: bilinear gather (tensorflow_addons.image.interpolate_bilinear, indexing='xy')
# --------------------------------------------------------------------------------------
def bilinear_taps(pix, height, width):
    """Integer taps and fp32 lerp factors of tfa interpolate_bilinear(indexing='xy').

    pix (...,2) = (x, y).  Returns x0, y0 (int32, clamped to [0, size-2]) and ax, ay (f32 in [0,1]).
    """
    x = pix[..., 0].astype(F32)
    y = pix[..., 1].astype(F32)
    fx = np.minimum(np.maximum(F32(0.0), np.floor(x)), F32(width - 2))
    fy = np.minimum(np.maximum(F32(0.0), np.floor(y)), F32(height - 2))
    ax = np.minimum(np.maximum(F32(0.0), (x - fx).astype(F32)), F32(1.0))
    ay = np.minimum(np.maximum(F32(0.0), (y - fy).astype(F32)), F32(1.0))
    return fx.astype(np.int32), fy.astype(np.int32), ax, ay


def tap_linear_indices(x0, y0, bv_index, height, width):
    """(tl, tr, bl, br) linear texel indices b*H*W + y*W + x as int32 (tfa `gather`)."""
    base = (bv_index * (height * width)).astype(np.int64)
    tl = base + y0.astype(np.int64) * width + x0
    return np.stack([tl, tl + 1, tl + width, tl + width + 1], axis=-1).astype(np.int32)


def interpolate_bilinear_xy(grid, query):
    """grid (N,H,W,C) f32, query (N,Q,2) [x,y] -> (N,Q,C).  Border clamp, never zero-pad."""
    n, h, w, c = grid.shape
    x0, y0, ax, ay = bilinear_taps(query, h, w)
    flat = grid.reshape(n * h * w, c)
    bidx = np.arange(n, dtype=np.int64)[:, None]
    idx = tap_linear_indices(x0, y0, bidx, h, w)                 # (N,Q,4)
    tl, tr, bl, br = (flat[idx[..., i]] for i in range(4))
    ax = ax[..., None]
    ay = ay[..., None]
    top = (ax * (tr - tl).astype(F32)).astype(F32) + tl
    bot = (ax * (br - bl).astype(F32)).astype(F32) + bl
    return ((ay * (bot - top).astype(F32)).astype(F32) + top).astype(F32)


def get_projection_features_mv(normalized_images, features, pixel_locations):
    """Reference: nerf_utils.py:277-285.  images (B,V,H,W,3) already *2-1, features (B,V,H,W,256),
    pixel_locations (B,V,R,S,2) -> (B,V,R,S,259) with channel order [rgb | features]."""
    b, v, h, w, _ = normalized_images.shape
    grid = np.concatenate([normalized_images, features], axis=-1).reshape(b * v, h, w, -1)
    r, s = pixel_locations.shape[2:4]
    q = pixel_locations.reshape(b * v, r * s, 2)
    out = interpolate_bilinear_xy(grid.astype(F32), q)
    return out.reshape(b, v, r, s, -1)


# --------------------------------------------------------------------------------------
# a8: positional encoding
# --------------------------------------------------------------------------------------
def position_encoding(position, n_freq=N_FREQ, pos_encoding_freq=np.pi):
    """Reference: nerf_utils.py:108-126.  Q4: layout (d, k, {sin,cos}), no identity term.

    The fp32 product x * fl32(pi*2^k) is formed first, then sin/cos of that rounded value.
    position (...,D) -> (..., D*2*n_freq).
    """
    freq = (F32(pos_encoding_freq) * np.power(F32(2.0), np.arange(n_freq, dtype=F32))).astype(F32)
    arg = (position.astype(F32)[..., None] * freq).astype(F32)             # (...,D,n)
    enc = np.stack([np.sin(arg), np.cos(arg)], axis=-1).astype(F32)        # (...,D,n,2)
    return enc.reshape(*position.shape[:-1], -1)


# --------------------------------------------------------------------------------------
# a9, a10: MLP trunk and read-out
# --------------------------------------------------------------------------------------
def unflatten_net(flat):
    """Split the 247 300-float Keras-order buffer into named arrays (kernel[in,out], bias[out]).

    Order: W0[379,128] b0[128] | 6 x (W1[128,128] b1[128] W2[128,128] b2[128]) | Wr[128,4] br[4].
    The first 3 blocks are the per-view feature blocks, the last 3 the fusion blocks
    (layers.py:345-352).
    """
    flat = np.asarray(flat, dtype=F32)
    assert flat.size == NET_PARAMS, flat.size
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
    assert pos == NET_PARAMS
    return net


def _relu(x):
    return np.maximum(x, F32(0.0))


def bf16_round(a):
    """Round fp32 to the nearest bfloat16 (ties to even), returned as fp32 - what v_cvt_pk_bf16_f32 does.
    Used to restate the bf16 variant of the field pass (bf16 MFMA inputs, fp32 accumulate)."""
    u = np.ascontiguousarray(a, dtype=F32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(F32).reshape(np.shape(a))


def resnet_block(x, w1, b1, w2, b2, rnd=None):
    """Pre-activation block x + W2.relu(W1.relu(x)+b1)+b2.  Reference: layers.py:262-298.
    rnd: optional rounding of the matmul inputs (bf16 variant); accumulation, bias and residual stay fp32."""
    q = rnd if rnd is not None else (lambda a: a)
    r = q(_relu(x)) @ q(w1) + b1
    r = q(_relu(r)) @ q(w2) + b2
    return (x + r).astype(F32)


def mv_embedding(net, cam_xyz, cam_dir, feat, n_views, complete_output=False, emulate_bf16=False):
    """MVResNetMLPNeRFEmbedding.call.  Reference: layers.py:354-379.

    cam_xyz, cam_dir: (B*V,R,S,3); feat: (B*V,R,S,259) -> (B,R,S,128).
    emulate_bf16: restate the bf16 kernel variant - every Dense input (activations and kernels) rounded to
    bfloat16 except the PE(cam dir) rows of layer 0, which that kernel keeps in fp32 (per-ray seed).
    """
    rnd = bf16_round if emulate_bf16 else None
    if emulate_bf16:
        pe_xyz, pe_dir = position_encoding(cam_xyz), position_encoding(cam_dir)
        rest = np.concatenate([pe_xyz, feat.astype(F32)], axis=-1)
        w_rest = np.concatenate([net['W0'][:60], net['W0'][120:]], axis=0)
        x = (bf16_round(rest) @ bf16_round(w_rest) + (pe_dir @ net['W0'][60:120] + net['b0'])).astype(F32)
    else:
        x = np.concatenate([position_encoding(cam_xyz), position_encoding(cam_dir), feat.astype(F32)], axis=-1)
        x = (x @ net['W0'] + net['b0']).astype(F32)
    outs = [x]
    for blk in net['blocks'][:N_BLOCKS // 2]:
        outs.append(resnet_block(outs[-1], *blk, rnd=rnd))
    bv = outs[-1].shape[0]
    pre = outs[-1].reshape(bv // n_views, n_views, *outs[-1].shape[1:])
    fusion = pre[:, 0].copy()
    for i in range(1, n_views):                  # reduce_mean over views: sequential sum / V
        fusion = (fusion + pre[:, i]).astype(F32)
    fusion = (fusion / F32(n_views)).astype(F32)
    outs.append(fusion)
    for blk in net['blocks'][N_BLOCKS // 2:]:
        outs.append(resnet_block(outs[-1], *blk, rnd=rnd))
    return outs if complete_output else outs[-1]


def sigmoid(x):
    x = x.astype(F32)
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def softplus(x):
    return np.logaddexp(F32(0.0), x.astype(F32)).astype(F32)


def render_readout(net, emb, emulate_bf16=False):
    """RenderReadout.call.  Reference: layers.py:392-397.  -> (rgb (...,3), sigma (...))."""
    q = bf16_round if emulate_bf16 else (lambda a: a)
    o = (q(_relu(emb)) @ q(net['Wr']) + net['br']).astype(F32)
    return sigmoid(o[..., :3]), softplus(o[..., 3])


# --------------------------------------------------------------------------------------
# a11, a12: alpha compositing
# --------------------------------------------------------------------------------------
def sigma_to_alpha(sigma, dists):
    """Reference: nerf_utils.py:129-140."""
    return (F32(1.0) - np.exp(-dists * _relu(sigma))).astype(F32)


def volumetric_render(zs, density, chromacity):
    """Reference: model_v0.py:89-100.  Q6: last delta duplicated, +1e-10 inside the cumprod.

    zs, density (B,R,S); chromacity (B,R,S,3) -> rgb (B,R,3), depth (B,R), weights (B,R,S).
    """
    zs = zs.astype(F32)
    dists = zs[..., 1:] - zs[..., :-1]
    dists = np.concatenate([dists, dists[..., -1:]], axis=-1)
    alpha = sigma_to_alpha(density.astype(F32), dists)
    t = ((F32(1.0) - alpha) + F32(1e-10)).astype(F32)
    trans = np.cumprod(t, axis=-1, dtype=F32)
    trans = np.concatenate([np.ones_like(trans[..., :1]), trans[..., :-1]], axis=-1)   # exclusive
    weights = (alpha * trans).astype(F32)
    rgb = np.zeros(zs.shape[:-1] + (3,), dtype=F32)
    depth = np.zeros(zs.shape[:-1], dtype=F32)
    for i in range(zs.shape[-1]):                # sequential sums, fp32
        rgb = (rgb + (weights[..., i, None] * chromacity[..., i, :]).astype(F32)).astype(F32)
        depth = (depth + (weights[..., i] * zs[..., i]).astype(F32)).astype(F32)
    return rgb, depth, weights


# --------------------------------------------------------------------------------------
# a13, a14: importance sampling and sort-merge
# --------------------------------------------------------------------------------------
def sample_pdf(bins, weights, u, q7_mode=Q7_ZERO, return_indices=False):
    """Inverse-CDF sampling.  Reference: nerf_utils.py:143-176 (u explicit, F11).

    bins (B,R,Nb), weights (B,R,Nb-1), u (B,R,N).  `above` = #{j : u >= cdf_j} over the Nb cdf
    entries (the tf.scan, :156-160); `below` = clip(above-1, 0, Nb-1); Q7: `above` is NOT clipped
    in the reference, so above == Nb indexes past the end; q7_mode picks TF-GPU (gather -> 0) or
    clamp behaviour.
    """
    bins = bins.astype(F32)
    stable = (weights.astype(F32) + F32(1e-5)).astype(F32)
    csum = np.cumsum(stable, axis=-1, dtype=F32)                 # sequential
    w_sum = csum[..., -1:]
    w_sum = np.where(np.abs(w_sum) == 0, np.ones_like(w_sum), w_sum)
    pdf = (stable / w_sum).astype(F32)
    cdf = np.cumsum(pdf, axis=-1, dtype=F32)                     # sequential
    cdf = np.concatenate([np.zeros_like(cdf[..., :1]), cdf], axis=-1)   # (B,R,Nb)
    nb = bins.shape[-1]
    assert cdf.shape[-1] == nb
    u = u.astype(F32)
    above = np.zeros(u.shape, dtype=np.int32)
    for j in range(nb):
        above += (u >= cdf[..., j:j + 1]).astype(np.int32)
    below = np.clip(above - 1, 0, nb - 1)

    def gather(tab, idx):
        if q7_mode == Q7_CLAMP:
            return np.take_along_axis(tab, np.minimum(idx, nb - 1).astype(np.int64), axis=-1)
        safe = np.minimum(idx, nb - 1).astype(np.int64)
        val = np.take_along_axis(tab, safe, axis=-1)
        return np.where(idx >= nb, F32(0.0), val).astype(F32)

    cdf_a, cdf_b = gather(cdf, above), gather(cdf, below)
    bins_a, bins_b = gather(bins, above), gather(bins, below)
    den = (cdf_a - cdf_b).astype(F32)
    den = np.where(den < F32(1e-5), np.ones_like(den), den)
    t = ((u - cdf_b).astype(F32) / den).astype(F32)
    samples = (bins_b + (t * (bins_a - bins_b).astype(F32)).astype(F32)).astype(F32)
    if return_indices:
        return samples, above, below
    return samples


def hierarchical_depths(z_coarse, weights, u_fine, q7_mode=Q7_ZERO, return_indices=False):
    """model_v0.py:150-156: z_mid, probs = w[1:-1], sample_pdf, concat, ascending sort."""
    z_mid = (F32(0.5) * (z_coarse[..., 1:] + z_coarse[..., :-1]).astype(F32)).astype(F32)
    probs = weights[..., 1:-1]
    res = sample_pdf(z_mid, probs, u_fine, q7_mode, return_indices=True)
    all_zs = np.sort(np.concatenate([z_coarse, res[0]], axis=-1), axis=-1)
    if return_indices:
        return all_zs, res[0], res[1], res[2]
    return all_zs


# --------------------------------------------------------------------------------------
# a15: the forward pass
# --------------------------------------------------------------------------------------
def field_eval(net, rays_o, rays_d, zs, images, features, k4, einv, ray_chunk=256, return_taps=False, emulate_bf16=False):
    """One pass (coarse or fine): project -> gather -> PE -> trunk -> read-out.

    Reference: model_v0.py:120-144 (coarse) / :157-180 (fine).  Returns rgb (B,R,S,3), sigma
    (B,R,S) and, on request, the int32 tap indices (B,V,R,S,4).
    """
    b, v, h, w, _ = images.shape
    r, s = zs.shape[1:3]
    norm_images = (images.astype(F32) * F32(2.0) - F32(1.0)).astype(F32)
    rgb = np.empty((b, r, s, 3), dtype=F32)
    sigma = np.empty((b, r, s), dtype=F32)
    taps = np.empty((b, v, r, s, 4), dtype=np.int32) if return_taps else None
    for r0 in range(0, r, ray_chunk):
        sl = slice(r0, min(r, r0 + ray_chunk))
        o, d, z = rays_o[:, sl], rays_d[:, sl], zs[:, sl]
        world = points_on_rays(o, d, z)
        pix, cam = compute_pixel_in_image_mv(world, k4, einv)
        feat = get_projection_features_mv(norm_images, features, pix)
        cdir = world_to_camera_direction_vector_mv(d, einv)                     # (B,V,Rc,3)
        cdir = np.broadcast_to(cdir[:, :, :, None, :], cam.shape[:-1] + (3,))
        rc = z.shape[1]
        emb = mv_embedding(net,
                           cam[..., :3].reshape(b * v, rc, s, 3),
                           cdir.reshape(b * v, rc, s, 3),
                           feat.reshape(b * v, rc, s, -1), v, emulate_bf16=emulate_bf16)
        c, sg = render_readout(net, emb, emulate_bf16)
        rgb[:, sl], sigma[:, sl] = c, sg
        if return_taps:
            x0, y0, _, _ = bilinear_taps(pix, h, w)
            bv = (np.arange(b)[:, None] * v + np.arange(v)[None, :])[:, :, None, None]
            taps[:, :, sl] = tap_linear_indices(x0, y0, bv, h, w)
    if return_taps:
        return rgb, sigma, taps
    return rgb, sigma


def render_call(coarse_net, fine_net, rays_o, rays_d, images, k4, einv, features, near, far,
                n_samples, u_coarse, u_fine, q7_mode=Q7_ZERO, ray_chunk=256, return_aux=False):
    """MVVNeRFRenderer._call.  Reference: model_v0.py:113-184.

    Returns (rgb, depth, fine_rgb, fine_depth); with return_aux also a dict of intermediates.
    """
    _, z = sample_along_ray(rays_o, rays_d, near, far, n_samples, u_coarse)
    c_rgb, c_sigma = field_eval(coarse_net, rays_o, rays_d, z, images, features, k4, einv, ray_chunk)
    rgb, depth, weights = volumetric_render(z, c_sigma, c_rgb)
    all_zs, z_fine, above, below = hierarchical_depths(z, weights, u_fine, q7_mode, return_indices=True)
    f_rgb, f_sigma = field_eval(fine_net, rays_o, rays_d, all_zs, images, features, k4, einv, ray_chunk)
    fine_rgb, fine_depth, fine_weights = volumetric_render(all_zs, f_sigma, f_rgb)
    if return_aux:
        aux = dict(z=z, coarse_rgb=c_rgb, coarse_sigma=c_sigma, weights=weights, z_fine=z_fine,
                   above=above, below=below, all_zs=all_zs, fine_rgbs=f_rgb, fine_sigma=f_sigma,
                   fine_weights=fine_weights)
        return rgb, depth, fine_rgb, fine_depth, aux
    return rgb, depth, fine_rgb, fine_depth


# --------------------------------------------------------------------------------------
# a17: full-image driver epilogue
# --------------------------------------------------------------------------------------
def finish_view(all_rgbs, all_depths, image_shape):
    """model_v0.py:275-281: rgb*255 clip -> uint8 ; depth min-max normalised -> uint8."""
    rgb = np.reshape(all_rgbs, (*image_shape, 3)) * 255
    rgb = np.clip(rgb, 0, 255).astype(np.uint8)
    dep = np.reshape(all_depths, (*image_shape, 1))
    norm = (dep - np.min(dep)) / (np.max(dep) - np.min(dep))
    return rgb, (norm * 255).astype(np.uint8)
