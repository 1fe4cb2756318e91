"""`LanguageNeRF` on the HIP trunk (reference: src/lib/lmvnerf/model_v4.py).

The reference's language/grasp model re-uses the NeRF trunk (`fine_embedding`, frozen) as a feature field on points
derived from grasp poses, puts a small `GraspReadout` MLP on its four fused activations and trains ONLY that read-out
with a loss on the prediction and a loss on d prediction / d pose (nested GradientTape, model_v4.py:277-322).  Here:

* the trunk is `TrunkField`: forward = `mvnerf_field_eval_stash` on the query points (activations read back from the
  stash), backward = `mvnerf_query_vjp`, and the backward of THAT backward w.r.t. its cotangent = `mvnerf_query_jvp`
  (the trunk is linear in nothing but the cotangent, so the double-backward the second tape needs is a JVP);
* `GraspReadout`, the pose algebra and the losses are ordinary torch (small, and torch's autograd supplies their
  double-backward); the CLIP / ViT / conv encoders are outside the hot path: `combined_features` is an input.

Names and argument meaning follow the reference (`_call`, `compute_matrices`, `set_pose`, `train_step`, `infer`).
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from . import ops

N_FUSED = 4


# ---- the trunk as an autograd function ----------------------------------------------------------------------
class TrunkState:
    """Everything the trunk needs besides the query points: source views, cameras, packed weights."""

    def __init__(self, images, features, intrinsics, extrinsics_inv, net_keras):
        self.geo = (images.contiguous(), features.contiguous(), intrinsics.contiguous(), extrinsics_inv.contiguous())
        self.packed = ops.pack_net(net_keras)
        self.packed_split = ops.pack_net_split(net_keras)          # value pass on the split-bf16 kernel (fp32-grade)
        self.bwd_streams = ops.pack_bwd_streams(net_keras)
        self.n_views = images.shape[1]


def _pad32(t, n_pad):
    return t if n_pad == 0 else torch.cat([t, t[:, -1:].expand(-1, n_pad, -1)], 1).contiguous()


class _TrunkVJP(torch.autograd.Function):
    """(g_acts) -> (d_points, d_dirs) = J^T g_acts; differentiable w.r.t. g_acts: its backward is J c."""

    @staticmethod
    def forward(ctx, g_acts, points, dirs, state, stash):
        ctx.state, ctx.points, ctx.dirs = state, points, dirs
        return ops.query_vjp(points, dirs, *state.geo, state.bwd_streams, stash, g_acts.contiguous())

    @staticmethod
    def backward(ctx, c_points, c_dirs):
        st = ctx.state
        zero = torch.zeros_like(ctx.points)
        c_points = zero if c_points is None else c_points.contiguous()
        c_dirs = zero if c_dirs is None else c_dirs.contiguous()
        t_acts = ops.query_jvp(ctx.points, ctx.dirs, c_points, c_dirs, *st.geo, st.packed)
        return t_acts, None, None, None, None          # second derivatives w.r.t. the poses are not propagated


class TrunkField(torch.autograd.Function):
    """points, dirs (B,N,3) -> acts (4,B,N,128) = (view mean, u1, u2, u3) of the frozen trunk (layers.py:364-377)."""

    @staticmethod
    def forward(ctx, points, dirs, state):
        b, n, _ = points.shape
        pad = (-n) % 32 if state.n_views > 1 else 0     # the multi-view training kernels want whole 32-point tiles per scene
        p, d = _pad32(points.detach().contiguous(), pad), _pad32(dirs.detach().contiguous(), pad)
        stash = ops.query_stash(p, d, *state.geo, state.packed, packed_split=state.packed_split)
        acts = ops.stash_fused_acts(stash, b, state.n_views, n + pad)       # tile layout -> (4, b, n + pad, 128) rows
        ctx.state, ctx.stash, ctx.pad, ctx.n = state, stash, pad, n
        ctx.save_for_backward(p, d)
        return acts[:, :, :n].contiguous() if pad else acts

    @staticmethod
    def backward(ctx, g_acts):
        p, d = ctx.saved_tensors
        if ctx.pad:
            g_acts = torch.cat([g_acts, g_acts.new_zeros(4, g_acts.shape[1], ctx.pad, 128)], 2)
        d_points, d_dirs = _TrunkVJP.apply(g_acts, p, d, ctx.state, ctx.stash)
        return d_points[:, :ctx.n], d_dirs[:, :ctx.n], None


# ---- GraspReadout (delta_ngf/layers.py:8-42) ----------------------------------------------------------------
def _skinny(m, n, k):
    """Products mvnerf_gemm_nt takes over from the library GEMM: every shape it accepts.  On the GraspReadout's shapes the library picks
    poor kernels - 4 to 36 workgroups for the weight gradients (64 x 128 outputs over K = 64 512 rows: 245 us) and for the 2688-input block
    (1536 x 128 over K = 2688), 16 x 16 tiles for 64 512 x 128 x 64 (590 us) - see profiles/r02_language_step_trace.md."""
    return ops.gemm_nt_ok(m, n, k) and k >= 64


def _mm_nt_raw(a, bt):
    if a.is_cuda and a.dtype == torch.float32 and bt.dtype == torch.float32 and _skinny(a.shape[0], bt.shape[0], a.shape[1]):
        return ops.gemm_nt(a.contiguous(), bt.contiguous())
    return a @ bt.t()


def _tn_skinny(m, n, k):
    return ops.gemm_tn_ok(m, n, k) and m >= 64


class _MatmulNT(torch.autograd.Function):
    """a (M,K) @ bt (N,K)^T with the skinny products on mvnerf_gemm_nt.  The backward is built from this function and _MatmulTN again, so
    the second derivatives LanguageNeRF.train_step takes (model_v4.py:290-322) choose their kernels the same way (a large forward
    product has a skinny weight gradient)."""

    @staticmethod
    def forward(ctx, a, bt):
        ctx.save_for_backward(a, bt)
        return _mm_nt_raw(a, bt)

    @staticmethod
    def backward(ctx, g):
        a, bt = ctx.saved_tensors
        ga = _MatmulNT.apply(g, bt.t()) if ctx.needs_input_grad[0] else None               # g (M,N) @ bt (N,K): bt^T is a small copy
        gbt = _MatmulTN.apply(g, a) if ctx.needs_input_grad[1] else None                   # g^T (N,M) @ a (M,K), operands as they lie
        return ga, gbt


class _MatmulTN(torch.autograd.Function):
    """g (M,N)^T @ a (M,K) -> (N,K): a weight gradient, without transposed copies of the M-row operands (mvnerf_gemm_tn)."""

    @staticmethod
    def forward(ctx, g, a):
        ctx.save_for_backward(g, a)
        if g.is_cuda and g.dtype == torch.float32 and a.dtype == torch.float32 and _tn_skinny(g.shape[0], g.shape[1], a.shape[1]):
            return ops.gemm_tn(g.contiguous(), a.contiguous())
        return g.t() @ a

    @staticmethod
    def backward(ctx, c):
        g, a = ctx.saved_tensors
        gg = _MatmulNT.apply(a, c) if ctx.needs_input_grad[0] else None                    # a (M,K) @ c (N,K)^T -> (M,N)
        ga = _MatmulNT.apply(g, c.t()) if ctx.needs_input_grad[1] else None                # g (M,N) @ c (N,K) -> (M,K): c^T is a small copy
        return gg, ga


class _LinearNT(torch.autograd.Function):
    """x (M,K) @ w (N,K)^T + b with the bias added in the GEMM's epilogue (mvnerf_gemm_nt_bias: one launch instead of two).  The backward is
    _MatmulNT / _MatmulTN and a column sum, so it can be differentiated again like the unfused form."""

    @staticmethod
    def forward(ctx, a, w, b):
        ctx.save_for_backward(a, w)
        return ops.gemm_nt(a.contiguous(), w.contiguous(), bias=b.contiguous())

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        ga = _MatmulNT.apply(g, w.t()) if ctx.needs_input_grad[0] else None
        if (ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and not torch.is_grad_enabled() and g.is_cuda and g.dtype == torch.float32
                and _tn_skinny(g.shape[0], g.shape[1], a.shape[1])):
            # the last backward of a step (nothing differentiates it again): weight and bias gradient from one pass over g
            gw, gb = (t[0] for t in ops.gemm_tn_batched(g.contiguous()[None], a.contiguous()[None], colsum_of=1))
            return ga, gw, gb
        gw = _MatmulTN.apply(g, a) if ctx.needs_input_grad[1] else None
        gb = g.sum(0) if ctx.needs_input_grad[2] else None
        return ga, gw, gb


def _mm_nt(a, bt):
    """a @ bt^T through _MatmulNT on the GPU (fp32), plain torch otherwise."""
    if a.is_cuda and a.dtype == torch.float32 and bt.dtype == torch.float32:
        return _MatmulNT.apply(a, bt)
    return a @ bt.t()


def _wide_linear(lin, x):
    """nn.Linear through _mm_nt (value, input / weight gradients and their derivatives)."""
    if not x.is_cuda:
        return lin(x)
    x2 = x.reshape(-1, lin.in_features)
    if lin.bias is not None and x2.dtype == torch.float32 and _skinny(x2.shape[0], lin.out_features, lin.in_features):
        y = _LinearNT.apply(x2, lin.weight, lin.bias)
    else:
        y = _mm_nt(x2, lin.weight)
        if lin.bias is not None:
            y = y + lin.bias
    return y.reshape(*x.shape[:-1], lin.out_features)


# ---- the per-point part of the read-out as fused HIP passes (csrc/grasp_head.hip) ----------------------------------------------------
def _gtn(g, a):
    """g (M,N)^T @ a (M,K) -> (N,K) on mvnerf_gemm_tn where it takes the shape (M = the number of query points)."""
    if _tn_skinny(g.shape[0], g.shape[1], a.shape[1]):
        return ops.gemm_tn(g.contiguous(), a.contiguous())
    return g.t() @ a


def _batched_ok(g_u):
    return g_u.is_cuda and g_u.dtype == torch.float32 and g_u.shape[0] % 8 == 0 and g_u.is_contiguous()


def _col_blocks(x):
    """(M, 4 * 64) -> the four (M, 64) column blocks as a (4, M, 64) view (strides 64, 256, 1): what mvnerf_gemm_tn_batched reads in place."""
    return x.view(x.shape[0], 4, 64).permute(1, 0, 2)


class _HeadVJP(torch.autograd.Function):
    """The vector-Jacobian product of the fused head as a differentiable function of its cotangent and of the weights: forward =
    mvnerf_grasp_head_vjp (+ the weight gradients as skinny GEMMs), backward = mvnerf_grasp_head_vjp_bwd - what the nested tape of
    LanguageNeRF.train_step needs (model_v4.py:290-322).  The activations' own second derivative is not formed (they depend on the pose
    only, which is not trained); cotangents on the weight-gradient outputs are refused."""

    @staticmethod
    def forward(ctx, g_y, acts, c, y, w4, b4, wc, bc, packed):
        g_y = g_y.contiguous()
        g_v, q, g_u, g_acts = ops.grasp_head_vjp(g_y, c, y, packed)
        if _batched_ok(g_u):
            # the five weight gradients and their bias gradients: two launches (+ fixed-order reduces), the column blocks of g_u read where they lie
            d_w4, d_b4 = ops.gemm_tn_batched(_col_blocks(g_u), acts, colsum_of=1)
            d_wc, d_bc = (t[0] for t in ops.gemm_tn_batched(g_v[None], c[None], colsum_of=1))
        else:
            d_wc, d_bc = _gtn(g_v, c), g_v.sum(0)
            d_w4 = torch.stack([_gtn(g_u[:, 64 * k:64 * k + 64], acts[k]) for k in range(4)])
            d_b4 = g_u.sum(0).reshape(4, 64)
        ctx.save_for_backward(g_y, acts, c, y, q, g_v, g_u, packed)
        ctx.set_materialize_grads(False)
        return g_acts, d_w4, d_b4, d_wc, d_bc

    @staticmethod
    def backward(ctx, t_acts, *weight_cotangents):
        if any(t is not None for t in weight_cotangents):
            raise NotImplementedError('_HeadVJP: derivatives of the weight gradients are not built (LanguageNeRF.train_step does not take them)')
        if t_acts is None:
            return (None,) * 9
        g_y, acts, c, y, q, g_v, g_u, packed = ctx.saved_tensors
        out_gy, r, m, p_ = ops.grasp_head_vjp_bwd(t_acts.contiguous(), g_y, c, y, q, packed)
        if _batched_ok(g_u):
            d_w4, d_b4 = ops.gemm_tn_batched(_col_blocks(g_u), t_acts.contiguous(), g2=_col_blocks(p_), a2=acts, colsum_of=2)
            d_wc, d_bc = (t[0] for t in ops.gemm_tn_batched(g_v[None], r[None], g2=m[None], a2=c[None], colsum_of=2))
        else:
            d_w4 = torch.stack([_gtn(g_u[:, 64 * k:64 * k + 64], t_acts[k]) + _gtn(p_[:, 64 * k:64 * k + 64], acts[k]) for k in range(4)])
            d_b4 = p_.sum(0).reshape(4, 64)
            d_wc = _gtn(g_v, r) + _gtn(m, c)
            d_bc = m.sum(0)
        return out_gy, None, None, None, d_w4, d_b4, d_wc, d_bc, None


class _HeadFn(torch.autograd.Function):
    """acts (4,N,128) -> (N,64): four Dense(128 -> 64) + elu, concatenation, Dense(256 -> 64) + elu in ONE launch (mvnerf_grasp_head_fwd);
    its backward is _HeadVJP, itself differentiable."""

    @staticmethod
    def forward(ctx, acts, w4, b4, wc, bc):
        acts, w4, b4, wc, bc = (t.contiguous() for t in (acts, w4, b4, wc, bc))
        packed = ops.grasp_head_pack(w4, wc)
        c, y = ops.grasp_head_fwd(acts, packed, b4, bc)
        ctx.save_for_backward(acts, c, y, w4, b4, wc, bc, packed)
        return y

    @staticmethod
    def backward(ctx, g_y):
        acts, c, y, w4, b4, wc, bc, packed = ctx.saved_tensors
        return _HeadVJP.apply(g_y, acts, c, y, w4, b4, wc, bc, packed)


def _he_normal_(w):
    fan_in = w.shape[1]
    nn.init.trunc_normal_(w, std=math.sqrt(2.0 / fan_in) / 0.87962566103423978, a=-2 * math.sqrt(2.0 / fan_in) / 0.87962566103423978,
                          b=2 * math.sqrt(2.0 / fan_in) / 0.87962566103423978)


class ResNetMLPBlock(nn.Module):
    """layers.py:262-298 with activation='elu' (the pre-activation form: act -> Dense -> act -> Dense, + shortcut)."""

    def __init__(self, in_size, hidden_size, output_size, transform_shortcut=False):
        super().__init__()
        self.layer_0 = nn.Linear(in_size, hidden_size)
        self.layer_1 = nn.Linear(hidden_size, output_size)
        self.shortcut = nn.Linear(in_size, output_size, bias=False) if transform_shortcut else None
        for lin in (self.layer_0, self.layer_1) + ((self.shortcut,) if self.shortcut is not None else ()):
            _he_normal_(lin.weight)
            if lin.bias is not None:
                nn.init.zeros_(lin.bias)

    def forward(self, x):
        r = _wide_linear(self.layer_1, nn.functional.elu(_wide_linear(self.layer_0, nn.functional.elu(x))))
        return (_wide_linear(self.shortcut, x) if self.shortcut is not None else x) + r


class GraspReadout(nn.Module):
    def __init__(self, n_offsets, use_bias=True):
        super().__init__()
        self.activation_downscale = nn.ModuleList([nn.Linear(128, 64) for _ in range(N_FUSED)])
        self.combined_activation_downscale = nn.Linear(4 * 64, 64)
        self.block_0 = ResNetMLPBlock(n_offsets * 64, 128, 64, transform_shortcut=True)
        self.block_1 = ResNetMLPBlock(64, 64, 64)
        self.output_layer = nn.Linear(64, 1, bias=use_bias)
        for lin in self.activation_downscale:
            _he_normal_(lin.weight)
            nn.init.zeros_(lin.bias)
        nn.init.xavier_uniform_(self.combined_activation_downscale.weight)      # Keras default glorot_uniform
        nn.init.zeros_(self.combined_activation_downscale.bias)
        _he_normal_(self.output_layer.weight)
        if self.output_layer.bias is not None:
            nn.init.zeros_(self.output_layer.bias)

    fused_head = True      # the per-point layers as fused HIP passes (csrc/grasp_head.hip); False: Linear by Linear (torch + gemm_ops)

    def forward(self, acts):
        """acts: 4 x (B, np, n5, 128), or the same stacked as one (4, B, np, n5, 128) tensor -> (B, np)."""
        stacked = acts if isinstance(acts, torch.Tensor) else None
        first = acts[0]
        b, n_p, n5 = first.shape[0], first.shape[1], first.shape[2]
        if self.fused_head and first.is_cuda and first.dtype == torch.float32:
            a = (stacked if stacked is not None else torch.stack(list(acts))).reshape(N_FUSED, -1, 128)
            w4 = torch.stack([lin.weight for lin in self.activation_downscale])
            b4 = torch.stack([lin.bias for lin in self.activation_downscale])
            x = _HeadFn.apply(a, w4, b4, self.combined_activation_downscale.weight, self.combined_activation_downscale.bias)
            x = x.reshape(b, n_p, n5 * 64)                                       # 'b np n5 d -> b np (n5 d)'
        else:
            acts = list(acts.unbind(0)) if stacked is not None else acts
            ds = [nn.functional.elu(_wide_linear(lin, a)) for lin, a in zip(self.activation_downscale, acts)]
            x = nn.functional.elu(_wide_linear(self.combined_activation_downscale, torch.cat(ds, -1)))
            x = x.reshape(x.shape[0], x.shape[1], -1)                            # 'b np n5 d -> b np (n5 d)'
        x = self.block_1(self.block_0(x))
        return self.output_layer(torch.relu(x))[..., 0]


# ---- pose algebra (model_v4.py:67-101, 192-206) -------------------------------------------------------------
def _rot_y(angle):
    c, s = math.cos(angle), math.sin(angle)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def grasp_offsets(n_5d_poses=7):
    """`transforms_to_check`: 6 gripper-part bases x n_5d_poses steps along the local z axis -> (6*n_5d_poses, 4, 4).
    (Affine(rotation=[0, +-pi/2, 0]) is scipy's extrinsic 'xyz' Euler = a rotation about y.)"""
    bx, by, bz = 0.02, 0.015, 0.0125
    step = (bx - 0.005) / ((n_5d_poses - 1) / 2)

    def affine(t, ry=0.0):
        m = np.eye(4)
        m[:3, :3] = _rot_y(ry)
        m[:3, 3] = t
        return m
    bases = [affine([0, by, 0]), affine([0, -by, 0]), affine([-bx, by, bz], math.pi / 2), affine([bx, by, bz], -math.pi / 2),
             affine([-bx, -by, bz], math.pi / 2), affine([bx, -by, bz], -math.pi / 2)]
    c = int((n_5d_poses - 1) / 2)
    steps = [affine([0.0, 0.0, i * step]) for i in range(-c, c + 1)]
    return np.array([b @ t for b in bases for t in steps], dtype=np.float32)


def rotation_from_quaternion(q):
    """tensorflow_graphics rotation_matrix_3d.from_quaternion (x, y, z, w; the input is used as given, not normalised)."""
    x, y, z, w = q.unbind(-1)
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    m = torch.stack([1 - (tyy + tzz), txy - twz, txz + twy,
                     txy + twz, 1 - (txx + tzz), tyz - twx,
                     txz - twy, tyz + twx, 1 - (txx + tyy)], -1)
    return m.reshape(q.shape[:-1] + (3, 3))


def rotation_from_6d(r6):
    """model_v4.py:196-204: both halves normalised (not orthogonalised), third column their cross product."""
    r1 = r6[..., :3] / r6[..., :3].norm(dim=-1, keepdim=True)
    r2 = r6[..., 3:] / r6[..., 3:].norm(dim=-1, keepdim=True)
    r3 = torch.linalg.cross(r1, r2)
    return torch.stack([r1, r2, r3], -1)


def t_m_to_h_matrix(translations, rot):
    top = torch.cat([rot, translations[..., None]], -1)
    # (0,0,0,1) built on the device: a host constant would be an H2D copy, which a graph capture refuses
    last = torch.cat([torch.zeros_like(translations), torch.ones_like(translations[..., :1])], -1)[..., None, :]
    return torch.cat([top, last], -2)


def cosine_similarity_loss(y_true, y_pred):
    """tf.keras.losses.CosineSimilarity(axis=-1): -mean(sum(l2n(y_true) * l2n(y_pred)))."""
    def l2n(x):
        return x * torch.rsqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-12))
    return -(l2n(y_true) * l2n(y_pred)).sum(-1).mean()


def kl_divergence(y_true, y_pred):
    """tf.keras.losses.KLDivergence(reduction=NONE): per batch element, inputs clipped to [1e-7, 1]."""
    y_true = torch.clamp(y_true, 1e-7, 1.0)
    y_pred = torch.clamp(y_pred, 1e-7, 1.0)
    return (y_true * torch.log(y_true / y_pred)).sum(-1)


def _quaternion_pose_map(offsets):
    """rotation_from_quaternion is a quadratic form in q = (x, y, z, w): R_ik = delta_ik + sum_ab q_a q_b A[ab, ik].  Folded with the gripper
    offsets, the query points and directions of ALL offsets are one product of the 16 pair products q_a q_b with a constant matrix:
        points[o, i] = off_t[o, i] + t_i + sum_ab q_a q_b (A[ab, i, :] . off_t[o]),     dirs[o, i] = off_z[o, i] + sum_ab q_a q_b (A[ab, i, :] . off_z[o])
    -> (16, 2 * n5 * 3) float32: columns [points (o, i) | dirs (o, i)].  Same polynomial as the reference's expression tree
    (tensorflow_graphics from_quaternion, then transforms @ offsets: model_v4.py:67-101, 222-226), different summation order."""
    a = np.zeros((4, 4, 3, 3))
    x, y, z, w = 0, 1, 2, 3
    def add(i, k, terms):
        for (p, q, c) in terms:
            a[p, q, i, k] += c
    add(0, 0, [(y, y, -2), (z, z, -2)]); add(0, 1, [(x, y, 2), (z, w, -2)]); add(0, 2, [(x, z, 2), (y, w, 2)])
    add(1, 0, [(x, y, 2), (z, w, 2)]); add(1, 1, [(x, x, -2), (z, z, -2)]); add(1, 2, [(y, z, 2), (x, w, -2)])
    add(2, 0, [(x, z, 2), (y, w, -2)]); add(2, 1, [(y, z, 2), (x, w, 2)]); add(2, 2, [(x, x, -2), (y, y, -2)])
    off_t, off_z = offsets[:, :3, 3].astype(np.float64), offsets[:, :3, 2].astype(np.float64)          # (n5, 3)
    pts = np.einsum('pqik,ok->pqoi', a, off_t).reshape(16, -1)
    drs = np.einsum('pqik,ok->pqoi', a, off_z).reshape(16, -1)
    return np.concatenate([pts, drs], 1).astype(np.float32)


class LanguageNeRF(nn.Module):
    """model_v4.py:37-330 without the encoders: `trunk_net` is the frozen fine_embedding (+ unused read-out) in the flat
    Keras order of MVVNeRFRenderer.fine_net; `combined_features` is passed in."""

    def __init__(self, trunk_net, n_points_train=5, n_views=1, n_5d_poses=7, batch_size=1, rotation_representation='quaternion',
                 softmax_before_loss=False, device='cuda:0'):
        super().__init__()
        if rotation_representation not in ('quaternion', '6d'):
            raise ValueError('Unknown rotation representation: ' + rotation_representation)
        self.device_ = torch.device(device)
        self.n_views, self.n_points_train, self.batch_size = n_views, n_points_train, batch_size
        self.rotation_representation = rotation_representation
        self.softmax_before_loss = softmax_before_loss
        self.register_buffer('trunk_net', torch.as_tensor(trunk_net, dtype=torch.float32).reshape(-1).clone())
        self.register_buffer('transforms_to_check', torch.from_numpy(grasp_offsets(n_5d_poses)))      # (n5,4,4)
        self.n_transforms_to_check = self.transforms_to_check.shape[0]
        self.register_buffer('_pose_map', torch.from_numpy(_quaternion_pose_map(self.transforms_to_check.numpy())))   # (16, 2 n5 3)
        self.grasp_readout = GraspReadout(self.n_transforms_to_check, use_bias=True)
        self.translations = nn.Parameter(torch.zeros(batch_size, n_points_train, 3))
        rot_dim = 4 if rotation_representation == 'quaternion' else 6
        self.rotations = nn.Parameter(torch.zeros(batch_size, n_points_train, rot_dim))
        self.pose_variables = [self.translations, self.rotations]
        self.loss = kl_divergence if softmax_before_loss else None
        self.optimizer = None
        self._graph_mode, self._graph, self._g_static, self._g_out, self._g_calls = False, None, None, None, 0
        self.to(self.device_)

    # -- reference API --
    def compile(self, optimizer=None, loss=None, learning_rate=1e-4, graph=False):
        """graph=True: `train_step` is captured ONCE as a HIP graph (torch.cuda.CUDAGraph) and replayed: the step is ~740 launches of
        3-500 us that the host cannot issue as fast as the GPU runs them - 6.6 ms per replay against 8-10.7 ms eager at the cfg3 shape
        (profiles/r02_language_step_graph.md).  The graph fixes shapes and addresses:
        inputs are staged into the buffers `graph_inputs()` returns (a producer that writes `combined_features` there in place skips the
        copy), the optimizer is this method's own Adam (capturable).  The first two steps run eagerly (they load every kernel and size
        the allocator pools), the third is captured."""
        if graph and optimizer is not None:
            raise ValueError('graph=True builds its own capturable Adam; pass learning_rate instead of an optimizer')
        self.optimizer = optimizer or torch.optim.Adam(self.grasp_readout.parameters(), lr=learning_rate, eps=1e-7, capturable=bool(graph))
        if loss is not None:
            self.loss = loss
        self._graph_mode, self._graph, self._g_static, self._g_out, self._g_calls = bool(graph), None, None, None, 0

    def set_pose(self, translations, rotations):
        with torch.no_grad():
            self.translations.copy_(torch.as_tensor(translations, dtype=torch.float32))
            self.rotations.copy_(torch.as_tensor(rotations, dtype=torch.float32))

    def compute_matrices(self):
        rot = (rotation_from_quaternion(self.rotations) if self.rotation_representation == 'quaternion'
               else rotation_from_6d(self.rotations))
        return t_m_to_h_matrix(self.translations, rot)

    def trunk_state(self, inputs, batched_features):
        """inputs[4:7] = src_images (B,V,H,W,3), src_intrinsics (B,V,4,4), src_extrinsics_inv (B,V,4,4) (model_v4.py:209-213)."""
        dev = self.device_
        f32 = lambda t: torch.as_tensor(t, dtype=torch.float32).to(dev)
        return TrunkState(f32(inputs[4]), f32(batched_features), f32(inputs[5]), f32(inputs[6]), self.trunk_net)

    def _query_points(self, transforms):
        """poses = transforms @ offsets (model_v4.py:222-226) reduced to what is used of them: points = the poses' translations, dirs =
        their z axes.  With affine offsets (last row 0 0 0 1) that is R t_o + t and R z_o - two (B np 3, 3) x (3, n5) products instead of
        B np n5 batched 4 x 4 products, for which the library GEMM takes 590 us (16 x 16 tiles over 64 512 batches)."""
        rot, trans = transforms[..., :3, :3], transforms[..., :3, 3]
        off_t, off_z = self.transforms_to_check[:, :3, 3], self.transforms_to_check[:, :3, 2]      # (n5, 3)
        points = torch.einsum('bpik,ok->bpoi', rot, off_t) + trans[:, :, None, :]
        dirs = torch.einsum('bpik,ok->bpoi', rot, off_z)
        b = transforms.shape[0]
        return points.reshape(b, -1, 3), dirs.reshape(b, -1, 3)                                     # query order (np, n5)

    def _query_points_from_pose(self, translations, rotations):
        """compute_matrices + _query_points for the quaternion representation in four launches instead of about thirty (and as many again
        in each of the two backward passes): q (x) q, one (B np, 16) x (16, 6 n5) product, the offsets' constant, the translation."""
        b, n_p = rotations.shape[:2]
        n5 = self.n_transforms_to_check
        qq = (rotations[..., :, None] * rotations[..., None, :]).reshape(b * n_p, 16)
        both = (qq @ self._pose_map).reshape(b, n_p, 2, n5, 3)
        off_t, off_z = self.transforms_to_check[:, :3, 3], self.transforms_to_check[:, :3, 2]
        points = both[:, :, 0] + off_t + translations[:, :, None, :]
        dirs = both[:, :, 1] + off_z
        return points.reshape(b, -1, 3), dirs.reshape(b, -1, 3)

    def _call(self, inputs, transforms, n_points, batched_features, state=None):
        """model_v4.py:211-265: poses = transforms @ offsets; points = their translations, directions = their z axes;
        trunk -> fused activations (b, np, n5, 128) x 4 -> GraspReadout -> (B, np)."""
        state = state or self.trunk_state(inputs, batched_features)
        if transforms is None:                           # the training step: straight from the pose variables (quaternion representation)
            points, dirs = self._query_points_from_pose(self.translations, self.rotations)
            b = self.translations.shape[0]
        else:
            points, dirs = self._query_points(transforms)
            b = transforms.shape[0]
        acts = TrunkField.apply(points, dirs, state)                                     # (4, B, np*n5, 128)
        acts = acts.reshape(N_FUSED, b, n_points, self.n_transforms_to_check, 128)
        return self.grasp_readout(acts)

    def infer(self, inputs, transforms, n_points_infer, batched_features, compute_dtype='f32'):
        """model_v4.py:208-209.  compute_dtype='bf16' evaluates the trunk on the bf16 MFMA kernel (no gradients there)."""
        transforms = torch.as_tensor(transforms, dtype=torch.float32).to(self.device_)
        with torch.no_grad():
            if compute_dtype == 'f32':
                return self._call(inputs, transforms, n_points_infer, batched_features)
            if compute_dtype != 'bf16':
                raise ValueError("compute_dtype must be 'f32' or 'bf16'")
            state = self.trunk_state(inputs, batched_features)
            b = transforms.shape[0]
            points, dirs = (t.contiguous() for t in self._query_points(transforms))
            z = torch.zeros(b, points.shape[1], 1, dtype=torch.float32, device=self.device_)
            _, acts = ops.field_eval_bf16(points, dirs, z, *state.geo, state.packed, ops.pack_net_bf16(self.trunk_net),
                                          return_fused_acts=True)
            acts = acts.reshape(N_FUSED, b, n_points_infer, self.n_transforms_to_check, 128)
            return self.grasp_readout(list(acts.unbind(0)))

    def loss_and_grads(self, data, combined_features):
        """The body of train_step (model_v4.py:277-318) up to the optimizer: returns (dict of losses, prediction)."""
        inputs, labels = data
        dev = self.device_
        lab = [torch.as_tensor(l, dtype=torch.float32).to(dev) for l in labels]
        state = self.trunk_state(inputs, combined_features)
        self.set_pose(inputs[0], inputs[1])
        fast = self.rotation_representation == 'quaternion'
        y_pred = self._call(inputs, None if fast else self.compute_matrices(), self.n_points_train, combined_features, state)
        if self.softmax_before_loss:
            y_pred = torch.softmax(y_pred, -1)
        landscape_loss = self.loss(lab[0], y_pred)
        self.set_pose(inputs[2], inputs[3])
        prediction = self._call(inputs, None if fast else self.compute_matrices(), self.n_points_train, combined_features, state)
        grads = torch.autograd.grad(prediction.sum(), self.pose_variables, create_graph=True)
        loss_t = cosine_similarity_loss(lab[1], grads[0])
        if self.rotation_representation == 'quaternion':
            loss_r = cosine_similarity_loss(lab[2], grads[1])
        else:
            loss_r = cosine_similarity_loss(lab[2][..., :3], grads[1][..., :3]) + cosine_similarity_loss(lab[2][..., 3:], grads[1][..., 3:])
        loss = loss_t + loss_r + landscape_loss
        for prm in self.grasp_readout.parameters():
            prm.grad = None
        loss.sum().backward(inputs=list(self.grasp_readout.parameters()))
        return {'landscape_loss': landscape_loss.detach().mean(), 'grad_loss_t': loss_t.detach(), 'grad_loss_r': loss_r.detach(),
                'pred': prediction.detach().mean()}, prediction.detach()

    def _clip_and_step(self):
        for prm in self.grasp_readout.parameters():                      # optimize(): clip-by-value 1.0, then Adam
            if prm.grad is not None:
                prm.grad.clamp_(-1.0, 1.0)
        self.optimizer.step()

    def train_step(self, data, combined_features):
        if self.optimizer is None:
            self.compile()
        if self._graph_mode:
            return self._train_step_graphed(data, combined_features)
        out, _ = self.loss_and_grads(data, combined_features)
        self._clip_and_step()
        return out

    # -- the step as one HIP graph (compile(graph=True)) --
    _GRAPH_INPUTS = ('translations_landscape', 'rotations_landscape', 'translations_grad', 'rotations_grad', 'src_images', 'src_intrinsics',
                     'src_extrinsics_inv', 'combined_features', 'label_landscape', 'label_grad_t', 'label_grad_r')

    def graph_inputs(self):
        """name -> the device buffer the captured step reads (None before the first step)."""
        return None if self._g_static is None else dict(zip(self._GRAPH_INPUTS, self._g_static))

    def bind_graph_inputs(self, data, combined_features):
        """Make the caller's own device tensors (fp32, contiguous) the buffers of the captured step: whatever they hold at the time of a
        `train_step` is what that step reads, nothing is copied (2.5 GB of features at the cfg3 shape).  Before the capture only."""
        if self._graph is not None:
            raise RuntimeError('the step is already captured; call compile(graph=True) again first')
        inputs, labels = data
        flat = [*inputs[:7], combined_features, *labels]
        for name, t in zip(self._GRAPH_INPUTS, flat):
            if not (isinstance(t, torch.Tensor) and t.device == self.device_ and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError(f'{name}: bind_graph_inputs wants contiguous float32 tensors on {self.device_}')
        self._g_static = flat

    def _stage(self, data, combined_features):
        inputs, labels = data
        flat = [torch.as_tensor(x, dtype=torch.float32) for x in (*inputs[:7], combined_features, *labels)]
        if self._g_static is None:
            self._g_static = [torch.empty(t.shape, dtype=torch.float32, device=self.device_) for t in flat]
        for t, s in zip(flat, self._g_static):
            if t.shape != s.shape:
                raise ValueError(f'compile(graph=True) fixed the input shapes at the first step: got {tuple(t.shape)}, captured '
                                 f'{tuple(s.shape)}; call compile(graph=True) again for a new shape')
            if not (t.device == s.device and t.data_ptr() == s.data_ptr()):
                s.copy_(t, non_blocking=True)
        st = self._g_static
        return (tuple(st[:7]), tuple(st[8:])), st[7]

    def _train_step_graphed(self, data, combined_features):
        data_s, feats = self._stage(data, combined_features)
        if self._g_calls < 2:
            self._g_calls += 1
            side = torch.cuda.Stream(self.device_)
            side.wait_stream(torch.cuda.current_stream(self.device_))
            with torch.cuda.stream(side):
                out, _ = self.loss_and_grads(data_s, feats)
                self._clip_and_step()
            torch.cuda.current_stream(self.device_).wait_stream(side)
            return out
        if self._graph is None:
            torch.cuda.synchronize(self.device_)
            for prm in self.grasp_readout.parameters():
                prm.grad = None
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out, _ = self.loss_and_grads(data_s, feats)
                self._clip_and_step()
            self._graph, self._g_out = graph, out
        self._graph.replay()
        return {k: v.clone() for k, v in self._g_out.items()}
