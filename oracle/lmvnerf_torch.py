"""Torch (CPU, float64) restatement of the language/grasp model's use of the trunk (lmvnerf/model_v4.py:67-101,
:192-265, :277-318; delta_ngf/layers.py:8-42; layers.py:262-298, :400-411).  TEST INFRASTRUCTURE ONLY.

Everything is plain differentiable torch on top of oracle/mvnerf_torch.query_acts, so autograd supplies the nested
gradient of the reference's train_step (d prediction / d pose inside the tape, then d loss / d read-out variables).
Parity status: unpinned by the reference (no tests, TensorFlow / tensorflow_graphics absent); third-party pieces
restated from their published definitions: tfg rotation_matrix_3d.from_quaternion (x,y,z,w, no normalisation),
scipy Rotation.from_euler('xyz', ...) inside manipulation_tasks.transform.Affine (imported here from scipy itself),
tf.keras.losses.CosineSimilarity / KLDivergence, keras ELU.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial.transform import Rotation

from . import mvnerf_torch as T


def affine(translation, rotation=(0, 0, 0, 1)):
    """manipulation_tasks/transform.py:11-23."""
    m = np.eye(4)
    m[:3, 3] = np.array(translation)
    m[:3, :3] = (Rotation.from_quat(rotation) if len(rotation) == 4 else Rotation.from_euler('xyz', rotation)).as_matrix()
    return m


def transforms_to_check(n_5d_poses=7):
    """model_v4.py:67-101 -> (6*n_5d_poses, 4, 4) in the reference's list order (base-major)."""
    bx, by, bz = 0.02, 0.015, 0.0125
    step = (bx - 0.005) / ((n_5d_poses - 1) / 2)
    bases = [affine([0, by, 0]), affine([0, -by, 0]),
             affine([-bx, by, bz], [0.0, np.pi / 2, 0.0]), affine([bx, by, bz], [0.0, -np.pi / 2, 0.0]),
             affine([-bx, -by, bz], [0.0, np.pi / 2, 0.0]), affine([bx, -by, bz], [0.0, -np.pi / 2, 0.0])]
    c = int((n_5d_poses - 1) / 2)
    steps = [affine([0.0, 0.0, i * step]) for i in range(-c, c + 1)]
    return np.array([b @ t for b in bases for t in steps])


def from_quaternion(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    rows = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    return torch.stack([torch.stack(r, -1) for r in rows], -2)


def compute_matrices(translations, rotations, representation):
    if representation == 'quaternion':
        rot = from_quaternion(rotations)
    else:
        r1 = rotations[..., :3] / torch.linalg.norm(rotations[..., :3], dim=-1, keepdim=True)
        r2 = rotations[..., 3:] / torch.linalg.norm(rotations[..., 3:], dim=-1, keepdim=True)
        rot = torch.stack([r1, r2, torch.linalg.cross(r1, r2)], -1)
    b, n = translations.shape[:2]
    m = torch.zeros(b, n, 4, 4, dtype=translations.dtype)
    m[..., :3, :3] = rot
    m[..., :3, 3] = translations
    m[..., 3, 3] = 1.0
    return m


def elu(x):
    return torch.where(x > 0, x, torch.expm1(x))


def grasp_readout(w, acts):
    """delta_ngf/layers.py:30-42.  w: dict of Keras-layout kernels [in,out] / biases; acts: 4 x (B, np, n5, 128)."""
    ds = [elu(a @ w[f'ds{i}.k'] + w[f'ds{i}.b']) for i, a in enumerate(acts)]
    x = elu(torch.cat(ds, -1) @ w['comb.k'] + w['comb.b'])
    x = x.reshape(x.shape[0], x.shape[1], -1)
    r = elu(elu(x) @ w['b0.l0.k'] + w['b0.l0.b']) @ w['b0.l1.k'] + w['b0.l1.b']
    x = x @ w['b0.sc.k'] + r
    r = elu(elu(x) @ w['b1.l0.k'] + w['b1.l0.b']) @ w['b1.l1.k'] + w['b1.l1.b']
    x = x + r
    return (torch.relu(x) @ w['out.k'] + w['out.b'])[..., 0]


def call(w, net, transforms, checks, n_points, images, features, k4, einv):
    """LanguageNeRF._call (model_v4.py:211-265) with the reference's tensor order: poses (B, n5, np, 4, 4)."""
    poses = transforms[:, None] @ checks[None, :, None]
    trans = poses[..., :3, 3]                                            # (B, n5, np, 3)
    dirs = (poses[..., :3, :3] @ torch.tensor([[0.0], [0.0], [1.0]], dtype=poses.dtype))[..., 0]
    b, n5 = trans.shape[:2]
    pts = trans.permute(0, 2, 1, 3).reshape(b, n_points * n5, 3)          # '(n5 np) -> np n5'
    drs = dirs.permute(0, 2, 1, 3).reshape(b, n_points * n5, 3)
    acts = T.query_acts(net, pts, drs, images, features, k4, einv)
    return grasp_readout(w, [a.reshape(b, n_points, n5, 128) for a in acts])


def cosine_similarity(y_true, y_pred):
    def l2n(x):
        return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-12))
    return -(l2n(y_true) * l2n(y_pred)).sum(-1).mean()


def train_losses(w, net, inputs, labels, checks, n_points, features, representation, softmax_before_loss=True):
    """model_v4.py:277-318 -> (total loss (B,), landscape, loss_t, loss_r, prediction); `w` leaves require grad."""
    images, k4, einv = inputs[4], inputs[5], inputs[6]
    tr = compute_matrices(inputs[0], inputs[1], representation)
    y = call(w, net, tr, checks, n_points, images, features, k4, einv)
    if softmax_before_loss:
        y = torch.softmax(y, -1)
    yt, yp = torch.clamp(labels[0], 1e-7, 1.0), torch.clamp(y, 1e-7, 1.0)
    landscape = (yt * torch.log(yt / yp)).sum(-1)
    t2 = inputs[2].clone().requires_grad_(True)
    r2 = inputs[3].clone().requires_grad_(True)
    pred = call(w, net, compute_matrices(t2, r2, representation), checks, n_points, images, features, k4, einv)
    g_t, g_r = torch.autograd.grad(pred.sum(), (t2, r2), create_graph=True)
    loss_t = cosine_similarity(labels[1], g_t)
    if representation == 'quaternion':
        loss_r = cosine_similarity(labels[2], g_r)
    else:
        loss_r = cosine_similarity(labels[2][..., :3], g_r[..., :3]) + cosine_similarity(labels[2][..., 3:], g_r[..., 3:])
    return loss_t + loss_r + landscape, landscape, loss_t, loss_r, pred
