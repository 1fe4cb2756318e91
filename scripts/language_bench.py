#!/usr/bin/env python3
"""cfg3-shaped workload: the trunk as a field on grasp-pose query points, as train_language.py uses it.
B scenes x (n_points x 42 offsets) query points, V source views of HxW, fp32.  Reports query points/s for
  forward bf16     mvnerf_field_eval_bf16 with the 4 fused activations written (cfg3's dtype)
  forward          mvnerf_field_eval with complete_output (8 activations written)
  forward+stash    what TrunkField.forward runs
  vjp              mvnerf_query_vjp (12 dX launches + layer-0 input gradient)
  jvp              mvnerf_query_jvp (fused primal + tangent pass)
  train_step       LanguageNeRF.train_step: 2 forwards, VJP, JVP, GraspReadout fwd/bwd/double-bwd in torch, Adam
  train_step (HIP graph)  the same step after compile(graph=True): one graph replay
Usage: python scripts/language_bench.py [--batch 8] [--points 192] [--size 480 640] [--views 1] [--steps 10]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thesis_clip_nerf_amd import ops  # noqa: E402
from thesis_clip_nerf_amd.lmvnerf import LanguageNeRF  # noqa: E402
from thesis_clip_nerf_amd.synthetic import make_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--points', type=int, default=192, help='poses per scene (x 42 offsets = query points)')
ap.add_argument('--size', type=int, nargs=2, default=[480, 640])
ap.add_argument('--views', type=int, default=1)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--only-train', action='store_true', help='time the train step only (for kernel traces)')
ap.add_argument('--train-mode', choices=['both', 'eager', 'graph'], default='both', help='which train_step legs to run')
args = ap.parse_args()
dev = 'cuda:0'
h, w = args.size
sc = make_scene(seed=0, batch=1, n_views=args.views, height=h, width=w, n_rays=4)
rng = np.random.default_rng(0)
b, npts = args.batch, args.points
rep = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).expand(b, *a.shape[1:]).contiguous()
images, feats, k4, einv = rep(sc['images']), rep(sc['features']), rep(sc['intrinsics']), rep(sc['extrinsics_inv'])
model = LanguageNeRF(sc['fine'], n_points_train=npts, n_views=args.views, batch_size=b, rotation_representation='6d',
                     softmax_before_loss=True, device=dev)
model.compile()


def poses():
    t = (np.array([0.0, 0.0, 0.8]) + 0.1 * rng.standard_normal((b, npts, 3))).astype(np.float32)
    return t, rng.standard_normal((b, npts, 6)).astype(np.float32)


t1, r1 = poses()
t2, r2 = poses()
lab0 = rng.random((b, npts)).astype(np.float32)
lab0 /= lab0.sum(-1, keepdims=True)
labels = (lab0, rng.standard_normal((b, npts, 3)).astype(np.float32), rng.standard_normal((b, npts, 6)).astype(np.float32))
inputs = (t1, r1, t2, r2, images, k4, einv)
model.set_pose(t1, r1)
with torch.no_grad():
    tr = model.compute_matrices()
    ps = tr[:, :, None] @ model.transforms_to_check[None, None]
    points = ps[..., :3, 3].reshape(b, -1, 3).contiguous()
    dirs = ps[..., :3, 2].reshape(b, -1, 3).contiguous()
n_q = points.shape[0] * points.shape[1]
state = model.trunk_state(inputs, feats)
geo = state.geo
g_acts = torch.randn(4, b, points.shape[1], 128, device=dev)
tp, td = torch.randn_like(points), torch.randn_like(dirs)
stash = ops.query_stash(points, dirs, *geo, state.packed)


def timed(fn, steps=args.steps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


z0 = torch.zeros(b, points.shape[1], 1, device=dev)
packed16 = ops.pack_net_bf16(model.trunk_net)
res = ({'train_step': timed(lambda: model.train_step((inputs, labels), feats), max(2, args.steps // 2))} if args.train_mode != 'graph' else {}) if args.only_train else {
    'forward bf16': timed(lambda: ops.field_eval_bf16(points, dirs, z0, *geo, state.packed, packed16, return_fused_acts=True)),
    'forward': timed(lambda: ops.query_field(points, dirs, *geo, state.packed, complete_output=True)),
    'forward+stash': timed(lambda: ops.query_stash(points, dirs, *geo, state.packed, stash)),
    'vjp': timed(lambda: ops.query_vjp(points, dirs, *geo, state.bwd_streams, stash, g_acts)),
    'jvp': timed(lambda: ops.query_jvp(points, dirs, tp, td, *geo, state.packed)),
    'train_step': timed(lambda: model.train_step((inputs, labels), feats), max(2, args.steps // 2)),
}
# the same step captured as one HIP graph (compile(graph=True)): device-resident inputs bound as the graph's buffers
if args.train_mode != 'eager':
    gmodel = LanguageNeRF(sc['fine'], n_points_train=npts, n_views=args.views, batch_size=b, rotation_representation='6d',
                          softmax_before_loss=True, device=dev)
    gmodel.compile(graph=True)
    dv = lambda a: torch.from_numpy(a).to(dev)
    g_inputs = (dv(t1), dv(r1), dv(t2), dv(r2), images, k4, einv)
    g_labels = tuple(dv(l) for l in labels)
    gmodel.bind_graph_inputs((g_inputs, g_labels), feats)
    for _ in range(3):                                   # two eager steps and the capture itself stay outside the timed region
        gmodel.train_step((g_inputs, g_labels), feats)
    res['train_step (HIP graph)'] = timed(lambda: gmodel.train_step((g_inputs, g_labels), feats), max(4, args.steps))
print(f'cfg3 shape: B={b} scenes x {points.shape[1]} query points ({npts} poses x 42 offsets), V={args.views}, {h}x{w}x256 fp32 features '
      f'({feats.numel() * 4 / 1e9:.2f} GB), {n_q} points per pass')
for k, dt in res.items():
    n = 2 * n_q if k.startswith('train_step') else n_q
    print(f'  {k:24s} {dt * 1e3:8.3f} ms   {n / dt / 1e6:8.2f} M query points/s')
