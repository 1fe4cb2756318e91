#!/usr/bin/env python3
"""Training-step throughput (model_v0.py:186-197) on cfg2 sizes: B=1, V=1, 64x64 source view, R rays.
Usage: python scripts/train_bench.py [--rays 4096] [--steps 10]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thesis_clip_nerf_amd import MVVNeRFRenderer  # noqa: E402
from thesis_clip_nerf_amd.synthetic import make_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--rays', type=int, default=4096)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--size', type=int, default=64)
ap.add_argument('--deterministic', action='store_true', help='weight gradients summed in a fixed order (mvnerf_set_deterministic)')
ap.add_argument('--feature-grad', action='store_true', help='also time loss_and_grads with dL/d(combined_features)')
args = ap.parse_args()
dev = 'cuda:0'
sc = make_scene(seed=0, height=args.size, width=args.size, n_rays=None if args.rays == args.size ** 2 else args.rays)
r = sc['rays_o'].shape[1]
y = torch.rand((1, r, 3), device=dev)
m = MVVNeRFRenderer(r, r, n_views=1, near=sc['near'], far=sc['far'], device=dev)
m.set_weights(sc['coarse'], sc['fine'])
m.compile(learning_rate=1e-4, deterministic=args.deterministic)
t = lambda k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev)
inputs = tuple(t(k) for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv'])
feats, uc, uf = t('features'), t('u_coarse'), t('u_fine')
for _ in range(2):
    out = m.train_step((inputs, y), combined_features=feats, u_coarse=uc, u_fine=uf)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    out = m.train_step((inputs, y), combined_features=feats, u_coarse=uc, u_fine=uf)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print(f'train_step{" (deterministic)" if args.deterministic else ""}: {r} rays, {dt*1e3:.2f} ms/step = {r/dt:.0f} rays/s (fwd+bwd+Adam), loss {float(out["loss"]):.5f}')

if args.feature_grad:
    for _ in range(2):
        m.loss_and_grads(inputs, y, feats, u_coarse=uc, u_fine=uf, return_d_features=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.loss_and_grads(inputs, y, feats, u_coarse=uc, u_fine=uf, return_d_features=True)
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / args.steps
    for _ in range(2):
        m.loss_and_grads(inputs, y, feats, u_coarse=uc, u_fine=uf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.loss_and_grads(inputs, y, feats, u_coarse=uc, u_fine=uf)
    torch.cuda.synchronize()
    dt1 = (time.perf_counter() - t0) / args.steps
    print(f'loss_and_grads: {dt1*1e3:.2f} ms; with dL/d(features) ({feats.numel()*4/1e6:.1f} MB map): {dt2*1e3:.2f} ms')
