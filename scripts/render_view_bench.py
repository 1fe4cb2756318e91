#!/usr/bin/env python3
"""Full-frame render at the reference's native size (480x640 source views, model_v0.py:243-281):
times render_view (one _call for the whole frame vs the reference's 512-ray chunks).  Parity at this size is a test
(tests/test_gpu_configs.py, cfg5: 256 strided rays against the oracle), not part of this script.  Usage: python scripts/render_view_bench.py [--views V]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thesis_clip_nerf_amd import MVVNeRFRenderer, ops, render_view  # noqa: E402
from thesis_clip_nerf_amd.model import camera_parameters  # noqa: E402
from thesis_clip_nerf_amd.synthetic import pinhole, ring_pose  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--views', type=int, default=1)
args = ap.parse_args()
h, w, v = 480, 640, args.views
dev = 'cuda:0'
rng = np.random.default_rng(0)
src_colors = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(v)]
k = pinhole(w, h)
cfgs = [{'pose': ring_pose(rng.uniform(0, 6.28)), 'intrinsics': k.reshape(-1)} for _ in range(v)]
tgt = {'pose': ring_pose(1.0), 'intrinsics': k.reshape(-1)}
feats = torch.randn((1, v, h, w, 256), device=dev) * 0.5
m = MVVNeRFRenderer(512, 512, n_views=v, near=0.3, far=1.3, device=dev, seed=3)

for chunk in (None, 16384, 512):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rgb8, d8 = render_view(m, src_colors, cfgs, tgt, combined_features=feats, chunk=chunk,
                               generator=torch.Generator(device=dev).manual_seed(0))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f'V={v} 480x640 frame, chunk={chunk}: {dt*1e3:.1f} ms  ({h*w/dt:.0f} rays/s incl. source upload + uint8 download)', flush=True)

# device-only timing of the whole-frame _call
images = torch.from_numpy(np.array([[c / 255.0 for c in src_colors]], dtype=np.float32)).to(dev)
cams = [camera_parameters(c) for c in cfgs]
einv = torch.from_numpy(np.array([[c[0] for c in cams]], dtype=np.float32)).to(dev)
k4 = torch.from_numpy(np.array([[c[1] for c in cams]], dtype=np.float32)).to(dev)
pose = tgt['pose']
ro, rd = ops.get_rays_device(pose[:3, :3] @ np.linalg.inv(k), pose[:3, 3], dev, width=w, height=h)
uc = torch.rand((1, h * w, 64), device=dev)
uf = torch.rand((1, h * w, 64), device=dev)
inputs = (ro[None], rd[None], images, k4, einv)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m._call(inputs, h * w, 1, feats, u_coarse=uc, u_fine=uf)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f'V={v} _call on {h*w} rays, inputs resident: {dt*1e3:.1f} ms = {h*w/dt:.0f} rays/s', flush=True)
