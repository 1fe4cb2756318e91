"""Compare the training stash (tile layout) with the complete_output activations of the inference kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from thesis_clip_nerf_amd import ops
from thesis_clip_nerf_amd.synthetic import make_scene
DEV = 'cuda:0'
for views in (1, 2):
    sc = make_scene(seed=3, n_views=views, height=16, width=16, n_rays=24, bias_scale=0.05)
    d = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(DEV) for k in ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'u_coarse', 'coarse']}
    z = ops.stratified_depths(d['u_coarse'], 0.3, 1.3)
    pk = ops.pack_net(d['coarse'])
    args = (d['rays_o'], d['rays_d'], z, d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'], pk)
    rgbs, acts = ops.field_eval(*args, complete_output=True)
    rgbs2, stash = ops.field_eval_stash(*args)
    torch.cuda.synchronize()
    print('views', views, 'rgbs diff', (rgbs - rgbs2).abs().max().item())
    st = stash.view(torch.float32)
    n = 24 * 64
    tiles = n // 32
    per_view = st[:7 * views * tiles * 4096].view(7, views * tiles, 128, 32)
    fused = st[7 * views * tiles * 4096:14 * views * tiles * 4096 // (views) * 1 + 7 * views * tiles * 4096][:7 * tiles * 4096].view(7, tiles, 128, 32)
    # per-view slots 0,2,4,6 = x0,x1,x2,x3 ; fused slots 0,2,4,6 = mean,x4,x5,x6
    for k in range(4):
        a = acts[k].reshape(views * n, 128)                       # (B*V, R, S, 128)
        s_ = per_view[2 * k].permute(0, 2, 1).reshape(views * n, 128)
        print('  view act', k, (a - s_).abs().max().item())
    for k in range(4):
        a = acts[4 + k].reshape(n, 128)
        s_ = fused[2 * k].permute(0, 2, 1).reshape(n, 128)
        print('  fused act', k, (a - s_).abs().max().item())
