import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thesis_clip_nerf_amd import ops
from thesis_clip_nerf_amd.synthetic import make_scene
DEV='cuda:0'
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
n_views, n_rays, s = 1, 4, 64
sc = make_scene(seed=81, n_views=n_views, height=24, width=28, n_rays=n_rays, bias_scale=0.1)
d = {k: dev(sc[k]) for k in ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'fine']}
z = np.sort(np.random.default_rng(0).uniform(0.3, 1.3, (1, n_rays, s)).astype(np.float32), -1)
packed, split = ops.pack_net(d['fine']), ops.pack_net_split(d['fine'])
args = (d['rays_o'], d['rays_d'], dev(z), d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'], packed)
rgbs, acts = ops.field_eval_split(*args, split, complete_output=True)
rgbs32, acts32 = ops.field_eval(*args, complete_output=True)
torch.cuda.synchronize()
os.makedirs('gpurun_out', exist_ok=True)
np.savez_compressed('gpurun_out/split_debug.npz', acts=np.stack([a.cpu().numpy() for a in acts[:2]]), acts32=np.stack([a.cpu().numpy() for a in acts32[:2]]),
                    split=split.cpu().numpy())
print('saved', (acts[1] - acts32[1]).abs().max().item())
