"""How the fp16 two-piece kernel behaves as activations grow: the trunk's weights scaled up until the residual stream reaches 1e2 .. 1e6,
embedding of the f16x3 / bf16x6 kernels against the fp32-MFMA kernel's (relative to max |embedding|)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from thesis_clip_nerf_amd import ops
from thesis_clip_nerf_amd.synthetic import make_scene
DEV = 'cuda:0'
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
sc = make_scene(seed=5, n_views=1, height=24, width=24, n_rays=256, bias_scale=0.1)
d = {k: dev(sc[k]) for k in ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'fine']}
z = dev(np.sort(np.random.default_rng(0).uniform(0.3, 1.3, (1, 256, 64)).astype(np.float32), -1))
for gain in (1.0, 1.5, 2.0, 2.5, 3.0, 3.6):
    net = d['fine'].clone()
    net[379 * 128 + 128:-516] *= gain                      # the 12 hidden Dense layers (kernels and biases)
    packed, split = ops.pack_net(net), ops.pack_net_split(net)
    args = (d['rays_o'], d['rays_d'], z, d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'], packed)
    _, ref = ops.field_eval(*args, return_embedding=True)
    out = {}
    for name in ('split_f16', 'split_bf16'):
        ops.set_split_kernel(name)
        _, emb = ops.field_eval_split(*args, split, return_embedding=True)
        out[name] = emb
    torch.cuda.synchronize()
    m = ref.abs().max().item()
    print(f'gain {gain}: max |embedding| {m:.3e}  f16x3 vs fp32 kernel {(out["split_f16"] - ref).abs().max().item() / m:.2e}  '
          f'bf16x6 vs fp32 kernel {(out["split_bf16"] - ref).abs().max().item() / m:.2e}  finite: {bool(torch.isfinite(out["split_f16"]).all())}')
