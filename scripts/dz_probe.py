import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from thesis_clip_nerf_amd import ops, MVVNeRFRenderer
from thesis_clip_nerf_amd.synthetic import make_scene
DEV='cuda:0'
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
for seed in (78, 5, 11):
    sc = make_scene(seed=seed, batch=1, n_views=1, height=16, width=16, n_rays=32, bias_scale=0.05)
    y = np.random.default_rng(3).random((1, 32, 3)).astype(np.float32)
    inputs = tuple(sc[k] for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv'])
    res = {}
    for mode in ('bf16x6', 'f16x3'):
        os.environ['MVNERF_SPLIT_MFMA'] = mode
        for use_table in (True, False):
            ops.texel_table_pays = (lambda *a, _u=use_table: _u)
            m = MVVNeRFRenderer(32, 32, n_views=1, batch_size=1, near=sc['near'], far=sc['far'], device=DEV)
            m.set_weights(sc['coarse'], sc['fine'])
            _, grad, _ = m.loss_and_grads(inputs, y, sc['features'], u_coarse=dev(sc['u_coarse']), u_fine=dev(sc['u_fine']), stop_fine_z=False)
            torch.cuda.synchronize()
            res[(mode, use_table)] = grad.cpu().numpy()[:247300].copy()
    n = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    print(seed, 'bf16x6 table vs direct %.2e' % n(res[('bf16x6', True)], res[('bf16x6', False)]), ' f16x3 table vs direct %.2e' % n(res[('f16x3', True)], res[('f16x3', False)]),
          ' f16x3 vs bf16x6 (table) %.2e (direct) %.2e' % (n(res[('f16x3', True)], res[('bf16x6', True)]), n(res[('f16x3', False)], res[('bf16x6', False)])))
