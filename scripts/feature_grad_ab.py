import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from thesis_clip_nerf_amd import MVVNeRFRenderer, ops
from thesis_clip_nerf_amd.synthetic import make_scene
dev='cuda:0'
for size, rays in ((64, 4096), (128, 16384)):
    sc = make_scene(seed=0, height=size, width=size)
    r = sc['rays_o'].shape[1]
    y = torch.rand((1, r, 3), device=dev)
    t = lambda k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev)
    inputs = tuple(t(k) for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv'])
    feats, uc, uf = t('features'), t('u_coarse'), t('u_fine')
    orig = ops.texel_table_pays
    for name, fn in (('table', orig), ('direct', lambda *a: False)):
        ops.texel_table_pays = fn
        m = MVVNeRFRenderer(r, r, n_views=1, near=sc['near'], far=sc['far'], device=dev)
        m.set_weights(sc['coarse'], sc['fine'])
        for flag in (False, True):
            for _ in range(2): m.loss_and_grads(inputs, y, feats, u_coarse=uc, u_fine=uf, return_d_features=flag)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): m.loss_and_grads(inputs, y, feats, u_coarse=uc, u_fine=uf, return_d_features=flag)
            torch.cuda.synchronize()
            print(f'{size}x{size} {r} rays, {name:6s} d_features={flag}: {(time.perf_counter()-t0)/5*1e3:.2f} ms')
    ops.texel_table_pays = orig
