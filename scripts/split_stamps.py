"""Debug (variants/lib_stamp.so, -DMVS_STAMP=1): per-wave cycle totals of the split kernel's fine pass at cfg2."""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thesis_clip_nerf_amd import ops
from thesis_clip_nerf_amd.synthetic import make_scene
DEV = 'cuda:0'
sc = make_scene(seed=0)
t = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(DEV) for k in ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'u_coarse', 'u_fine', 'coarse', 'fine']}
pf, pfs = ops.pack_net(t['fine']), ops.pack_net_split(t['fine'])
tab = ops.project_texels(t['features'], pf)
z = ops.stratified_depths(t['u_coarse'], 0.3, 1.3)
z_all = torch.sort(torch.cat([z, z + 0.001], -1), -1).values.contiguous()
for rep in range(3):
    rgbs, pix = ops.field_eval_split(t['rays_o'], t['rays_d'], z_all, t['images'], t['features'], t['intrinsics'], t['extrinsics_inv'], pf, pfs, return_pix=True, texel_table=tab)
    torch.cuda.synchronize()
dbg = pix.view(-1).view(torch.int64)[:8 * 2048].view(2048, 8).cpu().numpy().astype(np.float64)
ksteps = 8 * 102
body, wait, total = dbg[:, 0], dbg[:, 1], dbg[:, 2]
print('waves', len(body), 'k-steps per wave', ksteps)
print('per k-step: body cycles mean %.0f (min %.0f max %.0f), wait+barrier mean %.0f (min %.0f max %.0f)' % (body.mean() / ksteps, body.min() / ksteps, body.max() / ksteps, wait.mean() / ksteps, wait.min() / ksteps, wait.max() / ksteps))
print('whole kernel per wave: mean %.0f cycles, body %.1f %%, wait %.1f %%, rest %.1f %%' % (total.mean(), 100 * body.sum() / total.sum(), 100 * wait.sum() / total.sum(), 100 * (1 - (body.sum() + wait.sum()) / total.sum())))
start = dbg[:, 3]
print('start skew over waves: %.0f cycles' % (start.max() - start.min()))
by_wave = dbg.reshape(256, 8, 8)
print('wait per k-step by wave index in WG:', np.round(by_wave[:, :, 1].mean(0) / ksteps))
print('body per k-step by wave index in WG:', np.round(by_wave[:, :, 0].mean(0) / ksteps))
for q in range(4):
    print('group', q, 'cycles per k-step by wave index:', np.round(by_wave[:, :, 4 + q].mean(0) / ksteps))
