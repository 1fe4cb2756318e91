"""Worker of tests/test_gpu_distributed.py: one of `WORLD_SIZE` ranks sharing the box's single GPU (gloo rendezvous; on an
8-GPU node the same code runs one rank per GPU over RCCL).  Each rank trains on ITS scene of a `world`-scene batch with
`compile(grad_sync=allreduce_mean_)` and renders its shard of one image through the real `_call`; results go to
`<out>/rank<r>.pt` for the parent test to compare with the single-process run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from thesis_clip_nerf_amd import MVVNeRFRenderer  # noqa: E402
from thesis_clip_nerf_amd import distributed as D  # noqa: E402
from thesis_clip_nerf_amd.synthetic import make_scene  # noqa: E402

SCENE = dict(seed=90, n_views=2, height=16, width=16, n_rays=64, bias_scale=0.05)
FRAME = dict(seed=91, n_views=1, height=16, width=20, bias_scale=0.05)          # 320 rays: uneven 2-way split of a ragged tile count


def main(out_dir):
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    try:
        sc = make_scene(batch=world, **SCENE)
        y = np.random.default_rng(2).random((world, 64, 3)).astype(np.float32)
        mine = slice(rank, rank + 1)
        m = MVVNeRFRenderer(64, 64, n_views=2, batch_size=1, near=sc['near'], far=sc['far'], device=dev)
        m.set_weights(sc['coarse'], sc['fine'])
        m.compile(learning_rate=1e-3, grad_sync=D.allreduce_mean_)
        inputs = tuple(sc[k][mine] for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv'])
        kw = dict(u_coarse=t(sc['u_coarse'][mine]), u_fine=t(sc['u_fine'][mine]))
        loss, grad, _ = m.loss_and_grads(inputs, y[mine], sc['features'][mine], **kw)
        local_grad = grad.clone()
        synced = D.allreduce_mean_(grad.clone())
        step_loss = m.train_step((inputs, y[mine]), combined_features=sc['features'][mine], **kw)['loss']
        res = {'loss': loss.cpu(), 'local_grad': local_grad.cpu(), 'synced_grad': synced.cpu(), 'step_loss': step_loss.cpu(),
               'coarse_net': m.coarse_net.cpu(), 'fine_net': m.fine_net.cpu()}
        # the same step with the fine half of the gradient all-reduced on a second stream while the coarse backward runs
        # (distributed.OverlappedGradSync + mvnerf_train_call.fine_grad_event): the same sums, so the same weights bit for bit
        m2 = MVVNeRFRenderer(64, 64, n_views=2, batch_size=1, near=sc['near'], far=sc['far'], device=dev)
        m2.set_weights(sc['coarse'], sc['fine'])
        m2.compile(learning_rate=1e-3, grad_sync=D.OverlappedGradSync(dev))
        m2.train_step((inputs, y[mine]), combined_features=sc['features'][mine], **kw)
        res['coarse_net_overlap'], res['fine_net_overlap'] = m2.coarse_net.cpu(), m2.fine_net.cpu()

        # inference: one frame's rays sharded over the ranks, every rank renders its block through the real `_call`
        fr = make_scene(batch=1, **FRAME)
        r = MVVNeRFRenderer(320, 320, n_views=1, near=fr['near'], far=fr['far'], device=dev)
        r.set_weights(fr['coarse'], fr['fine'])
        geo = (t(fr['images']), t(fr['intrinsics']), t(fr['extrinsics_inv']))
        feats, uc, uf = t(fr['features']), t(fr['u_coarse']), t(fr['u_fine'])
        n = fr['rays_o'].shape[1]
        lo, hi = D.shard_bounds(n, rank, world)

        def render_fn(o, d):
            return r.infer((o[None], d[None], *geo), feats, u_coarse=uc[:, lo:hi].contiguous(), u_fine=uf[:, lo:hi].contiguous())

        outs = D.render_rays_sharded(lambda o, d: tuple(x[0] for x in render_fn(o, d)), t(fr['rays_o'][0]), t(fr['rays_d'][0]))
        res['sharded'] = [o.cpu() for o in outs]
        torch.cuda.synchronize()
        torch.save(res, os.path.join(out_dir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
