"""world_size-2 `gloo` tests (CPU) of the N>1 path: ray sharding, uneven all_gather, the single flat
gradient all-reduce and the max-over-ranks timing used by bench.py.  The kernels themselves are not
involved (no GPU here); `render_fn` is a per-ray stand-in, which is all the sharding logic can see."""
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from thesis_clip_nerf_amd import distributed as D


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 4096, 307200):
        for world in (1, 2, 3, 8):
            b = [D.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_bounds(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n):
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        rays_o = torch.rand((n, 3), generator=g)
        rays_d = torch.rand((n, 3), generator=g)

        def render_fn(o, d):                       # any per-ray function
            return (o * 2 + d, (o * d).sum(-1))

        rgb, depth = D.render_rays_sharded(render_fn, rays_o, rays_d)
        ref_rgb, ref_depth = render_fn(rays_o, rays_d)
        assert torch.equal(rgb, ref_rgb) and torch.equal(depth, ref_depth)

        flat = torch.full((494600,), float(rank + 1))                    # both MLPs' gradients, one buffer
        D.allreduce_mean_(flat)
        assert torch.allclose(flat, torch.full_like(flat, (1 + world) / 2))

        assert D.max_over_ranks(0.5 + rank) == 0.5 + (world - 1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n', [10, 7])
def test_world2_gloo(n):
    mp.spawn(_worker, args=(2, _free_port(), n), nprocs=2, join=True)


def test_single_process_passthrough():
    x = torch.arange(12.).reshape(4, 3)
    assert D.all_gather_rows(x, 4) is x
    assert D.max_over_ranks(1.25) == 1.25
    out = D.render_rays_sharded(lambda o, d: (o + d,), x, x)
    assert torch.equal(out[0], 2 * x)
