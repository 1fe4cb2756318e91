"""BASELINE.json full sizes (cfg2: B=1, V=1, 64x64 source, 4096 rays, 64+128 samples; and the V=3 variant): properties
that do not need the oracle to finish - ray independence (chunk invariance, permutation equivariance), determinism,
sortedness / permutation of the merged depths, ranges - on the default (texel table) and the direct path.
The oracle itself checks 512 of these rays in bench.py (`parity`)."""
import numpy as np
import pytest
import torch

from thesis_clip_nerf_amd import ops
from thesis_clip_nerf_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope='module', params=[1, 3])
def scene(request):
    sc = make_scene(seed=5, batch=1, n_views=request.param, height=64, width=64)
    d = {k: dev(sc[k]) for k in ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'u_coarse', 'u_fine', 'coarse', 'fine']}
    d['pc'], d['pf'] = ops.pack_net(d['coarse']), ops.pack_net(d['fine'])
    d['near'], d['far'] = sc['near'], sc['far']
    return d


def render(d, sl=slice(None), tables='auto', perm=None):
    pick = (lambda t: t[:, sl]) if perm is None else (lambda t: t[:, perm])
    return ops.render_fwd(pick(d['rays_o']).contiguous(), pick(d['rays_d']).contiguous(), d['images'], d['features'], d['intrinsics'],
                          d['extrinsics_inv'], d['pc'], d['pf'], pick(d['u_coarse']).contiguous(), pick(d['u_fine']).contiguous(),
                          d['near'], d['far'], texel_tables=tables)


@pytest.mark.parametrize('tables', ['auto', None])
def test_rays_are_independent_and_runs_deterministic(scene, tables):
    d = scene
    assert d['rays_o'].shape[1] == 4096
    whole = render(d, tables=tables)
    again = render(d, tables=tables)
    for a, b in zip(whole, again):
        assert torch.equal(a, b)                                            # deterministic: no atomics on the forward path
    # two halves of the rays, and an odd split that leaves a ragged last tile
    for cut in (2048, 1234):
        lo, hi = render(d, slice(0, cut), tables=tables), render(d, slice(cut, 4096), tables=tables)
        for w_, a, b in zip(whole, lo, hi):
            assert torch.equal(w_, torch.cat([a, b], 1))
    perm = torch.randperm(4096, generator=torch.Generator().manual_seed(0)).to(DEV)
    shuffled = render(d, perm=perm, tables=tables)
    for w_, s_ in zip(whole, shuffled):
        assert torch.equal(w_[:, perm], s_)


def test_table_and_direct_paths_agree(scene):
    d = scene
    a, b = render(d, tables='auto'), render(d, tables=None)
    for name, x, y in zip(['rgb', 'depth', 'fine_rgb', 'fine_depth'], a, b):
        assert (x - y).abs().max().item() < 2e-5, name                      # fp32 re-association of layer 0 only


def test_sampling_properties_at_full_size(scene):
    d = scene
    z = ops.stratified_depths(d['u_coarse'], d['near'], d['far'])
    assert (z[..., 1:] > z[..., :-1]).all() and z.min() >= d['near'] and z.max() <= d['far']
    rgbs = ops.field_eval(d['rays_o'], d['rays_d'], z, d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'], d['pc'])
    assert torch.isfinite(rgbs).all()
    assert (rgbs[..., :3] >= 0).all() and (rgbs[..., :3] <= 1).all() and (rgbs[..., 3] >= 0).all()      # sigmoid / softplus
    rgb, depth, w = ops.composite(z, rgbs)
    assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-5).all()
    assert (rgb >= 0).all() and (rgb <= 1 + 1e-5).all()
    z_all, z_fine, above, below = ops.resample(z, w, d['u_fine'], return_aux=True)
    assert (z_all[..., 1:] >= z_all[..., :-1]).all()                        # sorted
    merged = torch.sort(torch.cat([z, z_fine], -1), -1).values
    assert torch.equal(z_all, merged)                                       # a permutation of coarse + fine depths
    assert (above >= 0).all() and (above <= 63).all() and (below >= 0).all() and (below <= 62).all()
    assert ((above - below == 1) | (above == 0)).all()
