"""BASELINE.json full sizes (cfg2: B=1, V=1, 64x64 source, 4096 rays, 64+128 samples; and the V=3 variant), on BOTH fp32
field kernels - the default `field_eval_split_kernel` (the one bench.py times; `split` = the two three-piece bf16 operand
streams) and the fp32-MFMA `field_eval_kernel` - with and without the texel table:
* 512 strided rays of the 4096 against the NumPy oracle at the 1e-4 bar, bilinear tap indices `array_equal` (a6);
* properties that do not need the oracle to finish: ray independence (chunk invariance, permutation equivariance),
  determinism, sortedness / permutation of the merged depths, ranges."""
import numpy as np
import pytest
import torch

from oracle import mvnerf_oracle as O
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
    d['split'] = (ops.pack_net_split(d['coarse']), ops.pack_net_split(d['fine']))
    d['near'], d['far'] = sc['near'], sc['far']
    d['host'] = sc
    return d


GEMMS = ['split_f16', 'split_bf16', 'mfma_f32']                             # MVVNeRFRenderer(f32_gemm=...): default (what bench.py times) first


def _split(d, gemm):
    """The packed split image for the two split kernels (selecting which of them runs), None for the fp32-MFMA kernel."""
    if gemm == 'mfma_f32':
        return None
    ops.set_split_kernel(gemm)
    return d['split']


def render(d, sl=slice(None), tables='auto', perm=None, gemm='split_f16'):
    pick = (lambda t: t[:, sl]) if perm is None else (lambda t: t[:, perm])
    return ops.render_fwd(pick(d['rays_o']).contiguous(), pick(d['rays_d']).contiguous(), d['images'], d['features'], d['intrinsics'],
                          d['extrinsics_inv'], d['pc'], d['pf'], pick(d['u_coarse']).contiguous(), pick(d['u_fine']).contiguous(),
                          d['near'], d['far'], texel_tables=tables, split=_split(d, gemm))


@pytest.mark.parametrize('gemm', GEMMS)
def test_strided_rays_match_the_oracle(scene, gemm):
    """512 of the 4096 rays (every 8th) against the NumPy oracle, rendered as part of the full 4096-ray launch."""
    d, sc = scene, scene['host']
    sub = np.arange(0, 4096, 8)
    ref = O.render_call(O.unflatten_net(sc['coarse']), O.unflatten_net(sc['fine']), sc['rays_o'][:, sub], sc['rays_d'][:, sub],
                        sc['images'], sc['intrinsics'], sc['extrinsics_inv'], sc['features'], sc['near'], sc['far'], 64,
                        sc['u_coarse'][:, sub], sc['u_fine'][:, sub], return_aux=True)
    for tables in ('auto', None):
        got = render(d, tables=tables, gemm=gemm)
        torch.cuda.synchronize()
        for name, g, want in zip(['rgb', 'depth', 'fine_rgb', 'fine_depth'], got, ref[:4]):
            err = float(np.abs(g[:, sub].cpu().numpy() - want).max())
            print(f'cfg2 V={d["images"].shape[1]} {gemm} tables={tables} {name}: max|hip - oracle| over 512 rays = {err:.2e}')
            assert err < 1e-4, (name, gemm, tables, err)                    # north_star: rendered RGB within 1e-4 in fp32
    # bilinear tap indices of the fine pass (oracle's merged depths): int32, bit-exact, on the kernel under test
    aux = ref[4]
    geo = (d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'])
    sub_t = torch.from_numpy(sub).to(DEV)
    z_all = dev(aux['all_zs'])
    ro, rd = d['rays_o'][:, sub_t].contiguous(), d['rays_d'][:, sub_t].contiguous()
    if _split(d, gemm) is not None:
        _, taps = ops.field_eval_split(ro, rd, z_all, *geo, d['pf'], d['split'][1], return_taps=True)
    else:
        _, taps = ops.field_eval(ro, rd, z_all, *geo, d['pf'], return_taps=True)
    _, _, taps_ref = O.field_eval(O.unflatten_net(sc['fine']), sc['rays_o'][:, sub], sc['rays_d'][:, sub], aux['all_zs'], sc['images'],
                                  sc['features'], sc['intrinsics'], sc['extrinsics_inv'], return_taps=True)
    np.testing.assert_array_equal(taps.cpu().numpy(), taps_ref)


@pytest.mark.parametrize('gemm', GEMMS)
@pytest.mark.parametrize('tables', ['auto', None])
def test_rays_are_independent_and_runs_deterministic(scene, tables, gemm):
    d = scene
    assert d['rays_o'].shape[1] == 4096
    whole = render(d, tables=tables, gemm=gemm)
    again = render(d, tables=tables, gemm=gemm)
    for a, b in zip(whole, again):
        assert torch.equal(a, b)                                            # deterministic: no atomics on the forward path
    # two halves of the rays, and an odd split that leaves a ragged last tile
    for cut in (2048, 1234):
        lo, hi = render(d, slice(0, cut), tables=tables, gemm=gemm), render(d, slice(cut, 4096), tables=tables, gemm=gemm)
        for w_, a, b in zip(whole, lo, hi):
            assert torch.equal(w_, torch.cat([a, b], 1))
    perm = torch.randperm(4096, generator=torch.Generator().manual_seed(0)).to(DEV)
    shuffled = render(d, perm=perm, tables=tables, gemm=gemm)
    for w_, s_ in zip(whole, shuffled):
        assert torch.equal(w_[:, perm], s_)


@pytest.mark.parametrize('gemm', GEMMS)
def test_table_and_direct_paths_agree(scene, gemm):
    d = scene
    a, b = render(d, tables='auto', gemm=gemm), render(d, tables=None, gemm=gemm)
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
