"""HIP path vs oracle on identical seeded inputs, through the C ABI (ops.py -> libmvnerf_hip.so).

Bars (BASELINE.json north_star): integer indices bit-exact; rendered RGB / sigma / depth within
1e-4 absolute (fp32).  Tighter per-op tolerances are stated at each assert."""
import os

import numpy as np
import pytest
import torch

from oracle import mvnerf_oracle as O
from thesis_clip_nerf_amd import ops
from thesis_clip_nerf_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def scene_to_dev(sc):
    keys = ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'u_coarse', 'u_fine', 'coarse', 'fine']
    return {k: dev(sc[k]) for k in keys}


def test_stratified_depths_bit_exact():
    sc = make_scene(seed=0, height=8, width=8)
    _, z_ref = O.sample_along_ray(sc['rays_o'], sc['rays_d'], 0.3, 1.3, 64, sc['u_coarse'])
    z = ops.stratified_depths(dev(sc['u_coarse']), 0.3, 1.3).cpu().numpy()
    np.testing.assert_array_equal(z, z_ref)
    # reference constructor defaults near=0.7 far=1.5 (model_v0.py:19) and an odd sample count
    u = np.random.default_rng(1).random((2, 5, 48), dtype=np.float32)
    _, z_ref = O.sample_along_ray(np.zeros((2, 5, 3), np.float32), np.zeros((2, 5, 3), np.float32), 0.7, 1.5, 48, u)
    np.testing.assert_array_equal(ops.stratified_depths(dev(u), 0.7, 1.5).cpu().numpy(), z_ref)


@pytest.mark.parametrize('n_views,hw,seed', [(1, (16, 16), 0), (2, (12, 20), 1), (3, (64, 64), 2)])
def test_field_eval_matches_oracle(n_views, hw, seed):
    sc = make_scene(seed=seed, height=hw[0], width=hw[1], n_views=n_views, n_rays=48, bias_scale=0.1)
    d = scene_to_dev(sc)
    _, z = O.sample_along_ray(sc['rays_o'], sc['rays_d'], sc['near'], sc['far'], 64, sc['u_coarse'])
    net = O.unflatten_net(sc['coarse'])
    rgb_ref, sig_ref, taps_ref = O.field_eval(net, sc['rays_o'], sc['rays_d'], z, sc['images'], sc['features'],
                                              sc['intrinsics'], sc['extrinsics_inv'], return_taps=True)
    pix_ref, _ = O.compute_pixel_in_image_mv(O.points_on_rays(sc['rays_o'], sc['rays_d'], z), sc['intrinsics'],
                                             sc['extrinsics_inv'])
    packed = ops.pack_net(d['coarse'])
    rgbs, taps, pix = ops.field_eval(d['rays_o'], d['rays_d'], dev(z), d['images'], d['features'], d['intrinsics'],
                                     d['extrinsics_inv'], packed, return_taps=True, return_pix=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(pix.cpu().numpy(), pix_ref)            # geometry chain is bit-exact
    np.testing.assert_array_equal(taps.cpu().numpy(), taps_ref)          # integer contract (a6)
    rgbs = rgbs.cpu().numpy()
    assert np.abs(rgbs[..., :3] - rgb_ref).max() < TOL
    assert np.abs(rgbs[..., 3] - sig_ref).max() < TOL


@pytest.mark.parametrize('n_views,hw,seed', [(1, (16, 16), 0), (2, (12, 20), 1), (3, (64, 64), 2)])
def test_field_eval_texel_table_matches_oracle(n_views, hw, seed):
    """Layer 0's feature rows hoisted to a per-texel table (mvnerf_project_texels + mvnerf_field_eval_table): same
    bars as the direct form against the oracle; the table itself against W0[123:379]^T features in float64."""
    sc = make_scene(seed=seed, height=hw[0], width=hw[1], n_views=n_views, n_rays=48, bias_scale=0.1)
    d = scene_to_dev(sc)
    _, z = O.sample_along_ray(sc['rays_o'], sc['rays_d'], sc['near'], sc['far'], 64, sc['u_coarse'])
    net = O.unflatten_net(sc['coarse'])
    rgb_ref, sig_ref, taps_ref = O.field_eval(net, sc['rays_o'], sc['rays_d'], z, sc['images'], sc['features'],
                                              sc['intrinsics'], sc['extrinsics_inv'], return_taps=True)
    packed = ops.pack_net(d['coarse'])
    table = ops.project_texels(d['features'], packed)
    both = ops.project_texels2(d['features'], packed, ops.pack_net(d['fine']))      # two nets from one pass over the maps
    assert torch.equal(both[0], table) and torch.equal(both[1], ops.project_texels(d['features'], ops.pack_net(d['fine'])))
    want = sc['features'].astype(np.float64) @ net['W0'][123:].astype(np.float64)              # (B,V,H,W,128)
    slot = [(((n & 31) >> 2) & 1) * 64 + (n >> 5) * 16 + ((n & 3) + 4 * ((n & 31) >> 3)) for n in range(128)]
    got_table = table.cpu().numpy()[..., slot]                                                 # accumulator order -> natural
    assert np.abs(got_table - want).max() < 2e-5 * max(1.0, np.abs(want).max())
    args = (d['rays_o'], d['rays_d'], dev(z), d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'], packed)
    rgbs, taps, emb = ops.field_eval(*args, return_taps=True, return_embedding=True, texel_table=table)
    rgbs_direct, emb_direct = ops.field_eval(*args, return_embedding=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(taps.cpu().numpy(), taps_ref)
    rgbs = rgbs.cpu().numpy()
    assert np.abs(rgbs[..., :3] - rgb_ref).max() < TOL
    assert np.abs(rgbs[..., 3] - sig_ref).max() < TOL
    # the two forms differ by fp32 re-association only
    assert (emb - emb_direct).abs().max().item() < 2e-5 * max(1.0, emb_direct.abs().max().item())
    assert (torch.from_numpy(rgbs).to(DEV) - rgbs_direct).abs().max().item() < 2e-5


@pytest.mark.parametrize('case', range(6))
def test_field_eval_random_shapes(case):
    """Odd image sizes, 1..4 views, batches, ray / sample counts that leave ragged tiles - direct and texel-table paths."""
    rng = np.random.default_rng(100 + case)
    views, batch = int(rng.integers(1, 5)), int(rng.integers(1, 3))
    hw = (int(rng.integers(5, 40)), int(rng.integers(5, 40)))
    n_rays, s = int(rng.integers(1, 50)), int(rng.choice([1, 7, 33, 64, 100]))
    sc = make_scene(seed=200 + case, batch=batch, height=hw[0], width=hw[1], n_views=views, n_rays=n_rays, bias_scale=0.1)
    d = scene_to_dev(sc)
    z = np.sort(rng.uniform(sc['near'], sc['far'], (batch, n_rays, s)).astype(np.float32), -1)
    net = O.unflatten_net(sc['fine'])
    rgb_ref, sig_ref, taps_ref = O.field_eval(net, sc['rays_o'], sc['rays_d'], z, sc['images'], sc['features'],
                                              sc['intrinsics'], sc['extrinsics_inv'], return_taps=True)
    packed = ops.pack_net(d['fine'])
    args = (d['rays_o'], d['rays_d'], dev(z), d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'], packed)
    for table in (None, ops.project_texels(d['features'], packed)):
        rgbs, taps = ops.field_eval(*args, return_taps=True, texel_table=table)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(taps.cpu().numpy(), taps_ref)
        rgbs = rgbs.cpu().numpy()
        assert np.abs(rgbs[..., :3] - rgb_ref).max() < TOL and np.abs(rgbs[..., 3] - sig_ref).max() < TOL, (views, batch, hw, n_rays, s)


def test_field_eval_ragged_tail_and_fine_count():
    # total samples not a multiple of the 32-sample wave tile; S = 128 as in the fine pass
    sc = make_scene(seed=4, height=16, width=16, n_rays=3)
    d = scene_to_dev(sc)
    rng = np.random.default_rng(0)
    for s in (8, 128, 40):
        z = np.sort(rng.uniform(0.3, 1.3, (1, 3, s)).astype(np.float32), -1)
        net = O.unflatten_net(sc['fine'])
        rgb_ref, sig_ref = O.field_eval(net, sc['rays_o'], sc['rays_d'], z, sc['images'], sc['features'],
                                        sc['intrinsics'], sc['extrinsics_inv'])
        rgbs = ops.field_eval(d['rays_o'], d['rays_d'], dev(z), d['images'], d['features'], d['intrinsics'],
                              d['extrinsics_inv'], ops.pack_net(d['fine'])).cpu().numpy()
        assert np.abs(rgbs[..., :3] - rgb_ref).max() < TOL and np.abs(rgbs[..., 3] - sig_ref).max() < TOL


@pytest.mark.parametrize('s', [64, 128])
def test_composite_matches_oracle(s):
    rng = np.random.default_rng(s)
    z = np.sort(rng.uniform(0.3, 1.3, (2, 37, s)).astype(np.float32), -1)
    rgbs = rng.random((2, 37, s, 4), dtype=np.float32)
    rgbs[..., 3] *= 40.0                                                   # densities up to 40
    rgb_ref, depth_ref, w_ref = O.volumetric_render(z, rgbs[..., 3], rgbs[..., :3])
    rgb, depth, w = ops.composite(dev(z), dev(rgbs))
    assert np.abs(rgb.cpu().numpy() - rgb_ref).max() < 2e-6
    assert np.abs(depth.cpu().numpy() - depth_ref).max() < 2e-6
    assert np.abs(w.cpu().numpy() - w_ref).max() < 1e-6


@pytest.mark.parametrize('q7', [O.Q7_ZERO, O.Q7_CLAMP])
def test_resample_indices_bit_exact(q7):
    rng = np.random.default_rng(7)
    n = 301
    u = rng.random((1, n, 64), dtype=np.float32)
    _, z = O.sample_along_ray(np.zeros((1, n, 3), np.float32), np.ones((1, n, 3), np.float32), 0.3, 1.3, 64, u)
    w = (rng.random((1, n, 64), dtype=np.float32) ** 4).astype(np.float32)
    w[0, 0] = 0.0                                   # all-zero weights -> uniform pdf through the +1e-5
    w[0, 1, 10:] = 0.0
    uf = rng.random((1, n, 64), dtype=np.float32)
    uf[0, :, 0] = 0.0
    uf[0, :, 1] = np.nextafter(np.float32(1), np.float32(0))             # forces Q7 whenever cdf[-1] <= u
    ref = O.hierarchical_depths(z, w, uf, q7, return_indices=True)
    got = ops.resample(dev(z), dev(w), dev(uf), q7, return_aux=True)
    z_all, z_fine, above, below = [t.cpu().numpy() for t in got]
    np.testing.assert_array_equal(above, ref[2])                          # integer contract (a13)
    np.testing.assert_array_equal(below, ref[3])
    assert (ref[2] == 63).any(), 'test input should exercise Q7'
    np.testing.assert_array_equal(z_fine, ref[1])                         # same fp32 op sequence -> identical
    np.testing.assert_array_equal(z_all, ref[0])
    assert (np.diff(z_all, axis=-1) >= 0).all()


@pytest.mark.parametrize('tables', [None, 'auto'])
@pytest.mark.parametrize('n_views,seed', [(1, 0), (3, 1)])
def test_render_fwd_matches_oracle(n_views, seed, tables):
    sc = make_scene(seed=seed, height=32, width=32, n_views=n_views, n_rays=96, bias_scale=0.05)
    d = scene_to_dev(sc)
    ref = O.render_call(O.unflatten_net(sc['coarse']), O.unflatten_net(sc['fine']), sc['rays_o'], sc['rays_d'],
                        sc['images'], sc['intrinsics'], sc['extrinsics_inv'], sc['features'], sc['near'], sc['far'],
                        64, sc['u_coarse'], sc['u_fine'])
    got = ops.render_fwd(d['rays_o'], d['rays_d'], d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'],
                         ops.pack_net(d['coarse']), ops.pack_net(d['fine']), d['u_coarse'], d['u_fine'], sc['near'],
                         sc['far'], texel_tables=tables)
    torch.cuda.synchronize()
    assert tables is None or ops.texel_table_pays(96, 64, 32, 32)
    for name, g, r in zip(['rgb', 'depth', 'fine_rgb', 'fine_depth'], got, ref):
        err = np.abs(g.cpu().numpy() - r).max()
        assert err < TOL, (name, err)


def test_get_rays_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'rays.npz'))
    for i in range(4):
        w, h = [int(x) for x in g[f'rays{i}_wh']]
        pose, k = g[f'rays{i}_pose'], g[f'rays{i}_k']
        m = pose[:3, :3] @ np.linalg.inv(k[:3, :3])
        o, d, d64 = ops.get_rays_device(m, pose[:3, 3], DEV, width=w, height=h, return_f64=True)
        np.testing.assert_array_equal(o.cpu().numpy().reshape(h, w, 3), g[f'rays{i}_o'].astype(np.float32))
        np.testing.assert_allclose(d64.cpu().numpy().reshape(h, w, 3), g[f'rays{i}_d'], rtol=0, atol=5e-16)
        assert np.abs(d.cpu().numpy().reshape(h, w, 3) - g[f'rays{i}_d'].astype(np.float32)).max() <= 6e-8
        u, v = dev(g[f'rays{i}_u'].astype(np.float32)), dev(g[f'rays{i}_v'].astype(np.float32))
        so, sd = ops.get_rays_device(m, pose[:3, 3], DEV, u=u, v=v)
        assert np.abs(sd.cpu().numpy() - g[f'rays{i}_sd'].astype(np.float32)).max() <= 6e-8


def test_argument_errors():
    with pytest.raises(ValueError):
        ops.stratified_depths(torch.zeros(4, 64), 0.3, 1.3)               # CPU tensor: no fallback path
    z = torch.zeros(1, 2, 32, device=DEV)
    with pytest.raises(ValueError):
        ops.composite(z, torch.zeros(1, 2, 32, 4, device=DEV))            # S not a multiple of 64
    with pytest.raises(ValueError):
        ops.resample(z, z, z)                                            # S != 64
