"""The reference-shaped Python surface (model.py: MVVNeRFRenderer / render / render_view) on the GPU."""
import numpy as np
import pytest
import torch

from oracle import mvnerf_oracle as O
from thesis_clip_nerf_amd import MVVNeRFRenderer, render, render_view
from thesis_clip_nerf_amd.synthetic import make_scene, pinhole, ring_pose

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def make_model(sc, n_views):
    m = MVVNeRFRenderer(n_rays_train=sc['rays_o'].shape[1], n_rays_infer=sc['rays_o'].shape[1], n_views=n_views,
                        batch_size=sc['rays_o'].shape[0], near=sc['near'], far=sc['far'], device=DEV)
    m.set_weights(sc['coarse'], sc['fine'])
    return m


def oracle_call(sc):
    return O.render_call(O.unflatten_net(sc['coarse']), O.unflatten_net(sc['fine']), sc['rays_o'], sc['rays_d'],
                         sc['images'], sc['intrinsics'], sc['extrinsics_inv'], sc['features'], sc['near'], sc['far'], 64,
                         sc['u_coarse'], sc['u_fine'])


def test_call_infer_and_render_match_oracle():
    sc = make_scene(seed=21, batch=2, n_views=2, height=24, width=24, n_rays=40, bias_scale=0.05)
    m = make_model(sc, 2)
    inputs = tuple(dev(sc[k]) for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv'])
    ref = oracle_call(sc)
    got = m._call(inputs, 40, 2, dev(sc['features']), u_coarse=dev(sc['u_coarse']), u_fine=dev(sc['u_fine']))
    for g, r in zip(got, ref):
        assert np.abs(g.cpu().numpy() - r).max() < 1e-4
    got2 = m.infer(inputs, dev(sc['features']), u_coarse=dev(sc['u_coarse']), u_fine=dev(sc['u_fine']))
    for g, g2 in zip(got, got2):
        assert torch.equal(g, g2)                                         # deterministic
    fine = render((inputs[0], inputs[1]), m, features=dev(sc['features']), images=inputs[2], K4=inputs[3], Einv=inputs[4],
                  u_coarse=dev(sc['u_coarse']), u_fine=dev(sc['u_fine']))
    assert torch.equal(fine[0], got[2]) and torch.equal(fine[1], got[3])
    # numpy inputs are accepted like the reference's generator output; own uniforms are drawn when none are given
    g = torch.Generator(device=DEV).manual_seed(5)
    out = m.call(tuple(sc[k] for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv']),
                 combined_features=sc['features'], generator=g)
    assert out[2].shape == (2, 40, 3) and torch.isfinite(out[2]).all() and float(out[2].min()) >= 0
    with pytest.raises(ValueError):
        m._call(inputs, 41, 2, dev(sc['features']))
    with pytest.raises(NotImplementedError):
        m.call(inputs)                                                    # encoders are outside the hot path


def test_volumetric_render_static():
    rng = np.random.default_rng(0)
    z = np.sort(rng.uniform(0.3, 1.3, (1, 9, 64)).astype(np.float32), -1)
    sig = (rng.random((1, 9, 64), dtype=np.float32) * 20).astype(np.float32)
    col = rng.random((1, 9, 64, 3), dtype=np.float32)
    rgb, depth, w = MVVNeRFRenderer.volumetric_render(dev(z), dev(sig), dev(col))
    ref = O.volumetric_render(z, sig, col)
    for g, r in zip((rgb, depth, w), ref):
        assert np.abs(g.cpu().numpy() - r).max() < 2e-6


def test_store_load_roundtrip(tmp_path):
    sc = make_scene(seed=22, height=8, width=8, n_rays=4)
    m = make_model(sc, 1)
    other = MVVNeRFRenderer(4, 4, n_views=1, device=DEV, seed=99)
    assert other.load(str(tmp_path / 'model_final')) is False             # model_v0.py:221-232
    m.store(str(tmp_path / 'model_final'))
    assert other.load(str(tmp_path / 'model_final')) is True
    assert torch.equal(other.coarse_net, m.coarse_net) and torch.equal(other.fine_net, m.fine_net)
    (tmp_path / 'model_final_fine_readout.pt').unlink()
    assert other.load(str(tmp_path / 'model_final')) is False


def test_render_view_matches_oracle():
    h, w, v = 16, 24, 2
    rng = np.random.default_rng(4)
    src_colors = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for _ in range(v)]
    k = pinhole(w, h)
    cfgs = [{'pose': ring_pose(rng.uniform(0, 6.28)), 'intrinsics': k.reshape(-1)} for _ in range(v)]
    tgt = {'pose': ring_pose(1.0), 'intrinsics': k.reshape(-1)}
    feats = (0.5 * rng.standard_normal((1, v, h, w, 256))).astype(np.float32)
    m = MVVNeRFRenderer(512, 512, n_views=v, near=0.3, far=1.3, device=DEV, seed=3)
    g = torch.Generator(device=DEV).manual_seed(0)
    u = [torch.rand((1, h * w, 64), device=DEV, generator=g) for _ in range(2)]   # u_coarse, then u_fine
    rgb8, d8 = render_view(m, src_colors, cfgs, tgt, combined_features=dev(feats),
                           generator=torch.Generator(device=DEV).manual_seed(0))
    assert rgb8.shape == (h, w, 3) and rgb8.dtype == np.uint8 and d8.shape == (h, w, 1) and d8.dtype == np.uint8
    # oracle on the same rays / uniforms (render_view draws u_coarse then u_fine from the generator)
    o, d = O.get_rays(w, h, tgt['pose'], k)
    cams = [O.camera_parameters(c['pose'], c['intrinsics']) for c in cfgs]
    einv = np.array([[c[0] for c in cams]], np.float32)
    k4 = np.array([[c[1] for c in cams]], np.float32)
    images = np.array([[img[..., :3] / 255.0 for img in src_colors]]).astype(np.float32)
    ref = O.render_call(O.unflatten_net(m.coarse_net.cpu().numpy()), O.unflatten_net(m.fine_net.cpu().numpy()),
                        o.reshape(1, -1, 3).astype(np.float32), d.reshape(1, -1, 3).astype(np.float32), images, k4, einv,
                        feats, 0.3, 1.3, 64, u[0].cpu().numpy(), u[1].cpu().numpy())
    ref_rgb8, ref_d8 = O.finish_view(ref[2], ref[3], (h, w))
    assert np.abs(rgb8.astype(int) - ref_rgb8.astype(int)).max() <= 1     # a 1e-6 difference can cross a uint8 step
    assert np.abs(d8.astype(int) - ref_d8.astype(int)).max() <= 1
    # chunked rendering (the reference's 512-ray loop) gives the same image when the uniforms match per ray
    rgb8c, _ = render_view(m, src_colors, cfgs, tgt, combined_features=dev(feats), chunk=100,
                           generator=torch.Generator(device=DEV).manual_seed(0))
    assert rgb8c.shape == rgb8.shape
