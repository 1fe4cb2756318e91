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


def test_train_step_trains_the_feature_producer_end_to_end():
    """the reference's encoder prologue (encoders.FeatureProducer = VisualFeatures + CombineCLIPVisualV0 shape) in
    front of the HIP render path; `train_step` hands dL/d(combined_features) back to autograd.  The weight gradients of the
    conv encoder and of the ViT side equal autograd through the float64 twin of the whole chain (producer in float64 on the
    CPU -> oracle/mvnerf_torch.render_call), and the producer's own optimizer (Adam, 1e-5 warm-up group) moves its weights."""
    import copy
    from oracle import mvnerf_torch as T
    from thesis_clip_nerf_amd import encoders as E
    from thesis_clip_nerf_amd.synthetic import make_scene
    tiny = dict(transformer_image_size=(32, 32), patch_size=16, embed_dim=32, num_heads=4, hooks=(1, 2, 3, 4), features=(4, 8, 16, 32))
    sc = make_scene(seed=85, batch=1, n_views=2, height=16, width=24, n_rays=32, bias_scale=0.05)
    y = np.random.default_rng(4).random((1, 32, 3)).astype(np.float32)
    torch.manual_seed(0)
    prod = E.FeatureProducer(original_image_size=(16, 24), **tiny).to(DEV)
    twin = copy.deepcopy(prod).double().cpu()
    m = MVVNeRFRenderer(32, 32, n_views=2, batch_size=1, near=sc['near'], far=sc['far'], device=DEV, feature_encoder=prod)
    m.set_weights(sc['coarse'], sc['fine'])
    opt, sched = E.make_encoder_optimizer(prod, target_lr=1e-3, warmup_steps=1)
    m.compile(learning_rate=0.0, encoder_optimizer=opt)
    inputs = tuple(sc[k] for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv'])
    before = [p.detach().clone() for p in prod.trainable_parameters()]
    m.train_step((inputs, y), u_coarse=dev(sc['u_coarse']), u_fine=dev(sc['u_fine']), stop_fine_z=True)      # lr(0) = 0: gradients only
    torch.cuda.synchronize()
    got = {n: p.grad.double().cpu().clone() for n, p in prod.named_parameters() if p.grad is not None}
    # float64 twin of the whole chain
    t = lambda k: torch.as_tensor(sc[k]).double()
    imgs = t('images')
    feats = twin(imgs.reshape(2, 16, 24, 3)).reshape(1, 2, 16, 24, 256)
    out = T.render_call(t('coarse'), t('fine'), t('rays_o'), t('rays_d'), imgs, t('intrinsics'), t('extrinsics_inv'), feats,
                        sc['near'], sc['far'], sc['n_samples'], t('u_coarse'), t('u_fine'), stop_fine_z=True)
    yy = torch.as_tensor(y).double()
    (((yy - out[0]) ** 2).mean() + ((yy - out[2]) ** 2).mean()).backward()
    checked = 0
    for n, p in twin.named_parameters():
        if p.grad is None or n not in got or p.grad.norm() == 0:
            continue
        ref = p.grad.clamp(-1.0, 1.0)                                  # train_step clips by value before the optimizer step
        if ref.norm() < 1e-7:                                          # e.g. a conv bias in front of a batch-statistics BatchNorm:
            assert got[n].norm() < 1e-5, (n, float(got[n].norm()))     # its gradient is exactly zero, fp32 leaves rounding noise
            continue
        rel = float((got[n] - ref).norm() / ref.norm())
        assert rel < 2e-2, (n, rel)                                    # fp32 producer + fp32 HIP backward vs float64
        checked += 1
    assert checked >= 20
    assert all(torch.equal(a, b) for a, b in zip(before, prod.trainable_parameters()))       # lr was 0 at iteration 0 ...
    sched.step()
    m.train_step((inputs, y), u_coarse=dev(sc['u_coarse']), u_fine=dev(sc['u_fine']), stop_fine_z=True)
    assert any(not torch.equal(a, b) for a, b in zip(before, prod.trainable_parameters()))  # ... and is not after the warm-up step
    assert torch.equal(prod.combine_clip_visual.conv.weight, twin.combine_clip_visual.conv.weight.float().to(DEV))  # not in the optimizer list (Q9)


def test_device_resident_generator_matches_reference_goldens(golden_dir):
    """a3 caller side on the GPU: pixel indices from the host RNG, rays from mvnerf_get_rays, targets by a device
    gather - against what the reference's own generate_rays / get_target produced (tests/golden/datagen.npz)."""
    import os
    from thesis_clip_nerf_amd.train_nerf import MVNeRFDataGenerator

    class _OneView:
        n_perspectives = 1

        def __len__(self):
            return 1
    g = np.load(os.path.join(golden_dir, 'datagen.npz'))
    for i in range(3):
        gen = MVNeRFDataGenerator(_OneView(), n_rays_train=int(g[f'case{i}_n']), shuffle=False, device=DEV)
        cam = {'pose': g[f'case{i}_pose'], 'intrinsics': g[f'case{i}_intrinsics']}
        np.random.seed(int(g[f'case{i}_seed']))
        r_d, r_o, px = gen.generate_rays_device(g[f'case{i}_color'], cam)
        np.testing.assert_array_equal(px.cpu().numpy(), g[f'case{i}_rays'])                       # integer (row, col): bit-exact
        want_d = g[f'case{i}_r_d'].astype(np.float32)                                            # float64 math, float32 storage (Q2)
        assert np.abs(r_d.cpu().numpy() - want_d).max() <= 6e-8                                  # <= 1 ulp of a unit vector component
        np.testing.assert_array_equal(r_o.cpu().numpy(), g[f'case{i}_r_o'].astype(np.float32))
        tgt = gen.get_target_device(dev(g[f'case{i}_color']), px)
        np.testing.assert_array_equal(tgt.cpu().numpy(), g[f'case{i}_target'].astype(np.float32))     # float64 division, float32 storage
