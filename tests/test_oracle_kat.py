"""Closed-form known-answer tests for the oracle.  The TensorFlow half of the
reference cannot run here and ships no fixtures, so these are the only independent pins for
a4-a14: each expected value below is derived by hand from the cited reference lines."""
import numpy as np

from oracle import mvnerf_oracle as O
from thesis_clip_nerf_amd.synthetic import make_scene, glorot_net

F32 = np.float32


def test_volumetric_render_constant_sigma():
    # model_v0.py:89-100 with sigma const, uniform delta: w_i = (1-e^{-sd}) e^{-sd i}
    s, sig, delta = 64, F32(3.0), F32(1.0 / 64)
    z = (F32(0.3) + delta * np.arange(s, dtype=F32))[None, None]
    rgb, depth, w = O.volumetric_render(z, np.full((1, 1, s), sig, F32), np.full((1, 1, s, 3), 0.5, F32))
    a = 1 - np.exp(-float(sig) * float(delta))
    expect = a * (1 - a) ** np.arange(s)
    np.testing.assert_allclose(w[0, 0], expect, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(w.sum(), 1 - np.exp(-float(sig) * float(delta) * s), rtol=1e-5)
    np.testing.assert_allclose(rgb[0, 0], 0.5 * w.sum(), rtol=1e-5)
    np.testing.assert_allclose(depth[0, 0], (expect * z[0, 0]).sum(), rtol=1e-5)


def test_volumetric_render_last_delta_duplicated():
    # Q6: the last interval reuses the previous one (no 1e10 far cap)
    z = np.array([[[0.0, 0.1, 0.4]]], F32)
    _, _, w = O.volumetric_render(z, np.ones((1, 1, 3), F32), np.zeros((1, 1, 3, 3), F32))
    a = 1 - np.exp(-np.array([0.1, 0.3, 0.3]))
    np.testing.assert_allclose(w[0, 0], a * np.array([1, 1 - a[0], (1 - a[0]) * (1 - a[1])]), rtol=1e-5)


def test_sample_pdf_uniform_weights_is_linear():
    # nerf_utils.py:143-176: equal weights -> linear CDF -> z = bins[0] + u (bins[-1]-bins[0])
    bins = np.linspace(0.3, 1.3, 63, dtype=F32)[None, None]
    w = np.ones((1, 1, 62), F32)
    u = np.linspace(0, 0.999, 64, dtype=F32)[None, None]
    z, above, below = O.sample_pdf(bins, w, u, return_indices=True)
    np.testing.assert_allclose(z, bins[..., :1] + u * (bins[..., -1:] - bins[..., :1]), atol=2e-6)
    assert above.min() >= 1 and above.max() <= 62
    np.testing.assert_array_equal(below, above - 1)


def test_sample_pdf_one_hot_weight():
    bins = np.linspace(0.0, 6.2, 63, dtype=F32)[None, None]
    w = np.zeros((1, 1, 62), F32)
    w[..., 17] = 1.0
    u = (np.linspace(0.01, 0.99, 64, dtype=F32))[None, None]
    z = O.sample_pdf(bins, w, u)
    assert (z >= bins[0, 0, 17] - 1e-6).all() and (z <= bins[0, 0, 18] + 1e-6).all()


def test_sample_pdf_q7_out_of_range():
    # Q7: u >= cdf[-1] gives above == Nb (one past the end).  TF-GPU gather -> 0; clamp -> last bin.
    bins = np.linspace(0.3, 1.3, 63, dtype=F32)[None, None]
    w = np.random.default_rng(0).random((1, 1, 62)).astype(F32)
    # find the cdf the oracle builds and place u just at/above its last entry
    stable = w + F32(1e-5)
    pdf = stable / np.cumsum(stable, -1, dtype=F32)[..., -1:]
    last = np.cumsum(pdf, -1, dtype=F32)[0, 0, -1]
    u = np.array([[[min(float(last), float(np.nextafter(F32(1), F32(0))))]]], F32)
    if u[0, 0, 0] >= last:
        z0, a0, b0 = O.sample_pdf(bins, w, u, O.Q7_ZERO, True)
        z1, a1, b1 = O.sample_pdf(bins, w, u, O.Q7_CLAMP, True)
        assert a0[0, 0, 0] == 63 and b0[0, 0, 0] == 62 and a1[0, 0, 0] == 63
        # zero mode: cdf_a = bins_a = 0 -> den = -cdf_b < 1e-5 -> 1 ; t = u - cdf_b ; z = b + t*(0-b)
        t = u[0, 0, 0] - last
        np.testing.assert_allclose(z0[0, 0, 0], bins[0, 0, 62] * (1 - t), rtol=1e-6)
        np.testing.assert_allclose(z1[0, 0, 0], bins[0, 0, 62], rtol=1e-6)


def test_position_encoding_layout_and_values():
    # nerf_utils.py:108-126: index = d*20 + k*2 + {0:sin,1:cos}; no identity term
    x = np.zeros((1, 1, 1, 3), F32)
    pe = O.position_encoding(x)
    assert pe.shape == (1, 1, 1, 60)
    np.testing.assert_array_equal(pe[0, 0, 0, 0::2], 0.0)
    np.testing.assert_array_equal(pe[0, 0, 0, 1::2], 1.0)
    x = np.array([[[[0.5, 0.0, 0.25]]]], F32)
    pe = O.position_encoding(x)[0, 0, 0]
    np.testing.assert_allclose(pe[0], 1.0, atol=1e-6)          # sin(pi/2)
    np.testing.assert_allclose(pe[1], 0.0, atol=1e-6)          # cos(pi/2)
    np.testing.assert_allclose(pe[3], -1.0, atol=1e-6)         # cos(pi)
    np.testing.assert_allclose(pe[40:44], [np.sin(np.pi / 4), np.cos(np.pi / 4), 1.0, 0.0], atol=1e-6)
    # the fp32 product is rounded BEFORE sin: compare with float64 sin of that rounded product
    x = np.array([[[[1.2345678]]]], F32)
    pe = O.position_encoding(x)[0, 0, 0]
    for k in range(10):
        arg = np.float64(F32(x[0, 0, 0, 0]) * (F32(np.pi) * F32(2.0 ** k)))
        np.testing.assert_allclose(pe[2 * k], np.sin(arg), atol=3e-7)
        np.testing.assert_allclose(pe[2 * k + 1], np.cos(arg), atol=3e-7)


def test_bilinear_integer_half_and_border():
    rng = np.random.default_rng(0)
    grid = rng.standard_normal((1, 5, 7, 3)).astype(F32)
    q = np.array([[[2.0, 3.0], [2.5, 1.5], [-4.0, 2.0], [100.0, 100.0], [6.0, 4.0]]], F32)   # (x,y)
    out = O.interpolate_bilinear_xy(grid, q)[0]
    np.testing.assert_allclose(out[0], grid[0, 3, 2], atol=1e-6)
    np.testing.assert_allclose(out[1], grid[0, 1:3, 2:4].mean((0, 1)), atol=1e-6)
    np.testing.assert_allclose(out[2], grid[0, 2, 0], atol=1e-6)           # Q5 clamps to edge texel
    np.testing.assert_allclose(out[3], grid[0, 4, 6], atol=1e-6)
    np.testing.assert_allclose(out[4], grid[0, 4, 6], atol=1e-6)
    x0, y0, ax, ay = O.bilinear_taps(q, 5, 7)
    assert x0.max() <= 5 and y0.max() <= 3 and x0.min() >= 0 and y0.min() >= 0
    np.testing.assert_array_equal(ax[0, 3:], 1.0)


def test_projection_round_trip():
    # a2 + a5: project o + z d from the SAME camera -> pixel (u, v) (checks K/E conventions, Q1)
    sc = make_scene(seed=3, height=8, width=12)
    o, d = O.get_rays(12, 8, sc['tgt_pose'][0], sc['tgt_intrinsics'])
    o = o.reshape(1, -1, 3).astype(F32)
    d = d.reshape(1, -1, 3).astype(F32)
    z = np.full((1, 96, 4), 0.7, F32)
    einv, k4 = O.camera_parameters(sc['tgt_pose'][0], sc['tgt_intrinsics'])
    pix, cam = O.compute_pixel_in_image_mv(O.points_on_rays(o, d, z), k4[None, None].astype(F32),
                                           einv[None, None].astype(F32))
    uu, vv = np.meshgrid(np.arange(12), np.arange(8), indexing='xy')
    np.testing.assert_allclose(pix[0, 0, :, 0, 0], uu.reshape(-1), atol=2e-4)
    np.testing.assert_allclose(pix[0, 0, :, 0, 1], vv.reshape(-1), atol=2e-4)
    np.testing.assert_allclose(cam[..., 3], 1.0)


def test_direction_gets_translation_q3():
    sc = make_scene(seed=1, height=4, width=4)
    cd = O.world_to_camera_direction_vector_mv(sc['rays_d'], sc['extrinsics_inv'])
    e = sc['extrinsics_inv'][0, 0].astype(np.float64)
    expect = sc['rays_d'][0].astype(np.float64) @ e[:3, :3].T + e[:3, 3]
    np.testing.assert_allclose(cd[0, 0], expect, atol=1e-6)


def test_stratified_edges():
    o = np.zeros((1, 2, 3), F32)
    d = np.ones((1, 2, 3), F32)
    _, z = O.sample_along_ray(o, d, 0.3, 1.3, 64, np.zeros((1, 2, 64), F32))
    np.testing.assert_array_equal(z[0, 0], np.array([0.3 + i * (1.0 / 64) for i in range(64)], F32))
    _, z1 = O.sample_along_ray(o, d, 0.3, 1.3, 64, np.full((1, 2, 64), np.nextafter(F32(1), F32(0)), F32))
    assert (z1[0, 0, :-1] <= z[0, 0, 1:] + 1e-7).all() and (z1 > z).all()


def test_embedding_bias_passthrough_and_view_mean():
    rng = np.random.default_rng(0)
    flat = np.zeros(O.NET_PARAMS, F32)
    net = O.unflatten_net(flat)
    net['b0'][:] = rng.standard_normal(128)
    xyz = rng.standard_normal((3, 2, 4, 3)).astype(F32)
    feat = rng.standard_normal((3, 2, 4, 259)).astype(F32)
    out = O.mv_embedding(net, xyz, xyz, feat, n_views=3)
    assert out.shape == (1, 2, 4, 128)
    np.testing.assert_allclose(out, np.broadcast_to(net['b0'], out.shape), atol=1e-6)
    # selector W0: feature channel c -> hidden c ; mean over 3 views of the gathered features
    net['b0'][:] = 0
    net['W0'][123 + np.arange(128), np.arange(128)] = 1.0
    out = O.mv_embedding(net, xyz, xyz, feat, n_views=3)
    np.testing.assert_allclose(out[0], feat[..., 3:131].mean(0), atol=1e-6)
    outs = O.mv_embedding(net, xyz, xyz, feat, n_views=3, complete_output=True)
    assert len(outs) == 8 and outs[0].shape[0] == 3 and outs[4].shape[0] == 1    # layers.py:364-377


def test_readout_activations():
    net = O.unflatten_net(np.zeros(O.NET_PARAMS, F32))
    net['br'][:] = [0.0, 2.0, -2.0, 1.0]
    rgb, sig = O.render_readout(net, np.zeros((1, 1, 1, 128), F32))
    np.testing.assert_allclose(rgb[0, 0, 0], 1 / (1 + np.exp(-np.array([0.0, 2.0, -2.0]))), atol=1e-6)
    np.testing.assert_allclose(sig[0, 0, 0], np.log1p(np.exp(1.0)), atol=1e-6)


def test_hierarchical_depths_sorted_permutation():
    sc = make_scene(seed=2, height=4, width=4)
    _, z = O.sample_along_ray(sc['rays_o'], sc['rays_d'], 0.3, 1.3, 64, sc['u_coarse'])
    w = np.random.default_rng(5).random(z.shape).astype(F32)
    all_zs, z_fine, above, below = O.hierarchical_depths(z, w, sc['u_fine'], return_indices=True)
    assert all_zs.shape[-1] == 128 and (np.diff(all_zs, axis=-1) >= 0).all()
    np.testing.assert_array_equal(np.sort(np.concatenate([z, z_fine], -1), -1), all_zs)


def test_net_param_count():
    assert O.NET_PARAMS == 247300 and glorot_net(np.random.default_rng(0)).size == 247300
