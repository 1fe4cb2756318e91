"""BASELINE.json configs 3, 4 and 5 at their full workload sizes, through the C ABI on the GPU:

* cfg5: hierarchical 64 + 128 samples, V = 3 source views of 480x640 (944 MB of fp32 feature maps, texel tables 2 x 472 MB),
  16 384 random rays, fp32 and bf16;
* cfg3: the trunk as a field on B = 8 scenes x 8 064 query points (192 poses x 42 offsets), V = 1, 480x640 (2.5 GB of feature
  maps: tap byte offsets need 64 bits), bf16 fused activations vs the fp32 kernel, query_vjp / query_jvp;
* cfg4: one scene of 128x128 = 16 384 rays, one complete train step (forward with stash, backward, clip, Adam).

At these sizes the oracle cannot run the whole workload in seconds, so each test checks (i) a strided subset of rays /
points against the oracle (NumPy fp32 port or the float64 torch twin) at the usual bars, (ii) integer tap indices
bit-exact on that subset, and (iii) size-independent properties on the whole workload: ray independence (chunk
invariance), determinism, table-vs-direct agreement, the transpose identity <g, J t> = <J^T g, t>, gradient additivity
over ray halves.  The big feature maps are drawn on the device (torch.randn, seeded) and copied to the host for the
oracle; everything else comes from synthetic.make_scene."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import mvnerf_oracle as O
from oracle import mvnerf_torch as T
from thesis_clip_nerf_amd import MVVNeRFRenderer, _lib, ops
from thesis_clip_nerf_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
KEYS = ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv', 'u_coarse', 'u_fine', 'coarse', 'fine']


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def big_scene(seed, batch, n_views, height, width, n_rays, bias_scale=0.0):
    """make_scene without host feature maps + N(0, 0.5^2) features drawn on the device (seeded); `features` on the host is
    filled lazily by host_features() only where an oracle needs it."""
    sc = make_scene(seed=seed, batch=batch, n_views=n_views, height=height, width=width, n_rays=n_rays, bias_scale=bias_scale,
                    with_features=False)
    d = {k: dev(sc[k]) for k in KEYS}
    g = torch.Generator(device=DEV).manual_seed(seed)
    d['features'] = torch.randn((batch, n_views, height, width, 256), dtype=torch.float32, device=DEV, generator=g).mul_(0.5)
    d['near'], d['far'] = sc['near'], sc['far']
    return sc, d


# ---------------------------------------------------------------------------------------------------------------
# cfg5: hierarchical sampling, V = 3, 480 x 640 sources, 16 384 rays, fp32 and bf16
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def cfg5():
    sc, d = big_scene(seed=21, batch=1, n_views=3, height=480, width=640, n_rays=16384)
    d['pc'], d['pf'] = ops.pack_net(d['coarse']), ops.pack_net(d['fine'])
    d['split'] = (ops.pack_net_split(d['coarse']), ops.pack_net_split(d['fine']))
    d['pc16'], d['pf16'] = ops.pack_net_bf16(d['coarse']), ops.pack_net_bf16(d['fine'])
    sub = np.arange(0, 16384, 64)                                           # 256 rays, strided over the whole batch
    sc['features'] = d['features'].cpu().numpy()
    cn, fn = O.unflatten_net(sc['coarse']), O.unflatten_net(sc['fine'])
    ref = O.render_call(cn, fn, sc['rays_o'][:, sub], sc['rays_d'][:, sub], sc['images'], sc['intrinsics'], sc['extrinsics_inv'],
                        sc['features'], sc['near'], sc['far'], 64, sc['u_coarse'][:, sub], sc['u_fine'][:, sub], return_aux=True)
    yield sc, d, sub, ref
    del sc['features']


GEMMS = ['split_f16', 'split_bf16', 'mfma_f32']      # the three fp32 field kernels (MVVNeRFRenderer(f32_gemm=...)); the default, which bench.py times, first


def _split(d, gemm):
    """The packed split image for the two split kernels (selecting which of them runs), None for the fp32-MFMA kernel."""
    if gemm == 'mfma_f32':
        return None
    ops.set_split_kernel(gemm)
    return d['split']


def _render(d, sl=slice(None), tables='auto', gemm='split_f16'):
    pick = lambda t: t[:, sl].contiguous()
    return ops.render_fwd(pick(d['rays_o']), pick(d['rays_d']), d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'],
                          d['pc'], d['pf'], pick(d['u_coarse']), pick(d['u_fine']), d['near'], d['far'], texel_tables=tables,
                          split=_split(d, gemm))


@pytest.mark.parametrize('gemm', GEMMS)
def test_cfg5_fp32_render_matches_oracle_and_is_ray_independent(cfg5, gemm):
    sc, d, sub, ref = cfg5
    assert d['features'].numel() * 4 > 900e6                                # 944 MB of feature maps, 2 x 472 MB of texel tables
    assert ops.texel_table_pays(16384, 64, 480, 640)
    _render_g = lambda d_, sl=slice(None), tables='auto': _render(d_, sl, tables, gemm)
    whole = _render_g(d, tables='auto')
    direct = _render_g(d, tables=None)
    torch.cuda.synchronize()
    names = ['rgb', 'depth', 'fine_rgb', 'fine_depth']
    for name, got_t, got_d, want in zip(names, whole, direct, ref[:4]):
        for tag, got in (('table', got_t), ('direct', got_d)):
            err = float(np.abs(got[:, sub].cpu().numpy() - want).max())
            print(f'cfg5 fp32 {gemm} {tag} {name}: max|hip - oracle| over 256 rays = {err:.2e}')
            assert err < 1e-4, (name, gemm, tag, err)                              # north_star: rendered RGB within 1e-4 in fp32
        assert (got_t - got_d).abs().max().item() < 2e-5, name              # fp32 re-association of layer 0 only
    # determinism and ray independence (the cut leaves a ragged 32-sample tile), on the direct path and on the table path - the
    # latter with the tables forced, since 'auto' would build none for the smaller part (R*S < 2*H*W there)
    forced = torch.empty((2, 1, 3, 480, 640, 128), dtype=torch.float32, device=DEV)
    for tables, ref_out in ((None, direct), (forced, whole)):
        again = _render_g(d, tables=tables)
        lo, hi = _render_g(d, slice(0, 5001), tables), _render_g(d, slice(5001, 16384), tables)
        for w_, a_, l_, h_ in zip(ref_out, again, lo, hi):
            assert torch.equal(w_, a_)                                      # deterministic
            assert torch.equal(w_, torch.cat([l_, h_], 1))                  # rays are independent units


@pytest.mark.parametrize('gemm', GEMMS)
def test_cfg5_tap_indices_and_sample_indices_bit_exact(cfg5, gemm):
    sc, d, sub, ref = cfg5
    aux = ref[4]
    sub_t = torch.from_numpy(sub).to(DEV)
    z = ops.stratified_depths(d['u_coarse'], d['near'], d['far'])
    np.testing.assert_array_equal(z[:, sub_t].cpu().numpy(), aux['z'])      # fp32, identical bits
    geo = (d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'])
    if _split(d, gemm) is not None:
        rgbs_c, taps_c = ops.field_eval_split(d['rays_o'], d['rays_d'], z, *geo, d['pc'], d['split'][0], return_taps=True)
    else:
        rgbs_c, taps_c = ops.field_eval(d['rays_o'], d['rays_d'], z, *geo, d['pc'], return_taps=True)
    # per-sample (rgb, sigma) of the coarse pass on the strided rays against the oracle's (direct gather)
    got_rgbs = rgbs_c[:, sub_t].cpu().numpy()
    assert np.abs(got_rgbs[..., :3] - aux['coarse_rgb']).max() < 2e-5
    assert np.abs(got_rgbs[..., 3] - aux['coarse_sigma']).max() < 5e-5
    net = O.unflatten_net(sc['coarse'])
    _, _, taps_ref = O.field_eval(net, sc['rays_o'][:, sub], sc['rays_d'][:, sub], aux['z'], sc['images'], sc['features'],
                                  sc['intrinsics'], sc['extrinsics_inv'], return_taps=True)
    got = taps_c[:, :, sub_t].cpu().numpy()
    np.testing.assert_array_equal(got, taps_ref)                            # int32 linear texel indices (a6)
    assert got.max() > 2 * 480 * 640 and got.min() >= 0                     # all three views are addressed
    # importance sampling on the oracle's coarse weights: integer above/below and the merged depths
    w_ref = np.zeros((1, 16384, 64), np.float32)
    w_ref[:, sub] = aux['weights']
    z_all, z_fine, above, below = ops.resample(z, dev(w_ref), d['u_fine'], return_aux=True)
    np.testing.assert_array_equal(above[:, sub_t].cpu().numpy(), aux['above'])
    np.testing.assert_array_equal(below[:, sub_t].cpu().numpy(), aux['below'])
    np.testing.assert_array_equal(z_all[:, sub_t].cpu().numpy(), aux['all_zs'])
    assert (z_all[..., 1:] >= z_all[..., :-1]).all()


def test_cfg5_bf16_render_and_field(cfg5):
    sc, d, sub, ref = cfg5
    args = (d['rays_o'], d['rays_d'], d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'], d['pc'], d['pf'],
            d['pc16'], d['pf16'], d['u_coarse'], d['u_fine'], d['near'], d['far'])
    got = ops.render_fwd_bf16(*args)
    again = ops.render_fwd_bf16(*args)
    f32 = _render(d, tables='auto')
    torch.cuda.synchronize()
    for name, g, a, f, want in zip(['rgb', 'depth', 'fine_rgb', 'fine_depth'], got, again, f32, ref[:4]):
        assert torch.equal(g, a), name
        e_oracle = float(np.abs(g[:, sub].cpu().numpy() - want).max())
        e_f32 = (g - f).abs().max().item()
        print(f'cfg5 bf16 {name}: max|bf16 - fp32 oracle| (256 rays) = {e_oracle:.2e}, max|bf16 - fp32 kernel| (16384 rays) = {e_f32:.2e}')
        assert e_oracle < 3e-2 and e_f32 < 3e-2, (name, e_oracle, e_f32)   # the stated bf16 bound
    # per-sample field outputs on the subset against the oracle restating the bf16 arithmetic, taps bit-exact
    aux = ref[4]
    sub_t = torch.from_numpy(sub).to(DEV)
    z_all = dev(aux['all_zs'])
    net = O.unflatten_net(sc['fine'])
    oargs = (net, sc['rays_o'][:, sub], sc['rays_d'][:, sub], aux['all_zs'], sc['images'], sc['features'], sc['intrinsics'], sc['extrinsics_inv'])
    rgb_e, sig_e, taps_ref = O.field_eval(*oargs, return_taps=True, emulate_bf16=True)
    tab_f = ops.project_texels_bf16(d['features'], d['pf16'])
    for table in (None, tab_f):
        rgbs, taps = ops.field_eval_bf16(d['rays_o'][:, sub_t].contiguous(), d['rays_d'][:, sub_t].contiguous(), z_all, d['images'],
                                         d['features'], d['intrinsics'], d['extrinsics_inv'], d['pf'], d['pf16'], return_taps=True,
                                         texel_table=table)
        np.testing.assert_array_equal(taps.cpu().numpy(), taps_ref)
        rgbs = rgbs.cpu().numpy()
        ref_e = np.concatenate([rgb_e, sig_e[..., None]], -1)
        ref_f = np.concatenate([aux['fine_rgbs'], aux['fine_sigma'][..., None]], -1)
        mean_e, mean_f = np.abs(rgbs - ref_e).mean(), np.abs(rgbs - ref_f).mean()
        print(f'cfg5 bf16 field ({"table" if table is not None else "direct"}): mean|d| vs bf16-restated oracle {mean_e:.2e}, vs fp32 oracle {mean_f:.2e}, '
              f'max vs fp32 {np.abs(rgbs - ref_f).max():.2e}')
        assert np.abs(rgbs - ref_f).max() < 5e-2
        if table is None:                     # the restated oracle shares every rounding of the direct form
            assert mean_e < 0.35 * mean_f and np.abs(rgbs - ref_e).max() < 2e-2


# ---------------------------------------------------------------------------------------------------------------
# cfg3: trunk as a field, B = 8 x 8 064 query points, 480 x 640, bf16 fused activations
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def cfg3():
    b, n = 8, 192 * 42
    sc, d = big_scene(seed=31, batch=b, n_views=1, height=480, width=640, n_rays=n, bias_scale=0.05)
    rng = np.random.default_rng(7)
    zq = rng.uniform(sc['near'], sc['far'], (b, n, 1)).astype(np.float32)
    points = (sc['rays_o'] + zq * sc['rays_d']).astype(np.float32)           # in front of the cameras
    dirs = rng.standard_normal((b, n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    d['points'], d['dirs'] = dev(points), dev(dirs)
    d['pf'], d['pf16'] = ops.pack_net(d['fine']), ops.pack_net_bf16(d['fine'])
    return sc, d, points, dirs


def _oracle_acts_fn(sc, d, bi):
    """float64 torch twin on scene `bi` alone (its 315 MB feature map; the last scene sits at the largest offsets)."""
    t64 = lambda a: torch.as_tensor(np.asarray(a)).to(torch.float64)
    net = T.unflatten_net(t64(sc['fine']))
    geo = (t64(sc['images'][bi:bi + 1]), d['features'][bi:bi + 1].cpu().to(torch.float64), t64(sc['intrinsics'][bi:bi + 1]),
           t64(sc['extrinsics_inv'][bi:bi + 1]))
    return lambda p, q: torch.stack(T.query_acts(net, p, q, *geo), 0), t64


def _oracle_acts_np(sc, feats_bi, bi, points, dirs):
    """The NumPy fp32 oracle (the arithmetic contract: fp32 pixel coordinates and lerp factors) on query points of scene `bi`:
    project -> gather -> PE -> trunk with complete_output, as lmvnerf/model_v4.py:216-262 -> the 8 activations."""
    sl = slice(bi, bi + 1)
    world = points[:, :, None, :]                                            # one "sample" per point
    pix, cam = O.compute_pixel_in_image_mv(world, sc['intrinsics'][sl], sc['extrinsics_inv'][sl])
    norm = (sc['images'][sl] * np.float32(2.0) - np.float32(1.0)).astype(np.float32)
    feat = O.get_projection_features_mv(norm, feats_bi, pix)
    cdir = O.world_to_camera_direction_vector_mv(dirs, sc['extrinsics_inv'][sl])[:, :, :, None, :]
    n = points.shape[1]
    return O.mv_embedding(O.unflatten_net(sc['fine']), cam[..., :3].reshape(1, n, 1, 3), cdir.reshape(1, n, 1, 3),
                          feat.reshape(1, n, 1, -1), 1, complete_output=True)


def test_cfg3_query_field_bf16_and_fp32_at_full_size(cfg3):
    sc, d, points, dirs = cfg3
    geo = (d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'])
    assert d['points'].shape == (8, 8064, 3) and d['features'].numel() * 4 > 2 ** 31
    _, acts = ops.query_field(d['points'], d['dirs'], *geo, d['pf'], complete_output=True)
    z0 = torch.zeros((8, 8064, 1), device=DEV)
    _, emb16, fused16 = ops.field_eval_bf16(d['points'], d['dirs'], z0, *geo, d['pf'], d['pf16'], return_embedding=True, return_fused_acts=True)
    _, _, fused16_again = ops.field_eval_bf16(d['points'], d['dirs'], z0, *geo, d['pf'], d['pf16'], return_embedding=True, return_fused_acts=True)
    torch.cuda.synchronize()
    assert torch.equal(fused16, fused16_again) and torch.equal(fused16[3], emb16)
    for k in range(4):                                                       # bf16 kernel vs fp32 kernel, all 64 512 points
        ref = acts[4 + k]
        err = (fused16[k, :, :, 0] - ref).abs()
        print(f'cfg3 bf16 fused act {k}: mean|d| {err.mean().item():.2e} (mean|ref| {ref.abs().mean().item():.2e}), max {err.max().item():.2e}')
        assert err.mean().item() < 2e-2 * ref.abs().mean().item() and torch.isfinite(fused16[k]).all()
    sub = np.arange(0, 8064, 126)                                            # 64 points of the first and of the last scene
    for bi in (0, 7):
        want = _oracle_acts_np(sc, d['features'][bi:bi + 1].cpu().numpy(), bi, points[bi:bi + 1, sub], dirs[bi:bi + 1, sub])
        for k in range(8):                                                   # x0, f1..f3 (per view, V = 1), mean, u1..u3
            got = acts[k][bi, sub].cpu().numpy()
            err = np.abs(got - want[k][0, :, 0]).max()
            assert err < 2e-5 * max(1.0, np.abs(want[k]).max()), (bi, k, err)
        # the float64 twin differs from both by fp32 input rounding (the top positional-encoding octave's argument, ~1.6e3 rad,
        # carries 1e-4; the pixel coordinate, |x| up to 640, 4e-5 in the lerp factor on white-noise feature maps): a looser,
        # independent cross-check of the same activations
        f, t64 = _oracle_acts_fn(sc, d, bi)
        want64 = f(t64(points[bi:bi + 1, sub]), t64(dirs[bi:bi + 1, sub])).numpy()         # (4,1,64,128)
        for k in range(4):
            got = acts[4 + k][bi, sub].cpu().numpy()
            assert np.abs(got - want64[k, 0]).max() < 1e-3 * max(1.0, np.abs(want64[k]).max()), (bi, k)
    # ray (here: point) independence at this size: the second half alone equals the second half of the whole
    _, acts_hi = ops.query_field(d['points'][:, 4000:].contiguous(), d['dirs'][:, 4000:].contiguous(), *geo, d['pf'], complete_output=True)
    for k in range(4, 8):
        assert torch.equal(acts_hi[k], acts[k][:, 4000:])


def test_cfg3_query_vjp_jvp_at_full_size(cfg3):
    from tests.test_gpu_query import check_close
    sc, d, points, dirs = cfg3
    geo = (d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'])
    gen = torch.Generator(device=DEV).manual_seed(3)
    tp = torch.randn((8, 8064, 3), device=DEV, generator=gen)
    td = torch.randn((8, 8064, 3), device=DEV, generator=gen)
    g = torch.randn((4, 8, 8064, 128), device=DEV, generator=gen)
    t_acts = ops.query_jvp(d['points'], d['dirs'], tp, td, *geo, d['pf'])
    stash = ops.query_stash(d['points'], d['dirs'], *geo, d['pf'])
    dp, dd = ops.query_vjp(d['points'], d['dirs'], *geo, ops.pack_bwd_streams(d['fine']), stash, g)
    torch.cuda.synchronize()
    assert torch.isfinite(t_acts).all() and torch.isfinite(dp).all() and torch.isfinite(dd).all()
    # <g, J t> = <J^T g, t>, point by point (a wrong scene offset cannot hide in a total).  The two HIP paths evaluate the
    # primal in different kernels, so a pre-activation within rounding of zero may take the other relu branch in one of them,
    # which changes that point's Jacobian: a handful of the 64 512 points may disagree, all others must agree tightly.
    lhs = (t_acts.double() * g.double()).sum(dim=(0, 3))                                    # (8, 8064)
    rhs = (dp.double() * tp.double()).sum(-1) + (dd.double() * td.double()).sum(-1)
    scale = lhs.abs().median().item()
    bad = (lhs - rhs).abs() > 1e-3 * torch.maximum(torch.maximum(lhs.abs(), rhs.abs()), torch.tensor(scale, device=DEV))
    for bi in range(8):
        print(f'cfg3 scene {bi}: <g,Jt> = {lhs[bi].sum().item():.6e}, <JTg,t> = {rhs[bi].sum().item():.6e}, '
              f'points off by > 1e-3: {int(bad[bi].sum())} of 8064')
        assert int(bad[bi].sum()) <= 8, (bi, int(bad[bi].sum()))                           # < 0.1 % (relu flips)
    good = ~bad
    tot_l, tot_r = lhs[good].sum().item(), rhs[good].sum().item()
    assert abs(tot_l - tot_r) < 1e-4 * lhs[good].abs().sum().item(), (tot_l, tot_r)
    # and against float64 autograd on 64 points of the last scene
    sub = np.arange(0, 8064, 126)
    f, t64 = _oracle_acts_fn(sc, d, 7)
    gs = g[:, 7:8, sub].double().cpu()
    _, (dp_ref, dd_ref) = torch.autograd.functional.vjp(f, (t64(points[7:8, sub]), t64(dirs[7:8, sub])), gs)
    check_close(dp[7, sub].cpu().numpy(), dp_ref[0].numpy(), 'd_points')
    check_close(dd[7, sub].cpu().numpy(), dd_ref[0].numpy(), 'd_dirs')
    _, t_ref = torch.autograd.functional.jvp(f, (t64(points[7:8, sub]), t64(dirs[7:8, sub])),
                                             (tp[7:8, sub].double().cpu(), td[7:8, sub].double().cpu()))
    for k in range(4):
        check_close(t_acts[k, 7, sub].cpu().numpy(), t_ref[k, 0].numpy(), f't_acts[{k}]')


# ---------------------------------------------------------------------------------------------------------------
# cfg4: one scene per GPU, 128 x 128 = 16 384 rays, full train step
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def cfg4():
    sc = make_scene(seed=41, batch=1, n_views=1, height=128, width=128, bias_scale=0.05)
    y = np.random.default_rng(5).random((1, 16384, 3)).astype(np.float32)
    return sc, y


def _model(sc, n):
    m = MVVNeRFRenderer(n, n, n_views=1, batch_size=1, near=sc['near'], far=sc['far'], device=DEV)
    m.set_weights(sc['coarse'], sc['fine'])
    return m


def _lg(m, sc, y, sl, stop=False):
    inputs = (sc['rays_o'][:, sl], sc['rays_d'][:, sl], sc['images'], sc['intrinsics'], sc['extrinsics_inv'])
    loss, grad, out = m.loss_and_grads(inputs, y[:, sl], sc['features'], u_coarse=dev(sc['u_coarse'][:, sl]),
                                       u_fine=dev(sc['u_fine'][:, sl]), stop_fine_z=stop)
    torch.cuda.synchronize()
    return float(loss), grad.clone(), [o.clone() for o in out]


def test_cfg4_train_step_gradient_is_additive_over_ray_halves(cfg4):
    sc, y = cfg4
    assert sc['rays_o'].shape == (1, 16384, 3)
    loss, grad, out = _lg(_model(sc, 16384), sc, y, slice(None))
    assert np.isfinite(loss) and torch.isfinite(grad).all() and grad.abs().max().item() > 0
    m_half = _model(sc, 8192)
    la, ga, oa = _lg(m_half, sc, y, slice(0, 8192))
    lb, gb, ob = _lg(m_half, sc, y, slice(8192, 16384))
    # Keras MSE is a mean over all elements: the full-batch loss / gradient is the mean of the two halves'
    assert abs(loss - 0.5 * (la + lb)) < 1e-6 * max(1.0, loss)
    for w_, a_, b_ in zip(out, oa, ob):
        assert torch.equal(w_, torch.cat([a_, b_], 1))                       # forward: rays independent, deterministic
    want = 0.5 * (ga + gb)
    for name, sl in (('coarse', slice(0, 247300)), ('fine', slice(247300, 494600))):
        rel = ((grad[sl] - want[sl]).norm() / want[sl].norm()).item()
        print(f'cfg4 {name} gradient: |full - mean(halves)| / |.| = {rel:.2e}')
        assert rel < 1e-4, (name, rel)                                       # fp32 atomic accumulation order only
    # against the float64 torch twin on a strided 64-ray subset of the same scene (128 x 128 source map)
    sub = np.arange(0, 16384, 256)
    scs = dict(sc, rays_o=sc['rays_o'][:, sub], rays_d=sc['rays_d'][:, sub], u_coarse=sc['u_coarse'][:, sub], u_fine=sc['u_fine'][:, sub])
    loss_ref, gc_ref, gf_ref, outs = T.train_loss_and_grads(sc['coarse'], sc['fine'], y[:, sub], scs, dtype=torch.float64, stop_fine_z=True)
    ls, gs, os_ = _lg(_model(sc, 64), sc, y, sub, stop=True)
    assert abs(ls - loss_ref) < 1e-5
    for g_, r_ in zip(os_, outs):
        assert np.abs(g_.cpu().numpy() - r_).max() < 1e-4
    gs = gs.cpu().numpy()
    for name, got, ref in (('coarse', gs[:247300], gc_ref), ('fine', gs[247300:], gf_ref)):
        for lo, hi in ((0, 48512), (48512, 48640), (48640, 246784), (246784, 247300)):
            rel = np.linalg.norm(got[lo:hi] - ref[lo:hi]) / np.linalg.norm(ref[lo:hi])
            assert rel < 6e-3, (name, lo, hi, rel)


def test_cfg4_feature_map_gradient_through_the_table_matches_the_direct_scatter(cfg4, monkeypatch):
    """dL/d(combined_features) at the cfg4 shape (128 x 128 x 256 map, 16 384 rays, both passes): the texel-table path
    (texel_scatter_kernel + texel_grad_to_features_kernel) against the direct four-tap scatter (field_dz_kernel<false>)."""
    from thesis_clip_nerf_amd import ops
    sc, y = cfg4
    inputs = (sc['rays_o'], sc['rays_d'], sc['images'], sc['intrinsics'], sc['extrinsics_inv'])
    got = {}
    for table in (True, False):
        monkeypatch.setattr(ops, 'texel_table_pays', lambda *a, _t=table: _t)
        m = _model(sc, 16384)
        _, grad, _, d_feat = m.loss_and_grads(inputs, y, sc['features'], u_coarse=dev(sc['u_coarse']), u_fine=dev(sc['u_fine']),
                                              stop_fine_z=True, return_d_features=True)
        torch.cuda.synchronize()
        got[table] = (grad.clone(), d_feat.clone())
    a, b = got[True][1], got[False][1]
    assert a.shape == (1, 1, 128, 128, 256) and torch.isfinite(a).all() and b.abs().max().item() > 0
    rel = ((a - b).norm() / b.norm()).item()
    print(f'cfg4 d_features: |table - direct| / |direct| = {rel:.2e}')
    # the same sums in another order - and a forward that differs in the last bits (with / without the table), so single relu branches
    # of near-zero pre-activations differ: measured 2e-4
    assert rel < 1e-3, rel
    assert torch.equal(a == 0, b == 0) or a[b == 0].abs().max().item() < 1e-9     # texels no sample touches stay untouched
    for sl in (slice(0, 247300), slice(247300, 494600)):                     # the weight gradients do not depend on the path (up to the forward's table rounding)
        assert ((got[True][0][sl] - got[False][0][sl]).norm() / got[False][0][sl].norm()).item() < 1e-3


def test_cfg4_train_step_updates_weights(cfg4):
    sc, y = cfg4
    m = _model(sc, 16384)
    m.compile(learning_rate=1e-3)
    data = ((sc['rays_o'], sc['rays_d'], sc['images'], sc['intrinsics'], sc['extrinsics_inv']), y)
    uc, uf = dev(sc['u_coarse']), dev(sc['u_fine'])
    before = m.fine_net.clone()
    losses = [float(m.train_step(data, combined_features=sc['features'], u_coarse=uc, u_fine=uf)['loss']) for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert torch.isfinite(m.fine_net).all() and not torch.equal(before, m.fine_net)


def test_stash_tile_limit_is_reported():
    """The stash addresses tiles with 32-bit byte offsets: V * B*R*S/32 must stay below 2^18.  Just past the bound the entry point
    must refuse (before any launch) with a message naming the limit; just below it the size query answers."""
    lib = _lib.lib()
    dummy = torch.zeros(1024, dtype=torch.float32, device=DEV)
    p = ctypes.c_void_p(dummy.data_ptr())
    for v, r, s in ((1, 65536, 128), (3, 21846, 128), (2, 32768, 128)):     # 262144, 262152, 262144 tiles
        assert v * ((r * s + 31) // 32) >= 2 ** 18
        rc = lib.mvnerf_field_eval_stash(p, p, p, p, p, None, p, p, p, 1, v, r, s, 64, 64, p, p, p, None)
        assert rc < 0
        with pytest.raises(ValueError, match='at most 262143'):
            _lib.check(rc, 'field_eval_stash')
    assert ops.stash_bytes(1, 1, 65535, 128) == 14 * 262140 * 16384
    # cfg4's fine pass (65 536 tiles, 15 GB of stash) and cfg5's V = 3 train shape (196 608 per-view tiles) are inside the bound
    assert 1 * (16384 * 128 // 32) < 2 ** 18 and 3 * (16384 * 128 // 32) < 2 ** 18
