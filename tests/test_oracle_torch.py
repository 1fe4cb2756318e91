"""The differentiable torch twin (oracle/mvnerf_torch.py) reproduces the NumPy oracle's forward, and its
autograd gradients agree with central finite differences in float64 - it is the reference for the
backward pass.  CPU only."""
import numpy as np
import torch

from oracle import mvnerf_oracle as O
from oracle import mvnerf_torch as T
from thesis_clip_nerf_amd.synthetic import make_scene


def test_forward_matches_numpy_oracle():
    sc = make_scene(seed=31, batch=1, n_views=2, height=12, width=16, n_rays=6, bias_scale=0.05)
    ref = O.render_call(O.unflatten_net(sc['coarse']), O.unflatten_net(sc['fine']), sc['rays_o'], sc['rays_d'], sc['images'],
                        sc['intrinsics'], sc['extrinsics_inv'], sc['features'], sc['near'], sc['far'], 64, sc['u_coarse'], sc['u_fine'])
    t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    out = T.render_call(t32(sc['coarse']), t32(sc['fine']), t32(sc['rays_o']), t32(sc['rays_d']), t32(sc['images']),
                        t32(sc['intrinsics']), t32(sc['extrinsics_inv']), t32(sc['features']), sc['near'], sc['far'], 64,
                        t32(sc['u_coarse']), t32(sc['u_fine']))
    for g, r in zip(out, ref):
        assert np.abs(g.numpy() - r).max() < 2e-5


def test_gradients_match_finite_differences_fp64():
    sc = make_scene(seed=32, batch=1, n_views=1, height=8, width=8, n_rays=3, bias_scale=0.05)
    y = np.random.default_rng(0).random((1, 3, 3))
    for stop in (False, True):
        loss, gc, gf, _ = T.train_loss_and_grads(sc['coarse'], sc['fine'], y, sc, stop_fine_z=stop)
        assert np.isfinite(loss) and np.abs(gc).max() > 0 and np.abs(gf).max() > 0
        rng = np.random.default_rng(1)
        for which, g in (('coarse', gc), ('fine', gf)):
            idx = np.concatenate([rng.integers(0, 48512, 3), rng.integers(48640, 246784, 3), rng.integers(246784, 247300, 2)])
            for i in idx:
                eps = 1e-6
                nets = {'coarse': sc['coarse'].astype(np.float64), 'fine': sc['fine'].astype(np.float64)}
                lp = []
                for sgn in (+1, -1):
                    p = {k: v.copy() for k, v in nets.items()}
                    p[which][i] += sgn * eps
                    lp.append(_loss(p['coarse'], p['fine'], y, sc, stop))
                fd = (lp[0] - lp[1]) / (2 * eps)
                if not stop or which == 'fine':
                    assert abs(fd - g[i]) < 1e-6 + 1e-4 * abs(fd), (which, i, fd, g[i])
        if stop:
            gf_stop = gf
        else:
            gf_full = gf
    assert np.allclose(gf_full, gf_stop)        # the fine net's own gradient does not depend on the detach


def _loss(cf, ff, y, sc, stop):
    t = lambda a: torch.as_tensor(np.asarray(a)).double()
    with torch.no_grad():
        out = T.render_call(t(cf), t(ff), t(sc['rays_o']), t(sc['rays_d']), t(sc['images']), t(sc['intrinsics']),
                            t(sc['extrinsics_inv']), t(sc['features']), sc['near'], sc['far'], 64, t(sc['u_coarse']),
                            t(sc['u_fine']), stop_fine_z=stop)
    return float(((t(y) - out[0]) ** 2).mean() + ((t(y) - out[2]) ** 2).mean())


def test_query_acts_matches_numpy_oracle_and_finite_differences():
    """The torch twin of the trunk-as-a-field (LanguageNeRF's use, oracle/mvnerf_torch.query_acts) against the NumPy oracle's
    complete_output on the same points, and its input Jacobian against central differences in float64."""
    sc = make_scene(seed=33, batch=2, n_views=2, height=10, width=12, n_rays=5, bias_scale=0.05)
    rng = np.random.default_rng(0)
    pts = (sc['rays_o'] + 0.8 * sc['rays_d']).astype(np.float32)
    dirs = rng.standard_normal(pts.shape).astype(np.float32)
    net_np = O.unflatten_net(sc['fine'])
    # NumPy oracle: one sample per point (z = 0 along a zero-length step: world point = origin)
    world = pts[:, :, None, :]
    pix, cam = O.compute_pixel_in_image_mv(world, sc['intrinsics'], sc['extrinsics_inv'])
    b, v = sc['images'].shape[:2]
    n = pts.shape[1]
    feat = O.get_projection_features_mv((sc['images'] * np.float32(2.0) - np.float32(1.0)).astype(np.float32), sc['features'], pix)
    cdir = O.world_to_camera_direction_vector_mv(dirs, sc['extrinsics_inv'])
    cdir = np.broadcast_to(cdir[:, :, :, None, :], (b, v, n, 1, 3))
    outs = O.mv_embedding(net_np, cam[..., :3].reshape(b * v, n, 1, 3), cdir.reshape(b * v, n, 1, 3), feat.reshape(b * v, n, 1, -1), v,
                          complete_output=True)
    t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    net_t = T.unflatten_net(t32(sc['fine']))
    acts = T.query_acts(net_t, t32(pts), t32(dirs), t32(sc['images']), t32(sc['features']), t32(sc['intrinsics']), t32(sc['extrinsics_inv']))
    for k in range(4):
        assert np.abs(acts[k].numpy() - outs[4 + k][:, :, 0]).max() < 2e-5
    # Jacobian-vector product by autograd vs central differences (float64, small step: the PE has gain pi * 2^9)
    t64 = lambda a: torch.as_tensor(np.asarray(a)).double()
    net64 = T.unflatten_net(t64(sc['fine']))
    geo = (t64(sc['images']), t64(sc['features']), t64(sc['intrinsics']), t64(sc['extrinsics_inv']))
    f = lambda p_, d_: torch.stack(T.query_acts(net64, p_, d_, *geo), 0)
    tp, td = t64(rng.standard_normal(pts.shape)), t64(rng.standard_normal(pts.shape))
    _, jv = torch.autograd.functional.jvp(f, (t64(pts), t64(dirs)), (tp, td))
    eps = 1e-9                                              # small: a relu / floor kink inside the step spoils the quotient
    fd = (f(t64(pts) + eps * tp, t64(dirs) + eps * td) - f(t64(pts) - eps * tp, t64(dirs) - eps * td)) / (2 * eps)
    assert float((jv - fd).norm() / fd.norm()) < 1e-4
