"""The trunk as a differentiable field on query points: forward-mode product (mvnerf_query_jvp) and
input gradient (mvnerf_query_vjp) through the C ABI against autograd on the torch twin of the oracle in float64.
Bars: fp32 kernels vs an fp64 reference through a positional encoding whose derivative has gain pi*2^9: the fp32
argument of the top octave (~1e3 rad) carries ~6e-5 of absolute error, which the derivative passes on at relative
size, so first derivatives agree to ~1e-3 relative (measured 1-3e-3; torch's own fp32 autograd is no closer);
besides rounding, a pre-activation within ~1e-6 of zero may take the other relu branch, which moves single rows.
Hence the two-part bar of check_close here and the tight transpose identity between the two HIP paths below."""
import numpy as np
import pytest
import torch

from oracle import mvnerf_torch as T
from thesis_clip_nerf_amd import ops
from thesis_clip_nerf_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def make_query(seed, n_views, n_points, batch=1, hw=(16, 20)):
    sc = make_scene(seed=seed, batch=batch, n_views=n_views, height=hw[0], width=hw[1], n_rays=n_points, bias_scale=0.05)
    rng = np.random.default_rng(seed + 100)
    z = rng.uniform(sc['near'], sc['far'], (batch, n_points, 1)).astype(np.float32)
    points = (sc['rays_o'] + z * sc['rays_d']).astype(np.float32)          # points in front of the cameras
    dirs = rng.standard_normal((batch, n_points, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    return sc, points, dirs


def t64(a):
    return torch.as_tensor(np.asarray(a)).to(torch.float64)


def oracle_fn(sc):
    net = T.unflatten_net(t64(sc['fine']))
    geo = (t64(sc['images']), t64(sc['features']), t64(sc['intrinsics']), t64(sc['extrinsics_inv']))

    def f(points, dirs):
        return torch.stack(T.query_acts(net, points, dirs, *geo), 0)       # (4,B,N,128)
    return f


def rel(got, ref):
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))


def check_close(got, ref, name):
    """Relative L2 over the tensor (3e-2: a relu flip moves a whole row) and the median per-point relative error (3e-3:
    what rounding alone does)."""
    rows = np.linalg.norm((got - ref).reshape(-1, ref.shape[-1]), axis=-1) / np.maximum(
        np.linalg.norm(ref.reshape(-1, ref.shape[-1]), axis=-1), 1e-30)
    assert rel(got, ref) < 3e-2, (name, rel(got, ref))
    assert np.median(rows) < 3e-3, (name, float(np.median(rows)))


@pytest.mark.parametrize('n_views,n_points,batch', [(1, 40, 1), (2, 64, 2), (3, 32, 1)])
def test_query_jvp_matches_autograd(n_views, n_points, batch):
    sc, points, dirs = make_query(7 + n_views, n_views, n_points, batch)
    rng = np.random.default_rng(1)
    tp = rng.standard_normal(points.shape).astype(np.float32) * 1e-2
    td = rng.standard_normal(dirs.shape).astype(np.float32) * 1e-2
    f = oracle_fn(sc)
    acts_ref, t_ref = torch.autograd.functional.jvp(f, (t64(points), t64(dirs)), (t64(tp), t64(td)))
    d = {k: dev(sc[k]) for k in ['images', 'features', 'intrinsics', 'extrinsics_inv', 'fine']}
    packed = ops.pack_net(d['fine'])
    t_acts, acts = ops.query_jvp(dev(points), dev(dirs), dev(tp), dev(td), d['images'], d['features'], d['intrinsics'],
                                 d['extrinsics_inv'], packed, return_primal=True)
    torch.cuda.synchronize()
    assert np.abs(acts.cpu().numpy() - acts_ref.numpy()).max() < 2e-5 * max(1.0, float(acts_ref.abs().max()))
    # the primal also equals the inference kernel's complete_output on the same points
    _, acts_q = ops.query_field(dev(points), dev(dirs), d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'], packed,
                                complete_output=True)
    for k in range(4):
        assert (acts[k] - acts_q[4 + k]).abs().max().item() < 2e-5 * max(1.0, acts_q[4 + k].abs().max().item())
    got, ref = t_acts.cpu().numpy(), t_ref.numpy()
    for k in range(4):
        check_close(got[k], ref[k], f't_acts[{k}]')


@pytest.mark.parametrize('n_views,n_points,batch', [(1, 40, 1), (2, 64, 2), (3, 32, 1)])
def test_query_vjp_matches_autograd(n_views, n_points, batch):
    sc, points, dirs = make_query(17 + n_views, n_views, n_points, batch)
    rng = np.random.default_rng(2)
    g = rng.standard_normal((4, batch, n_points, 128)).astype(np.float32)
    f = oracle_fn(sc)
    _, (dp_ref, dd_ref) = torch.autograd.functional.vjp(f, (t64(points), t64(dirs)), t64(g))
    d = {k: dev(sc[k]) for k in ['images', 'features', 'intrinsics', 'extrinsics_inv', 'fine']}
    geo = (d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'])
    stash = ops.query_stash(dev(points), dev(dirs), *geo, ops.pack_net(d['fine']))
    dp, dd = ops.query_vjp(dev(points), dev(dirs), *geo, ops.pack_bwd_streams(d['fine']), stash, dev(g))
    torch.cuda.synchronize()
    check_close(dp.cpu().numpy(), dp_ref.numpy(), 'd_points')
    check_close(dd.cpu().numpy(), dd_ref.numpy(), 'd_dirs')


def test_query_jvp_is_transpose_of_vjp():
    """<g, J t> == <J^T g, t> for the two HIP paths themselves (fp32, same relu branches): tight bar."""
    sc, points, dirs = make_query(31, 2, 64)
    rng = np.random.default_rng(3)
    tp = rng.standard_normal(points.shape).astype(np.float32)
    td = rng.standard_normal(dirs.shape).astype(np.float32)
    g = rng.standard_normal((4, 1, 64, 128)).astype(np.float32)
    d = {k: dev(sc[k]) for k in ['images', 'features', 'intrinsics', 'extrinsics_inv', 'fine']}
    geo = (d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'])
    packed = ops.pack_net(d['fine'])
    t_acts = ops.query_jvp(dev(points), dev(dirs), dev(tp), dev(td), *geo, packed)
    stash = ops.query_stash(dev(points), dev(dirs), *geo, packed)
    dp, dd = ops.query_vjp(dev(points), dev(dirs), *geo, ops.pack_bwd_streams(d['fine']), stash, dev(g))
    lhs = float((t_acts.double() * dev(g).double()).sum())
    rhs = float((dp.double() * dev(tp).double()).sum() + (dd.double() * dev(td).double()).sum())
    assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)


# ---- LanguageNeRF train step on the HIP trunk vs the float64 restatement (oracle/lmvnerf_torch.py) ----------------
def _language_case(seed, n_views, batch, n_points, representation):
    from thesis_clip_nerf_amd.lmvnerf import LanguageNeRF
    sc = make_scene(seed=seed, batch=batch, n_views=n_views, height=16, width=20, n_rays=4, bias_scale=0.05)
    rng = np.random.default_rng(seed)
    rot_dim = 4 if representation == 'quaternion' else 6

    def poses():
        t = (np.array([0.0, 0.0, 0.8]) + 0.1 * rng.standard_normal((batch, n_points, 3))).astype(np.float32)
        r = rng.standard_normal((batch, n_points, rot_dim)).astype(np.float32)
        if representation == 'quaternion':
            r /= np.linalg.norm(r, axis=-1, keepdims=True)
        return t, r
    t1, r1 = poses()
    t2, r2 = poses()
    lab0 = rng.random((batch, n_points)).astype(np.float32)
    lab0 /= lab0.sum(-1, keepdims=True)
    labels = (lab0, rng.standard_normal((batch, n_points, 3)).astype(np.float32),
              rng.standard_normal((batch, n_points, rot_dim)).astype(np.float32))
    inputs = (t1, r1, t2, r2, sc['images'], sc['intrinsics'], sc['extrinsics_inv'])
    torch.manual_seed(seed)
    model = LanguageNeRF(sc['fine'], n_points_train=n_points, n_views=n_views, batch_size=batch,
                         rotation_representation=representation, softmax_before_loss=True, device=DEV)
    return sc, inputs, labels, model


@pytest.mark.parametrize('n_views,batch,representation', [(1, 1, '6d'), (2, 2, 'quaternion')])
def test_language_train_step_matches_restatement(n_views, batch, representation):
    from oracle import lmvnerf_torch as L
    from tests.test_oracle_lmvnerf import keras_weights
    n_points = 3
    sc, inputs, labels, model = _language_case(50 + n_views, n_views, batch, n_points, representation)
    out, pred = model.loss_and_grads((inputs, labels), sc['features'])
    torch.cuda.synchronize()
    # float64 restatement with the same read-out weights
    w = {k: v.detach().double().cpu().clone().requires_grad_(True) for k, v in keras_weights(model.grasp_readout).items()}
    net = T.unflatten_net(t64(sc['fine']))
    checks = torch.as_tensor(L.transforms_to_check(7))
    loss, landscape, loss_t, loss_r, pred_ref = L.train_losses(w, net, [t64(a) for a in inputs], [t64(a) for a in labels], checks,
                                                               n_points, t64(sc['features']), representation)
    loss.sum().backward()
    assert np.abs(pred.cpu().numpy() - pred_ref.detach().numpy()).max() < 1e-4 * max(1.0, float(pred_ref.detach().abs().max()))
    assert abs(float(out['landscape_loss']) - float(landscape.detach().mean())) < 1e-4 * max(1.0, abs(float(landscape.detach().mean())))
    # the two gradient losses are cosine similarities of d prediction / d pose: first derivatives through the PE (see the
    # module docstring for the 1e-3-level fp32 error of those)
    assert abs(float(out['grad_loss_t']) - float(loss_t.detach())) < 5e-3
    assert abs(float(out['grad_loss_r']) - float(loss_r.detach())) < 5e-3
    worst = 0.0
    for k, ref in w.items():                  # oracle name -> module parameter (Keras kernels are the transposed weights)
        if k.startswith('ds'):
            lin = model.grasp_readout.activation_downscale[int(k[2])]
        elif k.startswith('comb'):
            lin = model.grasp_readout.combined_activation_downscale
        elif k.startswith('out'):
            lin = model.grasp_readout.output_layer
        else:
            blk = model.grasp_readout.block_0 if k.startswith('b0') else model.grasp_readout.block_1
            lin = {'l0': blk.layer_0, 'l1': blk.layer_1, 'sc': blk.shortcut}[k.split('.')[1]]
        g = (lin.weight.grad.T if k.endswith('.k') else lin.bias.grad).double().cpu().numpy()
        r = ref.grad.numpy()
        # (the output bias has an exactly zero gradient - softmax and d/d pose are blind to it - hence the absolute term)
        e = np.linalg.norm(g - r)
        worst = max(worst, e)
        assert e < 3e-2 * np.linalg.norm(r) + 1e-6, (k, e, np.linalg.norm(r))
    assert worst > 0.0


def test_language_train_step_updates_readout_only():
    sc, inputs, labels, model = _language_case(60, 1, 1, 2, '6d')
    before = [p.detach().clone() for p in model.grasp_readout.parameters()]
    trunk_before = model.trunk_net.clone()
    out = model.train_step((inputs, labels), sc['features'])
    assert all(np.isfinite(float(v)) for v in out.values())
    assert any((a != b).any().item() for a, b in zip(before, model.grasp_readout.parameters()))
    assert torch.equal(trunk_before, model.trunk_net)
    scores = model.infer(inputs, model.compute_matrices().detach(), 2, sc['features'])
    assert scores.shape == (1, 2)
    scores16 = model.infer(inputs, model.compute_matrices().detach(), 2, sc['features'], compute_dtype='bf16')
    assert scores16.shape == (1, 2)
    assert (scores16 - scores).abs().max().item() < 5e-2 * max(1.0, scores.abs().max().item())     # bf16 trunk, fp32 read-out



def test_language_train_step_graph_replay_matches_eager():
    """compile(graph=True): two eager steps, then one capture, then replays - against a twin model stepping eagerly on the same data.
    The inputs change from step to step (staged into the graph's buffers), so a replay that ignored them would show.
    Phase A, learning rate 0: the weights stand still, every step's losses must agree to fp32 rounding.  Phase B, learning rate 1e-3: Adam's
    normalised update turns last-bit differences of small gradient entries into lr-sized weight differences, so two models drift apart
    however they are stepped; the bar there is the distance between the two against the distance both moved."""
    n_points, steps = 3, 5
    sc, inputs, labels, eager = _language_case(70, 2, 2, n_points, '6d')
    rng = np.random.default_rng(7)
    datas = [((*[(a + 0.05 * rng.standard_normal(a.shape)).astype(np.float32) for a in inputs[:4]], *inputs[4:]), labels) for _ in range(steps)]
    for lr in (0.0, 1e-3):
        eager, graphed, start = (_language_case(70, 2, 2, n_points, '6d')[3] for _ in range(3))
        eager.compile(learning_rate=lr)
        with pytest.raises(ValueError):
            graphed.compile(optimizer=torch.optim.SGD(graphed.grasp_readout.parameters(), lr=lr), graph=True)
        graphed.compile(learning_rate=lr, graph=True)
        seen = []
        for step, data in enumerate(datas):
            out_e = eager.train_step(data, sc['features'])
            out_g = graphed.train_step(data, sc['features'])
            bar = 1e-5 if lr == 0.0 else 5e-3
            for k in out_e:
                assert abs(float(out_e[k]) - float(out_g[k])) < bar * max(1.0, abs(float(out_e[k]))), (lr, step, k, float(out_e[k]), float(out_g[k]))
            seen.append(float(out_g['grad_loss_t']))
        assert graphed._graph is not None
        if lr == 0.0:
            assert min(abs(a - b) for a, b in zip(seen[2:], seen[3:])) > 2e-3, seen       # replays follow the staged inputs
        else:
            pe, pg, p0 = (list(m.grasp_readout.parameters()) for m in (eager, graphed, start))
            apart = sum((a - b).abs().sum().item() for a, b in zip(pe, pg))
            moved = sum((a - c).abs().sum().item() for a, c in zip(pe, p0))
            assert moved > 0 and apart < 0.1 * moved, (apart, moved)
    with pytest.raises(ValueError):                          # the capture fixed the shapes
        graphed.train_step(((*[a[:, :2] for a in inputs[:4]], *inputs[4:]), labels), sc['features'])
    with pytest.raises(RuntimeError):
        graphed.bind_graph_inputs(datas[0], sc['features'])


@pytest.mark.parametrize('n_views,n_points,batch', [(1, 40, 2), (2, 64, 2), (1, 33, 1)])
def test_stash_fused_acts_is_the_transposed_stash(n_views, n_points, batch):
    """mvnerf_stash_fused_acts against the same gather written with torch views of the stash (bit-exact: it only moves floats), and against
    the acts_fused output of the plain field pass."""
    sc = make_scene(seed=80 + n_points, batch=batch, n_views=n_views, height=16, width=20, n_rays=4)
    rng = np.random.default_rng(n_points)
    points = (np.array([0.0, 0.0, 0.8]) + 0.1 * rng.standard_normal((batch, n_points, 3))).astype(np.float32)
    dirs = rng.standard_normal((batch, n_points, 3)).astype(np.float32)
    d = {k: dev(sc[k]) for k in ('images', 'features', 'intrinsics', 'extrinsics_inv', 'fine')}
    geo = (d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'])
    packed = ops.pack_net(d['fine'])
    stash = ops.query_stash(dev(points), dev(dirs), *geo, packed)
    acts = ops.stash_fused_acts(stash, batch, n_views, n_points)
    rows = batch * n_points
    tiles = (rows + 31) // 32
    fused = stash.view(torch.float32)[7 * n_views * tiles * 4096:][:7 * tiles * 4096].view(7, tiles, 128, 32)
    ref = fused[0::2].permute(0, 1, 3, 2).reshape(4, tiles * 32, 128)[:, :rows].reshape(4, batch, n_points, 128)
    assert torch.equal(acts, ref)
    direct = torch.stack(ops.query_field(dev(points), dev(dirs), *geo, packed, complete_output=True)[1][4:])
    assert (acts - direct).abs().max().item() < 1e-5 * max(1.0, direct.abs().max().item())
    with pytest.raises(ValueError):
        ops.stash_fused_acts(stash[:1024], batch, n_views, n_points)
