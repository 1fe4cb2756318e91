"""The per-point part of GraspReadout as fused HIP passes (csrc/grasp_head.hip, lmvnerf._HeadFn / _HeadVJP): value, first derivatives and the
derivative of the vector-Jacobian product (what the nested tape of LanguageNeRF.train_step takes, lmvnerf/model_v4.py:290-322) against the
same layers written out in float64 torch (delta_ngf/layers.py:8-42: four Dense(128 -> 64) + elu, concat, Dense(256 -> 64) + elu)."""
import numpy as np
import pytest
import torch

from thesis_clip_nerf_amd import lmvnerf as L

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def head64(acts, w4, b4, wc, bc):
    ds = [torch.nn.functional.elu(acts[k] @ w4[k].t() + b4[k]) for k in range(4)]
    return torch.nn.functional.elu(torch.cat(ds, -1) @ wc.t() + bc)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize('n', [96, 41, 1536])          # 41: a ragged last tile; 1536: the weight gradients go through mvnerf_gemm_tn
def test_fused_head_value_first_and_second_derivatives(n):
    g = torch.Generator().manual_seed(n)
    acts = (torch.randn((4, n, 128), generator=g) * 0.7)
    w4 = torch.randn((4, 64, 128), generator=g) * 0.12
    b4 = torch.randn((4, 64), generator=g) * 0.1
    wc = torch.randn((64, 256), generator=g) * 0.1
    bc = torch.randn((64,), generator=g) * 0.1
    gy = torch.randn((n, 64), generator=g)
    tt = torch.randn((4, n, 128), generator=g)

    def run(fn, dtype, dev):
        p = [t.to(dev, dtype).clone().requires_grad_(True) for t in (acts, w4, b4, wc, bc, gy)]
        a_, w4_, b4_, wc_, bc_, gy_ = p
        y = fn(a_, w4_, b4_, wc_, bc_)
        l1 = (y * gy_).sum()
        first = torch.autograd.grad(l1, [a_, w4_, b4_, wc_, bc_], create_graph=True)
        l2 = (first[0] * tt.to(dev, dtype)).sum()                          # a function of d y / d acts, as the pose-gradient loss is
        second = torch.autograd.grad(l2, [w4_, b4_, wc_, bc_, gy_])
        return y, first, second

    y64, f64, s64 = run(head64, torch.float64, 'cpu')
    y32, f32, s32 = run(lambda *a: L._HeadFn.apply(*a), torch.float32, DEV)
    torch.cuda.synchronize()
    assert rel(y32, y64) < 2e-6
    for name, a, b in zip(['d acts', 'd w4', 'd b4', 'd wc', 'd bc'], f32, f64):
        assert rel(a, b) < 1e-5, (name, rel(a, b))
    for name, a, b in zip(['dd w4', 'dd b4', 'dd wc', 'dd bc', 'dd g_y'], s32, s64):
        assert rel(a, b) < 1e-4, (name, rel(a, b))


def test_grasp_readout_fused_head_equals_the_layer_by_layer_form():
    """GraspReadout.forward with fused_head on and off: same prediction, same first gradients of every read-out variable."""
    torch.manual_seed(3)
    ro = L.GraspReadout(42).to(DEV)
    acts = torch.randn((4, 2, 3, 42, 128), device=DEV) * 0.7
    outs = {}
    for fused in (True, False):
        ro.fused_head = fused
        for p in ro.parameters():
            p.grad = None
        a = acts.clone().requires_grad_(True)
        pred = ro(a)
        pred.square().sum().backward()
        outs[fused] = (pred.detach().clone(), a.grad.clone(), [p.grad.clone() for p in ro.parameters()])
    assert (outs[True][0] - outs[False][0]).abs().max().item() < 1e-4 * outs[False][0].abs().max().item() + 1e-6
    assert rel(outs[True][1], outs[False][1]) < 1e-4
    for x, y in zip(outs[True][2], outs[False][2]):
        assert rel(x, y) < 1e-4


def test_cotangents_on_the_weight_gradient_outputs_are_refused():
    a = torch.randn((4, 32, 128), device=DEV, requires_grad=True)
    w4 = (torch.randn((4, 64, 128), device=DEV) * 0.1).requires_grad_(True)
    b4 = torch.zeros((4, 64), device=DEV, requires_grad=True)
    wc = (torch.randn((64, 256), device=DEV) * 0.1).requires_grad_(True)
    bc = torch.zeros((64,), device=DEV, requires_grad=True)
    y = L._HeadFn.apply(a, w4, b4, wc, bc)
    gw = torch.autograd.grad(y.sum(), w4, create_graph=True)[0]
    with pytest.raises((NotImplementedError, RuntimeError)):
        gw.sum().backward()
