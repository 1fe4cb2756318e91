"""mvnerf_gemm_nt (the GraspReadout's wide Dense layers) against torch, value and first / second derivatives."""
import numpy as np
import pytest
import torch

from thesis_clip_nerf_amd import ops
from thesis_clip_nerf_amd.lmvnerf import _mm_nt, _wide_linear

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('m,n,k', [(32, 64, 8), (1536, 128, 2688), (128, 2688, 1536), (96, 2688, 64), (64, 128, 64512)])
def test_gemm_nt_matches_float64(m, n, k):
    g = torch.Generator().manual_seed(m + n + k)
    a, bt = torch.randn((m, k), generator=g), torch.randn((n, k), generator=g)
    got = ops.gemm_nt(a.to(DEV), bt.to(DEV)).cpu().double()
    want = a.double() @ bt.double().t()
    # fp32 products and accumulation over K terms
    assert (got - want).abs().max().item() < 2e-6 * k ** 0.5 * 16
    again = ops.gemm_nt(a.to(DEV), bt.to(DEV)).cpu().double()
    assert torch.equal(got, again)                                       # one wave per output block over all of K: deterministic


@pytest.mark.parametrize('m,n,k', [(8, 32, 64), (64512, 64, 128), (1536, 128, 2688), (1536, 64, 64)])
def test_gemm_tn_matches_float64(m, n, k):
    g = torch.Generator().manual_seed(m + n + k)
    gm, a = torch.randn((m, n), generator=g), torch.randn((m, k), generator=g)
    got = ops.gemm_tn(gm.to(DEV), a.to(DEV)).cpu().double()
    want = gm.double().t() @ a.double()
    assert (got - want).abs().max().item() < 2e-6 * m ** 0.5 * 16
    assert torch.equal(got, ops.gemm_tn(gm.to(DEV), a.to(DEV)).cpu().double())


@pytest.mark.parametrize('m,n,k', [(64512, 64, 128), (96, 128, 2688), (32, 64, 8)])
def test_gemm_nt_bias_equals_the_product_plus_a_broadcast_add(m, n, k):
    """mvnerf_gemm_nt_bias adds the bias after the products (epilogue of the unsplit kernel / of the split-K reduce): bit for bit the
    unfused result.  (64512, 64, 128) is the unsplit case, (96, 128, 2688) the split one."""
    g = torch.Generator().manual_seed(m + n + k + 1)
    a, bt, b = (torch.randn(sh, generator=g).to(DEV) for sh in ((m, k), (n, k), (n,)))
    assert torch.equal(ops.gemm_nt(a, bt, bias=b), ops.gemm_nt(a, bt) + b)
    with pytest.raises(ValueError):
        ops.gemm_nt(a, bt, bias=b[:-1].contiguous())


def test_gemm_nt_rejects_other_shapes():
    a = torch.zeros((30, 8), device=DEV)
    with pytest.raises(ValueError, match='needs M'):
        ops.gemm_nt(a, torch.zeros((64, 8), device=DEV))
    assert not ops.gemm_nt_ok(32, 32, 8) and ops.gemm_nt_ok(32, 64, 8)


def test_wide_linear_first_and_second_derivatives_match_torch():
    """The Function's backward is built from itself: gradients w.r.t. input and weight and a gradient-of-gradient (what
    LanguageNeRF.train_step takes through the GraspReadout) against plain torch in float64."""
    torch.manual_seed(3)
    lin = torch.nn.Linear(2688, 128).to(DEV)
    x = torch.randn((2, 48, 2688), device=DEV, requires_grad=True)        # M = 96 rows
    u = torch.randn((2, 48, 128), device=DEV)

    def second_order(fn, lin_, x_, u_):
        y = fn(lin_, x_)
        gx, = torch.autograd.grad((torch.tanh(y) * u_).sum(), x_, create_graph=True)
        loss = (gx ** 2).sum() + y.sum()
        gw, gb = torch.autograd.grad(loss, [lin_.weight, lin_.bias])
        return y.detach(), gx.detach(), gw, gb

    got = second_order(_wide_linear, lin, x, u)
    lin64 = torch.nn.Linear(2688, 128).to(DEV).double()
    lin64.load_state_dict({k_: v.double() for k_, v in lin.state_dict().items()})
    want = second_order(lambda l, xx: l(xx), lin64, x.detach().double().requires_grad_(True), u.double())
    for g_, w_ in zip(got, want):
        rel = ((g_.double() - w_).norm() / w_.norm()).item()
        assert rel < 2e-5, rel
    # shapes the kernel does not take stay on torch
    small = torch.nn.Linear(64, 64).to(DEV)
    xs = torch.randn((4, 64), device=DEV)
    assert torch.equal(_wide_linear(small, xs), small(xs))


@pytest.mark.parametrize('two_pairs,colsum_of', [(False, 0), (False, 1), (True, 2), (True, 1)])
def test_gemm_tn_batched_matches_float64(two_pairs, colsum_of):
    """mvnerf_gemm_tn_batched: a batch of weight gradients on column blocks of one cotangent, optionally the sum of two products, and
    the bias gradient (column sums) from the same pass; operands read where they lie (no copies), results deterministic."""
    g = torch.Generator(device=DEV).manual_seed(3)
    m, batch, n, k = 4000, 4, 64, 128
    gu = torch.randn((m, batch * n), device=DEV, generator=g)
    a = torch.randn((batch, m, k), device=DEV, generator=g)
    gu2 = torch.randn((m, batch * n), device=DEV, generator=g) if two_pairs else None
    a2 = torch.randn((batch, m, k), device=DEV, generator=g) if two_pairs else None
    blocks = lambda x: x.view(m, batch, n).permute(1, 0, 2)
    out = ops.gemm_tn_batched(blocks(gu), a, g2=None if gu2 is None else blocks(gu2), a2=a2, colsum_of=colsum_of)
    again = ops.gemm_tn_batched(blocks(gu), a, g2=None if gu2 is None else blocks(gu2), a2=a2, colsum_of=colsum_of)
    c, cs = out if colsum_of else (out, None)
    want = torch.einsum('bmn,bmk->bnk', blocks(gu).double(), a.double())
    if two_pairs:
        want = want + torch.einsum('bmn,bmk->bnk', blocks(gu2).double(), a2.double())
    assert c.shape == (batch, n, k)
    assert (c.double() - want).abs().max().item() < 2e-5 * want.abs().max().item()
    if colsum_of:
        src = gu if colsum_of == 1 else gu2
        want_s = src.double().sum(0).view(batch, n)
        assert (cs.double() - want_s).abs().max().item() < 2e-5 * want_s.abs().max().item()
        assert torch.equal(cs, again[1])
    assert torch.equal(c, again[0] if colsum_of else again)
    with pytest.raises(ValueError):
        ops.gemm_tn_batched(blocks(gu).transpose(1, 2), a)                       # rows not contiguous
