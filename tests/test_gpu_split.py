"""fp32-grade field pass on the bf16 matrix pipe (csrc/field_eval_split.hip: every GEMM operand cut exactly into three bf16
pieces, six MFMAs per product, fp32 accumulation).  It replaces the fp32-MFMA kernel at the SAME bars: per-sample outputs
and activations against the NumPy fp32 oracle, integer tap indices bit-exact, rendered outputs within the 1e-4 of
north_star, on the direct and the texel-table path, 1..3 views, ragged tiles."""
import os

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


def test_pack_net_split_pieces_sum_to_the_weights():
    """The three bf16 pieces of every streamed weight add up to the fp32 weight (to <= 2^-24 relative)."""
    sc = make_scene(seed=1, height=8, width=8, n_rays=4, bias_scale=0.1)
    net = dev(sc['fine'])
    sp = ops.pack_net_split(net).view(torch.bfloat16).view(-1, 64, 8).float()      # (chunks, lane, 8)
    n_k = (sp.shape[0] - 24) // 12
    body = sp[:n_k * 12].view(n_k, 4, 3, 64, 8).sum(2)                              # (kstep, nb, lane, jj): piece sum
    w0 = net[:379 * 128].view(379, 128)
    # k-step 4 (first feature k-step): rows 123 + 8h + jj, output 32nb + i
    want = torch.stack([w0[123 + 8 * (lane >> 5) + torch.arange(8, device=DEV), 32 * nb + (lane & 31)]
                        for nb in range(4) for lane in range(64)]).view(4, 64, 8)
    assert (body[4] - want).abs().max().item() <= 2 ** -24 * w0.abs().max().item()
    # hidden layer 0 (k-steps 20..27), k-step (kb, s): feature 32kb + 16s + 8(jj>>2) + 4h + (jj&3)
    w1 = net[48640:48640 + 128 * 128].view(128, 128)
    jj = torch.arange(8, device=DEV)
    for kbs in (0, 5):
        want = torch.stack([w1[32 * (kbs // 2) + 16 * (kbs % 2) + 8 * (jj >> 2) + 4 * (lane >> 5) + (jj & 3), 32 * nb + (lane & 31)]
                            for nb in range(4) for lane in range(64)]).view(4, 64, 8)
        assert (body[20 + kbs] - want).abs().max().item() <= 2 ** -24 * w1.abs().max().item()


@pytest.mark.parametrize('n_views,n_rays,s,table', [(1, 40, 64, False), (1, 40, 64, True), (3, 17, 128, True), (2, 300, 64, False),
                                                    (2, 33, 1, True)])
def test_field_eval_split_matches_oracle(n_views, n_rays, s, table):
    sc = make_scene(seed=80 + n_views, n_views=n_views, height=24, width=28, n_rays=n_rays, bias_scale=0.1)
    d = {k: dev(sc[k]) for k in ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'fine']}
    rng = np.random.default_rng(0)
    z = np.sort(rng.uniform(0.3, 1.3, (1, n_rays, s)).astype(np.float32), -1)
    net = O.unflatten_net(sc['fine'])
    rgb_ref, sig_ref, taps_ref = O.field_eval(net, sc['rays_o'], sc['rays_d'], z, sc['images'], sc['features'], sc['intrinsics'],
                                              sc['extrinsics_inv'], return_taps=True)
    packed, split = ops.pack_net(d['fine']), ops.pack_net_split(d['fine'])
    tab = ops.project_texels(d['features'], packed) if table else None
    args = (d['rays_o'], d['rays_d'], dev(z), d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'], packed)
    rgbs, taps, acts = ops.field_eval_split(*args, split, return_taps=True, complete_output=True, texel_table=tab)
    rgbs32, acts32 = ops.field_eval(*args, complete_output=True, texel_table=tab)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(taps.cpu().numpy(), taps_ref)                    # integer contract (a6)
    got = rgbs.cpu().numpy()
    e_rgb, e_sig = np.abs(got[..., :3] - rgb_ref).max(), np.abs(got[..., 3] - sig_ref).max()
    e32 = (rgbs - rgbs32).abs().max().item()
    print(f'split V={n_views} table={table}: max|rgb - oracle| {e_rgb:.2e}, max|sigma - oracle| {e_sig:.2e}, max|split - fp32 kernel| {e32:.2e}')
    assert e_rgb < 1e-6 * 5 and e_sig < 2e-5 * max(1.0, np.abs(sig_ref).max())    # the fp32 kernel's bars (test_gpu_ops / parity)
    for k in range(8):                                                              # all 8 complete_output activations
        ref = acts32[k]
        assert (acts[k] - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), k


@pytest.mark.parametrize('n_views,tables', [(1, 'auto'), (1, None), (3, 'auto')])
def test_render_fwd_split_matches_oracle(n_views, tables):
    sc = make_scene(seed=90 + n_views, n_views=n_views, height=32, width=32, n_rays=None if tables else 200, bias_scale=0.05)
    d = {k: dev(sc[k]) for k in ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'u_coarse', 'u_fine',
                                 'coarse', 'fine']}
    sub = np.arange(0, sc['rays_o'].shape[1], 4)
    ref = O.render_call(O.unflatten_net(sc['coarse']), O.unflatten_net(sc['fine']), sc['rays_o'][:, sub], sc['rays_d'][:, sub], sc['images'],
                        sc['intrinsics'], sc['extrinsics_inv'], sc['features'], sc['near'], sc['far'], 64, sc['u_coarse'][:, sub],
                        sc['u_fine'][:, sub])
    got = ops.render_fwd_split(d['rays_o'], d['rays_d'], d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'],
                               ops.pack_net(d['coarse']), ops.pack_net(d['fine']), ops.pack_net_split(d['coarse']),
                               ops.pack_net_split(d['fine']), d['u_coarse'], d['u_fine'], sc['near'], sc['far'], texel_tables=tables)
    again = ops.render_fwd_split(d['rays_o'], d['rays_d'], d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'],
                                 ops.pack_net(d['coarse']), ops.pack_net(d['fine']), ops.pack_net_split(d['coarse']),
                                 ops.pack_net_split(d['fine']), d['u_coarse'], d['u_fine'], sc['near'], sc['far'], texel_tables=tables)
    for name, g, a, r in zip(['rgb', 'depth', 'fine_rgb', 'fine_depth'], got, again, ref):
        assert torch.equal(g, a), name                                              # deterministic
        err = np.abs(g[:, sub].cpu().numpy() - r).max()
        print(f'render split V={n_views} tables={tables} {name}: max|hip - oracle| = {err:.2e}')
        assert err < 1e-4, (name, err)                                              # north_star bar, fp32


@pytest.mark.parametrize('views,table', [(1, True), (2, False), (3, True)])
def test_stash_of_split_forward_matches_fp32_stash(views, table):
    """Training forward on the split kernel: the 14 pre-activation slots it leaves in HBM (tile layout) against the fp32-MFMA
    kernel's, same per-sample outputs - the backward consumes either."""
    sc = make_scene(seed=3, n_views=views, height=16, width=16, n_rays=24, bias_scale=0.05)
    d = {k: dev(sc[k]) for k in ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'u_coarse', 'coarse']}
    z = ops.stratified_depths(d['u_coarse'], sc['near'], sc['far'])
    packed, split = ops.pack_net(d['coarse']), ops.pack_net_split(d['coarse'])
    tab = ops.project_texels(d['features'], packed) if table else None
    args = (d['rays_o'], d['rays_d'], z, d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'], packed)
    rgbs32, stash32 = ops.field_eval_stash(*args, texel_table=tab)
    rgbs, stash = ops.field_eval_stash(*args, texel_table=tab, packed_split=split)
    torch.cuda.synchronize()
    n = ops.stash_bytes(1, views, 24, 64) // 4
    slot = views * (24 * 64 // 32) * 4096                                      # floats per per-view slot
    keep = torch.ones(n, dtype=torch.bool, device=DEV)
    keep[6 * slot:7 * slot] = False                                              # per-view slot 6 (x3) is not written by either kernel
    a, b = stash.view(torch.float32)[:n][keep], stash32.view(torch.float32)[:n][keep]
    assert (rgbs - rgbs32).abs().max().item() < 1e-5
    assert (a - b).abs().max().item() < 2e-5 * max(1.0, b.abs().max().item())


def test_renderer_default_gemm_is_split_and_matches_mfma_f32():
    from thesis_clip_nerf_amd import MVVNeRFRenderer
    sc = make_scene(seed=71, height=16, width=16, n_rays=64, bias_scale=0.05)
    inputs = tuple(sc[k] for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv'])
    outs = {}
    for gemm in ('split_f16', 'split_bf16', 'mfma_f32'):
        m = MVVNeRFRenderer(64, 64, n_views=1, near=sc['near'], far=sc['far'], device=DEV, f32_gemm=gemm)
        m.set_weights(sc['coarse'], sc['fine'])
        outs[gemm] = m._call(inputs, 64, 1, sc['features'], u_coarse=dev(sc['u_coarse']), u_fine=dev(sc['u_fine']))
    assert MVVNeRFRenderer(64, 64, device=DEV).f32_gemm == 'split_f16'
    for gemm in ('split_f16', 'split_bf16'):
        for a, b in zip(outs[gemm], outs['mfma_f32']):
            assert float((a - b).abs().max()) < 2e-5
    if not os.environ.get('MVNERF_SPLIT_MFMA'):                                                    # (the variable pins ONE kernel for every call)
        assert not all(torch.equal(a, b) for a, b in zip(outs['split_f16'], outs['split_bf16']))   # really two kernels
    with pytest.raises(ValueError):
        MVVNeRFRenderer(64, 64, f32_gemm='tf32')


def test_products_of_the_three_fp32_grade_kernels_against_float64(monkeypatch):
    """How exact are the Dense layers of the three fp32 kernels?  Each ResNet block is recomputed in float64 FROM THE KERNEL'S OWN
    input activation (complete_output), so the difference is that block's two GEMMs alone - no geometry, no positional encoding.
    fp32 MFMA (v_mfma_f32_32x32x2_f32), six bf16 products on exactly cut operands (MVNERF_SPLIT_MFMA=bf16x6) and three fp16 products
    on two-piece operands (f16x3) must all sit at fp32 rounding level, and the fp16 form within 2x of the fp32 MFMA."""
    if os.environ.get('MVNERF_SPLIT_MFMA'):
        pytest.skip('MVNERF_SPLIT_MFMA pins one kernel for every call: nothing to compare')
    sc = make_scene(seed=91, n_views=1, height=24, width=28, n_rays=256, bias_scale=0.1)
    d = {k: dev(sc[k]) for k in ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'fine']}
    rng = np.random.default_rng(3)
    z = dev(np.sort(rng.uniform(0.3, 1.3, (1, 256, 64)).astype(np.float32), -1))
    packed, split = ops.pack_net(d['fine']), ops.pack_net_split(d['fine'])
    args = (d['rays_o'], d['rays_d'], z, d['images'], d['features'], d['intrinsics'], d['extrinsics_inv'], packed)
    net = O.unflatten_net(sc['fine'])

    def block64(x, blk):
        w1, b1, w2, b2 = (np.asarray(a, np.float64) for a in blk)
        x = x.astype(np.float64)
        return x + np.maximum(np.maximum(x, 0) @ w1 + b1, 0) @ w2 + b2

    def block_errors(acts):
        a = [t.cpu().numpy().reshape(-1, 128) for t in acts]
        errs = []
        for k, blk in zip((0, 1, 2, 4, 5, 6), net['blocks']):
            ref = block64(a[k], blk)
            errs.append(np.abs(a[k + 1] - ref).max() / np.abs(ref).max())
        return np.array(errs)

    out = {}
    _, acts = ops.field_eval(*args, complete_output=True)
    out['fp32 mfma'] = block_errors(acts)
    for mode, name in (('bf16x6', 'split_bf16'), ('f16x3', 'split_f16'), ('32x32x16', 'split_bf16_32x32x16')):
        ops.set_split_kernel(name)
        _, acts = ops.field_eval_split(*args, split, complete_output=True)
        out[mode] = block_errors(acts)
    monkeypatch.setenv('MVNERF_SPLIT_MFMA', 'bf16x6')                  # the environment variable overrides the setter
    ops.set_split_kernel('split_f16')
    _, acts = ops.field_eval_split(*args, split, complete_output=True)
    assert np.array_equal(block_errors(acts), out['bf16x6'])
    monkeypatch.delenv('MVNERF_SPLIT_MFMA')
    for name, e in out.items():
        print(f'{name:10s} per-block max |act - float64 block| / max |act|: ' + ' '.join(f'{v:.2e}' for v in e))
    eps = 2.0 ** -24
    for name, e in out.items():
        assert e.max() < 16 * eps, (name, e)                  # a 128-term fp32 dot product, twice: a few ulps
    assert out['f16x3'].max() < 2.0 * out['fp32 mfma'].max() + 2 * eps


def test_pack_net_split_f16_pieces_represent_the_weights():
    """The fp16 stream of the packed split image (behind the two bf16 streams; chunk = 24 kstep + 3 rb + piece): A0 = rn16(64 w),
    A0s = A0 / 64, A1 = rn16(64 w - A0); (A0 + A1) / 64 reproduces w to 2^-23 |w| (+ 2^-31 absolute where A1 is subnormal)."""
    sc = make_scene(seed=2, height=8, width=8, n_rays=4, bias_scale=0.1)
    net = dev(sc['fine'])
    raw = ops.pack_net_split(net)
    n16 = 58 * 24 * 1024                                                           # bytes of one 16x16x32 stream
    f16 = raw[-n16:].view(torch.float16).view(58, 8, 3, 64, 8).double()            # (kstep, rb, piece, lane, jj)
    a0, a0s, a1 = f16[:, :, 0], f16[:, :, 1], f16[:, :, 2]
    assert ((a0s - a0 / 64).abs() <= 2.0 ** -25).all()                            # exact unless A0 / 64 is subnormal (|w| < 6e-5)
    assert torch.equal(a0s[a0.abs() >= 2.0 ** -8], (a0 / 64)[a0.abs() >= 2.0 ** -8])
    w1 = net[48640:48640 + 128 * 128].view(128, 128).double()
    lane = torch.arange(64, device=DEV)
    i, g = lane & 15, lane >> 4
    jj = torch.arange(8, device=DEV)
    for t in range(4):
        f = 32 * t + torch.where(jj[None, :] < 4, 4 * g[:, None] + jj[None, :], 16 + 4 * g[:, None] + jj[None, :] - 4)   # (lane, jj)
        for rb in (0, 3, 7):
            want = w1[f, (16 * rb + i)[:, None].expand(64, 8)]
            got = (a0[10 + t, rb] + a1[10 + t, rb]) / 64
            assert ((got - want).abs() <= 2.0 ** -23 * want.abs() + 2.0 ** -31).all(), (t, rb)
            assert ((a0[10 + t, rb] / 64 - want).abs() <= 2.0 ** -11 * want.abs() + 2.0 ** -31).all()
