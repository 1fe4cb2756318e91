"""fp32-grade field pass on the bf16 matrix pipe (csrc/field_eval_split.hip: every GEMM operand cut exactly into three bf16
pieces, six MFMAs per product, fp32 accumulation).  It replaces the fp32-MFMA kernel at the SAME bars: per-sample outputs
and activations against the NumPy fp32 oracle, integer tap indices bit-exact, rendered outputs within the 1e-4 of
north_star, on the direct and the texel-table path, 1..3 views, ragged tiles."""
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
    for gemm in ('split_bf16', 'mfma_f32'):
        m = MVVNeRFRenderer(64, 64, n_views=1, near=sc['near'], far=sc['far'], device=DEV, f32_gemm=gemm)
        m.set_weights(sc['coarse'], sc['fine'])
        outs[gemm] = m._call(inputs, 64, 1, sc['features'], u_coarse=dev(sc['u_coarse']), u_fine=dev(sc['u_fine']))
    assert MVVNeRFRenderer(64, 64, device=DEV).f32_gemm == 'split_bf16'
    for a, b in zip(outs['split_bf16'], outs['mfma_f32']):
        assert float((a - b).abs().max()) < 2e-5
    with pytest.raises(ValueError):
        MVVNeRFRenderer(64, 64, f32_gemm='tf32')
