"""CPU checks of the language-model restatement (oracle/lmvnerf_torch.py) and of the host-side pose algebra /
read-out of thesis_clip_nerf_amd/lmvnerf.py against it (no GPU: the trunk is not called here)."""
import numpy as np
import torch

from oracle import lmvnerf_torch as L
from thesis_clip_nerf_amd import lmvnerf as P


def test_grasp_offsets_match_scipy_affine():
    ref = L.transforms_to_check(7)
    got = P.grasp_offsets(7)
    assert got.shape == (42, 4, 4)
    assert np.abs(got - ref).max() < 1e-7
    # 6 bases x 7 steps along the local z axis, centred
    assert np.allclose(ref[3][:3, 3], [0, 0.015, 0]) and np.allclose(ref[0][:3, 3], [0, 0.015, -0.015])


def test_pose_matrices():
    rng = np.random.default_rng(0)
    t = torch.tensor(rng.standard_normal((2, 5, 3)))
    q = torch.tensor(rng.standard_normal((2, 5, 4)))
    q = q / q.norm(dim=-1, keepdim=True)
    m = L.compute_matrices(t, q, 'quaternion')
    from scipy.spatial.transform import Rotation
    assert np.abs(m[0, 0, :3, :3].numpy() - Rotation.from_quat(q[0, 0].numpy()).as_matrix()).max() < 1e-12
    assert torch.allclose(P.t_m_to_h_matrix(t, P.rotation_from_quaternion(q)), m, atol=1e-12)
    r6 = torch.tensor(rng.standard_normal((2, 5, 6)))
    assert torch.allclose(P.t_m_to_h_matrix(t, P.rotation_from_6d(r6)), L.compute_matrices(t, r6, '6d'), atol=1e-12)


def keras_weights(readout):
    """GraspReadout parameters of the product module in the oracle's Keras [in,out] naming."""
    w = {}
    for i, lin in enumerate(readout.activation_downscale):
        w[f'ds{i}.k'], w[f'ds{i}.b'] = lin.weight.T, lin.bias
    w['comb.k'], w['comb.b'] = readout.combined_activation_downscale.weight.T, readout.combined_activation_downscale.bias
    for name, blk in (('b0', readout.block_0), ('b1', readout.block_1)):
        w[f'{name}.l0.k'], w[f'{name}.l0.b'] = blk.layer_0.weight.T, blk.layer_0.bias
        w[f'{name}.l1.k'], w[f'{name}.l1.b'] = blk.layer_1.weight.T, blk.layer_1.bias
    w['b0.sc.k'] = readout.block_0.shortcut.weight.T
    w['out.k'], w['out.b'] = readout.output_layer.weight.T, readout.output_layer.bias
    return w


def test_grasp_readout_matches_restatement():
    torch.manual_seed(0)
    ro = P.GraspReadout(42).double()
    acts = [torch.randn(2, 3, 42, 128, dtype=torch.float64) for _ in range(4)]
    got = ro(acts)
    ref = L.grasp_readout(keras_weights(ro), acts)
    assert got.shape == (2, 3)
    assert torch.allclose(got, ref, atol=1e-10)


def test_losses():
    a = torch.tensor([[1.0, 2.0, 2.0], [0.0, 0.0, 0.0]])
    b = torch.tensor([[2.0, 4.0, 4.0], [1.0, 0.0, 0.0]])
    assert abs(float(P.cosine_similarity_loss(a, b)) + 0.5) < 1e-6          # rows: cos = 1 and 0 -> -mean = -0.5
    assert abs(float(L.cosine_similarity(a.double(), b.double())) + 0.5) < 1e-12
    p = torch.tensor([[0.5, 0.5]])
    q = torch.tensor([[0.25, 0.75]])
    assert abs(float(P.kl_divergence(p, q)) - (0.5 * np.log(2) + 0.5 * np.log(0.5 / 0.75))) < 1e-6
