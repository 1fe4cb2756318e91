"""Feature-map producer (thesis_clip_nerf_amd/encoders.py): shapes and layout of the reference's encoder
prologue (layers.py:232-259, legacy_layers.py:154-191, model_v0.py:75-86) on a tiny configuration, the Keras-variable
importers, the feature-side optimizer schedule.  CPU only; the end-to-end gradient into the producer through the HIP
render path is in tests/test_gpu_model.py."""
import numpy as np
import pytest
import torch

from thesis_clip_nerf_amd import encoders as E

TINY = dict(transformer_image_size=(32, 32), patch_size=16, embed_dim=32, num_heads=4, hooks=(1, 2, 3, 4), features=(4, 8, 16, 32))


def test_tf_same_padding_rule():
    assert E._same_pad(480, 7, 2) == (2, 3) and E._same_pad(640, 7, 2) == (2, 3)       # more padding at the end, as TensorFlow
    assert E._same_pad(14, 3, 2) == (0, 1) and E._same_pad(15, 3, 1) == (1, 1) and E._same_pad(8, 1, 1) == (0, 0)
    conv = E.SameConv2d(3, 5, 7, stride=2, bias=False)
    assert conv(torch.zeros(1, 3, 48, 64)).shape == (1, 5, 24, 32)


def test_producer_shapes_layout_and_gradient_flow():
    torch.manual_seed(0)
    prod = E.FeatureProducer(original_image_size=(32, 48), **TINY)
    images = torch.rand(3, 32, 48, 3)
    out = prod(images)
    assert out.shape == (3, 32, 48, 256) and out.dtype == torch.float32 and out.is_contiguous()      # NHWC, what the gather reads
    vis = prod.visual_features(images)
    assert vis.shape == (3, 256, 16, 24)                                                          # half resolution, [ViT 128 | conv 128]
    out.square().mean().backward()
    trainable = prod.trainable_parameters()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in trainable)
    assert any(p.grad.abs().max() > 0 for p in prod.visual_features.conv_features.parameters())
    assert any(p.grad.abs().max() > 0 for p in prod.visual_features.vision_transformer.vit.parameters())
    assert prod.combine_clip_visual.conv.weight.grad is not None                                 # reachable, but not in the optimizer list (Q9)
    ids = {id(p) for p in trainable}
    assert id(prod.combine_clip_visual.conv.weight) not in ids
    prod16 = E.FeatureProducer(original_image_size=(32, 48), out_dtype=torch.bfloat16, **TINY)
    o16 = prod16(images)
    assert o16.dtype == torch.bfloat16 and o16.is_contiguous() and o16.shape == (3, 32, 48, 256)
    with pytest.raises(ValueError):
        E.FeatureProducer(original_image_size=(31, 48), **TINY)


def test_reference_sized_modules_have_the_reference_parameter_counts():
    """ViT-B/16 blocks as the reference builds them (layers.py:72-86): BN(768) + MHA(12 x 64) + LN + MLP(3072)."""
    blk = E.TransformerBlock(12, 768, 4)
    mha = 4 * (768 * 768 + 768)
    assert E.count_parameters(blk) == 2 * 768 + mha + 2 * 768 + (768 * 3072 + 3072) + (3072 * 768 + 768)
    conv = E.ConvolutionalEncoder(256)
    # stem 7x7x3x64 + BN, downsample 1x1x64x128 + BN, 3 blocks of 2 biased 3x3 convs + ONE BatchNormalization each
    want = 7 * 7 * 3 * 64 + 2 * 64 + 64 * 128 + 2 * 128 + (3 * 3 * 64 * 128 + 128 + 3 * 3 * 128 * 128 + 128 + 2 * 128) + \
        2 * (2 * (3 * 3 * 128 * 128 + 128) + 2 * 128)
    assert E.count_parameters(conv) == want
    assert E.count_parameters(E.CombineCLIPVisualV0()) == 512 * 256


def test_block_norm_is_shared_and_uses_batch_statistics():
    torch.manual_seed(1)
    blk = E.Block(8, 8).eval()                       # eval mode must not matter: the reference hard-codes training=True
    x = torch.randn(4, 8, 6, 6)
    y = blk(x)
    h = blk.conv_1(x)
    hn = (h - h.mean((0, 2, 3), keepdim=True)) / torch.sqrt(h.var((0, 2, 3), unbiased=False, keepdim=True) + 1e-3)
    h2 = blk.conv_2(torch.relu(hn * blk.norm_1.weight.view(1, -1, 1, 1) + blk.norm_1.bias.view(1, -1, 1, 1)))
    h2n = (h2 - h2.mean((0, 2, 3), keepdim=True)) / torch.sqrt(h2.var((0, 2, 3), unbiased=False, keepdim=True) + 1e-3)
    want = torch.relu(h2n * blk.norm_1.weight.view(1, -1, 1, 1) + blk.norm_1.bias.view(1, -1, 1, 1) + x)
    assert torch.allclose(y, want, atol=1e-5)


def test_transformer_block_residual_quirk():
    torch.manual_seed(2)
    blk = E.TransformerBlock(2, 8, 2).eval()
    x = torch.randn(3, 5, 8)
    n1 = blk.layer_norm_1(x.transpose(1, 2)).transpose(1, 2)
    mid = blk.layer_norm_2(x + blk.attention(n1))
    want = x + blk.dense_1(torch.nn.functional.gelu(blk.dense_0(mid)))          # + x, not + (x + attention)
    assert torch.allclose(blk(x), want, atol=1e-6)


def test_flat_net_importer_round_trip():
    from thesis_clip_nerf_amd.synthetic import glorot_net
    flat = glorot_net(np.random.default_rng(0), bias_scale=0.1)
    emb, ro = E.keras_from_flat_net(flat)
    assert [a.shape for a in emb] == E.MLP_EMBEDDING_SHAPES and [a.shape for a in ro] == E.MLP_READOUT_SHAPES
    assert emb[0].shape == (379, 128) and emb[2].shape == (128, 128) and ro[0].shape == (128, 4)
    np.testing.assert_array_equal(E.flat_net_from_keras(emb, ro), flat)
    # Dense semantics: y = x @ kernel + bias with the kernel as stored
    from oracle import mvnerf_oracle as O
    net = O.unflatten_net(flat)
    np.testing.assert_array_equal(net['W0'], emb[0])
    np.testing.assert_array_equal(net['blocks'][5][2], emb[2 + 4 * 5 + 2])
    with pytest.raises(ValueError):
        E.flat_net_from_keras(emb[1:], ro)


def test_keras_layout_importers():
    rng = np.random.default_rng(3)
    conv = E.SameConv2d(3, 5, 3)
    k = rng.standard_normal((3, 3, 3, 5)).astype(np.float32)                   # Keras HWIO
    E.load_conv(conv, k, np.zeros(5, np.float32))
    x = torch.randn(1, 3, 6, 6)
    y = conv(x)
    xp = np.pad(x[0].numpy(), ((0, 0), (1, 1), (1, 1)))
    want = sum(xp[c, 2 + a:3 + a, 3 + b:4 + b] * k[a, b, c, 1] for a in range(3) for b in range(3) for c in range(3))
    assert abs(float(y[0, 1, 2, 3]) - float(want)) < 1e-5
    lin = torch.nn.Linear(4, 6)
    kd, bd = rng.standard_normal((4, 6)).astype(np.float32), rng.standard_normal(6).astype(np.float32)
    E.load_dense(lin, kd, bd)
    v = rng.standard_normal(4).astype(np.float32)
    np.testing.assert_allclose(lin(torch.from_numpy(v)).detach().numpy(), v @ kd + bd, atol=1e-5)
    ct = torch.nn.ConvTranspose2d(2, 2, 4, stride=4)
    kt = rng.standard_normal((4, 4, 2, 2)).astype(np.float32)                  # Keras (kh, kw, out, in)
    E.load_conv_transpose(ct, kt, np.zeros(2, np.float32))
    xi = torch.zeros(1, 2, 2, 2)
    xi[0, 1, 1, 0] = 1.0
    yt = ct(xi)[0].detach().numpy()                                            # y[o, 4i + a, 4j + b] = K[a, b, o, c]
    np.testing.assert_allclose(yt[:, 4:8, 0:4], kt[:, :, :, 1].transpose(2, 0, 1), atol=1e-6)
    mha = E.KerasMHA(8, 2)
    qk = rng.standard_normal((8, 2, 4)).astype(np.float32)
    E.load_mha(mha, qk, np.zeros((2, 4)), qk, np.zeros((2, 4)), qk, np.zeros((2, 4)), rng.standard_normal((2, 4, 8)).astype(np.float32), np.zeros(8))
    np.testing.assert_allclose(mha.q.weight.detach().numpy(), qk.reshape(8, 8).T)
    enc = E.ConvolutionalEncoder(256)
    shapes = [(1, 1, 64, 128)] + [(128,)] * 4 + [(7, 7, 3, 64)] + [(64,)] * 4
    for cin in (64, 128, 128):
        shapes += [(3, 3, cin, 128), (128,), (3, 3, 128, 128), (128,)] + [(128,)] * 4
    E.load_convolutional_encoder(enc, [rng.standard_normal(s).astype(np.float32) if len(s) > 1 else np.abs(rng.standard_normal(s)).astype(np.float32) for s in shapes],
                                 order='creation')
    # the Keras `.weights` order (trainable before non-trainable inside a plain Layer; Block 1 carries the downsample branch): a list
    # exported in that order loads back onto the same variables, and the two orders really differ
    w_keras = E.convolutional_encoder_weights(enc, order='keras')
    assert [tuple(a.shape) for a in w_keras[:5]] == [(7, 7, 3, 64), (64,), (64,), (64,), (64,)]                  # stem first
    assert tuple(w_keras[11].shape) == (1, 1, 64, 128)                                                           # downsample kernel inside Block 1
    enc2 = E.ConvolutionalEncoder(256)
    E.load_convolutional_encoder(enc2, w_keras)                                                                  # default: order='keras'
    for a, b in zip(enc.state_dict().items(), enc2.state_dict().items()):
        if 'num_batches_tracked' not in a[0]:
            assert torch.equal(a[1], b[1]), a[0]
    with pytest.raises(Exception):
        E.load_convolutional_encoder(E.ConvolutionalEncoder(256), w_keras, order='creation')                     # wrong order: shape mismatch
    blk = E.TransformerBlock(2, 8, 2)
    E.load_transformer_block(blk, [np.ones(8)] * 4 + [qk, np.zeros((2, 4))] * 3 + [np.zeros((2, 4, 8)), np.zeros(8)] + [np.ones(8), np.zeros(8)] +
                             [np.zeros((8, 16)), np.zeros(16), np.zeros((16, 8)), np.zeros(8)])


def test_feature_optimizer_schedule_matches_warmup_scheduler():
    from thesis_clip_nerf_amd.nerf_utils import WarmupScheduler
    prod = E.FeatureProducer(original_image_size=(32, 48), **TINY)
    opt, sched = E.make_encoder_optimizer(prod, 1e-5, 10, 20)
    ref = WarmupScheduler(1e-5, 10, 20)
    for step in range(25):
        assert abs(opt.param_groups[0]['lr'] - ref(step)) < 1e-12, step       # lr used by the update taken at `step`
        opt.step()
        sched.step()
    assert opt.defaults['eps'] == 1e-7


def test_keras_adam_is_the_keras_update_not_torch_adam():
    """p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps): against the closed form over three steps, and different from
    torch.optim.Adam where it must be (eps added after the bias correction there)."""
    g = [np.array([1e-8, 2e-3, -0.5], np.float64), np.array([3e-8, -1e-3, 0.25], np.float64), np.array([-2e-8, 5e-4, 0.1], np.float64)]
    p = torch.nn.Parameter(torch.tensor([0.1, -0.2, 0.3], dtype=torch.float64))
    q = torch.nn.Parameter(p.detach().clone())
    opt, ref_opt = E.KerasAdam([p], lr=1e-2, eps=1e-7), torch.optim.Adam([q], lr=1e-2, eps=1e-7)
    want, m, v = p.detach().numpy().copy(), np.zeros(3), np.zeros(3)
    for t, gt in enumerate(g, 1):
        p.grad, q.grad = torch.from_numpy(gt.copy()), torch.from_numpy(gt.copy())
        opt.step()
        ref_opt.step()
        m, v = 0.9 * m + 0.1 * gt, 0.999 * v + 0.001 * gt * gt
        want = want - 1e-2 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-7)
        np.testing.assert_allclose(p.detach().numpy(), want, rtol=0, atol=1e-15)
    assert abs(p[0].item() - q[0].item()) > 1e-4          # tiny gradients: the two epsilons act differently
    assert abs(p[2].item() - q[2].item()) < 1e-6          # large gradients: the same step
