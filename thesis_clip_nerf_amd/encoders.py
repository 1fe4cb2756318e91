"""Feature-map producer in front of the render hot path: PyTorch-ROCm modules with the structure of the
reference's `VisualFeatures` (src/lib/mvnerf/layers.py:232-259: conv encoder :36-57 + ViT-B/16 with a DPT-style decoder
:60-229) and `CombineCLIPVisualV0` (src/lib/mvnerf/legacy_layers.py:154-191), emitting `combined_features`
(B, V, H, W, 256) NHWC - contiguous, fp32 or bf16 - i.e. exactly the layout `mvnerf_project_texels` / the gather read,
and a Keras-variable importer that takes plain arrays (no TensorFlow needed here).

What is and is not reproduced:
* layer structure, shapes, paddings (TensorFlow 'same' with stride 2 pads more at the END), Keras defaults
  (BatchNormalization eps 1e-3 / momentum 0.99, LayerNormalization eps 1e-3, exact-erf gelu, glorot / zeros init) and the
  reference's quirks: `Block` applies ONE BatchNormalization to both convolutions and always in training mode
  (layers.py:10-14,22,26), the transformer block's first norm is a BatchNormalization over the embedding axis and its second
  residual adds the block INPUT (layers.py:76,88-94).  Parity unpinned: TensorFlow is not available and the reference ships
  no fixtures for these layers; the restated numerics are checked for shape / layout / gradient flow only.
* CLIP RN50 (`clip_visual`, an external frozen SavedModel whose weights are not available offline) is NOT rebuilt: its
  stage-1 map (B, 56, 56, 256) is an input of :class:`CombineCLIPVisualV0`; :class:`SyntheticCLIPStage1` is a frozen random
  stand-in for synthetic runs.
* Sizes are constructor arguments (the reference hard-codes 480x640 / 224 / ViT-B): tests run a tiny configuration.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


def _same_pad(size, k, s):
    """TensorFlow 'same' padding for one spatial axis: (before, after)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


class SameConv2d(nn.Conv2d):
    """Conv2D(padding='same') with TensorFlow's asymmetric padding when the stride is > 1.  Keras init: glorot_uniform, zeros."""

    def __init__(self, c_in, c_out, k, stride=1, bias=True):
        super().__init__(c_in, c_out, k, stride=stride, padding=0, bias=bias)
        nn.init.xavier_uniform_(self.weight)
        if bias:
            nn.init.zeros_(self.bias)

    def forward(self, x):
        ph = _same_pad(x.shape[2], self.kernel_size[0], self.stride[0])
        pw = _same_pad(x.shape[3], self.kernel_size[1], self.stride[1])
        return super().forward(F.pad(x, (pw[0], pw[1], ph[0], ph[1])))


def _keras_bn(c):
    return nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)          # Keras momentum 0.99 = torch momentum 0.01


class Block(nn.Module):
    """layers.py:7-33.  One BatchNormalization serves both convolutions (the attribute is assigned twice in the reference)
    and it always normalises with batch statistics (`training=True` is hard-coded there)."""

    def __init__(self, c_in, n_features, downsample=None):
        super().__init__()
        self.conv_1 = SameConv2d(c_in, n_features, 3)
        self.conv_2 = SameConv2d(n_features, n_features, 3)
        self.norm_1 = _keras_bn(n_features)
        self.downsample = downsample

    def _norm(self, x):
        return F.batch_norm(x, None, None, self.norm_1.weight, self.norm_1.bias, True, 0.0, self.norm_1.eps)

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        out = F.relu(self._norm(self.conv_1(x)))
        out = self._norm(self.conv_2(out))
        return F.relu(out + skip)


class ConvolutionalEncoder(nn.Module):
    """layers.py:36-57: 7x7/2 conv -> BN -> ReLU -> 3 residual blocks of n_features / 2 channels; (H, W) -> (H/2, W/2)."""

    def __init__(self, n_features=256, stem=64):
        super().__init__()
        half = n_features // 2
        downsample = nn.Sequential(SameConv2d(stem, half, 1, bias=False), _keras_bn(half))
        self.conv_features = nn.Sequential(SameConv2d(3, stem, 7, stride=2, bias=False), _keras_bn(stem), nn.ReLU(),
                                           Block(stem, half, downsample), Block(half, half), Block(half, half))

    def forward(self, x):
        return self.conv_features(x)


class KerasMHA(nn.Module):
    """tf.keras.layers.MultiHeadAttention(num_heads, key_dim = value_dim = embed / heads) on (x, x): biased q/k/v/out projections."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.h, self.d = num_heads, embed_dim // num_heads
        self.q = nn.Linear(embed_dim, self.h * self.d)
        self.k = nn.Linear(embed_dim, self.h * self.d)
        self.v = nn.Linear(embed_dim, self.h * self.d)
        self.o = nn.Linear(self.h * self.d, embed_dim)
        for lin in (self.q, self.k, self.v, self.o):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)

    def forward(self, x):
        b, n, _ = x.shape
        split = lambda t: t.view(b, n, self.h, self.d).transpose(1, 2)
        out = F.scaled_dot_product_attention(split(self.q(x)), split(self.k(x)), split(self.v(x)))      # softmax(q k^T / sqrt(d)) v
        return self.o(out.transpose(1, 2).reshape(b, n, self.h * self.d))


class TransformerBlock(nn.Module):
    """layers.py:72-95, quirks kept: the first norm is a BatchNormalization over the embedding axis; `x = inputs + attn`,
    then `inputs + mlp(LayerNorm(x))` - the second residual adds the block input, not x."""

    def __init__(self, num_heads=12, embed_dim=768, mlp_ratio=4):
        super().__init__()
        self.layer_norm_1 = nn.BatchNorm1d(embed_dim, eps=1e-3, momentum=0.01)
        self.attention = KerasMHA(embed_dim, num_heads)
        self.layer_norm_2 = nn.LayerNorm(embed_dim, eps=1e-3)
        self.dense_0 = nn.Linear(embed_dim, embed_dim * mlp_ratio)
        self.dense_1 = nn.Linear(embed_dim * mlp_ratio, embed_dim)
        for lin in (self.dense_0, self.dense_1):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)

    def forward(self, inputs):
        x = self.layer_norm_1(inputs.transpose(1, 2)).transpose(1, 2)
        x = inputs + self.attention(x)
        x = self.layer_norm_2(x)
        return inputs + self.dense_1(F.gelu(self.dense_0(x)))


class VisionTransformer(nn.Module):
    """layers.py:98-157 (skip_classification=True): patch embedding, class token, learned positions, `sum(hooks)`-free grouping of
    the blocks at the hook depths; returns the token maps after each group."""

    def __init__(self, img_size=(224, 224), patch_size=16, embed_dim=768, mlp_ratio=4, hooks=(3, 6, 9, 12), num_heads=12):
        super().__init__()
        self.grid_size = (img_size[0] // patch_size, img_size[1] // patch_size)
        self.patch_embed = nn.Conv2d(3, embed_dim, patch_size, stride=patch_size)
        nn.init.xavier_uniform_(self.patch_embed.weight)
        nn.init.zeros_(self.patch_embed.bias)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embedding = nn.Parameter(0.02 * torch.randn(1, self.grid_size[0] * self.grid_size[1] + 1, embed_dim))
        depth = [hooks[0]] + [hooks[i] - hooks[i - 1] for i in range(1, len(hooks))]
        self.transformer_blocks = nn.ModuleList(
            nn.Sequential(*[TransformerBlock(num_heads, embed_dim, mlp_ratio) for _ in range(d)]) for d in depth)

    def forward(self, x):
        x = self.patch_embed(x).flatten(2).transpose(1, 2)                     # b (h w) c
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], 1) + self.pos_embedding
        feats = []
        for blocks in self.transformer_blocks:
            x = blocks(x)
            feats.append(x)
        return feats


def _up(x, s):
    return F.interpolate(x, scale_factor=s, mode='bilinear', align_corners=False)     # UpSampling2D(interpolation='bilinear')


def _resize(x, size):
    return F.interpolate(x, size=size, mode='bilinear', align_corners=False)          # Resizing(interpolation='bilinear')


class VisionTransformerEncoder(nn.Module):
    """layers.py:160-229: four token maps -> post-processing to 4 / 2 / 1 / 0.5 x the patch grid -> 3x3 decode convs to
    n_features -> bilinear up-sampling to 8 x the grid -> concat -> ReLU, conv, ReLU, conv to n_features / 2."""

    def __init__(self, img_size=(224, 224), patch_size=16, embed_dim=768, n_features=256, mlp_ratio=4, hooks=(3, 6, 9, 12),
                 features=(48, 96, 192, 384), num_heads=12):
        super().__init__()
        self.vit = VisionTransformer(img_size, patch_size, embed_dim, mlp_ratio, hooks, num_heads)
        f = features

        def convT(c, k):
            m = nn.ConvTranspose2d(c, c, k, stride=k)
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)
            return m
        self.post_process_1 = nn.Sequential(SameConv2d(embed_dim, f[0], 1), convT(f[0], 4))
        self.post_process_2 = nn.Sequential(SameConv2d(embed_dim, f[1], 1), convT(f[1], 2))
        self.post_process_3 = SameConv2d(embed_dim, f[2], 1)
        self.post_process_4 = nn.Sequential(SameConv2d(embed_dim, f[3], 1), SameConv2d(f[3], f[3], 3, stride=2))
        self.conv_decode = nn.ModuleList(SameConv2d(c, n_features, 3, bias=False) for c in f)
        self.output_conv = nn.Sequential(nn.ReLU(), SameConv2d(4 * n_features, n_features, 3), nn.ReLU(),
                                         SameConv2d(n_features, n_features // 2, 3))

    def forward(self, x):
        gh, gw = self.vit.grid_size
        feats = [t[:, 1:].transpose(1, 2).reshape(t.shape[0], -1, gh, gw) for t in self.vit(x)]
        post = (self.post_process_1, self.post_process_2, self.post_process_3, self.post_process_4)
        maps = [_up(dec(pp(t)), s) for t, pp, dec, s in zip(feats, post, self.conv_decode, (2, 4, 8, 16))]
        return self.output_conv(torch.cat(maps, 1))


class VisualFeatures(nn.Module):
    """layers.py:232-259: images (N, H, W, 3) in [0, 1] -> (N, H/2, W/2, n_features) = [ViT latents resized | conv features]."""

    def __init__(self, n_features=256, original_image_size=(480, 640), transformer_image_size=(224, 224), **vit_kw):
        super().__init__()
        self.conv_features = ConvolutionalEncoder(n_features)
        self.vision_transformer = VisionTransformerEncoder(img_size=transformer_image_size, n_features=n_features, **vit_kw)
        self.transformer_image_size = tuple(transformer_image_size)
        self.half_size = (original_image_size[0] // 2, original_image_size[1] // 2)

    def forward(self, images_nhwc):
        x = images_nhwc.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        latents = _resize(self.vision_transformer(_resize(x, self.transformer_image_size)), self.half_size)
        return torch.cat([latents, self.conv_features(x)], 1)                  # NCHW view, channels-last storage


class CombineCLIPVisualV0(nn.Module):
    """legacy_layers.py:154-191: [resize(clip stage-1 map, (H/2, W/2)) | visual features] -> 1x1 conv (no bias) -> x2 bilinear."""

    def __init__(self, half_size=(240, 320), clip_channels=256, visual_channels=256, filters=256):
        super().__init__()
        self.conv = SameConv2d(clip_channels + visual_channels, filters, 1, bias=False)
        self.half_size = tuple(half_size)

    def forward(self, clip_256, visual_features):
        fusion = torch.cat([_resize(clip_256, self.half_size), visual_features], 1)
        return _up(self.conv(fusion), 2)


class SyntheticCLIPStage1(nn.Module):
    """Frozen random stand-in for the stage-1 output (N, 256, 56, 56) of the CLIP RN50 visual trunk (clip/model.py:21-28; the
    SavedModel and its weights are not available offline).  Not trained, carries no semantics."""

    def __init__(self, channels=256, out_size=(56, 56), seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer('w', torch.randn(channels, 3, 7, 7, generator=g) * 0.1)
        self.out_size = tuple(out_size)

    @torch.no_grad()
    def forward(self, x_nchw):
        return F.relu(F.adaptive_avg_pool2d(F.conv2d(x_nchw, self.w, stride=2, padding=3), self.out_size))


class FeatureProducer(nn.Module):
    """`MVVNeRFRenderer.call`'s encoder prologue (model_v0.py:75-86) as one callable: images (N, H, W, 3) in [0, 1] ->
    combined_features (N, H, W, 256), NHWC-contiguous in `out_dtype` (fp32, or bf16 to halve the gather traffic).  Use it as
    `MVVNeRFRenderer(feature_encoder=producer)`; `trainable_parameters()` is the reference's optimizer list for the feature
    side (train_nerf.py:27-32: vision_transformer + conv_features at lr 1e-5; combine_clip_visual is NOT in it)."""

    def __init__(self, original_image_size=(480, 640), n_features=256, clip_stage1=None, out_dtype=torch.float32, **visual_kw):
        super().__init__()
        h, w = original_image_size
        if h % 2 or w % 2:
            raise ValueError('image height and width must be even (the encoders work at half resolution)')
        self.visual_features = VisualFeatures(n_features, original_image_size, **visual_kw)
        self.combine_clip_visual = CombineCLIPVisualV0((h // 2, w // 2), 256, n_features, 256)
        self.clip_stage1 = clip_stage1 if clip_stage1 is not None else SyntheticCLIPStage1()
        self.out_dtype = out_dtype

    def trainable_parameters(self):
        return list(self.visual_features.vision_transformer.parameters()) + list(self.visual_features.conv_features.parameters())

    def forward(self, images_nhwc):
        x = images_nhwc.permute(0, 3, 1, 2)
        visual = self.visual_features(images_nhwc)
        with torch.no_grad():
            clip_256 = self.clip_stage1(x)                                   # frozen (layers.py:550-561)
        fused = self.combine_clip_visual(clip_256, visual)                      # (N, 256, H, W), channels-last storage
        out = fused.permute(0, 2, 3, 1)
        out = out.to(self.out_dtype)
        return out if out.is_contiguous() else out.contiguous()


# ---- Keras variables -> this package (plain arrays; no TensorFlow needed) ------------------------------------------------
# The reference stores one TensorFlow checkpoint per sub-model (model_v0.py:199-214).  Dump each in the reference's own
# environment with, e.g.,
#     r = tf.train.load_checkpoint(f'{path}_coarse_embedding'); np.savez(out, **{k: r.get_tensor(k) for k, _ in tf.train.list_variables(...)})
# or simply `np.savez(out, *[w.numpy() for w in model.coarse_embedding.weights])` - the functions below take the arrays in
# `layer.weights` order (Keras creation order), which for these sub-models is:
#   *_embedding (MVResNetMLPNeRFEmbedding, layers.py:334-379): Dense0 kernel (379,128), bias (128); then per ResNetMLPBlock
#       (6 of them, :262-298): dense_1 kernel (128,128), bias, dense_2 kernel (128,128), bias            -> 246 784 floats
#   *_readout (RenderReadout, :382-397): Dense kernel (128,4), bias (4)                                   ->     516 floats
# which is exactly the flat Keras-order buffer `MVVNeRFRenderer.set_weights` takes (kernel[in,out] row-major, then bias).

MLP_EMBEDDING_SHAPES = [(379, 128), (128,)] + [s for _ in range(6) for s in ((128, 128), (128,), (128, 128), (128,))]
MLP_READOUT_SHAPES = [(128, 4), (4,)]


def flat_net_from_keras(embedding_weights, readout_weights):
    """`[w.numpy() for w in embedding.weights]`, `[... readout.weights]` -> the 247 300-float buffer of one MLP."""
    parts = []
    for arrays, shapes, what in ((embedding_weights, MLP_EMBEDDING_SHAPES, 'embedding'), (readout_weights, MLP_READOUT_SHAPES, 'readout')):
        arrays = [np.asarray(a, dtype=np.float32) for a in arrays]
        if [a.shape for a in arrays] != shapes:
            raise ValueError(f'{what}: variable shapes {[a.shape for a in arrays]} do not match the Keras creation order {shapes}')
        parts += [a.reshape(-1) for a in arrays]
    flat = np.concatenate(parts)
    assert flat.size == 247300
    return flat


def keras_from_flat_net(flat):
    """Inverse of :func:`flat_net_from_keras`: (embedding arrays, readout arrays) for `layer.set_weights`."""
    flat = np.asarray(flat, dtype=np.float32).reshape(-1)
    out, pos = [], 0
    for shape in MLP_EMBEDDING_SHAPES + MLP_READOUT_SHAPES:
        n = int(np.prod(shape))
        out.append(flat[pos:pos + n].reshape(shape).copy())
        pos += n
    return out[:len(MLP_EMBEDDING_SHAPES)], out[len(MLP_EMBEDDING_SHAPES):]


def _t(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float32))


def load_conv(conv, kernel_hwio, bias=None):
    """Keras Conv2D kernel (kh, kw, in, out) -> torch (out, in, kh, kw)."""
    conv.weight.data.copy_(_t(kernel_hwio).permute(3, 2, 0, 1))
    if bias is not None:
        conv.bias.data.copy_(_t(bias))


def load_conv_transpose(conv, kernel_hwoi, bias):
    """Keras Conv2DTranspose kernel (kh, kw, out, in) -> torch ConvTranspose2d (in, out, kh, kw)."""
    conv.weight.data.copy_(_t(kernel_hwoi).permute(3, 2, 0, 1))
    conv.bias.data.copy_(_t(bias))


def load_dense(lin, kernel_io, bias):
    """Keras Dense kernel (in, out) -> torch Linear (out, in)."""
    lin.weight.data.copy_(_t(kernel_io).T)
    lin.bias.data.copy_(_t(bias))


def load_batchnorm(bn, gamma, beta, moving_mean, moving_var):
    bn.weight.data.copy_(_t(gamma))
    bn.bias.data.copy_(_t(beta))
    bn.running_mean.data.copy_(_t(moving_mean))
    bn.running_var.data.copy_(_t(moving_var))


def load_mha(mha, q_k, q_b, k_k, k_b, v_k, v_b, o_k, o_b):
    """Keras MultiHeadAttention: query/key/value kernels (embed, heads, dim) + biases (heads, dim); output kernel (heads, dim, embed)."""
    for lin, k, b in ((mha.q, q_k, q_b), (mha.k, k_k, k_b), (mha.v, v_k, v_b)):
        k = np.asarray(k, dtype=np.float32)
        load_dense(lin, k.reshape(k.shape[0], -1), np.asarray(b).reshape(-1))
    o_k = np.asarray(o_k, dtype=np.float32)
    load_dense(mha.o, o_k.reshape(-1, o_k.shape[-1]), o_b)


def load_combine_clip_visual(module, weights):
    """CombineCLIPVisualV0.weights = [conv kernel (1, 1, 512, 256)]."""
    (kernel,) = weights
    load_conv(module.conv, kernel)


def load_convolutional_encoder(enc, weights, order='keras'):
    """ConvolutionalEncoder variables (layers.py:36-57) as plain arrays.

    order='keras' (default): the order of Keras' `ConvolutionalEncoder.weights`, i.e. `[w.numpy() for w in enc.weights]`: a
    container lists the weights of its sub-layers in attribute-creation order, and a plain `Layer` (the reference's `Block`)
    lists ALL its trainable variables first and its non-trainable ones (the BatchNormalization moving statistics) after them:
      Sequential: stem conv kernel; stem BN gamma, beta, moving_mean, moving_var;
      Block 1 (has the downsample branch): conv_1 kernel, bias; conv_2 kernel, bias; norm_1 gamma, beta; downsample conv kernel;
                                           downsample BN gamma, beta | norm_1 moving_mean, moving_var; downsample BN moving_mean, moving_var;
      Blocks 2, 3: conv_1 kernel, bias; conv_2 kernel, bias; norm_1 gamma, beta | norm_1 moving_mean, moving_var.
    order='creation' (round 2's reading): downsample conv kernel, downsample BN (gamma, beta, mean, var); stem conv kernel, stem BN (4);
    per Block conv_1 kernel, bias, conv_2 kernel, bias, norm_1 (gamma, beta, mean, var).
    Neither order could be checked against a TensorFlow dump here (no TF, no checkpoint in the reference): PARITY UNPINNED.  Prefer
    keying by variable name (tf.train.list_variables) when a real checkpoint is at hand; a wrong order fails on a shape mismatch
    for the stem / downsample variables but NOT between same-shaped Block variables."""
    w = list(weights)
    seq = enc.conv_features
    blocks = (seq[3], seq[4], seq[5])
    down = seq[3].downsample
    if order == 'creation':
        load_conv(down[0], w.pop(0))
        load_batchnorm(down[1], *[w.pop(0) for _ in range(4)])
        load_conv(seq[0], w.pop(0))
        load_batchnorm(seq[1], *[w.pop(0) for _ in range(4)])
        for blk in blocks:
            load_conv(blk.conv_1, w.pop(0), w.pop(0))
            load_conv(blk.conv_2, w.pop(0), w.pop(0))
            load_batchnorm(blk.norm_1, *[w.pop(0) for _ in range(4)])
    elif order == 'keras':
        load_conv(seq[0], w.pop(0))
        load_batchnorm(seq[1], *[w.pop(0) for _ in range(4)])
        for i, blk in enumerate(blocks):
            load_conv(blk.conv_1, w.pop(0), w.pop(0))
            load_conv(blk.conv_2, w.pop(0), w.pop(0))
            gamma, beta = w.pop(0), w.pop(0)
            if i == 0:
                load_conv(down[0], w.pop(0))
                d_gamma, d_beta = w.pop(0), w.pop(0)
            mean, var = w.pop(0), w.pop(0)
            load_batchnorm(blk.norm_1, gamma, beta, mean, var)
            if i == 0:
                load_batchnorm(down[1], d_gamma, d_beta, w.pop(0), w.pop(0))
    else:
        raise ValueError(f"order: {order!r}, expected 'keras' or 'creation'")
    if w:
        raise ValueError(f'{len(w)} unused variables')


def convolutional_encoder_weights(enc, order='keras'):
    """The inverse of load_convolutional_encoder: this module's variables as Keras-layout arrays in the same order."""
    k_conv = lambda c: c.weight.detach().permute(2, 3, 1, 0).cpu().numpy()
    bn4 = lambda b: [b.weight.detach().cpu().numpy(), b.bias.detach().cpu().numpy(), b.running_mean.cpu().numpy(), b.running_var.cpu().numpy()]
    seq = enc.conv_features
    blocks = (seq[3], seq[4], seq[5])
    down = seq[3].downsample
    out = []
    if order == 'creation':
        out += [k_conv(down[0])] + bn4(down[1]) + [k_conv(seq[0])] + bn4(seq[1])
        for blk in blocks:
            out += [k_conv(blk.conv_1), blk.conv_1.bias.detach().cpu().numpy(), k_conv(blk.conv_2), blk.conv_2.bias.detach().cpu().numpy()] + bn4(blk.norm_1)
        return out
    out += [k_conv(seq[0])] + bn4(seq[1])
    for i, blk in enumerate(blocks):
        out += [k_conv(blk.conv_1), blk.conv_1.bias.detach().cpu().numpy(), k_conv(blk.conv_2), blk.conv_2.bias.detach().cpu().numpy()]
        n = bn4(blk.norm_1)
        out += n[:2]
        if i == 0:
            d = bn4(down[1])
            out += [k_conv(down[0])] + d[:2]
        out += n[2:]
        if i == 0:
            out += d[2:]
    return out


def load_transformer_block(blk, weights):
    """TransformerBlock.weights (layers.py:73-86): BN (gamma, beta, mean, var), MHA (q k, q b, k k, k b, v k, v b, out k, out b),
    LayerNorm (gamma, beta), dense_0 (kernel, bias), dense_1 (kernel, bias)."""
    w = list(weights)
    load_batchnorm(blk.layer_norm_1, *[w.pop(0) for _ in range(4)])
    load_mha(blk.attention, *[w.pop(0) for _ in range(8)])
    blk.layer_norm_2.weight.data.copy_(_t(w.pop(0)))
    blk.layer_norm_2.bias.data.copy_(_t(w.pop(0)))
    load_dense(blk.dense_0, w.pop(0), w.pop(0))
    load_dense(blk.dense_1, w.pop(0), w.pop(0))
    if w:
        raise ValueError(f'{len(w)} unused variables')


def warmup_lr_lambda(warmup_steps=10000, scale_down_after=450000):
    """`torch.optim.lr_scheduler.LambdaLR` factor of nerf_utils.WarmupScheduler (nerf_utils.py:288-300) relative to the target
    rate: the reference's feature optimizer is Adam(WarmupScheduler(1e-5, 10000, 450000)) (train_nerf.py:24-26)."""
    warm = max(1.0, float(warmup_steps))

    def factor(step):
        if step <= warm:
            return step / warm
        return 1.0 if step <= scale_down_after else 0.1
    return factor


class KerasAdam(torch.optim.Optimizer):
    """tf.keras.optimizers.Adam's update (the reference's optimizer_feature, train_nerf.py:24-26), which is NOT torch.optim.Adam's:
        lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);   p -= lr_t * m / (sqrt(v) + eps)
    torch adds eps to sqrt(v_hat) after the bias correction, i.e. an effective epsilon larger by 1 / sqrt(1 - beta2^t) (31x at t = 1).
    The MLP group of the same train_step (mvnerf_adam_clip) uses the Keras form too."""

    def __init__(self, params, lr=1e-5, betas=(0.9, 0.999), eps=1e-7):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['m'] = torch.zeros_like(p)
                    st['v'] = torch.zeros_like(p)
                st['step'] += 1
                t = st['step']
                st['m'].mul_(b1).add_(p.grad, alpha=1.0 - b1)
                st['v'].mul_(b2).addcmul_(p.grad, p.grad, value=1.0 - b2)
                lr_t = group['lr'] * (1.0 - b2 ** t) ** 0.5 / (1.0 - b1 ** t)
                p.addcdiv_(st['m'], st['v'].sqrt().add_(group['eps']), value=-lr_t)
        return loss


def make_encoder_optimizer(producer, target_lr=1e-5, warmup_steps=10000, scale_down_after=450000):
    """The reference's `optimizer_feature` (train_nerf.py:24-31): Keras Adam (eps 1e-7, Keras' update form: KerasAdam above) on
    vision_transformer + conv_features with the warm-up schedule.  Returns (optimizer, scheduler); call `scheduler.step()` after every
    `train_step`."""
    opt = KerasAdam(producer.trainable_parameters(), lr=target_lr, betas=(0.9, 0.999), eps=1e-7)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, warmup_lr_lambda(warmup_steps, scale_down_after))
    return opt, sched


def count_parameters(module):
    return sum(p.numel() for p in module.parameters())


__all__ = ['FeatureProducer', 'VisualFeatures', 'CombineCLIPVisualV0', 'ConvolutionalEncoder', 'VisionTransformerEncoder',
           'VisionTransformer', 'TransformerBlock', 'SyntheticCLIPStage1', 'flat_net_from_keras', 'keras_from_flat_net',
           'make_encoder_optimizer', 'KerasAdam', 'convolutional_encoder_weights', 'warmup_lr_lambda', 'load_conv', 'load_conv_transpose', 'load_dense', 'load_batchnorm', 'load_mha',
           'load_combine_clip_visual', 'load_convolutional_encoder', 'load_transformer_block', 'count_parameters']
