"""Host-side mirror of the reference renderer (src/lib/mvnerf/model_v0.py) on top of the HIP library.

Same names and argument meaning as the reference so that a `train_nerf.py`-style script (or a
test) reads the same:

* :class:`MVVNeRFRenderer` - ``_call`` / ``infer`` / ``call`` / ``volumetric_render`` / ``store`` /
  ``load`` (model_v0.py:16-240).  The two MLPs are held as flat Keras-order fp32 buffers
  (``kernel[in,out]``, ``bias[out]``); the image encoders (``VisualFeatures``, CLIP,
  ``CombineCLIPVisual*``) are NOT part of the hot path: their output ``combined_features``
  (B,V,H,W,256) is an input here, or comes from a user-supplied ``feature_encoder`` callable.
* :func:`render_view` (model_v0.py:243-281) - full-image driver, device resident (no per-chunk
  host round trip, source images uploaded once).
* :func:`render` - the ``render(rays, model) -> rgb, depth`` surface named in BASELINE.json.

Differences that are deliberate: image size and ray count are not baked into
signatures; the uniforms that the reference draws inside the graph with ``tf.random.uniform``
(nerf_utils.py:57,151) can be passed explicitly (``u_coarse``, ``u_fine``) and otherwise come from
``torch.rand`` on the device (optionally with a ``generator``).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops
from ._lib import NET_PARAMS, Q7_ZERO
from .synthetic import glorot_net

_SUBMODELS = ('coarse_embedding', 'coarse_readout', 'fine_embedding', 'fine_readout')
_EMB_PARAMS = NET_PARAMS - (128 * 4 + 4)      # trunk part of the flat buffer; the rest is the read-out


class MVVNeRFRenderer:
    """model_v0.py:16-44.  Holds the coarse and fine MLP weights and runs `_call` on the GPU."""

    def __init__(self, n_rays_train, n_rays_infer, n_views=2, n_samples=64, n_features=256,
                 embed_direction_vector=True, batch_size=1, near=0.7, far=1.5, original_image_size=(480, 640),
                 device='cuda', seed=0, feature_encoder=None, q7_mode=Q7_ZERO, compute_dtype='f32', f32_gemm='split_f16'):
        if n_features != 256:
            raise ValueError('n_features must be 256 (the HIP kernels are built for the reference feature width)')
        if not embed_direction_vector:
            raise ValueError('embed_direction_vector=False is not built (every reference config sets True)')
        if n_samples != 64:
            raise ValueError('n_samples must be 64 (reference config nerf_model/default.yaml:3)')
        self.n_views = n_views
        self.n_samples = n_samples
        self.n_rays_train = n_rays_train
        self.n_rays_infer = n_rays_infer
        self.batch_size = batch_size
        self.infer_batch_size = 1
        self.near = near
        self.far = far
        self.original_image_size = tuple(original_image_size)
        self.device = torch.device(device)
        self.feature_encoder = feature_encoder
        self.q7_mode = q7_mode
        if compute_dtype not in ('f32', 'bf16'):
            raise ValueError("compute_dtype must be 'f32' (reference precision) or 'bf16' (bf16 MFMA inputs, fp32 accumulate)")
        self.compute_dtype = compute_dtype       # inference only; training always runs the fp32 path
        if f32_gemm not in ('split_f16', 'split_bf16', 'mfma_f32'):
            raise ValueError("f32_gemm must be 'split_f16' (two fp16 pieces per operand, three MFMAs per product block: the default), "
                             "'split_bf16' (exact three-piece bf16 cut, six MFMAs) - both fp32-grade - or 'mfma_f32'")
        self.f32_gemm = f32_gemm                 # how compute_dtype='f32' inference runs its Dense layers (same results to ~1e-6)
        rng = np.random.default_rng(seed)
        self.coarse_net = torch.from_numpy(glorot_net(rng)).to(self.device)      # Keras glorot_uniform, zero bias
        self.fine_net = torch.from_numpy(glorot_net(rng)).to(self.device)
        self._packed = None
        self._packed16 = None
        self._packed_split = None
        self._packed_bwd = None
        self._workspace = None
        self._tables = None             # (2,B,V,H,W,128) texel tables [coarse | fine] of the last scene, see _call
        self._tables_key = None
        self._last_call = (None, None)  # the mvnerf_train_call of the last loss_and_grads (and the tensors it points to)

    # ---- weights -------------------------------------------------------------------------
    def set_weights(self, coarse_net=None, fine_net=None):
        """Replace the flat Keras-order buffers (247 300 fp32 each) and drop the packed images."""
        for name, val in (('coarse_net', coarse_net), ('fine_net', fine_net)):
            if val is not None:
                val = torch.as_tensor(val, dtype=torch.float32).reshape(-1).to(self.device).contiguous()
                if val.numel() != NET_PARAMS:
                    raise ValueError(f'{name}: {val.numel()} parameters, expected {NET_PARAMS}')
                setattr(self, name, val)
        self._packed = None
        self._packed_bwd = None
        self._tables_key = None
        self._last_call = (None, None)

    def weights_changed(self):
        """Call after updating coarse_net / fine_net in place (e.g. an optimizer step)."""
        self._packed = None
        self._packed_bwd = None
        self._tables_key = None
        self._last_call = (None, None)

    def packed(self):
        if self._packed is None:
            self._packed = (ops.pack_net(self.coarse_net), ops.pack_net(self.fine_net))
            self._packed16 = None
            self._packed_split = None
        return self._packed

    def packed_split(self):
        self.packed()
        if self._packed_split is None:
            self._packed_split = (ops.pack_net_split(self.coarse_net), ops.pack_net_split(self.fine_net))
        return self._packed_split

    def packed16(self):
        self.packed()
        if self._packed16 is None:
            self._packed16 = (ops.pack_net_bf16(self.coarse_net), ops.pack_net_bf16(self.fine_net))
        return self._packed16

    # ---- forward -------------------------------------------------------------------------
    def encode(self, image):
        """model_v0.py:47-49 - image encoder hook; out of the hot-path scope."""
        if self.feature_encoder is None:
            raise NotImplementedError('no feature_encoder configured: pass combined_features explicitly '
                                      '(the conv/ViT/CLIP encoders are outside the render hot path)')
        return self.feature_encoder(image)

    def _uniforms(self, b, r, u_coarse, u_fine, generator):
        shape = (b, r, self.n_samples)
        if u_coarse is None:
            u_coarse = torch.rand(shape, dtype=torch.float32, device=self.device, generator=generator)
        if u_fine is None:
            u_fine = torch.rand(shape, dtype=torch.float32, device=self.device, generator=generator)
        return u_coarse, u_fine

    def _call(self, inputs, n_rays, batch_size, combined_features, u_coarse=None, u_fine=None, generator=None,
              scene_key=None):
        """model_v0.py:113-184.  inputs = (ray_origins (B,R,3), ray_directions (B,R,3),
        images (B,V,H,W,3) in [0,1], intrinsics (B,V,4,4), extrinsics_inv (B,V,4,4)).
        Returns (rgb, depth, fine_rgb, fine_depth).
        The feature rows of layer 0 go through per-texel tables (include/mvnerf_hip.h, "Texel table") when that is
        cheaper: always rebuilt for a call with R*S >= 2*H*W; with `scene_key` (any hashable naming the feature maps,
        e.g. one frame rendered in chunks) they are built by the first call and re-used while key and weights last."""
        rays_o, rays_d, images, k4, einv = [self._dev(t) for t in inputs]
        if self.compute_dtype == 'bf16' and isinstance(combined_features, torch.Tensor) and combined_features.dtype == torch.bfloat16:
            # bf16 feature maps stay bf16 (encoders.FeatureProducer(out_dtype=torch.bfloat16)): the bf16 passes read them as stored
            features = combined_features.to(self.device).contiguous()
        else:
            features = self._dev(combined_features)
        if rays_o.dim() != 3 or rays_o.shape[0] != batch_size or rays_o.shape[1] != n_rays:
            raise ValueError(f'ray_origins: shape {tuple(rays_o.shape)}, expected ({batch_size}, {n_rays}, 3)')
        u_coarse, u_fine = self._uniforms(batch_size, n_rays, u_coarse, u_fine, generator)
        pc, pf = self.packed()
        if self.compute_dtype == 'bf16':
            pc16, pf16 = self.packed16()
            return ops.render_fwd_bf16(rays_o, rays_d, images, features, k4, einv, pc, pf, pc16, pf16, self._dev(u_coarse),
                                       self._dev(u_fine), self.near, self.far, self.q7_mode)
        need = ops.render_workspace_bytes(batch_size, images.shape[1], n_rays, self.n_samples)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        v, h, w = images.shape[1:4]
        tables, ready = None, False
        if scene_key is not None or ops.texel_table_pays(n_rays, self.n_samples, h, w):
            shape = (2, batch_size, v, h, w, 128)
            if self._tables is None or tuple(self._tables.shape) != shape:
                self._tables = torch.empty(shape, dtype=torch.float32, device=self.device)
                self._tables_key = None
            tables = self._tables
            ready = scene_key is not None and self._tables_key == scene_key
            self._tables_key = scene_key
        if self.f32_gemm != 'mfma_f32':
            ops.set_split_kernel(self.f32_gemm)          # process-wide: which split kernel the calls below run
        return ops.render_fwd(rays_o, rays_d, images, features, k4, einv, pc, pf, self._dev(u_coarse), self._dev(u_fine),
                              self.near, self.far, self.q7_mode, workspace=self._workspace, texel_tables=tables,
                              tables_ready=ready, split=self.packed_split() if self.f32_gemm != 'mfma_f32' else None)

    def infer(self, inputs, batched_features, **kw):
        """model_v0.py:61-63 (any ray count, not only n_rays_infer=512)."""
        return self._call(inputs, inputs[0].shape[1], inputs[0].shape[0], batched_features, **kw)

    def call(self, inputs, training=False, combined_features=None, **kw):
        """model_v0.py:75-87: encoder prologue (pluggable) + `_call`."""
        if combined_features is None:
            b, v = inputs[2].shape[:2]
            feats = self.encode(self._dev(inputs[2]).reshape(b * v, *inputs[2].shape[2:]))
            combined_features = feats.reshape(b, v, *feats.shape[1:])
        return self._call(inputs, inputs[0].shape[1], inputs[0].shape[0], combined_features, **kw)

    __call__ = call

    @staticmethod
    def volumetric_render(zs, density, chromacity):
        """model_v0.py:89-100 -> (rgb, depth, weights)."""
        rgbs = torch.cat([chromacity, density[..., None]], dim=-1).contiguous()
        return ops.composite(zs.contiguous(), rgbs, return_weights=True)

    # ---- training (model_v0.py:186-197, train_nerf.py:20-34, nerf_utils.py:8-12) -----------------------------
    def compile(self, learning_rate=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7, gradients_clip=1.0,
                train_readout=False, grad_sync=None, encoder_optimizer=None, deterministic=None):
        """train_nerf.py:20-34: MSE loss, Adam(1e-4) on the coarse and fine embeddings.
        `learning_rate` may be a callable of the step (e.g. nerf_utils.WarmupScheduler).
        train_readout=False mirrors the reference's MultiOptimizer list, which names only the two embeddings
        (SURVEY.md Q9); set True to update the RenderReadout kernels as well.
        grad_sync: optional callable on the single flat gradient buffer (494 600 fp32), e.g.
        distributed.allreduce_mean_ for data-parallel training (one collective per step).
        deterministic: kept for callers (ops.set_deterministic): the weight gradients are summed in a fixed order in any case.
        """
        if deterministic is not None:
            ops.set_deterministic(deterministic)
        self._opt = dict(lr=learning_rate, b1=beta_1, b2=beta_2, eps=epsilon, clip=gradients_clip, step=0)
        self._grad = torch.zeros(2 * NET_PARAMS, dtype=torch.float32, device=self.device)
        self._adam_m = torch.zeros_like(self._grad)
        self._adam_v = torch.zeros_like(self._grad)
        mask = torch.ones(NET_PARAMS, dtype=torch.uint8, device=self.device)
        if not train_readout:
            mask[_EMB_PARAMS:] = 0
        self._update_mask = torch.cat([mask, mask]).contiguous()
        self._grad_sync = grad_sync
        self._train_bufs = {}
        # optional torch optimizer over the parameters of `feature_encoder` (a torch.nn.Module): train_step then also
        # back-propagates dL/d(combined_features) into the encoder (used only when train_step encodes the images itself)
        self._encoder_optimizer = encoder_optimizer

    def loss_and_grads(self, inputs, labels, combined_features, u_coarse=None, u_fine=None, generator=None,
                       stop_fine_z=False, return_d_features=False):
        """Forward + backward of loss = MSE(labels, rgb) + MSE(labels, fine_rgb) (model_v0.py:190-194).
        Returns (loss 1-element device tensor, flat gradient (2 x 247300): [coarse | fine], outputs 4-tuple).
        Gradient scope: all MLP variables, including the path the reference leaves open (no stop_gradient on the
        importance samples): fine loss -> fine sample positions -> sample_pdf -> coarse weights ->
        coarse network.  stop_fine_z=True cuts that path (cheaper).
        return_d_features: also return dL/d(combined_features) (B,V,H,W,256) as a 4th element - the cotangent an
        upstream feature encoder (trained in the reference, train_nerf.py:27-32) continues from."""
        if not hasattr(self, '_grad'):
            self.compile()
        call, keep = self._train_call(inputs, labels, combined_features, u_coarse, u_fine, generator, stop_fine_z, return_d_features)
        ops.loss_and_grads(call, keep['rays_o'])                 # ONE C call: mvnerf_loss_and_grads (csrc/train_api.hip)
        self._last_call = (call, keep)
        if return_d_features:
            return keep['loss'], self._grad, keep['outputs'], keep['d_features']
        return keep['loss'], self._grad, keep['outputs']

    def _train_call(self, inputs, labels, combined_features, u_coarse, u_fine, generator, stop_fine_z, return_d_features):
        """The mvnerf_train_call of one step: device tensors checked and bound, workspace and weight images cached per shape."""
        rays_o, rays_d, images, k4, einv = [self._dev(t) for t in inputs]
        feats = self._dev(combined_features)
        y = self._dev(labels)
        b, r, _ = rays_o.shape
        v, h, w = images.shape[1:4]
        u_coarse, u_fine = self._uniforms(b, r, u_coarse, u_fine, generator)
        u_coarse, u_fine = self._dev(u_coarse), self._dev(u_fine)
        pc, pf = self.packed()
        if self._packed_bwd is None:
            self._packed_bwd = (ops.pack_bwd_streams(self.coarse_net), ops.pack_bwd_streams(self.fine_net))
        split = self.packed_split() if self.f32_gemm != 'mfma_f32' else None
        if split is not None:
            ops.set_split_kernel(self.f32_gemm)          # the training forward runs on the same split kernel as inference; the backward keeps the exact cut
        use_tables = ops.texel_table_pays(r, self.n_samples, h, w)      # layer 0's feature rows per texel
        tb = self._train_bufs
        key = (b, v, r, h, w, bool(return_d_features))
        if tb.get('key') != key:
            tb.clear()
            tb['key'] = key
            tb['workspace'] = torch.empty(ops.train_workspace_bytes(b, v, r, self.n_samples, h, w, use_tables, return_d_features),
                                          dtype=torch.uint8, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        outputs = (torch.empty((b, r, 3), dtype=torch.float32, device=self.device), torch.empty((b, r), dtype=torch.float32, device=self.device),
                   torch.empty((b, r, 3), dtype=torch.float32, device=self.device), torch.empty((b, r), dtype=torch.float32, device=self.device))
        d_feat = torch.empty_like(feats) if return_d_features else None
        keep = dict(rays_o=rays_o, rays_d=rays_d, images=images, feats=feats, k4=k4, einv=einv, u=(u_coarse, u_fine), y=y, loss=loss,
                    outputs=outputs, d_features=d_feat, packed=(pc, pf), split=split, bwd=self._packed_bwd)
        call = ops.train_call(rays_o, rays_d, images, feats, k4, einv, u_coarse, u_fine, y, self.near, self.far,
                              (self.coarse_net, self.fine_net), (pc, pf), split, self._packed_bwd, loss, self._grad, outputs, tb['workspace'],
                              q7_mode=self.q7_mode, stop_fine_z=stop_fine_z, use_tables=use_tables, d_features=d_feat)
        if hasattr(self._grad_sync, 'event_handle'):         # distributed.OverlappedGradSync: the fine half's all-reduce starts early
            call.fine_grad_event = self._grad_sync.event_handle()
        return call, keep

    def train_step(self, data, combined_features=None, u_coarse=None, u_fine=None, generator=None, stop_fine_z=False):
        """model_v0.py:186-197: one optimisation step on (inputs, labels); returns {'loss': 1-element tensor}."""
        inputs, labels = data
        enc_opt = getattr(self, '_encoder_optimizer', None)
        train_encoder = combined_features is None and enc_opt is not None
        if combined_features is None:
            bsz, v = inputs[2].shape[:2]
            with torch.set_grad_enabled(train_encoder):
                feats = self.encode(self._dev(inputs[2]).reshape(bsz * v, *inputs[2].shape[2:]))
                combined_features = feats.reshape(bsz, v, *feats.shape[1:])
        if train_encoder:
            # the reference's optimizer list also names the feature encoders (train_nerf.py:27-32): the HIP backward hands
            # dL/d(combined_features) back to torch autograd, which continues into the encoder's variables
            loss, grad, _, d_feat = self.loss_and_grads(inputs, labels, combined_features.detach().contiguous(), u_coarse, u_fine,
                                                        generator, stop_fine_z, return_d_features=True)
            enc_opt.zero_grad(set_to_none=True)
            combined_features.backward(d_feat.to(combined_features.dtype))
            if self._grad_sync is not None:              # data parallel: the encoder's gradients ride in ONE more flat collective
                grads = [prm.grad for group in enc_opt.param_groups for prm in group['params'] if prm.grad is not None]
                if grads:
                    flat = torch._utils._flatten_dense_tensors(grads)
                    self._grad_sync(flat)
                    for g_, f_ in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
                        g_.copy_(f_)
            for group in enc_opt.param_groups:           # optimize(): clip-by-value, then the optimizer step
                for prm in group['params']:
                    if prm.grad is not None:
                        prm.grad.clamp_(-self._opt['clip'], self._opt['clip'])
            enc_opt.step()
        else:
            loss, grad, _ = self.loss_and_grads(inputs, labels, combined_features, u_coarse, u_fine, generator, stop_fine_z)
        if self._grad_sync is not None:
            self._grad_sync(grad)                         # one flat collective for both MLPs
        self.apply_gradients(grad)
        return {'loss': loss}

    def apply_gradients(self, grad):
        """optimize() (nerf_utils.py:8-12) on the flat gradient [coarse | fine]: clip-by-value, Adam (tf.keras, eps 1e-7)."""
        o = self._opt
        # Keras evaluates the schedule at `optimizer.iterations` BEFORE the increment (0 on the first step: the warm-up's
        # first update has lr = 0); only the Adam bias correction uses iterations + 1
        lr = o['lr'](o['step']) if callable(o['lr']) else o['lr']
        o['step'] += 1
        lr_t = lr * np.sqrt(1.0 - o['b2'] ** o['step']) / (1.0 - o['b1'] ** o['step'])
        if grad.data_ptr() != self._grad.data_ptr():
            self._grad.copy_(grad)
        adam = ops.adam_state(self._adam_m, self._adam_v, lr_t, o['b1'], o['b2'], o['eps'], o['clip'], self._update_mask, repack=True)
        call = getattr(self, '_last_call', (None, None))[0]
        if call is None:                                     # gradients that did not come from loss_and_grads: only the variables move
            call = ops._lib.TrainCall()
            call.net_coarse, call.net_fine, call.grad = self.coarse_net.data_ptr(), self.fine_net.data_ptr(), self._grad.data_ptr()
            ops.apply_gradients(call, adam, self._grad)
            self.weights_changed()
            return
        # mvnerf_apply_gradients: clip + Adam on both MLPs, then the weight images of the call (packed, split, transposed streams) are
        # rebuilt in place - they are this model's cached ones, so they stay valid; the derived caches are dropped
        ops.apply_gradients(call, adam, self._grad)
        self._packed16 = None
        self._tables_key = None
        if not call.split_coarse:
            self._packed_split = None

    # ---- checkpoint (model_v0.py:199-240; per-sub-model files, load() -> False if any is missing) ----
    def _split(self, flat):
        return flat[:_EMB_PARAMS], flat[_EMB_PARAMS:]

    def store(self, path):
        parts = dict(zip(_SUBMODELS, (*self._split(self.coarse_net), *self._split(self.fine_net))))
        for name, t in parts.items():
            torch.save(t.detach().cpu(), f'{path}_{name}.pt')

    def load(self, path, old=False):
        files = [f'{path}_{name}.pt' for name in _SUBMODELS]
        if not all(os.path.exists(f) for f in files):
            return False
        ce, cr, fe, fr = [torch.load(f, weights_only=True) for f in files]
        self.set_weights(torch.cat([ce, cr]), torch.cat([fe, fr]))
        return True

    def _dev(self, t):
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
        return t.to(self.device, dtype=torch.float32).contiguous()


def camera_parameters(camera_config):
    """data_generator/util.py:4-10: (extrinsics_inv, intrinsics padded to 4x4), float64 NumPy."""
    k = np.reshape(camera_config['intrinsics'], (3, 3))
    k4 = np.concatenate((k, np.zeros((3, 1))), axis=1)
    k4 = np.concatenate((k4, np.array([[0, 0, 0, 1]])), axis=0)
    return np.linalg.inv(camera_config['pose']), k4


def render(rays, model, *, features, images, K4, Einv, near=None, far=None, n_samples=64, u_coarse=None, u_fine=None,
           generator=None, return_coarse=False):
    """render(rays, model) -> (rgb, depth): the fine pair by default, the reference 4-tuple
    (rgb, depth, fine_rgb, fine_depth) (model_v0.py:184) with return_coarse=True.
    rays = (origins (B,R,3), directions (B,R,3))."""
    if n_samples != model.n_samples:
        raise ValueError(f'n_samples={n_samples} but the model was built with {model.n_samples}')
    saved = model.near, model.far
    if near is not None:
        model.near = near
    if far is not None:
        model.far = far
    try:
        out = model._call((rays[0], rays[1], images, K4, Einv), rays[0].shape[1], rays[0].shape[0], features,
                          u_coarse=u_coarse, u_fine=u_fine, generator=generator)
    finally:
        model.near, model.far = saved
    return out if return_coarse else (out[2], out[3])


def render_view(model, src_colors, src_camera_configs, tgt_camera_config, combined_features=None, generator=None,
                chunk=None):
    """model_v0.py:243-281.  src_colors: list of (H,W,>=3) uint8 images; camera configs: dicts with
    'pose' (4x4) and 'intrinsics' (9,).  Returns (rgb (H,W,3) uint8, depth (H,W,1) uint8) as NumPy.
    Everything between the upload of the source views and the download of the two uint8 images stays on
    the device; `chunk` rays per `_call` (default: the whole image in one call)."""
    dev = model.device
    tgt_k = np.reshape(tgt_camera_config['intrinsics'], (3, 3)).astype(np.float32)
    h, w = src_colors[0].shape[:2]
    pose = np.asarray(tgt_camera_config['pose'], dtype=np.float64)
    m = pose[:3, :3] @ np.linalg.inv(tgt_k[:3, :3])                      # nerf_utils.py:30, host LAPACK as the reference
    rays_o, rays_d = ops.get_rays_device(m, pose[:3, 3], dev, width=w, height=h)
    src = np.array([[img[..., :3] / 255.0 for img in src_colors]])         # (1,V,H,W,3), model_v0.py:253
    images = torch.from_numpy(src.astype(np.float32)).to(dev)
    cams = [camera_parameters(c) for c in src_camera_configs]              # data_generator/mvnerf.py:28-43
    einv = torch.from_numpy(np.array([[c[0] for c in cams]], dtype=np.float32)).to(dev)
    k4 = torch.from_numpy(np.array([[c[1] for c in cams]], dtype=np.float32)).to(dev)
    if combined_features is None:
        feats = model.encode(images.reshape(-1, h, w, 3))
        combined_features = feats.reshape(1, len(src_colors), *feats.shape[1:])
    n = h * w
    chunk = n if chunk is None else int(chunk)
    rgbs = torch.empty((n, 3), dtype=torch.float32, device=dev)
    depths = torch.empty((n,), dtype=torch.float32, device=dev)
    frame = object() if model.compute_dtype == 'f32' else None           # texel tables: built by the first chunk, then re-used
    for i in range(0, n, chunk):
        sl = slice(i, min(n, i + chunk))
        kw = {'scene_key': frame} if frame is not None else {}
        out = model.infer((rays_o[None, sl], rays_d[None, sl], images, k4, einv), combined_features, generator=generator, **kw)
        rgbs[sl], depths[sl] = out[2][0], out[3][0]
    rgb8, depth8 = ops.finish_view(rgbs, depths)
    return rgb8.reshape(h, w, 3).cpu().numpy(), depth8.reshape(h, w, 1).cpu().numpy()
