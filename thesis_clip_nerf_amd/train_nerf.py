"""`train_nerf.py`-shaped training loop on the HIP renderer (reference: src/train_nerf.py).

Same structure and names as the reference script - `compile_model`, `train_model`, `validate`,
`init_training_session`, `MVNeRFDataGenerator` - with the pieces that are not in this repository's scope
replaced by explicit stand-ins:

* the dataset submodule (`src/lib/dataset`, absent from the reference tree) -> :class:`SyntheticSceneDataset`,
  random views with known cameras;
* the image encoders -> a fixed per-pixel feature map supplied by the dataset (`features`), since
  `combined_features` is an *input* of the hot path;
* hydra -> keyword arguments / argparse; cv2 PNG dumps -> binary PPM files.

    python -m thesis_clip_nerf_amd.train_nerf --model-path /tmp/nerf_run --epochs 4 --eval-after 2
"""
from __future__ import annotations

import argparse
import json
import os

import numpy as np
import torch

from .model import MVVNeRFRenderer, camera_parameters, render_view
from .nerf_utils import WarmupScheduler, bbox_biased_sample
from .synthetic import pinhole, ring_pose


class SyntheticSceneDataset:
    """Stand-in for `load_dataset_nerf(n_perspectives, path)`: `n_scenes` scenes x `n_perspectives` views."""

    def __init__(self, n_scenes=4, n_perspectives=6, height=32, width=32, seed=0):
        rng = np.random.default_rng(seed)
        self.n_perspectives = n_perspectives
        self.k = pinhole(width, height)
        proj = rng.standard_normal((3, 256)).astype(np.float32)
        self.colors, self.cameras, self.features = [], [], []
        for _ in range(n_scenes):
            base = rng.random((height, width, 3))
            cols, cams, feats = [], [], []
            for p in range(n_perspectives):
                img = np.clip(base + 0.05 * rng.standard_normal(base.shape), 0, 1)
                cols.append((img * 255).astype(np.uint8))
                cams.append({'pose': ring_pose(2 * np.pi * p / n_perspectives + rng.uniform(-0.1, 0.1)),
                             'intrinsics': self.k.reshape(-1).copy()})
                feats.append(np.tanh((img.astype(np.float32) * 2 - 1) @ proj))
            self.colors.append(cols)
            self.cameras.append(cams)
            self.features.append(feats)

    def __len__(self):
        return len(self.colors)


class MVNeRFDataGenerator:
    """data_generator/mvnerf.py + base.py: keras-Sequence semantics, NumPy RNG exactly as the reference
    (np.random.choice of views, bbox_biased_sample of pixels).  Returns ((inputs 5-tuple, features), targets)."""

    def __init__(self, dataset, n_rays_train=512, batch_size=1, n_views=2, shuffle=True, device=None):
        """device: keep every view (colours, features, cameras) resident on that GPU after its first use and build the
        batch there - pixel indices still come from the host NumPy RNG (the reference's stream, mvnerf.py:16-25), rays
        from mvnerf_get_rays on their (u, v), targets from a device gather; per step only the (n, 2) indices cross PCIe
        instead of V x H x W x (3 + 256) floats."""
        self.device = torch.device(device) if device is not None else None
        self._resident = {}
        self.dataset = dataset
        self.n_rays_train = n_rays_train
        self.batch_size = batch_size
        self.n_views = n_views
        self.shuffle = shuffle
        self.n_perspectives = dataset.n_perspectives
        self.indices = np.arange(len(dataset))
        self.on_epoch_end()

    def on_epoch_end(self):
        if self.shuffle:
            np.random.shuffle(self.indices)

    def __len__(self):
        return len(self.indices) // self.batch_size

    def __getitem__(self, index):
        return self.get_data(self.indices[index * self.batch_size:(index + 1) * self.batch_size])

    def generate_rays(self, color, camera_config):
        """mvnerf.py:16-25 (u = col, v = row); ray directions via the same float64 formula, on the host."""
        k = np.reshape(camera_config['intrinsics'], (3, 3)).astype(np.float32)
        rays = bbox_biased_sample(self.n_rays_train, np.array([0, 0, color.shape[0], color.shape[1]]), color.shape[0],
                                  color.shape[1])
        u, v = rays[:, 1], rays[:, 0]
        pose = camera_config['pose']
        d = (pose[:3, :3] @ np.linalg.inv(k) @ np.stack((u, v, np.ones_like(u)), axis=0)).T
        d = d / np.linalg.norm(d, axis=1, keepdims=True)
        return d, np.broadcast_to(pose[:3, -1], d.shape), rays

    @staticmethod
    def get_target(color, rays):
        return np.array(color[rays[:, 0], rays[:, 1], :3]) / 255.0                       # mvnerf.py:45-48

    @staticmethod
    def get_input(colors, camera_configs, r_d, r_o):
        """mvnerf.py:27-43: one scene's network inputs (rays_o, rays_d, images in [0,1], K4, E^-1), float32, batch axis 1."""
        cams = [camera_parameters(c) for c in camera_configs]
        f32 = lambda a: np.array([a], dtype=np.float32)
        return (f32(r_o), f32(r_d), f32(np.array(colors) / 255.0), f32([c[1] for c in cams]), f32([c[0] for c in cams]))

    def _view(self, i, p):
        """Device-resident (image float (H,W,3) in [0,1], colour uint8, features, E^-1, K4) of scene i, perspective p."""
        key = (int(i), int(p))
        if key not in self._resident:
            dev = self.device
            color = np.ascontiguousarray(self.dataset.colors[i][p][..., :3])
            einv, k4 = camera_parameters(self.dataset.cameras[i][p])
            self._resident[key] = (torch.from_numpy((color / 255.0).astype(np.float32)).to(dev),
                                   torch.from_numpy(color).to(dev),
                                   torch.from_numpy(np.asarray(self.dataset.features[i][p], dtype=np.float32)).to(dev),
                                   torch.from_numpy(einv.astype(np.float32)).to(dev), torch.from_numpy(k4.astype(np.float32)).to(dev))
        return self._resident[key]

    def generate_rays_device(self, color, camera_config):
        """generate_rays (mvnerf.py:16-25) with the rays made on the GPU: pixel indices from the host NumPy RNG (the reference's
        stream), (row, col) -> (u = col, v = row), mvnerf_get_rays on those.  Returns (r_d, r_o, px (n,2) int64 device)."""
        from . import ops
        dev = self.device
        rays = bbox_biased_sample(self.n_rays_train, np.array([0, 0, color.shape[0], color.shape[1]]), color.shape[0], color.shape[1])
        k = np.reshape(camera_config['intrinsics'], (3, 3)).astype(np.float32)
        m = camera_config['pose'][:3, :3] @ np.linalg.inv(k)                              # as generate_rays (host LAPACK)
        px = torch.from_numpy(np.ascontiguousarray(rays)).to(dev)                          # (n,2) int64 (row, col)
        r_o, r_d = ops.get_rays_device(m, camera_config['pose'][:3, -1], dev, u=px[:, 1].to(torch.float32).contiguous(),
                                       v=px[:, 0].to(torch.float32).contiguous())
        return r_d, r_o, px

    @staticmethod
    def get_target_device(color_u8, px):
        """get_target (mvnerf.py:45-48) as a device gather: color_u8 (H,W,3) uint8, px (n,2) (row, col)."""
        # float64 division, then float32: the reference divides in NumPy float64 and Keras casts the labels (bit-identical)
        return (color_u8[px[:, 0], px[:, 1], :3].to(torch.float64) / 255.0).to(torch.float32)

    def get_data_device(self, batch):
        """get_data with the batch assembled on the GPU; consumes the NumPy RNG exactly like get_data."""
        ro, rd, imgs, ks, es, feats, targets = [], [], [], [], [], [], []
        for i in batch:
            idx = np.random.choice(range(self.n_perspectives), size=self.n_views + 1, replace=False)
            src, tgt = idx[:-1], idx[-1]
            r_d, r_o, px = self.generate_rays_device(self.dataset.colors[i][tgt], self.dataset.cameras[i][tgt])
            targets.append(self.get_target_device(self._view(i, tgt)[1], px))
            views = [self._view(i, s_) for s_ in src]
            ro.append(r_o)
            rd.append(r_d)
            imgs.append(torch.stack([v_[0] for v_ in views]))
            feats.append(torch.stack([v_[2] for v_ in views]))
            es.append(torch.stack([v_[3] for v_ in views]))
            ks.append(torch.stack([v_[4] for v_ in views]))
        st = torch.stack
        return ((st(ro), st(rd), st(imgs), st(ks), st(es)), st(feats)), st(targets)

    def get_data(self, batch):
        if self.device is not None:
            return self.get_data_device(batch)
        ro, rd, imgs, ks, es, feats, targets = [], [], [], [], [], [], []
        for i in batch:
            idx = np.random.choice(range(self.n_perspectives), size=self.n_views + 1, replace=False)
            src, tgt = idx[:-1], idx[-1]
            r_d, r_o, rays = self.generate_rays(self.dataset.colors[i][tgt], self.dataset.cameras[i][tgt])
            targets.append(self.get_target(self.dataset.colors[i][tgt], rays))
            cams = [camera_parameters(self.dataset.cameras[i][s]) for s in src]
            ro.append(r_o)
            rd.append(r_d)
            imgs.append([self.dataset.colors[i][s][..., :3] / 255.0 for s in src])
            es.append([c[0] for c in cams])
            ks.append([c[1] for c in cams])
            feats.append([self.dataset.features[i][s] for s in src])
        f32 = lambda a: np.array(a, dtype=np.float32)
        return ((f32(ro), f32(rd), f32(imgs), f32(ks), f32(es)), f32(feats)), f32(targets)


def compile_model(nerf_renderer, grad_sync=None):
    """train_nerf.py:20-34: MSE; Adam with WarmupScheduler(1e-4, 10000, 450000) on the two embeddings."""
    nerf_renderer.compile(learning_rate=WarmupScheduler(1e-4, 10000, 450000), gradients_clip=1.0, train_readout=False,
                          grad_sync=grad_sync)


def init_training_session(model_log_dir):
    """utils/util.py:27-37: resume bookkeeping through training_progress.json."""
    start_epoch = 0
    training_progress_file = os.path.join(model_log_dir, 'training_progress.json')
    if os.path.exists(training_progress_file):
        with open(training_progress_file) as f:
            start_epoch = json.load(f).get('epoch', 0)
    return start_epoch, training_progress_file


def fit(nerf_renderer, data_generator, epochs, initial_epoch=0, log=print):
    """Keras `Model.fit(generator, epochs=, initial_epoch=)` over the custom train_step."""
    history = []
    for epoch in range(initial_epoch, epochs):
        losses = []
        for step in range(len(data_generator)):
            (inputs, features), targets = data_generator[step]
            losses.append(nerf_renderer.train_step((inputs, targets), combined_features=features)['loss'])
        data_generator.on_epoch_end()
        mean = float(torch.cat(losses).mean()) if losses else float('nan')
        history.append(mean)
        log(f'Epoch {epoch + 1}/{epochs} - loss: {mean:.6f}')
    return history


def write_ppm(path, image):
    with open(path, 'wb') as f:
        f.write(f'P6 {image.shape[1]} {image.shape[0]} 255\n'.encode())
        f.write(np.ascontiguousarray(image[..., :3], dtype=np.uint8).tobytes())


def validate(nerf_renderer, tgt_color, valid_data):
    """train_nerf.py:68-81: [sources | target | rendered rgb | rendered depth] side by side."""
    rgb, depth = render_view(nerf_renderer, **valid_data)
    src = np.concatenate([c[..., :3] for c in valid_data['src_colors']], axis=1)
    return np.concatenate([src, tgt_color[..., :3], rgb, np.repeat(depth, 3, axis=2)], axis=1)


def train_model(nerf_renderer, data_generator, n_epochs, eval_after_epochs, model_log_dir, model_checkpoint_name, valid_data,
                log=print):
    """train_nerf.py:37-65."""
    start_epoch, training_progress_file = init_training_session(model_log_dir)
    start_n_fit, n_fits = start_epoch // eval_after_epochs, n_epochs // eval_after_epochs
    valid_data = dict(valid_data)
    tgt_color = valid_data.pop('tgt_colors')
    os.makedirs(f'{model_log_dir}/valid', exist_ok=True)
    if start_epoch == 0:
        write_ppm(f'{model_log_dir}/valid/valid-0.ppm', validate(nerf_renderer, tgt_color, valid_data))
    history = []
    for k in range(start_n_fit, n_fits):
        e_epoch = (k + 1) * eval_after_epochs
        history += fit(nerf_renderer, data_generator, epochs=e_epoch, initial_epoch=k * eval_after_epochs, log=log)
        write_ppm(f'{model_log_dir}/valid/valid-{e_epoch}.ppm', validate(nerf_renderer, tgt_color, valid_data))
        with open(training_progress_file, 'w') as f:
            json.dump({'epoch': e_epoch}, f)
        nerf_renderer.store(model_checkpoint_name)
    return history


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument('--model-path', default='/tmp/mvnerf_run')
    ap.add_argument('--epochs', type=int, default=4)
    ap.add_argument('--eval-after', type=int, default=2)
    ap.add_argument('--n-views', type=int, default=1)
    ap.add_argument('--batch-size', type=int, default=1)
    ap.add_argument('--n-rays', type=int, default=512)
    ap.add_argument('--size', type=int, default=32)
    ap.add_argument('--host-batches', action='store_true', help='assemble batches in NumPy on the host (default: on the GPU)')
    args = ap.parse_args(argv)
    train = SyntheticSceneDataset(n_scenes=8, height=args.size, width=args.size, seed=0)
    valid = SyntheticSceneDataset(n_scenes=1, height=args.size, width=args.size, seed=1)
    src = list(range(args.n_views))
    valid_data = {'src_colors': [valid.colors[0][i] for i in src],
                  'src_camera_configs': [valid.cameras[0][i] for i in src],
                  'tgt_camera_config': valid.cameras[0][args.n_views],
                  'tgt_colors': valid.colors[0][args.n_views],
                  'combined_features': torch.from_numpy(np.array([[valid.features[0][i] for i in src]], dtype=np.float32))}
    gen = MVNeRFDataGenerator(train, n_rays_train=args.n_rays, batch_size=args.batch_size, n_views=args.n_views,
                              device=None if args.host_batches else 'cuda:0')
    model = MVVNeRFRenderer(args.n_rays, 512, n_views=args.n_views, batch_size=args.batch_size, near=0.3, far=1.3)
    compile_model(model)
    ckpt = f'{args.model_path}/model_final'
    os.makedirs(args.model_path, exist_ok=True)
    print('Model loaded from checkpoint.' if model.load(ckpt) else 'New model initialized')
    train_model(model, gen, args.epochs, args.eval_after, args.model_path, ckpt, valid_data)


if __name__ == '__main__':
    main()
