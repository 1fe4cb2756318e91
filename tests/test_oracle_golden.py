"""Pin the oracle's NumPy functions (a1-a3) against vectors produced by the reference's own code
(tests/golden/make_golden.py; reference: nerf_utils.py:15-46, data_generator/util.py:4-10)."""
import os

import numpy as np

from oracle import mvnerf_oracle as O


def test_get_rays_and_specific_rays(golden_dir):
    g = np.load(os.path.join(golden_dir, 'rays.npz'))
    for i in range(4):
        w, h = g[f'rays{i}_wh']
        o, d = O.get_rays(int(w), int(h), g[f'rays{i}_pose'], g[f'rays{i}_k'])
        assert o.dtype == np.float64 and d.dtype == np.float64          # Q2
        np.testing.assert_array_equal(o, g[f'rays{i}_o'])
        np.testing.assert_allclose(d, g[f'rays{i}_d'], rtol=0, atol=1e-15)
        so, sd = O.get_specific_rays(g[f'rays{i}_u'], g[f'rays{i}_v'], g[f'rays{i}_pose'], g[f'rays{i}_k'])
        np.testing.assert_array_equal(np.array(so), g[f'rays{i}_so'])
        np.testing.assert_allclose(sd, g[f'rays{i}_sd'], rtol=0, atol=1e-15)
        einv, k4 = O.camera_parameters(g[f'rays{i}_pose'], g[f'rays{i}_k'].reshape(-1))
        np.testing.assert_array_equal(einv, g[f'rays{i}_einv'])
        np.testing.assert_array_equal(k4, g[f'rays{i}_k4'])


def test_bbox_biased_sample_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, 'pixel_idx.npz'))
    for seed in range(4):
        for n, (h, w) in [(512, (480, 640)), (4096, (64, 64)), (10, (7, 5))]:
            np.random.seed(seed)
            s = O.bbox_biased_sample(n, np.array([0, 0, h, w]), h, w)
            assert s.dtype == np.int64 and s.shape == (n, 2)
            np.testing.assert_array_equal(s, g[f'seed{seed}_n{n}_h{h}_w{w}'])
        np.random.seed(seed)
        s = O.bbox_biased_sample(512, np.array([100, 200, 300, 500]), 480, 640)
        np.testing.assert_array_equal(s, g[f'seed{seed}_bbox'])
