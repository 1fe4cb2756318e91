"""The scalar source the HIP kernels compile (thesis_clip_nerf_amd/csrc/mvnerf_math.h), built for the
host by tests/cpu_math/Makefile and checked against the oracle: the geometry chain must be
bit-identical (it decides the integer tap indices), transcendental helpers within stated ulps."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import mvnerf_oracle as O
from thesis_clip_nerf_amd.synthetic import make_scene

HERE = os.path.dirname(os.path.abspath(__file__))
F32 = np.float32


@pytest.fixture(scope='module')
def cpu():
    d = os.path.join(HERE, 'cpu_math')
    subprocess.run(['make', '-C', d], check=True, capture_output=True)
    return ctypes.CDLL(os.path.join(d, 'libmvnerf_math_cpu.so'))


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_sincos_accuracy_over_pe_range(cpu):
    rng = np.random.default_rng(0)
    x = rng.uniform(-2.5, 2.5, 200000).astype(F32)
    worst = 0.0
    for k in range(10):
        arg = (x * (F32(np.pi) * F32(2.0 ** k))).astype(F32)          # nerf_utils.py:120-123 product, rounded
        s, c = np.empty_like(arg), np.empty_like(arg)
        cpu.mv_sincos(ptr(arg), arg.size, ptr(s), ptr(c))
        worst = max(worst, np.abs(s - np.sin(arg.astype(np.float64))).max(), np.abs(c - np.cos(arg.astype(np.float64))).max())
    assert worst < 2.5e-7, worst                                       # ~2 ulp at |value| <= 1
    # large / special arguments take the libm branch
    arg = np.array([1e5, -3e6, 1e9, 0.0, -0.0], F32)
    s, c = np.empty_like(arg), np.empty_like(arg)
    cpu.mv_sincos(ptr(arg), arg.size, ptr(s), ptr(c))
    np.testing.assert_allclose(s, np.sin(arg.astype(np.float64)), atol=3e-7)
    np.testing.assert_allclose(c, np.cos(arg.astype(np.float64)), atol=3e-7)


def test_sincos_matches_oracle_pe(cpu):
    x = np.random.default_rng(1).uniform(-1.5, 1.5, (1, 1, 4096, 3)).astype(F32)
    pe = O.position_encoding(x).reshape(-1, 3, 10, 2)
    arg = (x.reshape(-1, 3, 1) * (F32(np.pi) * np.power(F32(2), np.arange(10, dtype=F32)))).astype(F32)
    s, c = np.empty_like(arg), np.empty_like(arg)
    cpu.mv_sincos(ptr(np.ascontiguousarray(arg)), arg.size, ptr(s), ptr(c))
    assert np.abs(s - pe[..., 0]).max() < 4e-7 and np.abs(c - pe[..., 1]).max() < 4e-7


def test_geometry_chain_bit_exact(cpu):
    sc = make_scene(seed=5, height=24, width=40, n_views=2, n_rays=500)
    _, z = O.sample_along_ray(sc['rays_o'], sc['rays_d'], 0.3, 1.3, 64, sc['u_coarse'])
    zc = np.empty_like(z)
    cpu.mv_stratified(ptr(sc['u_coarse']), 500, 64, ctypes.c_double(0.3), ctypes.c_double(1.3), ptr(zc))
    np.testing.assert_array_equal(zc, z)
    world = O.points_on_rays(sc['rays_o'], sc['rays_d'], z)
    pix_ref, cam_ref = O.compute_pixel_in_image_mv(world, sc['intrinsics'], sc['extrinsics_inv'])
    for v in range(2):
        e = np.ascontiguousarray(sc['extrinsics_inv'][0, v])
        k = np.ascontiguousarray(sc['intrinsics'][0, v])
        w = np.ascontiguousarray(world[0].reshape(-1, 3))
        cam = np.empty((w.shape[0], 4), F32)
        cpu.mv_matvec_rows(ptr(e), ptr(w), w.shape[0], ctypes.c_float(1.0), ptr(cam))
        np.testing.assert_array_equal(cam, cam_ref[0, v].reshape(-1, 4))
        n = w.shape[0]
        pix, x0, y0 = np.empty((n, 2), F32), np.empty(n, np.int32), np.empty(n, np.int32)
        ax, ay = np.empty(n, F32), np.empty(n, F32)
        cpu.mv_project(ptr(k), ptr(cam), n, 24, 40, ptr(pix), ptr(x0), ptr(y0), ptr(ax), ptr(ay))
        np.testing.assert_array_equal(pix, pix_ref[0, v].reshape(-1, 2))
        rx0, ry0, rax, ray_ = O.bilinear_taps(pix_ref[0, v].reshape(-1, 2), 24, 40)
        np.testing.assert_array_equal(x0, rx0)
        np.testing.assert_array_equal(y0, ry0)
        np.testing.assert_array_equal(ax, rax)
        np.testing.assert_array_equal(ay, ray_)
        # Q3: direction vectors get w = 1
        d = np.ascontiguousarray(sc['rays_d'][0])
        cd = np.empty((d.shape[0], 4), F32)
        cpu.mv_matvec_rows(ptr(e), ptr(d), d.shape[0], ctypes.c_float(1.0), ptr(cd))
        np.testing.assert_array_equal(cd[:, :3], O.world_to_camera_direction_vector_mv(sc['rays_d'], sc['extrinsics_inv'])[0, v])


def test_bilerp_bit_exact(cpu):
    rng = np.random.default_rng(2)
    t = rng.standard_normal((5000, 6)).astype(F32)
    t[:, 4:] = rng.random((5000, 2), dtype=F32)
    out = np.empty(5000, F32)
    cpu.mv_bilerp(ptr(t), 5000, ptr(out))
    tl, tr, bl, br, ax, ay = t.T
    top = (ax * (tr - tl)).astype(F32) + tl
    bot = (ax * (br - bl)).astype(F32) + bl
    np.testing.assert_array_equal(out, (ay * (bot - top)).astype(F32) + top)


def test_readout_activations(cpu):
    x = np.linspace(-30, 30, 4001).astype(F32)
    sg, sp = np.empty_like(x), np.empty_like(x)
    cpu.mv_sigmoid_softplus(ptr(x), x.size, ptr(sg), ptr(sp))
    assert np.abs(sg - O.sigmoid(x)).max() < 1.5e-7
    assert np.abs(sp - O.softplus(x)).max() / 1.0 < 4e-6 and np.abs((sp - O.softplus(x)) / np.maximum(O.softplus(x), 1e-30)).max() < 5e-7
