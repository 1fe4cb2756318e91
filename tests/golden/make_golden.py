"""Generate golden vectors from the reference's own NumPy functions.  Runs ONLY in the build
container (needs /root/reference); the GPU box and the test-suite use the committed .npz files.

The reference module `src/lib/mvnerf/nerf_utils.py` imports TensorFlow at the top, which is not
installed here (ordinary ModuleNotFoundError), so the three pure-NumPy functions
(`get_rays`, `get_specific_rays`, `bbox_biased_sample`, nerf_utils.py:15-46) are pulled out of the
parsed source with `ast` and executed with only {np, rearrange, repeat} in scope.  Likewise
`camera_parameters` from data_generator/util.py:4-10.  Only inputs and outputs are stored.

    python tests/golden/make_golden.py
"""
import ast
import os
import sys

import numpy as np
from einops import rearrange, repeat

REF = '/root/reference/src/lib'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))


def extract(path, names):
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == len(names), [n.name for n in keep]
    scope = {'np': np, 'rearrange': rearrange, 'repeat': repeat}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, 'exec'), scope)
    return [scope[n] for n in names]


def main():
    from thesis_clip_nerf_amd.synthetic import ring_pose, pinhole

    get_rays, get_specific_rays, bbox_biased_sample = extract(
        os.path.join(REF, 'mvnerf', 'nerf_utils.py'), ['get_rays', 'get_specific_rays', 'bbox_biased_sample'])
    (camera_parameters,) = extract(os.path.join(REF, 'data_generator', 'util.py'), ['camera_parameters'])

    # a1/a2: full ray grids and specific rays, several cameras and image sizes
    out = {}
    rng = np.random.default_rng(1234)
    for i, (w, h) in enumerate([(64, 64), (16, 8), (128, 128), (40, 30)]):
        pose = ring_pose(rng.uniform(0, 2 * np.pi), radius=rng.uniform(0.6, 1.0), elevation=rng.uniform(0.3, 1.2))
        k = pinhole(w, h, focal_scale=rng.uniform(0.7, 1.2))
        k[0, 2] += np.float32(rng.uniform(-2, 2))       # off-centre principal point
        o, d = get_rays(w, h, pose, k)
        out[f'rays{i}_wh'] = np.array([w, h])
        out[f'rays{i}_pose'] = pose
        out[f'rays{i}_k'] = k
        out[f'rays{i}_o'] = o
        out[f'rays{i}_d'] = d
        u = rng.integers(0, w, size=37)
        v = rng.integers(0, h, size=37)
        so, sd = get_specific_rays(u, v, pose, k)
        out[f'rays{i}_u'] = u
        out[f'rays{i}_v'] = v
        out[f'rays{i}_so'] = np.array(so)
        out[f'rays{i}_sd'] = sd
        einv, k4 = camera_parameters({'pose': pose, 'intrinsics': k.reshape(-1)})
        out[f'rays{i}_einv'] = einv
        out[f'rays{i}_k4'] = k4
    np.savez_compressed(os.path.join(HERE, 'rays.npz'), **out)

    # a3: pixel indices under fixed global seeds (integer, bit-exact contract)
    out = {}
    for seed in range(4):
        for n, (h, w) in [(512, (480, 640)), (4096, (64, 64)), (10, (7, 5))]:
            np.random.seed(seed)
            s = bbox_biased_sample(n, np.array([0, 0, h, w]), h, w)
            out[f'seed{seed}_n{n}_h{h}_w{w}'] = s
        np.random.seed(seed)
        s = bbox_biased_sample(512, np.array([100, 200, 300, 500]), 480, 640)
        out[f'seed{seed}_bbox'] = s
    np.savez_compressed(os.path.join(HERE, 'pixel_idx.npz'), **out)
    print('wrote', os.listdir(HERE))


if __name__ == '__main__':
    main()
