"""Generate golden vectors from the reference's own NumPy functions.  Runs ONLY in the build
container (needs /root/reference); the GPU box and the test-suite use the committed .npz files.

The reference module `src/lib/mvnerf/nerf_utils.py` imports TensorFlow at the top, which is not
installed here (ordinary ModuleNotFoundError), so the three pure-NumPy functions
(`get_rays`, `get_specific_rays`, `bbox_biased_sample`, nerf_utils.py:15-46) are pulled out of the
parsed source with `ast` and executed with only {np, rearrange, repeat} in scope.  Likewise
`camera_parameters` from data_generator/util.py:4-10 and the three NumPy methods `generate_rays`,
`get_input`, `get_target` of data_generator/mvnerf.py:16-48 (the module imports a package whose
submodule is absent).  Only inputs and outputs are stored.

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


def extract_methods(path, cls_name, names, scope_extra):
    """The named methods of one class of the reference, as a bare class (no base, no __init__)."""
    tree = ast.parse(open(path).read())
    (cls,) = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls_name]
    keep = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == len(names), [n.name for n in keep]
    bare = ast.ClassDef(name=cls_name, bases=[], keywords=[], body=keep, decorator_list=[])
    scope = {'np': np, **scope_extra}
    exec(compile(ast.fix_missing_locations(ast.Module(body=[bare], type_ignores=[])), path, 'exec'), scope)
    return scope[cls_name]


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

    # a3 caller side (data_generator/mvnerf.py:16-48): generate_rays ((row, col) -> (u, v) -> rays), get_target and
    # get_input of the reference's MVNeRFDataGenerator, executed as plain methods of a bare class
    gen_cls = extract_methods(os.path.join(REF, 'data_generator', 'mvnerf.py'), 'MVNeRFDataGenerator',
                              ['generate_rays', 'get_input', 'get_target'],
                              {'bbox_biased_sample': bbox_biased_sample, 'get_specific_rays': get_specific_rays,
                               'camera_parameters': camera_parameters})
    out = {}
    rng = np.random.default_rng(4321)
    for i, (h, w, n, n_src) in enumerate([(16, 20, 48, 2), (64, 64, 512, 1), (30, 40, 100, 3)]):
        gen = object.__new__(gen_cls)
        gen.n_rays_train = n
        color = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        cam = {'pose': ring_pose(rng.uniform(0, 2 * np.pi), radius=rng.uniform(0.6, 1.0)),
               'intrinsics': pinhole(w, h, focal_scale=rng.uniform(0.7, 1.2)).reshape(-1)}
        np.random.seed(100 + i)
        r_d, r_o, rays = gen.generate_rays(color, cam)
        target = gen_cls.get_target(color, rays)
        src_colors = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(n_src)]
        src_cams = [{'pose': ring_pose(rng.uniform(0, 2 * np.pi)), 'intrinsics': cam['intrinsics']} for _ in range(n_src)]
        inputs = gen_cls.get_input(src_colors, src_cams, r_d, r_o)
        out[f'case{i}_seed'] = np.array(100 + i)
        out[f'case{i}_n'] = np.array(n)
        out[f'case{i}_color'] = color
        out[f'case{i}_pose'] = cam['pose']
        out[f'case{i}_intrinsics'] = cam['intrinsics']
        out[f'case{i}_r_d'] = r_d
        out[f'case{i}_r_o'] = np.array(r_o)
        out[f'case{i}_rays'] = rays
        out[f'case{i}_target'] = target
        out[f'case{i}_src_colors'] = np.array(src_colors)
        out[f'case{i}_src_poses'] = np.array([c['pose'] for c in src_cams])
        for j, a in enumerate(inputs):
            out[f'case{i}_input{j}'] = a
    np.savez_compressed(os.path.join(HERE, 'datagen.npz'), **out)
    print('wrote', os.listdir(HERE))


if __name__ == '__main__':
    main()
