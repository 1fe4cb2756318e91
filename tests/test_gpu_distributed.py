"""N > 1 with the real kernels: two ranks sharing this box's one GPU (gloo; RCCL needs one GPU per rank and
is what bench.py uses on a multi-GPU node).  Data parallel training: the mean over ranks of the per-scene gradients is the
gradient of the two-scene batch (Keras MSE = mean over every element, model_v0.py:190-194), both ranks end a step with
bit-identical weights, and those equal a single-process Adam step taken on that mean gradient.  Inference:
`render_rays_sharded` over the real `_call` reproduces the unsharded image bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from thesis_clip_nerf_amd import MVVNeRFRenderer
from thesis_clip_nerf_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope='module')
def ranks(tmp_path_factory):
    out = tmp_path_factory.mktemp('dist')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'dist_gpu_worker.py'), str(out)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    return [torch.load(out / f'rank{r}.pt', weights_only=True) for r in range(2)]


def test_data_parallel_gradient_equals_two_scene_batch(ranks):
    from tests.dist_gpu_worker import SCENE
    sc = make_scene(batch=2, **SCENE)
    y = np.random.default_rng(2).random((2, 64, 3)).astype(np.float32)
    m = MVVNeRFRenderer(64, 64, n_views=2, batch_size=2, near=sc['near'], far=sc['far'], device=DEV)
    m.set_weights(sc['coarse'], sc['fine'])
    inputs = tuple(sc[k] for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv'])
    loss, grad, _ = m.loss_and_grads(inputs, y, sc['features'], u_coarse=dev(sc['u_coarse']), u_fine=dev(sc['u_fine']))
    torch.cuda.synchronize()
    grad = grad.cpu()
    assert torch.equal(ranks[0]['synced_grad'], ranks[1]['synced_grad'])               # the all-reduce gives every rank the same bits
    assert not torch.equal(ranks[0]['local_grad'], ranks[1]['local_grad'])             # ... from different per-scene gradients
    mean = 0.5 * (ranks[0]['local_grad'] + ranks[1]['local_grad'])
    assert (ranks[0]['synced_grad'] - mean).abs().max().item() <= 1e-7 * mean.abs().max().item()
    assert abs(float(loss) - 0.5 * (float(ranks[0]['loss']) + float(ranks[1]['loss']))) < 1e-6
    for name, sl in (('coarse', slice(0, 247300)), ('fine', slice(247300, 494600))):
        rel = ((ranks[0]['synced_grad'][sl] - grad[sl]).norm() / grad[sl].norm()).item()
        print(f'data-parallel {name} gradient vs the single-process two-scene batch: rel L2 {rel:.2e}')
        assert rel < 1e-4, (name, rel)                                                  # fp32 atomic accumulation order only


def test_data_parallel_step_leaves_identical_weights(ranks):
    from tests.dist_gpu_worker import SCENE
    for k in ('coarse_net', 'fine_net'):
        assert torch.equal(ranks[0][k], ranks[1][k]), k
        # the overlapped all-reduce (two halves on two streams) applies the same averaged gradient
        assert torch.equal(ranks[0][k + '_overlap'], ranks[1][k + '_overlap']), k
        # ... which a second model recomputed: for V = 2 the sample-position gradient accumulates with fp32 atomics, so the two runs agree
        # to rounding, not bit for bit (Adam's first step is ~lr = 1e-3 per weight)
        assert (ranks[0][k + '_overlap'] - ranks[0][k]).abs().max().item() < 5e-5, k
    # a single-process optimizer step on the mean of the two gradients lands on the same weights.  (train_step recomputed the
    # gradient: the backward accumulates with fp32 atomics, so it equals `synced_grad` only to rounding - hence a tolerance
    # here, while the two ranks above must agree bit for bit because they apply the SAME all-reduced buffer.)
    sc = make_scene(batch=2, **SCENE)
    m = MVVNeRFRenderer(64, 64, n_views=2, batch_size=1, near=sc['near'], far=sc['far'], device=DEV)
    m.set_weights(sc['coarse'], sc['fine'])
    m.compile(learning_rate=1e-3)
    m.apply_gradients(ranks[0]['synced_grad'].to(DEV))
    torch.cuda.synchronize()
    moved = (m.fine_net.cpu() - torch.from_numpy(sc['fine'])).abs().max().item()
    assert moved > 1e-4                                                               # Adam's first step moves weights by ~lr
    for k, ref in (('coarse_net', m.coarse_net), ('fine_net', m.fine_net)):
        assert (ranks[0][k] - ref.cpu()).abs().max().item() < 2e-2 * moved, k


def test_sharded_render_equals_unsharded_bit_for_bit(ranks):
    from tests.dist_gpu_worker import FRAME
    fr = make_scene(batch=1, **FRAME)
    r = MVVNeRFRenderer(320, 320, n_views=1, near=fr['near'], far=fr['far'], device=DEV)
    r.set_weights(fr['coarse'], fr['fine'])
    whole = r.infer(tuple(fr[k] for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv']), fr['features'],
                    u_coarse=dev(fr['u_coarse']), u_fine=dev(fr['u_fine']))
    torch.cuda.synchronize()
    for rk in ranks:
        for got, want in zip(rk['sharded'], whole):
            assert torch.equal(got, want[0].cpu())


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher in the environment (the driver's command shape): the parent starts the two
    ranks itself (here on one GPU through gloo), relays ONE JSON line, and the data-parallel training leg leaves both ranks with
    identical weights."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(MVNERF_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--train-steps', '1'],
                          env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['value'] > 0 and d['metric'].startswith('rendered rays/sec')
    assert d['train_cfg4']['weights_identical_across_ranks'] is True and d['train_cfg4']['allreduce_ms'] > 0
    assert 'cpu_baseline' not in d                                     # rank 0 at N = 1 only
    gs = d['train_cfg4']['grad_sync']
    assert gs['overlapped_ms_per_step'] > 0 and gs['flat_ms_per_step'] > 0 and gs['no_collective_ms_per_step'] > 0


def test_bench_strong_scaling_leg(tmp_path):
    """`bench.py --gpus 2 --scaling strong`: ONE 128x128 scene's 16 384 rays split over the ranks by shard_bounds (SURVEY.md 8d cfg4,
    strong leg), rehearsed on one GPU through gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(MVNERF_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--scaling', 'strong', '--steps', '2', '--warmup', '1',
                           '--train-steps', '1'], env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    d = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith('{')][0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['config']['rays_per_gpu'] == 8192
    assert abs(d['value'] * d['ms_per_step'] * 1e-3 - 16384) < 1.0            # value = the whole job's 16 384 rays / step time
    assert d['train_cfg4']['scaling'] == 'strong' and d['train_cfg4']['rays_per_gpu'] == 8192
    assert d['train_cfg4']['weights_identical_across_ranks'] is True
