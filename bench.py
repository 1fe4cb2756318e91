#!/usr/bin/env python3
"""Benchmark of the MVNeRF render hot path on MI355X (BASELINE.json metric: rendered rays/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full `_call` (model_v0.py:113-184) over one batch of synthetic rays: stratified
depths -> coarse field (64 samples/ray) -> composite -> resample -> fine field (128 samples/ray) ->
composite.  Workload at every N: BASELINE.json configs[1] as restated in SURVEY.md 8d (cfg2):
B=1 scene, V=1 source view of 64x64 (3+256 channels), R=4096 rays (every pixel of a 64x64 target
view), fp32, inputs resident in HBM before the timed region.  N>1: every rank renders its own
scene (rays/scenes are independent units, no data-path collective) -> weak scaling.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (field_eval_kernel, fp32
MFMA bound): algorithmic FLOPs per launch (491 264 FLOP/sample) / its average
duration measured with HIP events inside the timed region, against the 157.3 TFLOP/s fp32 MFMA
peak.  `cpu_baseline` times the NumPy oracle (a port of the reference's TF graph; the reference
itself cannot run here) on a bounded sample of the same rays on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from thesis_clip_nerf_amd import ops  # noqa: E402
from thesis_clip_nerf_amd.distributed import max_over_ranks  # noqa: E402
from thesis_clip_nerf_amd.synthetic import make_scene  # noqa: E402

FLOP_PER_SAMPLE_V1 = 491264          # BASELINE.md 3 (2 x 245 632 MAC), one source view
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md, chip-level parameters


def flop_per_sample(v, table=False):
    """MFMA FLOPs per sample: per view Dense 379->128 + 3 blocks, then 3 blocks + read-out.  With the texel table the
    256 feature rows of layer 0 are not multiplied per sample (they are per texel, in project_texels_kernel)."""
    return 2 * (v * ((379 - (256 if table else 0)) * 128 + 6 * 128 * 128) + 6 * 128 * 128 + 128 * 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--views', type=int, default=1)
    ap.add_argument('--size', type=int, default=64, help='source/target image side (rays = size^2)')
    ap.add_argument('--height', type=int, default=0, help='source image height (default: --size)')
    ap.add_argument('--width', type=int, default=0, help='source image width (default: --size)')
    ap.add_argument('--rays', type=int, default=0, help='random target pixels instead of every pixel of a size x size view (e.g. cfg5: 16384 rays, 480x640 sources)')
    ap.add_argument('--cpu-rays', type=int, default=512, help='rays of the bounded CPU-baseline sample (0 = skip)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='bf16: both field passes on the bf16 MFMA kernel (configs 3/5; not the headline)')
    ap.add_argument('--texel-table', default='auto', choices=['auto', 'on', 'off'],
                    help="hoist layer 0's feature rows to a per-texel table rebuilt every step (auto: when R*S >= 2*H*W)")
    ap.add_argument('--train-steps', type=int, default=5,
                    help='N=1 only: also time this many train_step calls (fwd + bwd + clip + Adam) on the same scene, reported as an extra object (0 = skip)')
    ap.add_argument('--fused-call', action='store_true', help='time mvnerf_render_fwd (one C call) instead of the op sequence')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    backend = os.environ.get('MVNERF_BENCH_BACKEND', 'nccl')      # 'gloo' only to rehearse N>1 on a one-GPU box
    n_dev = torch.cuda.device_count()
    dev_index = local_rank if backend == 'nccl' else local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)       # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    img_h, img_w = args.height or args.size, args.width or args.size
    sc = make_scene(seed=rank, batch=1, n_views=args.views, height=img_h, width=img_w, n_rays=args.rays or None)
    t = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in
         ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'u_coarse', 'u_fine', 'coarse', 'fine']}
    pc, pf = ops.pack_net(t['coarse']), ops.pack_net(t['fine'])
    bf16 = args.dtype == 'bf16'
    if bf16:
        pc16, pf16 = ops.pack_net_bf16(t['coarse']), ops.pack_net_bf16(t['fine'])
    b, r, s = t['u_coarse'].shape
    ws = torch.empty(ops.render_workspace_bytes(b, args.views, r, s), dtype=torch.uint8, device=dev)
    near, far = sc['near'], sc['far']
    field_args = (t['images'], t['features'], t['intrinsics'], t['extrinsics_inv'])
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(args.steps)]
    use_table = args.texel_table == 'on' or (args.texel_table == 'auto' and ops.texel_table_pays(r, s, img_h, img_w))
    tables = torch.empty((2, b, args.views, img_h, img_w, 128), dtype=torch.float32, device=dev) if use_table else None

    def step_ops(e=None):
        tab_c = tab_f = None
        if e: e[4].record()
        if use_table:                      # inside the step: a new _call brings new feature maps
            if bf16:
                tab_c, tab_f = ops.project_texels_bf16(t['features'], pc16, out=tables, packed16_b=pf16).unbind(0)
            else:
                tab_c, tab_f = ops.project_texels2(t['features'], pc, pf, out=tables).unbind(0)
        if e: e[5].record()
        z = ops.stratified_depths(t['u_coarse'], near, far)
        if e: e[0].record()
        rgbs_c = (ops.field_eval_bf16(t['rays_o'], t['rays_d'], z, *field_args, pc, pc16, texel_table=tab_c) if bf16 else
                  ops.field_eval(t['rays_o'], t['rays_d'], z, *field_args, pc, texel_table=tab_c))
        if e: e[1].record()
        rgb, depth, w = ops.composite(z, rgbs_c)
        z_all = ops.resample(z, w, t['u_fine'])
        if e: e[2].record()
        rgbs_f = (ops.field_eval_bf16(t['rays_o'], t['rays_d'], z_all, *field_args, pf, pf16, texel_table=tab_f) if bf16 else
                  ops.field_eval(t['rays_o'], t['rays_d'], z_all, *field_args, pf, texel_table=tab_f))
        if e: e[3].record()
        fine_rgb, fine_depth, _ = ops.composite(z_all, rgbs_f, return_weights=False)
        return rgb, depth, fine_rgb, fine_depth

    def step_fused(e=None):
        return ops.render_fwd(t['rays_o'], t['rays_d'], *field_args, pc, pf, t['u_coarse'], t['u_fine'], near, far,
                              workspace=ws, texel_tables=tables)

    step = step_fused if (args.fused_call and not bf16) else step_ops
    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(ev[i])
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dev if backend == 'nccl' else 'cpu')   # the slowest rank's clock (no-op at world 1)

    rays_per_step = b * r * world
    ms_per_step = 1e3 * elapsed / args.steps
    value = rays_per_step * args.steps / elapsed

    result = {
        'metric': 'rendered rays/sec (64 samples/ray)', 'value': value, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': f'cfg2: _call on B=1 scene/GPU, V={args.views} source view {img_h}x{img_w}x(3+256) fp32, '
                               f'R={r} rays ({"random pixels" if args.rays else "all pixels"} of a {img_h}x{img_w} target), 64 coarse + 128 fine samples/ray, '
                               'two 247300-param ResNet-MLPs (379->128, 3+3 blocks), explicit uniforms',
                   'ray_definition': 'one full _call row: 64 stratified coarse samples + 128 merged fine samples through both MLPs '
                                     '(the conservative reading; the coarse pass alone is roofline.coarse_only_rays_per_sec)',
                   'rays_per_gpu': b * r, 'samples_per_ray': [s, 2 * s], 'n_views': args.views,
                   'call': 'mvnerf_render_fwd' if args.fused_call else f'op sequence ({7 if use_table else 6} C-ABI calls/step)',
                   'layer0_features': 'texel table, rebuilt every step' if use_table else 'gathered per sample',
                   'parallelism': f'ray/scene sharding x{world}, no data-path collective'},
    }

    if rank == 0:
        fps = flop_per_sample(args.views, use_table)
        fps_ref = flop_per_sample(args.views)
        if not args.fused_call:
            coarse_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
            fine_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))
            flops_c, flops_f = fps * b * r * s, fps * b * r * 2 * s
            # the fine-pass launch is the dominant kernel instance (2/3 of the FLOPs)
            achieved = flops_f / (fine_ms * 1e-3) / 1e12
            peak = 2500.0 if bf16 else PEAK_FP32_MFMA_TFLOPS          # dense bf16 MFMA peak ~2.5 PFLOP/s
            kname = (('field_eval_bf16_kernel' if bf16 else 'field_eval_kernel') + ('<true' if args.views > 1 else '<false') +
                     ((',true>' if use_table else ',false>') if bf16 else (',false,true>' if use_table else ',false,false>')))
            result['roofline'] = {
                'bound': 'mfma', 'kernel': kname + ' (fine pass, S=128)',
                'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                'traffic': None, 'flop_per_launch': flops_f, 'avg_launch_ms': fine_ms,
                'coarse_launch': {'flop_per_launch': flops_c, 'avg_launch_ms': coarse_ms,
                                  'achieved': flops_c / (coarse_ms * 1e-3) / 1e12},
                'field_kernel_share_of_step': (coarse_ms + fine_ms) / ms_per_step,
                # the literal '64 samples/ray' reading of the metric = the coarse pass alone
                'coarse_only_rays_per_sec': b * r / (coarse_ms * 1e-3),
            }
            if use_table:
                # `achieved` counts the FLOPs the kernel executes; the reference's graph multiplies the 256 feature
                # rows per sample, so the same launch stands for more "reference FLOPs" than it runs
                result['roofline']['reference_flop_per_launch'] = fps_ref * b * r * 2 * s
                result['roofline']['reference_equiv_tflops'] = fps_ref * b * r * 2 * s / (fine_ms * 1e-3) / 1e12
                # SURVEY.md 8d's algorithmic figure (491 264 FLOP/sample at V=1) over the same duration; > 1 is possible because
                # 13 % of those FLOPs are not executed per sample any more - `frac` above is the hardware utilisation
                result['roofline']['frac_reference_equiv'] = result['roofline']['reference_equiv_tflops'] / peak
                result['roofline']['project_texels_ms_per_step'] = float(np.mean([e[4].elapsed_time(e[5]) for e in ev]))
            pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
            if os.path.exists(pmc) and not bf16 and args.views == 1 and (img_h, img_w, r) == (64, 64, 4096):   # counters were collected on cfg2
                try:
                    key = 'field_eval_table_fine_hbm_bytes_per_launch' if use_table else 'field_eval_fine_hbm_bytes_per_launch'
                    counters = json.load(open(pmc))
                    result['roofline']['traffic'] = counters.get(key)
                    if counters.get(key):                  # rocprof counters (profiles/): HBM-side GB/s of this launch, matrix pipe busy
                        result['roofline']['hbm_gbps_from_pmc_traffic'] = counters[key] / (fine_ms * 1e-3) / 1e9
                        result['roofline']['mfma_busy_pmc'] = counters.get(key.replace('hbm_bytes_per_launch', 'mfma_busy'))
                except Exception:
                    pass
        if world == 1 and args.train_steps > 0 and not bf16:
            result['train_step'] = train_throughput(sc, t, args.views, dev, args.train_steps)
        if world == 1 and args.cpu_rays > 0:
            result['cpu_baseline'], result['parity'] = cpu_baseline(sc, out, args.cpu_rays)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def train_throughput(sc, t, n_views, dev, steps):
    """cfg2 'fwd+bwd rays/s' (SURVEY.md 8d): MVVNeRFRenderer.train_step (model_v0.py:186-197) = forward with stash,
    full backward (incl. the gradient through the importance samples), clip-by-value, Adam; not part of `value`."""
    from thesis_clip_nerf_amd import MVVNeRFRenderer
    r = t['rays_o'].shape[1]
    m = MVVNeRFRenderer(r, r, n_views=n_views, near=sc['near'], far=sc['far'], device=dev)
    m.set_weights(sc['coarse'], sc['fine'])
    m.compile(learning_rate=1e-4)
    inputs = tuple(t[k] for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv'])
    y = torch.rand((1, r, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    kw = dict(combined_features=t['features'], u_coarse=t['u_coarse'], u_fine=t['u_fine'])
    for _ in range(2):
        m.train_step((inputs, y), **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train_step((inputs, y), **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {'rays_per_sec': r / dt, 'ms_per_step': 1e3 * dt, 'steps': steps,
            'what': 'train_step: fwd (activations stashed) + bwd of both nets incl. d/d(sample depth) + clip + Adam, fp32'}


def cpu_baseline(sc, gpu_out, n_rays):
    """Oracle (oracle/mvnerf_oracle.py, NumPy fp32 port of the TF graph) on the first `n_rays` rays of
    the benchmark scene, on this box's host cores; also the checker for the GPU result on those rays."""
    from oracle import mvnerf_oracle as O
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count()
    cn, fn = O.unflatten_net(sc['coarse']), O.unflatten_net(sc['fine'])
    sl = slice(0, n_rays)

    def run():
        return O.render_call(cn, fn, sc['rays_o'][:, sl], sc['rays_d'][:, sl], sc['images'], sc['intrinsics'],
                             sc['extrinsics_inv'], sc['features'], sc['near'], sc['far'], sc['n_samples'],
                             sc['u_coarse'][:, sl], sc['u_fine'][:, sl])
    run()                                   # warm-up (BLAS thread pool, page faults)
    times = []
    ref = None
    for _ in range(3):
        t0 = time.perf_counter()
        ref = run()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    cpu_model = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            cpu_model = next((l.split(':', 1)[1].strip() for l in f if l.startswith('model name')), 'unknown')
    except OSError:
        pass
    base = {'value': n_rays / med, 'unit': 'rays/s', 'cores': int(blas_threads), 'kind': 'port', 'cpu_model': cpu_model,
            'sample': f'first {n_rays} of the 4096 rays of the same scene, full 64+128 samples, NumPy fp32 oracle '
                      f'(op-for-op port of the TF graph; BLAS sgemm uses {blas_threads} threads, the rest is single-threaded), '
                      f'median of 3 after 1 warm-up, {med:.2f} s per run; host has {os.cpu_count()} logical cores'}
    names = ['rgb', 'depth', 'fine_rgb', 'fine_depth']
    parity = {}
    for n, g, rf in zip(names, gpu_out, ref):
        diff = np.abs(g.cpu().numpy()[:, sl] - rf)
        parity[n + '_max_abs'] = float(diff.max())
        parity[n + '_mean_l1'] = float(diff.mean())
    parity['checked_rays'] = n_rays
    parity['tolerance'] = 1e-4
    return base, parity


if __name__ == '__main__':
    main()
