#!/usr/bin/env python3
"""Benchmark of the MVNeRF render hot path on MI355X (BASELINE.json metric: rendered rays/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
`torch.distributed.run`, spawned before this process touches the GPU or loads the HIP library) and relays rank 0's line.

One "step" = one full `_call` (model_v0.py:113-184) over one batch of synthetic rays: stratified
depths -> coarse field (64 samples/ray) -> composite -> resample -> fine field (128 samples/ray) ->
composite.  Workload at every N: BASELINE.json configs[1] as restated in SURVEY.md 8d (cfg2):
B=1 scene, V=1 source view of 64x64 (3+256 channels), R=4096 rays (every pixel of a 64x64 target
view), fp32, inputs resident in HBM before the timed region.  N>1: every rank renders its own
scene (rays/scenes are independent units, no data-path collective) -> weak scaling.

`--scaling strong` (SURVEY.md 8d cfg4, strong leg): ONE fixed job - the 16 384 rays of a 128x128 scene - split over the N ranks by
`distributed.shard_bounds` (contiguous ray blocks, replicated weights and source view, no data-path collective); `value` is then
16 384 x K / time and `scaling` reads "strong".  The default (and what the driver runs) is the weak leg above.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, the fine-pass launch of the field kernel.  Default
(`--f32-gemm split_f16`): `field_eval_split16h_kernel` - every fp32 operand as two fp16 pieces (round-to-nearest twice: 22-24
significant bits), three v_mfma_f32_16x16x32_f16 per product block, fp32 accumulation; fp32-grade (per ResNet block as close to a
float64 evaluation as the fp32 MFMA kernel, tests/test_gpu_split.py) - so the pipe it is bound by is the 16-bit matrix pipe:
`peak` = 2 500 TFLOP/s dense fp16 / bf16 (`peak_dtype`), `achieved` / `frac` = `frac_executed` count the FLOPs the launch EXECUTES
there (tiles x MFMAs per 32-sample tile x 16 384 FLOP per MFMA; the count is checked against SQ_INSTS_MFMA in profiles/), and
`frac_algorithmic` restates the same duration in SURVEY.md 8d's algorithmic figure (491 264 FLOP per sample at V = 1) against the
same peak - 2.6x lower, the price of fp32-grade products on a 16-bit pipe.  `--f32-gemm split_bf16` times `field_eval_split16_kernel`
(exact three-piece bf16 cut, six v_mfma_f32_16x16x32_bf16 per block: round 3's first default).  `--f32-gemm mfma_f32` times `field_eval_kernel` on the
fp32 MFMA (v_mfma_f32_32x32x2_f32, 4 096 FLOP each, peak 157.3 TFLOP/s).  Durations are HIP events inside the timed region;
`traffic` / `traffic_ratio` / `mfma_busy_pmc` come from the committed rocprofv3 counter passes (`traffic_source`).
`cpu_baseline` times the op-for-op torch-CPU restatement of the reference's TF graph (oracle/mvnerf_torch.py, fp32; the reference
itself cannot run here) on all 4096 rays on the host cores; `parity` checks the GPU result against the NumPy oracle on all 4096
rays.  `train_cfg4` is the data-parallel training leg (BASELINE.json configs[3]): one 128x128 scene = 16 384 rays per GPU,
forward with stash + backward + ONE flat all-reduce of the 494 600 gradients + clip + Adam.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE_V1 = 491264          # BASELINE.md 3 (2 x 245 632 MAC), one source view
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md, chip-level parameters
METRIC = 'rendered rays/sec (64 samples/ray) at 1/2/4/8 MI355X; RGB L1 vs ref'


def reference_flop_per_sample(v):
    """per view Dense 379->128 + 3 blocks, then 3 blocks + read-out, 2 FLOP per MAC."""
    return 2 * (v * (379 * 128 + 6 * 128 * 128) + 6 * 128 * 128 + 128 * 4)


def mfma_per_tile(v, table, bf16=False, split=False):
    """Matrix instructions one 32-sample tile issues: layer 0 streams K = 64 rows (PE(cam xyz) 60 + rgb 3,
    padded; the 60 PE(cam dir) rows are a per-ray seed computed on the vector ALU) + 256 feature rows unless those come from
    the texel table; 12 hidden layers of K = 128; the 128->4 read-out runs on the vector ALU.  One k-step covers K = 2 (fp32
    32x32x2) for each of the 4 output blocks of 32 features."""
    if split in (16, '16h'):
        # field_eval_split16[h].hip: k-steps of K = 32 rows x 8 row blocks x 2 column blocks x 6 products of v_mfma_f32_16x16x32_bf16
        # (three bf16 pieces per operand) or 3 products of v_mfma_f32_16x16x32_f16 (two fp16 pieces: the default), 16 384 FLOP each;
        # layer 0: 2 k-steps (+ 8 for the feature rows without the texel table); 6 Dense layers x 4 k-steps per view, 6 x 4 fused;
        # the read-out on the vector ALU
        return (48 if split == '16h' else 96) * (v * ((2 if table else 10) + 24) + 24)
    if split:      # field_eval_split.hip: per k-step of 16 rows 4 output blocks x 6 products; PE(dir) in the per-ray seed
        return 24 * (v * ((4 if table else 20) + 48) + 48) + 48
    if bf16 == 'x':  # field_eval_bf16x.hip (texel-table form at V = 1): v_mfma_f32_16x16x32_bf16, 8 row blocks x 2 column blocks per t-step of K = 32;
        #              layer 0: 2 t-steps (PE(cam xyz) + rgb), 12 hidden layers x 4 t-steps; read-out on the vector ALU in fp32
        return 16 * (v * (2 + 24) + 24)
    if bf16:       # field_eval_bf16.hip: K = 16 per MFMA; direct form streams PE(xyz) 64 + PE(dir) 64 + 256 feature rows, the table
        #            form 64 rows (PE(dir) in the per-ray seed, features from the table); read-out = 2 k-steps x 4 blocks on the MFMA
        l0 = (64 if table else 384) // 16 * 4
        return v * (l0 + 6 * 32) + 6 * 32 + 8
    l0 = (64 + (0 if table else 256)) // 2 * 4
    hidden = 128 // 2 * 4
    return v * (l0 + 6 * hidden) + 6 * hidden


def flop_per_mfma(bf16=False, split=False):
    if split in (16, '16h'):
        return 2 * 16 * 16 * 32
    return 2 * 32 * 32 * (16 if bf16 else 2)


def self_launch(args):
    """N > 1 without a launcher: start the ranks as a child process tree.  Nothing in this process has touched the GPU or
    loaded libmvnerf_hip.so yet (no exec of a GPU-initialised process, no GPU context in the parent)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return proc.returncode if proc.returncode != 0 or line is not None else 1


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
    ap.add_argument('--cpu-baseline', default='on', choices=['on', 'off'], help='N=1: time the torch-CPU restatement and check parity on all rays')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='bf16: both field passes on the bf16 MFMA kernel (configs 3/5; not the headline)')
    ap.add_argument('--f32-gemm', default='split_f16', choices=['mfma_f32', 'split_bf16', 'split_f16'],
                    help='fp32 Dense layers on the fp32 MFMA, or fp32-grade on the 16-bit matrix pipe: three fp16 MFMAs per product block on two-piece operands (default) or six bf16 MFMAs on exactly cut three-piece operands')
    ap.add_argument('--texel-table', default='auto', choices=['auto', 'on', 'off'],
                    help="hoist layer 0's feature rows to a per-texel table rebuilt every step (auto: when R*S >= 2*H*W)")
    ap.add_argument('--train-steps', type=int, default=5,
                    help='timed train_step calls of the two training legs (cfg2 scene at N=1; cfg4 128x128 scene per GPU with the gradient all-reduce at every N); 0 = skip')
    ap.add_argument('--fused-call', action='store_true', help='time mvnerf_render_fwd (one C call) instead of the op sequence')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak: one cfg2 scene per GPU (default, the driver\'s run); strong: the 16 384 rays of ONE 128x128 scene (cfg4) split over the ranks')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))

    import numpy as np
    import torch
    from thesis_clip_nerf_amd import ops
    from thesis_clip_nerf_amd.distributed import max_over_ranks, shard_bounds
    from thesis_clip_nerf_amd.synthetic import make_scene

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
    dist = None
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

    strong = args.scaling == 'strong'
    if strong and args.size == 64 and not (args.height or args.width or args.rays):
        args.size = 128                                           # cfg4's scene: 128 x 128 = 16 384 rays in total
    img_h, img_w = args.height or args.size, args.width or args.size
    sc = make_scene(seed=0 if strong else rank, batch=1, n_views=args.views, height=img_h, width=img_w, n_rays=args.rays or None)
    total_rays = sc['rays_o'].shape[1]
    if strong:                                                    # this rank's contiguous block of the one job's rays
        lo, hi = shard_bounds(total_rays, rank, world)
        for k in ('rays_o', 'rays_d', 'u_coarse', 'u_fine'):
            sc[k] = np.ascontiguousarray(sc[k][:, lo:hi])
    t = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in
         ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'u_coarse', 'u_fine', 'coarse', 'fine']}
    pc, pf = ops.pack_net(t['coarse']), ops.pack_net(t['fine'])
    bf16 = args.dtype == 'bf16'
    split = args.f32_gemm != 'mfma_f32' and not bf16
    if split:
        ops.set_split_kernel(args.f32_gemm)
    if bf16:
        pc16, pf16 = ops.pack_net_bf16(t['coarse']), ops.pack_net_bf16(t['fine'])
    if split:
        pcs, pfs = ops.pack_net_split(t['coarse']), ops.pack_net_split(t['fine'])
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
                  ops.field_eval_split(t['rays_o'], t['rays_d'], z, *field_args, pc, pcs, texel_table=tab_c) if split else
                  ops.field_eval(t['rays_o'], t['rays_d'], z, *field_args, pc, texel_table=tab_c))
        if e: e[1].record()
        rgb, depth, w = ops.composite(z, rgbs_c)
        z_all = ops.resample(z, w, t['u_fine'])
        if e: e[2].record()
        rgbs_f = (ops.field_eval_bf16(t['rays_o'], t['rays_d'], z_all, *field_args, pf, pf16, texel_table=tab_f) if bf16 else
                  ops.field_eval_split(t['rays_o'], t['rays_d'], z_all, *field_args, pf, pfs, texel_table=tab_f) if split else
                  ops.field_eval(t['rays_o'], t['rays_d'], z_all, *field_args, pf, texel_table=tab_f))
        if e: e[3].record()
        fine_rgb, fine_depth, _ = ops.composite(z_all, rgbs_f, return_weights=False)
        return rgb, depth, fine_rgb, fine_depth

    def step_fused(e=None):
        return ops.render_fwd(t['rays_o'], t['rays_d'], *field_args, pc, pf, t['u_coarse'], t['u_fine'], near, far,
                              workspace=ws, texel_tables=tables, split=(pcs, pfs) if split else None)

    step = step_fused if (args.fused_call and not bf16) else step_ops
    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(ev[i])
    barrier()
    elapsed = time.perf_counter() - t0
    red_dev = dev if backend == 'nccl' else 'cpu'
    elapsed = max_over_ranks(elapsed, red_dev)   # the slowest rank's clock (no-op at world 1)

    rays_per_step = total_rays if strong else b * r * world
    ms_per_step = 1e3 * elapsed / args.steps
    value = rays_per_step * args.steps / elapsed

    result = {
        'metric': METRIC, 'value': value, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': args.scaling,
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic', 'f32_gemm': None if bf16 else args.f32_gemm,
        'config': {'workload': (f'cfg4 strong leg: ONE scene of {total_rays} rays split over {world} rank(s) by shard_bounds ({r} rays on rank 0), ' if strong else 'cfg2: ') +
                               f'_call on B=1 scene/GPU, V={args.views} source view {img_h}x{img_w}x(3+256) fp32, '
                               f'R={r} rays ({"random pixels" if args.rays else "all pixels"} of a {img_h}x{img_w} target), 64 coarse + 128 fine samples/ray, '
                               'two 247300-param ResNet-MLPs (379->128, 3+3 blocks), explicit uniforms',
                   'ray_definition': 'one full _call row: 64 stratified coarse samples + 128 merged fine samples through both MLPs '
                                     '(the conservative reading; the coarse pass alone is roofline.coarse_only_rays_per_sec)',
                   'rays_per_gpu': b * r, 'samples_per_ray': [s, 2 * s], 'n_views': args.views,
                   'call': 'mvnerf_render_fwd' if args.fused_call else f'op sequence ({7 if use_table else 6} C-ABI calls/step)',
                   'layer0_features': 'texel table, rebuilt every step' if use_table else 'gathered per sample',
                   'parallelism': f'ray/scene sharding x{world}, no data-path collective',
                   'launcher': 'torch.distributed.run' if world > 1 else 'single process'},
    }

    if rank == 0 and not args.fused_call:
        coarse_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        fine_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))
        n_tiles_c, n_tiles_f = (b * r * s + 31) // 32, (b * r * 2 * s + 31) // 32
        # which split kernel runs the inference passes (csrc/field_eval_split.hip, split_kernel_choice): field_eval_split16h_kernel (two fp16
        # pieces, three products) unless MVNERF_SPLIT_MFMA pins field_eval_split16_kernel ("bf16x6") or round 2's kernel ("32x32x16")
        env_split = os.environ.get('MVNERF_SPLIT_MFMA', '') or {'split_f16': 'f16x3', 'split_bf16': 'bf16x6'}.get(args.f32_gemm, '')
        split_shape = (32 if env_split.startswith('3') else 16 if env_split.startswith('b') else '16h') if split else False
        # bf16: the layer-ring kernel (field_eval_bf16x.hip) runs the texel-table form at V = 1 unless MVNERF_BF16_KERNEL=segments pins round 2's
        bf16x = bf16 and use_table and args.views == 1 and not os.environ.get('MVNERF_BF16_KERNEL', '').startswith('s')
        mpt = mfma_per_tile(args.views, use_table, 'x' if bf16x else bf16, split_shape)
        fpm = flop_per_mfma(bf16 or bool(split), 16 if bf16x else split_shape)
        flops_c, flops_f = n_tiles_c * mpt * fpm, n_tiles_f * mpt * fpm
        fps_ref = reference_flop_per_sample(args.views)
        # the fine-pass launch is the dominant kernel instance (2/3 of the FLOPs)
        achieved = flops_f / (fine_ms * 1e-3) / 1e12
        peak = 2500.0 if (bf16 or split) else PEAK_FP32_MFMA_TFLOPS   # dense bf16 MFMA peak ~2.5 PFLOP/s
        kname = ((('field_eval_bf16x_kernel' if bf16x else 'field_eval_bf16_kernel') if bf16 else ('field_eval_split16h_kernel' if split_shape == '16h' else 'field_eval_split16_kernel' if split_shape == 16 else 'field_eval_split_kernel') if split else 'field_eval_kernel') +
                 ('<true' if args.views > 1 else '<false') +
                 ((',false>' if bf16x else ',true>' if use_table else ',false>') if bf16 else (',true,false,false>' if use_table else ',false,false,false>') if split_shape in (16, '16h') else (',true,false>' if use_table else ',false,false>') if split else
                  (',false,true>' if use_table else ',false,false>')))
        ref_tflops = fps_ref * b * r * 2 * s / (fine_ms * 1e-3) / 1e12
        peak_dtype = ('f16' if split_shape == '16h' else 'bf16') if (bf16 or split) else 'f32'
        result['roofline'] = {
            'bound': 'mfma', 'kernel': kname + ' (fine pass, S=128)',
            'achieved': achieved, 'peak': peak, 'peak_dtype': peak_dtype, 'unit': 'TFLOP/s', 'frac': achieved / peak,
            # frac_executed: FLOPs issued on the matrix pipe named by peak_dtype / its dense peak (= frac);
            # frac_algorithmic: SURVEY.md 8d's reference-graph FLOPs (491 264 per sample at V = 1) over the same duration / the same peak
            'frac_executed': achieved / peak, 'frac_algorithmic': ref_tflops / peak,
            'peak_note': ('dense bf16 MFMA peak 2.5 PFLOP/s assumes 2.4 GHz; under sustained bf16-MFMA load this chip holds about 1.9-2.0 GHz '
                          '(in-kernel s_memtime / s_memrealtime, DESIGN.md 4.0), i.e. a matrix pipe that never idles reads about 0.8 here'
                          if (bf16 or split) else 'fp32 MFMA peak 157.3 TFLOP/s (MI355X_MICROARCH.md)'),
            'traffic': None, 'traffic_ratio': None, 'traffic_source': None,
            'flop_per_launch': flops_f, 'avg_launch_ms': fine_ms,
            'flop_count': f'executed on the matrix pipe: {n_tiles_f} tiles x {mpt} MFMAs x {fpm} FLOP',
            'tiles_per_launch': n_tiles_f, 'mfma_per_tile': mpt,
            'coarse_launch': {'flop_per_launch': flops_c, 'avg_launch_ms': coarse_ms,
                              'achieved': flops_c / (coarse_ms * 1e-3) / 1e12},
            'field_kernel_share_of_step': (coarse_ms + fine_ms) / ms_per_step,
            # the literal '64 samples/ray' reading of the metric = the coarse pass alone
            'coarse_only_rays_per_sec': b * r / (coarse_ms * 1e-3),
            # SURVEY.md 8d's algorithmic figure (491 264 FLOP/sample at V=1) over the same duration; against the fp32 peak it can
            # exceed 1 because the PE(dir) rows are hoisted to a per-ray seed and, with the table, the feature rows to per-texel products
            'reference_flop_per_launch': fps_ref * b * r * 2 * s,
            'reference_equiv_tflops': ref_tflops, 'frac_reference_equiv': ref_tflops / peak,
        }
        if split:
            # the split kernel delivers fp32-grade products on the bf16 pipe: six MFMAs per fp32 product block.  `frac` above is
            # the matrix-pipe utilisation (executed bf16 FLOPs / 2.5 PFLOP/s); this is the same launch measured in the
            # reference graph's fp32 FLOPs against the fp32 MFMA peak the round-1 kernel was bound by
            result['roofline']['reference_equiv_vs_fp32_mfma_peak'] = ref_tflops / PEAK_FP32_MFMA_TFLOPS
            result['roofline']['arithmetic'] = (
                'fp32 operands as two fp16 pieces (round-to-nearest twice, the remainder scaled by 64: 22-24 significant bits), 3 v_mfma_f32_16x16x32_f16 '
                'per product block, fp32 accumulate; per ResNet block as close to a float64 evaluation as the fp32 MFMA kernel '
                '(tests/test_gpu_split.py::test_products_of_the_three_fp32_grade_kernels_against_float64)' if split_shape == '16h' else
                'fp32 operands cut exactly into 3 bf16 pieces, 6 ' + ('v_mfma_f32_16x16x32_bf16' if split_shape == 16 else 'v_mfma_f32_32x32x16_bf16') +
                ' per product block, fp32 accumulate (dropped terms <= 2^-24 relative)')
        if use_table:
            result['roofline']['project_texels_ms_per_step'] = float(np.mean([e[4].elapsed_time(e[5]) for e in ev]))
        pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc) and not bf16 and (use_table or not split) and args.views == 1 and (img_h, img_w, r) == (64, 64, 4096):   # counters were collected on cfg2
            try:
                key = ('field_eval_split16h_table_fine_hbm_bytes_per_launch' if split_shape == '16h' else 'field_eval_split16_table_fine_hbm_bytes_per_launch' if split_shape == 16 else
                       'field_eval_split_table_fine_hbm_bytes_per_launch' if split else
                       'field_eval_table_fine_hbm_bytes_per_launch' if use_table else 'field_eval_fine_hbm_bytes_per_launch')
                counters = json.load(open(pmc))
                rl = result['roofline']
                rl['traffic'] = counters.get(key)
                if counters.get(key):                  # rocprof counters (profiles/): HBM-side GB/s of this launch, matrix pipe busy
                    alg = counters.get('algorithmic_bytes_per_launch')
                    rl['traffic_algorithmic_bytes'] = alg
                    rl['traffic_ratio'] = counters[key] / alg if alg else None
                    rl['traffic_source'] = 'profiles/pmc_traffic.json <- ' + str(counters.get('source_split16h' if split_shape == '16h' else 'source_split16' if split_shape == 16 else 'source_split' if split else 'source_table' if use_table else 'source'))[:200]
                    rl['traffic_note'] = ('HBM-side bytes (FETCH_SIZE x2 + WRITE_SIZE) exceed the algorithmic bytes because each of the 8 XCD L2s pulls '
                                          'its own copy of the texel table and the weight stream, and because of the kernel\'s remaining register spills '
                                          '(scratch); at the rate below that is about 1 % of the 8 TB/s HBM peak')
                    rl['hbm_gbps_from_pmc_traffic'] = counters[key] / (fine_ms * 1e-3) / 1e9
                    rl['mfma_busy_pmc'] = counters.get(key.replace('hbm_bytes_per_launch', 'mfma_busy'))
                    mops = counters.get(key.replace('hbm_bytes_per_launch', 'mfma_flop_per_launch_pmc'))
                    if mops:
                        rl['flop_per_launch_pmc'] = mops
            except Exception:
                pass

    if args.train_steps > 0 and not bf16:
        # the training legs are reported beside the headline, never instead of it: a failure in one of them (it would be the same on every
        # rank: they run the same code on the same shapes) is recorded and the line is still printed
        if world == 1 and not strong:
            try:
                result['train_step'] = train_throughput(sc, t, args.views, dev, args.train_steps)
            except Exception as exc:               # noqa: BLE001
                result['train_step'] = {'error': f'{type(exc).__name__}: {exc}'[:300]}
        try:
            leg = train_cfg4(rank, world, dev, backend, args.train_steps, barrier, red_dev, strong)
        except Exception as exc:                   # noqa: BLE001
            leg = {'error': f'{type(exc).__name__}: {exc}'[:300]}
        if rank == 0:
            result['train_cfg4'] = leg
    if rank == 0:
        if world == 1 and args.cpu_baseline == 'on' and (img_h, img_w) == (64, 64) and args.views == 1 and not args.rays and not strong:
            result['cpu_baseline'], result['parity'] = cpu_baseline(sc, out)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def train_throughput(sc, t, n_views, dev, steps):
    """cfg2 'fwd+bwd rays/s' (SURVEY.md 8d): MVVNeRFRenderer.train_step (model_v0.py:186-197) = forward with stash,
    full backward (incl. the gradient through the importance samples), clip-by-value, Adam; not part of `value`."""
    import torch
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


def train_cfg4(rank, world, dev, backend, steps, barrier, red_dev, strong=False):
    """BASELINE.json configs[3] as restated in SURVEY.md 8d (cfg4): one scene per GPU, 128x128 source view, 16 384 rays (every
    pixel of a 128x128 target), forward with stash + backward + ONE flat all-reduce (mean) of the 494 600 fp32 gradients of both
    MLPs + clip + Adam (model_v0.py:186-197 per rank; the mean over ranks is the gradient of the `world`-scene batch).  Timed
    like the main leg: barrier + synchronize on both sides, max over ranks; rays/s is the whole job's.
    strong: ONE scene's 16 384 rays split over the ranks by shard_bounds (equal blocks for N = 1, 2, 4, 8: the mean over ranks of the
    per-block mean-squared-error gradients is the whole scene's gradient)."""
    import numpy as np
    import torch
    from thesis_clip_nerf_amd import MVVNeRFRenderer
    from thesis_clip_nerf_amd.distributed import allreduce_mean_, max_over_ranks, shard_bounds
    from thesis_clip_nerf_amd.synthetic import make_scene
    sc = make_scene(seed=1000 if strong else 1000 + rank, batch=1, n_views=1, height=128, width=128)
    total = sc['rays_o'].shape[1]
    if strong:
        lo_r, hi_r = shard_bounds(total, rank, world)
        for k in ('rays_o', 'rays_d', 'u_coarse', 'u_fine'):
            sc[k] = np.ascontiguousarray(sc[k][:, lo_r:hi_r])
    t = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in
         ['rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'u_coarse', 'u_fine']}
    r = t['rays_o'].shape[1]
    job = total if strong else world * r
    m = MVVNeRFRenderer(r, r, n_views=1, near=sc['near'], far=sc['far'], device=dev, seed=0)        # same initial weights on every rank
    m.compile(learning_rate=1e-4, grad_sync=allreduce_mean_ if world > 1 else None)
    inputs = tuple(t[k] for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv'])
    y = torch.rand((1, r, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
    kw = dict(combined_features=t['features'], u_coarse=t['u_coarse'], u_fine=t['u_fine'])

    def timed(fn, n, warm):
        for _ in range(warm):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        barrier()
        return max_over_ranks((time.perf_counter() - t0) / n, red_dev)

    dt_fwd = timed(lambda: m.infer(inputs, t['features'], u_coarse=t['u_coarse'], u_fine=t['u_fine']), steps, 1)
    # data parallel: the fine half of the flat gradient is all-reduced on a second stream while the coarse net's backward still runs
    # (distributed.OverlappedGradSync, mvnerf_train_call.fine_grad_event); the single flat collective is timed beside it
    overlap = None
    if world > 1:
        from thesis_clip_nerf_amd.distributed import OverlappedGradSync
        overlap = OverlappedGradSync(dev)
        m._grad_sync = overlap
    dt_train = timed(lambda: m.train_step((inputs, y), **kw), steps, 2)
    leg = {'workload': ('cfg4 strong leg: ONE scene of 16384 rays split over the ranks, ' if strong else 'cfg4: B=1 scene/GPU, ') +
                       'V=1 source view 128x128x(3+256) fp32, 16384 rays (all pixels), 64+128 samples/ray',
           'scaling': 'strong' if strong else 'weak', 'rays_per_gpu': r, 'steps': steps,
           'forward_rays_per_sec': job / dt_fwd, 'forward_ms_per_step': 1e3 * dt_fwd,
           'train_rays_per_sec': job / dt_train, 'train_ms_per_step': 1e3 * dt_train,
           'what': 'train_step = fwd (stash) + bwd of both nets incl. d/d(sample depth) + gradient all-reduce (mean; fine half overlapped with the '
                   'coarse backward) + clip + Adam, fp32; one C call each for mvnerf_loss_and_grads and mvnerf_apply_gradients'}
    if world > 1:
        buf = torch.zeros(2 * 247300, dtype=torch.float32, device=dev)
        dt_ar = timed(lambda: allreduce_mean_(buf), 20, 3)
        leg['allreduce_ms'] = 1e3 * dt_ar
        leg['allreduce'] = f'one {buf.numel() * 4} B fp32 all-reduce per step ({backend}), {1e3 * dt_ar:.3f} ms = {dt_ar / dt_train:.1%} of the step'
        # every rank must hold the same weights after the same averaged updates
        chk = torch.stack([m.coarse_net.double().sum(), m.fine_net.double().sum()])
        lo, hi = chk.clone(), chk.clone()
        import torch.distributed as dist
        if backend != 'nccl':
            lo, hi = lo.cpu(), hi.cpu()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        leg['weights_identical_across_ranks'] = bool(torch.equal(lo, hi))
        # the same step with ONE flat all-reduce behind the whole backward, and with no collective at all (ranks then diverge: last)
        m._grad_sync = allreduce_mean_
        dt_flat = timed(lambda: m.train_step((inputs, y), **kw), steps, 1)
        m._grad_sync = None
        dt_none = timed(lambda: m.train_step((inputs, y), **kw), steps, 1)
        leg['grad_sync'] = {'overlapped_ms_per_step': 1e3 * dt_train, 'flat_ms_per_step': 1e3 * dt_flat, 'no_collective_ms_per_step': 1e3 * dt_none,
                            'exposed_ms_overlapped': 1e3 * (dt_train - dt_none), 'exposed_ms_flat': 1e3 * (dt_flat - dt_none),
                            'note': 'exposed = step time minus the step without any collective; overlapped: fine half (0.99 MB) reduced on a second '
                                    'stream behind mvnerf_train_call.fine_grad_event while the coarse backward runs, coarse half after it'}
    return leg


def host_cores():
    """Threads the CPU baseline may use: physical cores visible to this process (SMT siblings collapsed), capped by the
    scheduler affinity and a cgroup CPU quota if there is one."""
    cores = set()
    try:
        phys = core = None
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.startswith('physical id'):
                    phys = ln.split(':')[1].strip()
                elif ln.startswith('core id'):
                    core = ln.split(':')[1].strip()
                elif not ln.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    n_phys = len(cores) or (os.cpu_count() or 1)
    n_aff = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            quota = max(1, int(float(q) / float(p)))
    except (OSError, ValueError):
        pass
    n = min(n_phys, n_aff, quota or n_phys)
    return max(1, n), {'physical_cores': n_phys, 'affinity': n_aff, 'cgroup_quota': quota, 'logical': os.cpu_count()}


def cpu_baseline(sc, gpu_out):
    """SURVEY.md 8d / BASELINE.md 4: the reference cannot run here (TensorFlow), so "the reference CPU path" is the op-for-op,
    unfused torch-CPU restatement of its graph (oracle/mvnerf_torch.py: separate gather, PE, 13 GEMMs per pass, cumprod, the
    63-step compare-accumulate of sample_pdf), fp32, torch.set_num_threads(physical cores), on ALL rays of the benchmark scene:
    forward (2 warm-ups, median of 5) and forward + backward of the training loss (1 warm-up, median of 3).  `parity` compares
    the GPU result with the NumPy oracle (oracle/mvnerf_oracle.py, the arithmetic contract) on all rays."""
    import numpy as np
    import torch
    from oracle import mvnerf_oracle as O
    from oracle import mvnerf_torch as T
    n_threads, core_info = host_cores()
    torch.set_num_threads(n_threads)
    n_rays = sc['rays_o'].shape[1]
    tt = lambda k: torch.as_tensor(np.asarray(sc[k])).to(torch.float32)
    args = [tt(k) for k in ['rays_o', 'rays_d', 'images', 'intrinsics', 'extrinsics_inv', 'features']]
    uc, uf = tt('u_coarse'), tt('u_fine')

    def fwd(cf, ff):
        return T.render_call(cf, ff, *args, sc['near'], sc['far'], sc['n_samples'], uc, uf, unfused=True)

    def run_fwd():
        with torch.no_grad():
            return fwd(tt('coarse'), tt('fine'))

    y = torch.rand((1, n_rays, 3), generator=torch.Generator().manual_seed(0))

    def run_train():
        cf, ff = tt('coarse').requires_grad_(True), tt('fine').requires_grad_(True)
        out = fwd(cf, ff)
        (((y - out[0]) ** 2).mean() + ((y - out[2]) ** 2).mean()).backward()
        return cf.grad, ff.grad

    def median_time(fn, warm, n):
        for _ in range(warm):
            res = fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            res = fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), res

    t_fwd, twin = median_time(run_fwd, 2, 5)
    t_train, _ = median_time(run_train, 1, 3)
    cpu_model = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            cpu_model = next((ln.split(':', 1)[1].strip() for ln in f if ln.startswith('model name')), 'unknown')
    except OSError:
        pass
    base = {'value': n_rays / t_fwd, 'unit': 'rays/s', 'cores': n_threads, 'kind': 'port', 'cpu_model': cpu_model,
            'forward_s': t_fwd, 'train_rays_per_sec': n_rays / t_train, 'train_s': t_train, 'host': core_info,
            'sample': f'all {n_rays} rays of the same scene, full 64+128 samples: op-for-op torch-CPU restatement of the TF graph '
                      f'(oracle/mvnerf_torch.py, fp32, unfused, 63-step compare-accumulate in sample_pdf), torch.set_num_threads({n_threads}); '
                      f'forward: median of 5 after 2 warm-ups = {t_fwd:.2f} s; forward+backward of the training loss: median of 3 after 1 '
                      f'warm-up = {t_train:.2f} s'}
    # parity: the NumPy oracle on all rays
    cn, fn = O.unflatten_net(sc['coarse']), O.unflatten_net(sc['fine'])
    ref = O.render_call(cn, fn, sc['rays_o'], sc['rays_d'], sc['images'], sc['intrinsics'], sc['extrinsics_inv'], sc['features'],
                        sc['near'], sc['far'], sc['n_samples'], sc['u_coarse'], sc['u_fine'])
    names = ['rgb', 'depth', 'fine_rgb', 'fine_depth']
    parity = {}
    for n, g, rf, tw in zip(names, gpu_out, ref, twin):
        diff = np.abs(g.cpu().numpy() - rf)
        parity[n + '_max_abs'] = float(diff.max())
        parity[n + '_mean_l1'] = float(diff.mean())
        parity[n + '_max_abs_vs_torch_cpu'] = float(np.abs(g.cpu().numpy() - tw.numpy()).max())
    parity['checked_rays'] = n_rays
    parity['tolerance'] = 1e-4
    parity['oracle'] = 'oracle/mvnerf_oracle.py (NumPy fp32); *_vs_torch_cpu: the timed torch-CPU restatement'
    return base, parity


if __name__ == '__main__':
    main()
