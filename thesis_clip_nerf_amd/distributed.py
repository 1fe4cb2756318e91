"""Multi-GPU plumbing for the render path: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on ROCm, "gloo" on CPU for the tests).

The path shards by rays: no operator in `_call` mixes rays, so the forward pass needs
no collective at all - every rank renders its own contiguous block of rays (or its own scenes) with
replicated weights and source views.  Collectives appear only at the two ends:
  * assembling a full image from per-rank row blocks: one all_gather of (rows, 3) + (rows,) fp32;
  * training: ONE all-reduce over a single flat fp32 buffer holding both MLPs' gradients
    (2 x 247 300 floats = 1.98 MB) - latency-bound on xGMI, hence one message, not per-tensor calls.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*.
    Returns (rank, world, local_rank).  World size 1 needs no process group."""
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def shard_bounds(n, rank, world):
    """Contiguous balanced split of n units: the first n % world ranks get one extra."""
    if not 0 <= rank < world:
        raise ValueError(f'rank {rank} outside world {world}')
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local, n_total, group=None):
    """Concatenate per-rank row blocks (shard_bounds order) of possibly unequal length along dim 0."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    longest = -(-n_total // world)
    padded = local.new_zeros((longest,) + tuple(local.shape[1:]))
    padded[:local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    out = []
    for r, p in enumerate(parts):
        lo, hi = shard_bounds(n_total, r, world)
        out.append(p[:hi - lo])
    return torch.cat(out, dim=0)


def allreduce_mean_(flat, group=None):
    """In-place mean over ranks of one flat gradient buffer (a single collective per step)."""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= dist.get_world_size(group)
    return flat


class OverlappedGradSync:
    """The flat gradient all-reduce (mean) of a training step, split at the coarse | fine boundary so that the fine half travels while
    the coarse net's backward is still running: `mvnerf_loss_and_grads` records `event` on the compute stream as soon as the fine
    half of the flat buffer is final (mvnerf_train_call.fine_grad_event); that half is reduced on a second stream behind the event,
    the coarse half on the compute stream after the whole backward, and the compute stream then waits for the second one.  Two
    messages of 0.99 MB instead of one of 1.98 MB: both are latency-bound on xGMI, the first is hidden.
    Use: `model.compile(grad_sync=OverlappedGradSync(device))`; without a process group it does nothing."""

    def __init__(self, device, group=None):
        self.device = torch.device(device)
        self.group = group
        self.side = torch.cuda.Stream(self.device)
        self.event = torch.cuda.Event()
        with torch.cuda.device(self.device):
            self.event.record()                          # creates the underlying hipEvent_t (lazily created by torch)

    def event_handle(self):
        return self.event.cuda_event

    def __call__(self, flat):
        if not (dist.is_available() and dist.is_initialized()):
            return flat
        world = dist.get_world_size(self.group)
        half = flat.numel() // 2
        coarse, fine = flat[:half], flat[half:]
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.event)             # the fine half is final here; the coarse backward is still in flight on `main`
            dist.all_reduce(fine, op=dist.ReduceOp.SUM, group=self.group)
            fine /= world
        dist.all_reduce(coarse, op=dist.ReduceOp.SUM, group=self.group)
        coarse /= world
        main.wait_stream(self.side)
        return flat


def max_over_ranks(value, device='cpu', group=None):
    """Max of a Python float over ranks (the benchmark's step time is the slowest rank's)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def render_rays_sharded(render_fn, rays_o, rays_d, group=None):
    """Split the ray axis (dim 0 of (n,3) tensors) over ranks, run `render_fn(o_block, d_block)` ->
    tuple of per-ray tensors on this rank's block, and all_gather every output back to length n."""
    n = rays_o.shape[0]
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(n, rank, world)
    outs = render_fn(rays_o[lo:hi], rays_d[lo:hi])
    return tuple(all_gather_rows(o.contiguous(), n, group) for o in outs)
