#!/usr/bin/env python3
"""Debug aid: norms of the trunk VJP's operands after each replay of the captured LanguageNeRF step."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thesis_clip_nerf_amd import ops
from thesis_clip_nerf_amd import lmvnerf
from tests.test_gpu_query import _language_case

sc, inputs, labels, model = _language_case(70, 2, 2, 3, '6d')
model.compile(learning_rate=0.0, graph=True)
rec = {}
orig_vjp, orig_jvp, orig_stash = ops.query_vjp, ops.query_jvp, ops.query_stash
def vjp(points, dirs, *a):
    out = orig_vjp(points, dirs, *a)
    rec.setdefault('vjp', []).append(dict(points=points, g_acts=a[-1], stash=a[-2], d_points=out[0], d_dirs=out[1]))
    return out
def jvp(*a, **k):
    out = orig_jvp(*a, **k)
    rec.setdefault('jvp', []).append(dict(c_points=a[2], t_acts=out))
    return out
ops.query_vjp, ops.query_jvp = vjp, jvp
for step in range(6):
    rec.clear() if model._graph is None else None
    out = model.train_step((inputs, labels), sc['features'])
    torch.cuda.synchronize()
    print(step, {k: round(float(v), 6) for k, v in out.items()})
    for kind, lst in rec.items():
        for i, d in enumerate(lst):
            print('   ', kind, i, {k: (float(v.float().abs().sum()) if v.dtype != torch.uint8 else int(v.view(torch.int32).sum())) for k, v in d.items()})
