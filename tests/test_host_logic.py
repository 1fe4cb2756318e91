"""Host-side logic that needs no GPU: the learning-rate schedule (nerf_utils.py:288-300) and the step at which
`train_step` evaluates it (Keras: `optimizer.iterations` before the increment)."""
import numpy as np

from thesis_clip_nerf_amd.nerf_utils import WarmupScheduler


def test_warmup_scheduler_values():
    """WarmupScheduler(1e-4, 10000, 450000) (train_nerf.py:23): float32 closed form at the branch edges."""
    s = WarmupScheduler(1e-4, 10000, 450000)
    f = np.float32
    tgt = f(1e-4)
    want = {0: f(0.0), 1: f(f(1) / f(10000)) * tgt, 5000: f(f(5000) / f(10000)) * tgt, 10000: f(f(10000) / f(10000)) * tgt,
            10001: tgt, 450000: tgt, 450001: f(0.1) * tgt, 10 ** 6: f(0.1) * tgt}
    for step, lr in want.items():
        assert s(step) == float(lr), (step, s(step), float(lr))
    assert s(0) == 0.0 and s(10000) == float(tgt)                       # first update has lr = 0; full rate AT warmup_steps
    assert WarmupScheduler(1e-3, 0)(0) == 0.0 and WarmupScheduler(1e-3, 0)(1) == float(f(1e-3))   # warmup clamps to >= 1


def test_train_step_calls_schedule_with_pre_increment_step(monkeypatch):
    """model.train_step: lr(iterations) with iterations = 0 on the first step, bias correction with iterations + 1."""
    import torch
    from thesis_clip_nerf_amd import model as M
    seen, lr_ts = [], []
    r = M.MVVNeRFRenderer.__new__(M.MVVNeRFRenderer)
    r.device = torch.device('cpu')
    r._opt = dict(lr=lambda step: seen.append(step) or 1e-3, b1=0.9, b2=0.999, eps=1e-7, clip=1.0, step=0)
    r._grad_sync = None
    r._encoder_optimizer = None
    r._adam_m = r._adam_v = r._update_mask = torch.zeros(2 * M.NET_PARAMS)
    r.coarse_net = r.fine_net = torch.zeros(M.NET_PARAMS)
    r._grad = torch.zeros(2 * M.NET_PARAMS)
    r._last_call = (None, None)
    r.loss_and_grads = lambda *a, **k: (torch.zeros(1), r._grad, None)
    r.weights_changed = lambda: None
    # the optimizer step is ONE C call (mvnerf_apply_gradients) on a mvnerf_adam_state: record the bias-corrected rate it is given
    monkeypatch.setattr(M.ops, 'adam_state', lambda m, v, lr_t, *a, **k: lr_ts.append(lr_t) or object())
    monkeypatch.setattr(M.ops, 'apply_gradients', lambda call, adam, stream_of: None)
    for _ in range(3):
        r.train_step((None, None), combined_features=object())
    assert seen == [0, 1, 2]
    want = [1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) for t in (1, 2, 3)]
    assert np.allclose(lr_ts, want, rtol=1e-12)
