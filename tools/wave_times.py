#!/usr/bin/env python3
"""Distribution of per-rollout (= per-wavefront) durations inside one rollout launch (needs the -DDIAL_PROFILE build):
how long does the slowest rollout take compared with the median, and how many rounds does the launch need?
    DIAL_HIP_LIB=.../libdialhip_prof.so python tools/wave_times.py <example> <N> <H>"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import seeded_inputs, setup_case  # noqa: E402
from dial_mpc_amd import _lib  # noqa: E402

example, N, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dc, env, model, task, cfg = setup_case(example, N, H)
ctx = _lib.Context(model, task, cfg)
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device="cuda")  # noqa: E731
s0, _, _ = ctx.env_reset(dev(env._init_q), dev(np.zeros(model.nv)))
eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=0, Ybar_scale=0.2)
for _ in range(2):
    ctx.reverse_once(s0, dev(Ybar), dev(sigma), dev(eps))
torch.cuda.synchronize()
B = N + 1
buf = (ctypes.c_ulonglong * (6 * B))()
ctx.lib.dial_debug_wave_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
assert ctx.lib.dial_debug_wave_times(ctx.h, buf, B) == 0
raw = np.array(list(buf), dtype=np.float64).reshape(B, 6)
t = raw[:, :2] / 100.0     # microseconds (100 MHz clock)
t0 = t[:, 0].min()
dur, start, end = t[:, 1] - t[:, 0], t[:, 0] - t0, t[:, 1] - t0
print(f"{example} N={N} H={H}: launch span {end.max():.0f} us; rollout duration min / median / p90 / p99 / max = "
      f"{dur.min():.0f} / {np.median(dur):.0f} / {np.quantile(dur, 0.9):.0f} / {np.quantile(dur, 0.99):.0f} / {dur.max():.0f} us")
print(f"  start time: {np.sum(start < 50)} rollouts start in the first 50 us, last start at {start.max():.0f} us "
      f"({np.sum(start > 0.25 * end.max())} rollouts start after 25 % of the span: later rounds)")
print(f"  mean-trajectory rollout (index {N}): start {start[N]:.0f} us, duration {dur[N]:.0f} us")
n_on, calls, ls_it, nw_it = raw[:, 2], raw[:, 3], raw[:, 4], raw[:, 5]
if nw_it.max() > 0:
    print(f"  Newton iterations per rollout min / median / max = {nw_it.min():.0f} / {np.median(nw_it):.0f} / {nw_it.max():.0f}; "
          f"line-search iterations {ls_it.min():.0f} / {np.median(ls_it):.0f} / {ls_it.max():.0f}; "
          f"corr(duration, Newton iters) = {np.corrcoef(dur, nw_it)[0, 1]:.3f}, corr(duration, LS iters) = {np.corrcoef(dur, ls_it)[0, 1]:.3f}")
    A = np.stack([np.ones(B), nw_it, ls_it, n_on], 1)
    coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
    print(f"  least squares: duration ~ {coef[0]:.0f} us + {coef[1]:.2f} us/Newton iter + {coef[2]:.2f} us/LS iter + {coef[3]:.3f} us/on-unit-call; "
          f"residual rms {np.sqrt(np.mean((A @ coef - dur) ** 2)):.0f} us")
if len(sys.argv) > 4:
    wpb = int(sys.argv[4])
    slot = np.arange(B) % wpb
    print("  mean duration by wavefront slot in the workgroup:", " ".join(f"{dur[slot == k].mean():.0f}" for k in range(wpb)))
if os.environ.get("WAVE_TIMES_OUT"):
    np.savez(os.environ["WAVE_TIMES_OUT"], raw=raw)
