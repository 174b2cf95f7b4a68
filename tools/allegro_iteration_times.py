#!/usr/bin/env python3
"""Rollout-kernel time of every annealing iteration of a run that starts from a zero plan (hipEvents of the library's timing hook, one
iteration at a time): where the rollouts' length depends on the iterate, averages over different stretches of the run are different
numbers (bench.py: why the Allegro example's "lean" iteration looked slower than its "full" one in round 5).
usage: allegro_iteration_times.py [example] [iterations]"""
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dial_mpc_amd.core.dial_core import MBDPI, load_dial_and_env  # noqa: E402
from dial_mpc_amd.utils.io_utils import get_example_path  # noqa: E402

example = sys.argv[1] if len(sys.argv) > 1 else "allegro_reorient"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 36
cfgd = yaml.safe_load(open(get_example_path(example + ".yaml")))
dc, _, env = load_dial_and_env(cfgd)
for want_bars in (True, False):
    pl = MBDPI(dc, env, kernel_rng=True)
    st = env.reset(0).packed
    Y = torch.zeros((dc.Hnode + 1, pl.nu), dtype=torch.float32, device=pl.device)
    ms = []
    for i in range(iters):
        pl.ctx.set_timing(True)
        _, Y, _ = pl.reverse_once(st, None, Y, pl.sigma_control, eps=None, want_bars=want_bars)
        torch.cuda.synchronize()
        t, n = pl.ctx.rollout_ms()
        pl.ctx.set_timing(False)
        ms.append(t / max(n, 1))
    f = lambda a: " ".join(f"{x:.2f}" for x in a)  # noqa: E731
    print(f"{example} N={dc.Nsample} H={dc.Hsample} {'full' if want_bars else 'lean'} iteration, rollout kernel ms per iteration (from a zero plan):")
    print("  ", f(ms))
    print(f"   mean of iterations 3..17 {sum(ms[3:18]) / 15:.3f}   3..32 {sum(ms[3:33]) / 30:.3f}   18..32 {sum(ms[18:33]) / 15:.3f}")
