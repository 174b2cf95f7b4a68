#!/usr/bin/env python3
"""A/B timing of library builds / launch options in ONE process (a fresh GPU box pays ~1 min per `import torch`):
     python tools/ab_time.py CASES.txt [reps]
   every line of CASES.txt:  label  lib(.so under dial_mpc_amd/csrc, or '-' = product)  example  N  H(0 = the example's)  steps  [opt=val ...]
   The cases run round-robin `reps` times (drift cancels); per case: wall-clock ms per full reverse_once (in-kernel noise, every
   output) and the rollout kernel's average launch duration (hipEvents on the launch stream)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import setup_case  # noqa: E402
from dial_mpc_amd import _lib  # noqa: E402

cases = []
for line in open(sys.argv[1]):
    f = line.split()
    if not f or f[0].startswith("#"):
        continue
    cases.append(dict(label=f[0], lib=None if f[1] == "-" else os.path.join(ROOT, "dial_mpc_amd", "csrc", f[1]), example=f[2], N=int(f[3]),
                      H=int(f[4]), steps=int(f[5]), opts={k: int(v) for k, v in (o.split("=") for o in f[6:])}))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device="cuda")  # noqa: E731
res = {c["label"]: [] for c in cases}
setups = {}
for rep in range(reps):
    for c in cases:
        key = (c["example"], c["N"], c["H"])
        if key not in setups:
            import yaml
            from dial_mpc_amd.utils.io_utils import get_example_path
            H = c["H"] or int(yaml.safe_load(open(get_example_path(c["example"] + ".yaml")))["Hsample"])
            setups[key] = setup_case(c["example"], c["N"], H)
        dc, env, model, task, cfg = setups[key]
        ctx = _lib.Context(model, task, cfg, lib_path=c["lib"], options=c["opts"])
        s0, _, _ = ctx.env_reset(dev(env._init_q), dev(np.zeros(model.nv)))
        Ybar = torch.zeros((dc.Hnode + 1, model.nu), device="cuda")
        sigma = dev(np.full(dc.Hnode + 1, 0.3))
        out = None
        for i in range(3):
            out = ctx.reverse_once_rng(s0, Ybar, sigma, 7, i, out=out)
        ctx.set_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(c["steps"]):
            out = ctx.reverse_once_rng(s0, Ybar, sigma, 7, 3 + i, out=out)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        k_ms, n = ctx.rollout_ms()
        ctx.status()
        assert torch.isfinite(out["Ybar"]).all()
        res[c["label"]].append(((t1 - t0) * 1e3 / c["steps"], k_ms / max(n, 1)))
        del ctx
for c in cases:
    r = np.array(res[c["label"]])
    print(f"{c['label']:34s} {c['example']:24s} N={c['N']:6d} ms/iter {r[:, 0].mean():8.4f} (runs {np.round(r[:, 0], 4).tolist()})  kernel {r[:, 1].mean():8.4f} ms"
          f"  rollouts/s {c['N'] / r[:, 0].mean() * 1e3:10.0f}")
