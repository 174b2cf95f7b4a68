#!/usr/bin/env python3
"""Where do the two-samples-per-wavefront kernel and the one-sample kernel first differ?  (debugging aid, GPU)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
from conftest import seeded_inputs, setup_case  # noqa: E402
from dial_mpc_amd import _lib  # noqa: E402

example, N, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
per_rollout = len(sys.argv) > 4 and sys.argv[4] == "swap"
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device="cuda:0")  # noqa: E731
dc, env, model, task, cfg = setup_case(example, N, H, per_rollout=per_rollout)
eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=3, Ybar_scale=0.2)
res = []
for opts in (dict(pair_mode=1), dict()):
    ctx = _lib.Context(model, task, cfg, options=opts)
    s0, _, _ = ctx.env_reset(dev(env._init_q), dev(np.zeros(model.nv)))
    ctx.reverse_once(s0, dev(Ybar), dev(sigma), dev(eps))
    torch.cuda.synchronize()
    ctx.status()
    res.append({k: np.array(v) for k, v in ctx.debug_scratch().items()})
a, b = res
for k in ("Y0s", "rewss", "qss", "qdss", "xss"):
    d = a[k].view(np.uint32) != b[k].view(np.uint32)
    print(k, "differing entries", int(d.sum()), "of", d.size, " max abs", float(np.abs(a[k] - b[k]).max()))
d = (a["qdss"].view(np.uint32) != b["qdss"].view(np.uint32)).any(axis=2) | (a["qss"].view(np.uint32) != b["qss"].view(np.uint32)).any(axis=2)
first = np.where(d.any(axis=1), d.argmax(axis=1), -1)
print("rollouts that differ:", int((first >= 0).sum()), "of", len(first))
print("first differing step histogram:", np.bincount(first[first >= 0], minlength=H + 1).tolist())
bad = np.where(first >= 0)[0][:6]
for n in bad:
    t = first[n]
    dq = np.abs(a["qdss"][n, t] - b["qdss"][n, t])
    print(f"rollout {n} step {t}: qd diff max {dq.max():.3g} at dof {int(dq.argmax())}; dofs differing {np.nonzero(dq)[0].tolist()}; rew {a['rewss'][n, t]} vs {b['rewss'][n, t]}")
print("odd vs even rollouts differing:", int((first[1::2] >= 0).sum()), int((first[0::2] >= 0).sum()))
