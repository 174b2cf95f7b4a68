#!/usr/bin/env python3
"""GPU vs the fp32 oracle over perturbed start states for every built env (N = 192, each example's own horizon):
a wider net than the pytest parity cases, same checker.  The fp64 oracle runs next to it, because a truncated
Newton solver makes discrete choices (warm start, line-search bracket): wherever two fp32 evaluations of the same
rollout take different branches the fp32 and fp64 oracles disagree just as much, which separates rounding-induced
decision flips from kernel bugs.

    python tools/stress_parity.py          (needs an MI355X)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle as O  # noqa: E402
from conftest import perturbed_state, seeded_inputs, setup_case  # noqa: E402
from dial_mpc_amd import _lib  # noqa: E402


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device="cuda")


def rel(a, b):
    return np.abs(a - b) / (1.0 + np.abs(b))


for ex, H in (("unitree_go2_trot", 16), ("unitree_go2_seq_jump", 20), ("unitree_h1_jog", 25), ("unitree_h1_loco", 20)):
    dc, env, model, task, cfg = setup_case(ex, 192, H)
    ctx = _lib.Context(model, task, cfg)
    o32, o64 = O.Oracle(model, task, cfg, np.float32), O.Oracle(model, task, cfg, np.float64)
    worst = dict(rew_gpu_o32=0.0, rew_o32_o64=0.0, frac_gpu_o32=0.0, frac_o32_o64=0.0, Ybar=0.0, qbar=0.0, xbar=0.0)
    for seed in range(6):
        if seed == 0:
            s0, _, _ = o32.env_reset(env._init_q, np.zeros(model.nv))
        else:
            s0, _, _ = o32.env_reset(*perturbed_state(env, seed))
        eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=seed, Ybar_scale=0.3)
        r32 = o32.reverse_once(s0, Ybar, sigma, eps, full=True)
        r64 = o64.reverse_once(s0.astype(np.float64), Ybar, sigma, eps, full=True)
        out = ctx.reverse_once(dev(s0), dev(Ybar), dev(sigma), dev(eps))
        g = ctx.debug_scratch()["rewss"]
        eg, eo = rel(g, r32["rewss"]), rel(r32["rewss"], r64["rewss"])
        worst["rew_gpu_o32"] = max(worst["rew_gpu_o32"], float(eg.max()))
        worst["rew_o32_o64"] = max(worst["rew_o32_o64"], float(eo.max()))
        worst["frac_gpu_o32"] = max(worst["frac_gpu_o32"], float((eg > 1e-3).mean()))
        worst["frac_o32_o64"] = max(worst["frac_o32_o64"], float((eo > 1e-3).mean()))
        for k in ("Ybar", "qbar", "xbar"):
            worst[k] = max(worst[k], float(np.abs(out[k].cpu().numpy() - r32[k]).max()))
    print(f"{ex:22s} per-step reward, max rel. dev: GPU vs fp32 oracle {worst['rew_gpu_o32']:.1e} (fraction > 1e-3: "
          f"{worst['frac_gpu_o32']:.4f}), fp32 vs fp64 oracle {worst['rew_o32_o64']:.1e} ({worst['frac_o32_o64']:.4f}); "
          f"max abs dev of Ybar {worst['Ybar']:.1e}, qbar {worst['qbar']:.1e}, xbar {worst['xbar']:.1e}")
