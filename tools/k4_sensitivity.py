#!/usr/bin/env python3
"""How far can the device's fp32 K4 be from the fp64 K4 of its own rollouts?  (tests/conftest.py: distribution_parity, first gate.)
Prints, for one reverse_once: std of the mean rewards, the logit error an fp32 mean reward carries (ulp(rews) / (std temp)), the
effective sample size, the leading weights, and the measured |Ybar_device - Ybar_fp64|.
usage: k4_sensitivity.py [example] [N] [H] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import k4_fp64, seeded_inputs, setup_case, perturbed_state  # noqa: E402
from dial_mpc_amd import _lib  # noqa: E402

example = sys.argv[1] if len(sys.argv) > 1 else "unitree_go2_seq_jump"
N, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 16)
dc, env, model, task, cfg = setup_case(example, N, H)
ctx = _lib.Context(model, task, cfg)
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device="cuda")  # noqa: E731
for seed in (0, 1):
    q, qd = (env._init_q, np.zeros(model.nv)) if seed == 0 else perturbed_state(env, seed)
    s0, _, _ = ctx.env_reset(dev(q), dev(qd))
    eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=seed, Ybar_scale=0.2)
    out = ctx.reverse_once(s0, dev(Ybar), dev(sigma), dev(eps))
    sc = ctx.debug_scratch()
    g = k4_fp64(sc["rewss"], sc["Y0s"], sc["qss"], sc["qdss"], sc["xss"], cfg.temp_sample)
    rews32 = out["rews"].cpu().numpy()
    w = np.sort(g["weights"])[::-1]
    dl = np.spacing(np.float32(np.abs(rews32).max())) / (g["std"] * cfg.temp_sample)
    spread = np.abs(np.asarray(sc["Y0s"], np.float64).reshape(N + 1, -1) - g["Ybar"]).max()
    print(f"{example} N={N} seed={seed}: mean reward {g['mean']:.4f}, std {g['std']:.5f}, temp {cfg.temp_sample}, ulp(rews) {np.spacing(np.float32(np.abs(rews32).max())):.2e} "
          f"-> logit error per ulp {dl:.2e}; ESS {g['ess']:.2f}, leading weights {w[:4].round(4)}; "
          f"|rews32 - fp64 mean| max {np.abs(rews32 - g['rews']).max():.2e}; |Ybar_dev - Ybar_fp64| max {np.abs(out['Ybar'].cpu().numpy().reshape(-1) - g['Ybar']).max():.2e} "
          f"(bound 2 ulp: {2 * dl * spread:.2e})")
