#!/usr/bin/env python3
"""Measure (not gate) the GPU-vs-oracle error distributions at the BASELINE sizes, so that the gates in
tests/conftest.py can be set ~10x above what is measured, and dump the per-entry data of the seq-jump stress
runs for the root-cause analysis of its outliers (tools/seq_jump_flips.py reads the dump).

    python tools/parity_survey.py [--dump gpurun_out/parity]      (needs an MI355X)
"""
import argparse
import json
import os
import sys
import time

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


def stats(got, ref):
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64)).ravel()
    r = d / (1.0 + np.abs(ref.astype(np.float64)).ravel())
    return dict(max_abs=float(d.max()), p999_abs=float(np.quantile(d, 0.999)), p99_abs=float(np.quantile(d, 0.99)),
                max_rel=float(r.max()), p999_rel=float(np.quantile(r, 0.999)),
                frac_gt_2e4=float((r > 2e-4).mean()), frac_gt_1e3=float((r > 1e-3).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dump", default=os.path.join(ROOT, "gpurun_out", "parity"))
    args = ap.parse_args()
    os.makedirs(args.dump, exist_ok=True)
    rows = []
    cases = [("unitree_go2_trot", 2048, 16, [0, 1, 2]), ("unitree_go2_seq_jump", 1024, 16, [0, 1, 2]),
             ("unitree_h1_jog", 2048, 16, [0, 1]), ("unitree_h1_loco", 1024, 20, [0, 1]),
             ("unitree_go2_trot", 64, 8, [0]), ("unitree_go2_seq_jump", 48, 16, [0]), ("unitree_h1_jog", 32, 16, [0]),
             ("unitree_h1_loco", 32, 20, [0])]
    if "allegro_reorient" in sys.argv:
        cases.append(("allegro_reorient", 4096, 24, [0]))
    for ex, N, H, seeds in cases:
        dc, env, model, task, cfg = setup_case(ex, N, H, per_rollout=True)
        ctx = _lib.Context(model, task, cfg)
        o32, o64 = O.Oracle(model, task, cfg, np.float32), O.Oracle(model, task, cfg, np.float64)
        for seed in seeds:
            q, qd = (env._init_q, np.zeros(model.nv)) if seed == 0 else perturbed_state(env, seed)
            s0, _, _ = o32.env_reset(q, qd)
            eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=seed, Ybar_scale=0.2)
            t0 = time.time()
            r32 = o32.reverse_once(s0, Ybar, sigma, eps, full=True)
            t32 = time.time() - t0
            r64 = o64.reverse_once(s0.astype(np.float64), Ybar, sigma, eps, full=True)
            out = ctx.reverse_once(dev(s0), dev(Ybar), dev(sigma), dev(eps))
            sc = ctx.debug_scratch()
            us = r32["us"]
            ro = o32.rollout(s0, us)            # full q / qd / x of the oracle at this size
            row = dict(example=ex, N=N, H=H, seed=seed, oracle_f32_s=t32)
            row["rewss"] = stats(sc["rewss"], r32["rewss"])
            row["rewss_o32_o64"] = stats(r32["rewss"], r64["rewss"])
            row["q"] = stats(sc["qss"], ro[1])
            row["qd"] = stats(sc["qdss"], ro[2])
            row["x"] = stats(sc["xss"], ro[3])
            row["weights"] = stats(sc["weights"], r32["weights"])
            for k in ("Ybar", "qbar", "qdbar", "xbar", "rews"):
                row[k] = stats(out[k].cpu().numpy(), r32[k])
            rows.append(row)
            print(json.dumps(row), flush=True)
            if ex == "unitree_go2_seq_jump" and N == 1024:
                np.savez_compressed(os.path.join(args.dump, f"seq_jump_N{N}_H{H}_seed{seed}.npz"), state=s0, eps=eps,
                                    sigma=sigma, Ybar=Ybar, g_rewss=sc["rewss"], g_qss=sc["qss"], g_qdss=sc["qdss"],
                                    g_xss=sc["xss"], o_rewss=r32["rewss"], o64_rewss=r64["rewss"], o_qss=ro[1],
                                    o_qdss=ro[2])
    json.dump(rows, open(os.path.join(args.dump, "parity_survey.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
