#!/usr/bin/env python3
"""Counted (not estimated) floating-point work of one env.step in the DENSE formulation the reference runs (MJX below
60 dofs: dense efc_J, dense J^T D J, dense Cholesky) -- the oracle's C source compiled as C++ with an operation-counting
arithmetic type (tools/opcount/counted.h).  Replaces SURVEY 8d's estimate of 6e4 FLOP per Go2 env.step.

    python tools/opcount/count_flops.py            -> table + profiles/r02_opcount.json

Counts are averages over the env.steps of a few rollouts from the home keyframe and from perturbed states (the solver's
iteration count depends on the contact state).  FLOP = add + mul + div + sqrt (one each); transcendental calls and
comparisons are listed separately.  The HIP kernel does LESS arithmetic than this (branch-sparse factorisations,
contact-sparse H, implicit Jacobian rows); its own instruction counts come from the PMC passes (profiles/)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle as O  # noqa: E402
from conftest import perturbed_state, seeded_inputs, setup_case  # noqa: E402


def build():
    so = os.path.join(HERE, "libopcount.so")
    srcs = [os.path.join(HERE, "opcount.cpp"), os.path.join(HERE, "counted.h"), os.path.join(ROOT, "oracle", "dial_oracle.c")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-fpermissive", "-w", "-o", so, srcs[0]])
    return ctypes.CDLL(so)


class CountingOracle(O.Oracle):
    def __init__(self, lib, model, task, cfg):
        self.dtype = np.dtype(np.float64)          # CReal wraps one double
        self.lib = lib
        self.model, self.task, self.cfg = model, task, cfg
        self.nq, self.nv, self.nu, self.nbody = model.nq, model.nv, model.nu, model.nbody
        self.nx = (model.nbody - 1) * 3
        from dial_mpc_amd import _abi
        self.state_size = _abi.state_size(model.nq, model.nv)


def main():
    lib = build()
    out = {}
    rows = [("unitree_go2_trot", 16), ("unitree_go2_seq_jump", 16), ("unitree_h1_jog", 16), ("unitree_h1_loco", 20), ("allegro_reorient", 24)]
    print(f"{'env':22s} {'FLOP/env.step':>14s} {'add':>9s} {'mul':>9s} {'div':>7s} {'sqrt':>6s} {'transc.':>8s} {'cmp':>8s}  physics steps per env.step")
    for ex, H in rows:
        dc, env, model, task, cfg = setup_case(ex, 8, H)
        orc = CountingOracle(lib, model, task, cfg)
        tot = np.zeros(6)
        nsteps = 0
        for seed in range(4):
            q, qd = (env._init_q, np.zeros(model.nv)) if seed == 0 else perturbed_state(env, seed)
            s0, _, _ = orc.env_reset(q, qd)
            eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=seed, Ybar_scale=0.2)
            W = np.array([[cfg.W[t][k] for k in range(dc.Hnode + 1)] for t in range(H + 1)])
            us = np.einsum("tk,nka->nta", W, np.clip(eps[:2] * sigma[None, :, None] + Ybar, -1, 1))
            for n in range(2):
                st = s0.copy()
                for t in range(H + 1):
                    lib.opcount_reset()
                    st, _, _, _ = orc.env_step(st, us[n, t])
                    c = (ctypes.c_ulonglong * 6)()
                    lib.opcount_get(c)
                    tot += np.array(list(c), dtype=np.float64)
                    nsteps += 1
        avg = tot / nsteps
        flop = avg[0] + avg[1] + avg[2] + avg[3]
        out[ex] = dict(flop_per_env_step=flop, add=avg[0], mul=avg[1], div=avg[2], sqrt=avg[3], transcendental=avg[4],
                       compare=avg[5], physics_steps_per_env_step=int(task.n_frames), env_steps_counted=nsteps)
        print(f"{ex:22s} {flop:14.0f} {avg[0]:9.0f} {avg[1]:9.0f} {avg[2]:7.0f} {avg[3]:6.0f} {avg[4]:8.0f} {avg[5]:8.0f}  {task.n_frames}")
    path = os.path.join(ROOT, "profiles", "r02_opcount.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
