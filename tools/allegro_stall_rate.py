#!/usr/bin/env python3
"""How often does the Allegro's constraint solver STOP SHORT inside a rollout, and does the rate depend on the arithmetic?
(the closed loop loses the ball in ~3 % of the runs of the product build, in 0 of 256 of the strict-IEEE build.)

The elliptic-cone Newton solver (iterations 100, tolerance 1e-8) stops when `scale * (prev_cost - cost) < tolerance`: in fp32 that is
"the line search found nothing" -- at the optimum, or at a point where the Newton direction is useless in fp32 (a stall).  A stalled
step leaves a wrong acceleration in that rollout.  Cold-started (qacc_warmstart = 0) the solve is robust: fp32 oracle, fp64 oracle and
the oracle under 4-ulp jitter agree to 6e-3 rad/s, so `oracle.env_step from the rollout's own (q, qd)` is the reference answer and

    stall  :=  max |qd_next(rollout) - qd_next(cold-started oracle step from the rollout's (q, qd))|  >  --thresh (0.05 rad/s)

Phase 1 (once per seed): the ORACLE's closed loop (oracle plant + oracle planner, as tools/allegro_closed_loop_study.py --mode oracle)
records, per control tick, the state, the mean trajectory and the noise of the last annealing iteration -> build/study/.
Phase 2 (per engine): the recorded batches are rolled out by
    oracle          oracle/dial_oracle.c (gcc, no contraction)
    emu[off]        the KERNEL's code on the host wave emulator, -ffp-contract=off            (~ libdialhip_ieee.so)
    emu[fma]        the same with -march=native -ffp-contract=fast                            (~ the contraction of the product build)
    emu[fast]       the same with -ffast-math -ffp-contract=off
    emu[fastfma]    both                                                                      (~ the product build)
and every transition of every rollout is checked.  CPU only.

    python tools/allegro_stall_rate.py --seed 0 --ticks 40 --nsample 512
"""
import argparse
import os
import sys
import time
from multiprocessing import Pool

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from allegro_closed_loop_study import philox_normal, setup  # noqa: E402
from dial_mpc_amd.core.dial_core import make_cfg  # noqa: E402

VARIANTS = {"emu[off]": ((), ""), "emu[fma]": (("-march=native", "-ffp-contract=fast"), "_fma"),
            "emu[fast]": (("-ffast-math", "-ffp-contract=off"), "_fastmath"), "emu[fastfma]": (("-march=native", "-ffast-math"), "_fastfma")}
_G = {}


def _init(nsample):
    import oracle as O
    dc, ec, env = setup(nsample, 0)
    model, task, cfg = env.make_model(), env.make_task(), make_cfg(dc)
    _G.update(o32=O.Oracle(model, task, cfg, np.float32), nq=model.nq, nv=model.nv)


def _check(job):
    """one rollout: the cold-started oracle step from every state of it; returns max |dqd| per transition"""
    s0, us, qss, qdss = job
    o32, nq, nv = _G["o32"], _G["nq"], _G["nv"]
    T = us.shape[0]
    out = np.zeros(T - 1, np.float32)
    for t in range(T - 1):
        st = np.array(s0, dtype=np.float32)
        st[:nq], st[nq:nq + nv], st[nq + nv:nq + 2 * nv], st[nq + 2 * nv] = qss[t], qdss[t], 0.0, t + 1
        out[t] = np.abs(o32.env_step(st, us[t + 1])[0][nq:nq + nv] - qdss[t + 1]).max()
    return out


def record(nsample, seed, ticks, path):
    import oracle as O
    dc, ec, env = setup(nsample, seed)
    model, task, cfg = env.make_model(), env.make_task(), make_cfg(dc)
    o32 = O.Oracle(model, task, cfg, np.float32)
    state, _, _ = o32.env_reset(env._init_q, np.zeros(model.nv))
    nu, Hn1 = model.nu, dc.Hnode + 1
    sigma = (dc.horizon_diffuse_factor ** np.arange(Hn1)[::-1] * dc.sigma_scale).astype(np.float32)
    Y = np.zeros((Hn1, nu), np.float32)
    counter, rec = 0, dict(state=[], Y=[], ns=[], eps=[])
    for t in range(ticks):
        state = o32.env_step(state, Y[0])[0]
        Y = np.asarray(o32.shift(Y), np.float32)
        n_it = dc.Ndiffuse_init if t == 0 else dc.Ndiffuse
        for i in range(n_it):
            eps = philox_normal(int(dc.seed), counter, dc.Nsample, Hn1 * nu).reshape(dc.Nsample, Hn1, nu)
            counter += 1
            ns = (sigma * np.float32(dc.traj_diffuse_factor ** i)).astype(np.float32)
            if i == n_it - 1:
                for k, v in zip(("state", "Y", "ns", "eps"), (state, Y, ns, eps)):
                    rec[k].append(np.array(v, np.float32))
            Y = np.asarray(o32.reverse_once(state, Y, ns, eps)["Ybar"], np.float32)
    np.savez_compressed(path, **{k: np.stack(v) for k, v in rec.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--nsample", type=int, default=512)
    ap.add_argument("--ticks", type=int, default=40)
    ap.add_argument("--thresh", type=float, default=0.05)
    ap.add_argument("--engines", default="oracle,emu[off],emu[fma],emu[fast],emu[fastfma]")
    args = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "build", "study"), exist_ok=True)
    path = os.path.join(ROOT, "build", "study", f"allegro_ticks_N{args.nsample}_seed{args.seed}_{args.ticks}.npz")
    if not os.path.exists(path):
        t0 = time.time()
        record(args.nsample, args.seed, args.ticks, path)
        print(f"recorded the oracle's closed loop (seed {args.seed}, {args.ticks} ticks) in {time.time() - t0:.0f} s -> {path}", flush=True)
    rec = np.load(path)
    import oracle as O
    from emu_lib import Emu
    dc, ec, env = setup(args.nsample, args.seed)
    model, task, cfg = env.make_model(), env.make_task(), make_cfg(dc)
    o32 = O.Oracle(model, task, cfg, np.float32)
    W = np.array([[cfg.W[t][k] for k in range(dc.Hnode + 1)] for t in range(dc.Hsample + 1)], np.float32)
    pool = Pool(os.cpu_count(), initializer=_init, initargs=(args.nsample,))
    for name in args.engines.split(","):
        emu = None if name == "oracle" else Emu(model, task, cfg, defines=list(VARIANTS[name][0]), tag=VARIANTS[name][1])
        t0, n_tr, stalls, worst, per_tick = time.time(), 0, 0, 0.0, []
        hist = np.zeros(5, int)   # > 0.01, 0.05, 0.2, 1, 5 rad/s
        for k in range(rec["state"].shape[0]):
            s0, Y, ns, eps = rec["state"][k], rec["Y"][k], rec["ns"][k], rec["eps"][k]
            Y0s = np.concatenate([Y[None] + eps * ns[None, :, None], Y[None]]).astype(np.float32)
            us = np.einsum("tk,nka->nta", W, Y0s).astype(np.float32)
            if emu is None:
                _, qss, qdss, _ = o32.rollout(s0, us)
            else:
                _, qss, qdss, _, _ = emu.rollout(s0, us)
            errs = np.stack(pool.map(_check, [(s0, us[n], qss[n], qdss[n]) for n in range(us.shape[0])], chunksize=8))
            errs = np.where(np.isfinite(errs), errs, 1e9)
            n_tr += errs.size
            stalls += int((errs > args.thresh).sum())
            per_tick.append(int((errs > args.thresh).sum()))
            worst = max(worst, float(errs.max()))
            hist += np.array([(errs > x).sum() for x in (0.01, 0.05, 0.2, 1.0, 5.0)])
        print(f"{name:14s} seed {args.seed}: {n_tr} transitions, {stalls} off the cold-started oracle step by > {args.thresh} rad/s "
              f"({1e4 * stalls / n_tr:.2f} per 10^4), worst {worst:.3g}; counts > [0.01, 0.05, 0.2, 1, 5] rad/s = {hist.tolist()}; "
              f"per tick {per_tick}  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
