#!/usr/bin/env python3
"""Why does the Allegro closed loop sometimes drop the ball?  

The synchronous driver loop of dial_core.py:245-266 (env.step, shift, Ndiffuse annealing iterations) on `allegro_reorient`:

  --mode gpu     the product: HIP plant + HIP planner (MBDPI with the in-kernel Philox noise), many seeds, any N      [needs a GPU]
  --mode oracle  the CPU checker as plant AND as the planner's rollout engine (fp32 oracle, oracle/dial_oracle.c), fed with the
                 SAME noise: Philox4x32-10 + Box-Muller restated in NumPy below (csrc/philox.h; the device's approximate log /
                 sin / cos differ from NumPy's in the last bits: the draws agree to ~1e-6), same seed / call counter sequence.

Both print one line per seed (ball height at ticks 10 / 20 / ... and whether it left the hand) and a summary.  If the oracle-driven
loop drops the ball at the rate the HIP-driven loop does, the drops are the TASK at this sample count (a chaotic toss planned with N
samples), not the kernel.  Run the oracle mode anywhere (CPU only, OpenMP); it costs about 2 s per annealing iteration at N = 512.

    python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 0:64 --ticks 40
    python tools/allegro_closed_loop_study.py --mode oracle --nsample 512 --seeds 0:32 --ticks 40
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from dial_mpc_amd.core.dial_core import load_dial_and_env, make_cfg  # noqa: E402
from dial_mpc_amd.utils.io_utils import get_example_path  # noqa: E402


def philox_normal(seed, counter, n_count, C):
    """csrc/philox.h + rng_fill_kernel in NumPy: eps[n, c], n < n_count, c < C (= (Hnode + 1) * nu), counter = annealing-call index."""
    nq = (C + 3) // 4
    n = np.repeat(np.arange(n_count, dtype=np.uint32), nq)
    q = np.tile(np.arange(nq, dtype=np.uint32), n_count)
    c0, c1, c2, c3 = n.copy(), q.copy(), np.full_like(n, counter), np.zeros_like(n)
    k0, k1 = np.uint32(seed & 0xffffffff), np.uint32((seed >> 32) & 0xffffffff)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0.astype(np.uint64)
        p1 = np.uint64(0xCD9E8D57) * c2.astype(np.uint64)
        n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
        n1 = p1.astype(np.uint32)
        n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
        n3 = p0.astype(np.uint32)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = np.uint32((int(k0) + 0x9E3779B9) & 0xffffffff)
        k1 = np.uint32((int(k1) + 0xBB67AE85) & 0xffffffff)
    u = [c0, c1, c2, c3]
    z = np.empty((n.size, 4), np.float32)
    for h in range(2):
        u1 = ((u[2 * h] >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        u2 = ((u[2 * h + 1] >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
        z[:, 2 * h] = r * np.cos(np.float32(6.283185307179586) * u2)
        z[:, 2 * h + 1] = r * np.sin(np.float32(6.283185307179586) * u2)
    return z.reshape(n_count, nq * 4)[:, :C]


def setup(nsample, seed):
    d = yaml.safe_load(open(get_example_path("allegro_reorient.yaml")))
    d["Nsample"], d["seed"] = nsample, seed
    return load_dial_and_env(d)


PLANT_JITTER = 0     # --plant-jitter K: the plant's (q, qd) moved by up to K ulp (random sign per element) after every env.step
JITTER_TICKS = 10 ** 9   # --jitter-ticks K: only on the first K ticks (1: a one-off perturbation, the loop is self-consistent afterwards)
BITCHECK = False     # --bitcheck: per run, on how many ticks the plant's step equals the planner's own first predicted step BIT FOR BIT
DRIFT = {}            # gpu mode: per seed, the ball's distance from its tick-1 position in the palm's plane, per tick
PLANT_LIB = None     # --plant-lib: another build of the library for the PLANT's env.step only (hybrid runs: which side matters?)


def loop_gpu(nsample, seed, ticks):
    import torch
    from dial_mpc_amd import _lib
    from dial_mpc_amd.core.dial_core import MBDPI
    dc, ec, env = setup(nsample, seed)
    mbdpi = MBDPI(dc, env, kernel_rng=True)
    if PLANT_LIB:
        env._ctx = _lib.Context(env.make_model(), env.make_task(), None, mbdpi.ctx.device, lib_path=PLANT_LIB)
    state = env.reset(0)
    Y = torch.zeros((dc.Hnode + 1, mbdpi.nu), device=mbdpi.device)
    zs = []
    nqv = env.sys.nq + env.sys.nv
    gen = torch.Generator(device="cpu").manual_seed(4242 + seed)
    pred, same, worst = None, 0, 0.0
    for t in range(ticks):
        state = env.step(state, Y[0])
        if BITCHECK and pred is not None:
            got = state.packed[:nqv].cpu().numpy()
            same += int(np.array_equal(got, pred))
            worst = max(worst, float(np.abs(got - pred).max()))
        if PLANT_JITTER and t < JITTER_TICKS:
            x = state.packed[:nqv]
            ulp = torch.nextafter(x.abs(), torch.full_like(x, float("inf"))) - x.abs()
            k = torch.randint(-PLANT_JITTER, PLANT_JITTER + 1, (nqv,), generator=gen).to(x.device, x.dtype)
            state.packed[:nqv] = x + k * ulp
        Y = mbdpi.shift(Y)
        n_it = dc.Ndiffuse_init if t == 0 else dc.Ndiffuse
        for i in range(n_it):
            _, Y, _ = mbdpi.reverse_once(state, None, Y, mbdpi.sigma_control * dc.traj_diffuse_factor ** i, want_bars=(i == n_it - 1))
        if BITCHECK:   # the mean trajectory's rollout (row Nsample) of the last iteration started with Ybar[0], which the update leaves in place:
            sc = mbdpi.ctx.debug_scratch()    # its first step is the planner's prediction of the plant's next step
            pred = np.concatenate([sc["qss"][-1, 0], sc["qdss"][-1, 0]])
            pred_u0 = float(np.abs(Y[0].cpu().numpy() - sc["Y0s"][-1, 0]).max())
            worst = max(worst, 0.0 if pred_u0 == 0.0 else worst)
        zs.append(float(state.pipeline_state.q[2]))
        xy = state.pipeline_state.q[:2].cpu().numpy()
        xy0 = xy if t == 0 else xy0
        DRIFT.setdefault(seed, []).append(float(np.hypot(*(xy - xy0))))
    mbdpi.ctx.status()
    if BITCHECK:
        print(f"   bitcheck seed {seed}: plant step == the planner's predicted first step bit for bit on {same} of {ticks - 1} ticks; largest |difference| {worst:.3g}", flush=True)
    return zs


def loop_oracle(nsample, seed, ticks):
    import oracle as O
    dc, ec, env = setup(nsample, seed)
    model, task, cfg = env.make_model(), env.make_task(), make_cfg(dc)
    o32 = O.Oracle(model, task, cfg, np.float32)
    state, _, _ = o32.env_reset(env._init_q, np.zeros(model.nv))
    nu, Hn1 = model.nu, dc.Hnode + 1
    sigma = (dc.horizon_diffuse_factor ** np.arange(Hn1)[::-1] * dc.sigma_scale).astype(np.float32)
    Y = np.zeros((Hn1, nu), np.float32)
    counter, zs = 0, []
    rng = np.random.default_rng(4242 + seed)
    nqv = model.nq + model.nv
    for t in range(ticks):
        state = o32.env_step(state, Y[0])[0]
        if PLANT_JITTER and t < JITTER_TICKS:     # as in loop_gpu: the plant's (q, qd) moved by up to K ulp
            state = np.array(state, np.float32)
            state[:nqv] += (rng.integers(-PLANT_JITTER, PLANT_JITTER + 1, nqv) * np.spacing(np.abs(state[:nqv]))).astype(np.float32)
        Y = np.asarray(o32.shift(Y), np.float32)
        n_it = dc.Ndiffuse_init if t == 0 else dc.Ndiffuse
        for i in range(n_it):
            eps = philox_normal(int(dc.seed), counter, dc.Nsample, Hn1 * nu).reshape(dc.Nsample, Hn1, nu)
            counter += 1
            Y = np.asarray(o32.reverse_once(state, Y, (sigma * np.float32(dc.traj_diffuse_factor ** i)).astype(np.float32), eps)["Ybar"], np.float32)
        zs.append(float(state[2]))
    return zs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=("gpu", "oracle", "philox-check"), required=True)
    ap.add_argument("--nsample", type=int, default=512)
    ap.add_argument("--seeds", default="0:8", help="a:b or a comma-separated list")
    ap.add_argument("--ticks", type=int, default=40)
    ap.add_argument("--json", default=None)
    ap.add_argument("--plant-jitter", type=int, default=0, help="move the plant's (q, qd) by up to this many ulp after every env.step")
    ap.add_argument("--jitter-ticks", type=int, default=10 ** 9, help="apply --plant-jitter only on the first K ticks")
    ap.add_argument("--bitcheck", action="store_true", help="gpu mode: count the ticks on which the plant's step equals the planner's predicted first step bitwise")
    ap.add_argument("--plant-lib", default=None, help="gpu mode: the plant's env.step from this build of the library (the planner's: DIAL_HIP_LIB)")
    args = ap.parse_args()
    global PLANT_LIB, PLANT_JITTER, BITCHECK, JITTER_TICKS
    PLANT_LIB, PLANT_JITTER, BITCHECK, JITTER_TICKS = args.plant_lib, args.plant_jitter, args.bitcheck, args.jitter_ticks
    seeds = list(range(*map(int, args.seeds.split(":")))) if ":" in args.seeds else [int(s) for s in args.seeds.split(",")]
    if args.mode == "philox-check":   # the NumPy restatement against dial_rng_fill (needs a GPU)
        from dial_mpc_amd import _lib
        dc, ec, env = setup(args.nsample, 3)
        ctx = _lib.Context(env.make_model(), env.make_task(), make_cfg(dc))
        for counter in (0, 7):
            dev = ctx.rng_fill(12345678901, counter, 0, args.nsample).cpu().numpy()
            ref = philox_normal(12345678901, counter, args.nsample, dev.shape[1] * dev.shape[2]).reshape(dev.shape)
            print(f"philox restatement vs dial_rng_fill (counter {counter}): max |diff| = {np.abs(dev - ref).max():.3g}, draws {dev.size}")
        return
    loop = loop_gpu if args.mode == "gpu" else loop_oracle
    marks = [k for k in (9, 19, 29, 39, 59, 79, 99) if k < args.ticks]
    res = []
    for seed in seeds:
        t0 = time.time()
        zs = loop(args.nsample, seed, args.ticks)
        dropped = min(zs) < 0.08 or not np.isfinite(zs).all()
        res.append(dict(seed=seed, dropped=bool(dropped), z=[round(zs[k], 4) for k in marks], z_min=round(float(np.nanmin(zs)), 4)))
        if seed in DRIFT:   # sideways drift while the ball is still in the hand (z >= 0.08), in cm
            dr = [d for d, zz in zip(DRIFT[seed], zs) if zz >= 0.08]
            res[-1]["drift_cm"] = [round(100 * d, 2) for d in DRIFT[seed]]
            res[-1]["drift_max_in_hand_cm"] = round(100 * max(dr), 2)
        print(f"{args.mode} N={args.nsample} seed {seed}: ball z at ticks {[k + 1 for k in marks]} = {[round(zs[k], 3) for k in marks]}"
              f"{'  DROPPED' if dropped else ''}  ({time.time() - t0:.0f} s)", flush=True)
    nd = sum(r["dropped"] for r in res)
    print(f"== {args.mode} planner, N={args.nsample}, {args.ticks} ticks: {nd} of {len(res)} runs lose the ball "
          f"(seeds {[r['seed'] for r in res if r['dropped']]})", flush=True)
    if args.json:
        json.dump(dict(mode=args.mode, nsample=args.nsample, ticks=args.ticks, runs=res), open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
