#!/usr/bin/env python3
"""Record what the planner saw in a closed-loop run that loses the ball: per control tick the plant state, and for
the LAST annealing iteration the node sequences Y0s and the per-step rewards of every rollout -> one .npz per seed.  [needs a GPU]
Analysed offline (the oracle rolls the same controls out from the same state): tools/allegro_drop_autopsy.py --analyse FILE

    python tools/allegro_drop_autopsy.py --seeds 0,11,31,36 --out gpurun_out/r05t/product
"""
import argparse
import os
import sys

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def record(seed, nsample, ticks, out):
    import torch
    from dial_mpc_amd.core.dial_core import MBDPI, load_dial_and_env
    from dial_mpc_amd.utils.io_utils import get_example_path
    d = yaml.safe_load(open(get_example_path("allegro_reorient.yaml")))
    d["Nsample"], d["seed"] = nsample, seed
    dc, ec, env = load_dial_and_env(d)
    mbdpi = MBDPI(dc, env, kernel_rng=True)
    state = env.reset(0)
    Y = torch.zeros((dc.Hnode + 1, mbdpi.nu), device=mbdpi.device)
    rec = dict(state=[], Y_in=[], Y_out=[], Y0s=[], rewss=[], sigma=[], weights=[])
    for t in range(ticks):
        state = env.step(state, Y[0])
        Y = mbdpi.shift(Y)
        n_it = dc.Ndiffuse_init if t == 0 else dc.Ndiffuse
        for i in range(n_it):
            if i == n_it - 1:
                rec["Y_in"].append(Y.cpu().numpy().copy())
                rec["sigma"].append((mbdpi.sigma_control * dc.traj_diffuse_factor ** i).cpu().numpy().copy())
            _, Y, _ = mbdpi.reverse_once(state, None, Y, mbdpi.sigma_control * dc.traj_diffuse_factor ** i, want_bars=True)
        mbdpi.ctx.status()
        sc = mbdpi.ctx.debug_scratch()
        rec["state"].append(state.packed.cpu().numpy().copy())
        rec["Y_out"].append(Y.cpu().numpy().copy())
        rec["Y0s"].append(np.array(sc["Y0s"], np.float32))
        rec["rewss"].append(np.array(sc["rewss"], np.float32))
        rec["weights"].append(np.array(sc["weights"], np.float32))
    os.makedirs(out, exist_ok=True)
    z = np.array([s[2] for s in rec["state"]])
    np.savez_compressed(os.path.join(out, f"seed{seed}.npz"), **{k: np.stack(v) for k, v in rec.items()})
    print(f"seed {seed}: ball z min {z.min():+.3f} at tick {int(z.argmin()) + 1}; first tick below 0.08: "
          f"{int(np.argmax(z < 0.08)) + 1 if (z < 0.08).any() else None}", flush=True)


def analyse(path, nsample):
    """CPU only.  (1) every plant transition against the oracle; (2) the tick the ball leaves the hand (the rollouts' rewards stop depending
    on the controls: their spread collapses); (3) around it: the planner's per-rollout rewards against the oracle's for the same controls, what
    the plan the update settled on predicts for the ball (oracle rollout of it) against what the plant then did; (4) the ORACLE's loop (oracle
    plant + oracle planner, same Philox noise) continued from the recorded state four ticks before the toss."""
    import oracle as O
    from allegro_closed_loop_study import philox_normal, setup
    from dial_mpc_amd.core.dial_core import make_cfg
    seed = int(os.path.basename(path).replace("seed", "").replace(".npz", ""))
    dc, ec, env = setup(nsample, seed)
    model, task, cfg = env.make_model(), env.make_task(), make_cfg(dc)
    o32 = O.Oracle(model, task, cfg, np.float32)
    nq, nv, nu, Hn1 = model.nq, model.nv, model.nu, dc.Hnode + 1
    W = np.array([[cfg.W[t][k] for k in range(Hn1)] for t in range(dc.Hsample + 1)], np.float32)
    p = np.load(path)
    S, Yo, ticks = p["state"], p["Y_out"], p["state"].shape[0]
    z = S[:, 2]
    print(f"== {path}: ball z min {z.min():+.3f} (tick {int(z.argmin()) + 1})")
    worst = 0.0
    for t in range(ticks - 1):
        cold = S[t].copy(); cold[nq + nv:nq + 2 * nv] = 0
        e = min(np.abs(o32.env_step(st, Yo[t][0])[0][nq:nq + nv] - S[t + 1][nq:nq + nv]).max() for st in (S[t], cold))
        worst = max(worst, float(e))
    print(f"(1) plant: {ticks - 1} env.step transitions against the oracle from the same state and action: worst |d qd| {worst:.3g} rad/s")
    std = np.array([p["rewss"][t].mean(1).std() for t in range(ticks)])
    if z.min() >= 0.08:
        print("(2) the ball stays in the hand")
        return
    tt = int(np.argmax(z < 0.08))            # the tick the ball is below the hand; back to the start of that flight
    while tt > 1 and std[tt - 1] < 0.04:
        tt -= 1
    print(f"(2) the reward spread of the rollouts collapses at tick {tt + 1} (std {std[tt - 1]:.3f} -> {std[tt]:.3f}): the ball is in flight from there on")
    for t in range(max(0, tt - 4), min(ticks - 7, tt + 2)) if not os.environ.get("AUTOPSY_SKIP3") else ():
        us = np.einsum("tk,nka->nta", W, p["Y0s"][t]).astype(np.float32)
        g, o = p["rewss"][t].mean(1), np.asarray(o32.rollout(S[t], us)[0]).mean(1)
        w = p["weights"][t]
        lp = (o - o[-1]) / o.std() / dc.temp_sample
        wo = np.exp(lp - lp.max()); wo /= wo.sum()
        Yor = np.einsum("n,nka->ka", wo, p["Y0s"][t])
        plan = (W @ Yo[t]).astype(np.float32)
        rew, qs, _, _ = o32.rollout(S[t], plan[None])
        print(f"(3) tick {t + 1}: z {S[t][2]:.3f} | rollouts whose reward differs from the oracle's by > 0.01: {(np.abs(g - o) > 0.01).sum()} of {g.size}; "
              f"effective sample size {1 / np.sum(w ** 2):.1f}; |plan(device) - plan(update on the oracle's rewards)| {np.abs(Yor - Yo[t]).max():.3f} | "
              f"the plan's mean reward {np.asarray(rew)[0].mean():+.3f} (best sample {g.max():+.3f}); ball z it predicts {np.asarray(qs)[0][:6, 2].round(3)} "
              f"realised {np.round([S[t + k + 1][2] for k in range(6)], 3)}")
    sigma = (dc.horizon_diffuse_factor ** np.arange(Hn1)[::-1] * dc.sigma_scale).astype(np.float32)
    backs = tuple(int(b) for b in os.environ.get("AUTOPSY_BACKS", "1,2,3,4,6").split(","))   # how many ticks before the flight the oracle takes over
    for back in backs:
        k0 = tt - back          # the oracle takes over after tick k0 (1-based): state S[k0 - 1], plan Yo[k0 - 1]
        if k0 < 1:
            continue
        state, Y = S[k0 - 1].copy(), Yo[k0 - 1].copy()
        counter = dc.Ndiffuse_init + (k0 - 1) * dc.Ndiffuse
        zs = []
        for t in range(k0, ticks):
            state = o32.env_step(state, Y[0])[0]
            Y = np.asarray(o32.shift(Y), np.float32)
            for i in range(dc.Ndiffuse):
                eps = philox_normal(int(dc.seed), counter, dc.Nsample, Hn1 * nu).reshape(dc.Nsample, Hn1, nu)
                counter += 1
                Y = np.asarray(o32.reverse_once(state, Y, (sigma * np.float32(dc.traj_diffuse_factor ** i)).astype(np.float32), eps)["Ybar"], np.float32)
            zs.append(float(state[2]))
        print(f"(4) the oracle's own loop (oracle plant + planner, same Philox counters) taking over after tick {k0} = {back} tick(s) before the ball's "
              f"flight starts: {'LOSES THE BALL TOO' if min(zs) < 0.08 else 'keeps the ball'}  (ball z every 4th tick {np.round(zs[::4], 3)}; recorded "
              f"run {np.round(z[k0::4], 3)})", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--analyse", default=None, help="a recorded seedN.npz: the CPU-side analysis against the oracle")
    ap.add_argument("--seeds", default="0,11,31,36")
    ap.add_argument("--nsample", type=int, default=512)
    ap.add_argument("--ticks", type=int, default=40)
    ap.add_argument("--out", default="gpurun_out/autopsy")
    args = ap.parse_args()
    if args.analyse:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        return analyse(args.analyse, args.nsample)
    for s in args.seeds.split(","):
        record(int(s), args.nsample, args.ticks, args.out)


if __name__ == "__main__":
    main()
