#!/usr/bin/env python3
"""Survey behind the thresholds of the shipped-rule parity gates (tests/conftest.py: transition_parity, distribution_parity):
for every env at its BASELINE / example size, under the solver settings the model ships with (`_in_bracket`, truncated),
  * per transition: share of the (rollout, step) transitions of 96 trajectories that the oracle reproduces directly from the
    device's own traced state at 1 x TOL, share that needs a <= 64 ulp witness (and at how many ulp), unwitnessed ones;
  * distribution level: the GPU's aggregate deviations over the p95 / max of a 32-member 1-ulp jitter ensemble of the oracle.
`--ieee` repeats a case on libdialhip_ieee.so (no device fast-math) -- how much of a deviation is the fast-math rounding.
Needs a GPU.  Output: profiles/r04_transition_parity.txt (collected by tools/collect_profiles_r04.sh)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

CASES = [("unitree_go2_trot", 2048, 16), ("unitree_go2_seq_jump", 1024, 16), ("unitree_h1_jog", 2048, 16), ("unitree_h1_loco", 1024, 20),
         ("allegro_reorient", 4096, 24), ("unitree_go2_crate_climb", 2048, 25), ("unitree_h1_push_crate", 2048, 24)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--members", type=int, default=32)
    ap.add_argument("--traj", type=int, default=96)
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--ieee", action="store_true", help="also run every case on libdialhip_ieee.so")
    ap.add_argument("--no-dist", action="store_true")
    args = ap.parse_args()
    import torch
    import oracle as O
    from conftest import distribution_parity, perturbed_state, seeded_inputs, setup_case, transition_parity, transition_sample
    from dial_mpc_amd import _lib
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device="cuda")  # noqa: E731
    for example, N, H in CASES:
        if args.only and args.only not in example:
            continue
        dc, env, model, task, cfg = setup_case(example, N, H)
        o32 = O.Oracle(model, task, cfg, np.float32)
        libs = [("product", None)] + ([("ieee", _lib.IEEE_LIB_PATH)] if args.ieee else [])
        for seed in range(args.seeds):
            q, qd = (env._init_q, np.zeros(model.nv)) if seed == 0 else perturbed_state(env, seed)
            s0, _, _ = o32.env_reset(q, qd)
            eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=seed, Ybar_scale=0.2)
            for tag, path in libs:
                ctx = _lib.Context(model, task, cfg, lib_path=path)
                tr = ctx.set_state_trace(N + 1)
                out = ctx.reverse_once(dev(s0), dev(Ybar), dev(sigma), dev(eps))
                sc = ctx.debug_scratch()
                W = np.array([[cfg.W[t][k] for k in range(dc.Hnode + 1)] for t in range(H + 1)], np.float32)
                us = np.einsum("tk,nka->nta", W, sc["Y0s"]).astype(np.float32)
                got = (sc["rewss"], sc["qss"], sc["qdss"], sc["xss"])
                t0 = time.time()
                trep = transition_parity(o32, s0, us, got, tr.cpu().numpy(), transition_sample(N, args.traj, seed), model.nq, model.nv,
                                         example=example, check=False)
                print(f"{example} N={N} H={H} seed={seed} [{tag}] transitions {trep['transitions']}: direct {100 * trep['direct_share']:.2f} % "
                      f"(worst {trep['direct_worst']:.2f}), witnessed {trep['witnessed']} = {100 * trep['witnessed_share']:.2f} % {trep['witness_ulp']}, "
                      f"unwitnessed {trep['unwitnessed']}  [{time.time() - t0:.0f} s]", flush=True)
                if not args.no_dist:
                    t0 = time.time()
                    prod = {k: out[k].cpu().numpy() for k in ("Ybar", "qbar", "qdbar", "xbar")}
                    rep = distribution_parity(o32, s0, us, sc["Y0s"], got, prod, cfg.temp_sample, members=args.members, check=False)
                    r95 = {k: round(v, 2) for k, v in rep["ratio"].items()}
                    rmax = {k: round(rep["gpu"][k] / max(rep["envelope_max"][k], 1e-30), 2) for k in rep["ratio"]}
                    print(f"   distribution [{tag}]: ESS oracle {rep['ess_oracle']:.1f} / GPU {rep['ess_gpu']:.1f}; GPU / p95-of-{args.members} {r95}\n"
                          f"      GPU / max-of-{args.members} {rmax}\n      outside: GPU {rep['gpu']['outside']:.3f}, ensemble max {rep['envelope_max']['outside']:.3f}"
                          f"  [{time.time() - t0:.0f} s]", flush=True)
                del ctx


if __name__ == "__main__":
    main()
