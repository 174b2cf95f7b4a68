#!/usr/bin/env python3
"""Per-section cycle breakdown of the rollout kernel (needs the -DDIAL_PROFILE build):
     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDIAL_PROFILE -o libdialhip_prof.so dial_hip.hip
     DIAL_HIP_LIB=.../libdialhip_prof.so python tools/profile_sections.py
Prints shader-clock cycles accumulated by sample 0 over one rollout for B = 1 and B = N+1."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import seeded_inputs, setup_case  # noqa: E402
from dial_mpc_amd import _lib  # noqa: E402

NAMES = {24: "per-step output stores", 25: "ctrl (act2tau) + gait clock", 0: "kinematics (levels)", 1: "M + qfs + collision", 16: "frames (bodies, geoms, sites)", 17: "subtree COM", 18: "cinert + cdof", 19: "cvel", 20: "cdof_dot", 21: "cacc", 22: "crb + local forces", 23: "F_i + cfrc", 2: "Jc + efc rows", 3: "chol(M)+solve",
         4: "warmstart select + constraint_grad", 5: "H build", 6: "chol(H)+solve", 7: "linesearch", 8: "post-ls update/sums",
         9: "euler", 10: "ctrl + reward", 11: "act/output/IO", 14: "(newton_dir entry)", 15: "(forward entry)",
         28: "EVENTS (all samples, cumulative): 2nd Newton iterations", 29: "  ... with an unchanged active set",
         30: "  line-search iterations", 31: "  Newton iterations"}
if len(sys.argv) > 1 and "crate" in sys.argv[1]:
    NAMES.update({14: "H: contact weights", 12: "H: GEMM loop (2 MFMA per touching contact)", 5: "H: accumulator tile -> packed H"})
if len(sys.argv) > 1 and sys.argv[1] == "allegro_reorient":
    NAMES.update({27: "EVENTS (all samples): contributing units (sum over solves)", 28: "  constraint solves (physics sub-steps)",
                  29: "  Newton iterations on the 3-points-per-pass line search (<= 16 units)",
                  14: "  solver: warm-start selection (per solve)", 12: "  solver: unit zones / forces / weights (per Newton it)", 13: "  solver: J^T f + gradient", 4: "  solver: sums + convergence test (+ warm start)",
                  24: "  solver: H = M copy + limit rows (+ per-step output stores)", 5: "  solver: H contact blocks", 25: "  solver: LS set-up (J v, M v, sums, unit registers) (+ ctrl)",
                  26: "  solver: LS opening points p0, p1", 7: "  solver: LS bracketing iterations + update"})

example = sys.argv[1] if len(sys.argv) > 1 else "unitree_go2_trot"
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
HS = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dc, env, model, task, cfg = setup_case(example, NS, HS)
ctx = _lib.Context(model, task, cfg)
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device="cuda")  # noqa: E731
s0, _, _ = ctx.env_reset(dev(env._init_q), dev(np.zeros(model.nv)))
eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=0)
lib = ctx.lib
lib.dial_debug_prof.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
W = np.array([[cfg.W[t][k] for k in range(dc.Hnode + 1)] for t in range(dc.Hsample + 1)], np.float32)
for label, B in (("B=1", 1), (f"B={NS + 1}", NS + 1)):
    if B == 1:
        us = dev((W @ np.clip(Ybar, -1, 1))[None])
        ctx.rollout(s0, us)
        # dial_rollout does not pass the profile buffer; use the shard path with n_local = 0 (mean sample only)
        rews = torch.zeros(1, device="cuda")
        ctx.shard_rollout(s0, dev(Ybar), dev(sigma), dev(eps[:1]), 0, True, rews)
    else:
        ctx.reverse_once(s0, dev(Ybar), dev(sigma), dev(eps))
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    assert lib.dial_debug_prof(ctx.h, out) == 0
    tot = sum(out)
    print(f"--- {example} {label}: total {tot} cycles over {dc.Hsample + 1} steps = {tot / (dc.Hsample + 1):.0f} / step")
    for k in range(32):
        if out[k]:
            print(f"  [{k:2d}] {NAMES.get(k, '?'):42s} {out[k]:10d}  {100.0 * out[k] / tot:5.1f}%  {out[k] / (dc.Hsample + 1):9.0f}/step")
