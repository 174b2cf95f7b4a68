#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ with the fp64 CPU oracle.

These fixtures pin the ORACLE's outputs (so that the GPU tests do not depend on rebuilding it), not the
JAX reference: the reference cannot run here.  tools/export_reference_vectors.py is the hook
that produces reference-generated vectors on a machine with the JAX stack."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
from conftest import seeded_inputs, setup_case  # noqa: E402

for name, example, N, H in [("go2_trot_N64_H8", "unitree_go2_trot", 64, 8),
                            ("go2_seq_jump_N48_H16", "unitree_go2_seq_jump", 48, 16),
                            ("h1_jog_N32_H16", "unitree_h1_jog", 32, 16),
                            ("h1_loco_N32_H20", "unitree_h1_loco", 32, 20),
                            ("allegro_reorient_N64_H8", "allegro_reorient", 64, 8)]:
    dc, env, model, task, cfg = setup_case(example, N, H, per_rollout=True)   # the fixtures pin the per-rollout-comparable rule
    o64 = O.Oracle(model, task, cfg, np.float64)
    s0, _, _ = o64.env_reset(env._init_q, np.zeros(model.nv))
    eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=0, Ybar_scale=0.2)
    r = o64.reverse_once(s0, Ybar, sigma, eps, full=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), state=s0.astype(np.float32),
                        eps=eps, noise_scale=sigma, Ybar_in=Ybar, rewss=r["rewss"].astype(np.float32),
                        rews=r["rews"].astype(np.float32), weights=r["weights"].astype(np.float32),
                        Ybar=r["Ybar"].astype(np.float32), qbar=r["qbar"].astype(np.float32),
                        xbar=r["xbar"].astype(np.float32))
    print("wrote", name)
