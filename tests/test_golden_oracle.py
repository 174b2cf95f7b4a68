"""The committed fixtures are reproduced by the current oracle (fp64 and fp32) and the wave emulator."""
import os

import numpy as np
import pytest

import oracle as O
from conftest import TOL, setup_case

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,example,N,H", [("go2_trot_N64_H8", "unitree_go2_trot", 64, 8),
                                              ("go2_seq_jump_N48_H16", "unitree_go2_seq_jump", 48, 16),
                                              ("h1_jog_N32_H16", "unitree_h1_jog", 32, 16),
                                              ("h1_loco_N32_H20", "unitree_h1_loco", 32, 20),
                                              ("allegro_reorient_N64_H8", "allegro_reorient", 64, 8)])
def test_oracle_reproduces_fixture(name, example, N, H):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    dc, env, model, task, cfg = setup_case(example, N, H, per_rollout=True)   # the fixtures pin the per-rollout-comparable rule
    for dt, tol in ((np.float64, 1e-5), (np.float32, 2e-3)):
        orc = O.Oracle(model, task, cfg, dt)
        r = orc.reverse_once(g["state"], g["Ybar_in"], g["noise_scale"], g["eps"], full=True)
        ok = (np.abs(r["rewss"] - g["rewss"]) <= tol + tol * np.abs(g["rewss"])).all(1)
        # fp32 vs the fp64 fixture: every rollout on the fp64 branch for the legged robots (measured: all of them, Ybar within
        # 3e-6); ONLY Allegro -- 32 impact-rich physics sub-steps per rollout -- may lose a rollout to another branch
        chaotic = example == "allegro_reorient"
        assert ok.all() if (dt == np.float64 or not chaotic) else ok.mean() >= 0.9, (name, dt, float(ok.mean()))
        assert np.allclose(r["Ybar"], g["Ybar"], atol=max(tol, 1e-4) if ok.all() else 2e-2)
