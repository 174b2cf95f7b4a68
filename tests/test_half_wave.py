"""The Go2's two-samples-per-wavefront kernel, CPU side: its 32-lane layouts (csrc/smooth_quad2.h, csrc/solver_reg2.h) compiled
for the host wave emulator must reproduce the 64-lane layouts (smooth_quad.h, solver_reg.h) BIT FOR BIT once both sum in the
GPU's association (wave.h: tree_sums) -- they are the same arithmetic on a different lane map -- and match the fp32 oracle like
every other instantiation.  The GPU half of the claim (two samples in one wavefront == one sample per wavefront, on the device)
is tests/test_gpu_parity.py::test_two_samples_per_wavefront_is_bit_identical."""
import numpy as np
import pytest

import emu_lib
import oracle as O
from conftest import perturbed_state, seeded_inputs, setup_case, witness_parity

KEYS = ("Y0s", "rewss", "rews", "qss", "qdss", "xss")


@pytest.mark.parametrize("example,N,H,per_rollout", [("unitree_go2_trot", 64, 8, True), ("unitree_go2_trot", 64, 8, False),
                                                     ("unitree_go2_seq_jump", 48, 16, False)])
def test_half_wave_layout_is_bit_identical_to_the_full_wave_layout(example, N, H, per_rollout):
    dc, env, model, task, cfg = setup_case(example, N, H, per_rollout=per_rollout)
    o32 = O.Oracle(model, task, cfg, np.float32)
    half, full = emu_lib.Emu(model, task, cfg, path=2), emu_lib.Emu(model, task, cfg, path=3)
    states = [o32.env_reset(env._init_q, np.zeros(model.nv))[0]]
    for seed in range(2):
        q, qd = perturbed_state(env, seed)
        states.append(o32.env_reset(q, qd)[0])
    for k, s0 in enumerate(states):
        eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=k, Ybar_scale=0.3 if k else 0.0)
        rh = half.rollout_nodes(s0, Ybar, sigma, eps, check_races=(k == 0))   # asserts: zero races
        rf = full.rollout_nodes(s0, Ybar, sigma, eps, check_races=False)
        for key in KEYS:
            assert np.array_equal(rh[key].view(np.uint32), rf[key].view(np.uint32)), (example, k, key, np.abs(rh[key] - rf[key]).max())


def test_half_wave_layout_matches_oracle():
    dc, env, model, task, cfg = setup_case("unitree_go2_trot", 64, 8, per_rollout=True)
    o32 = O.Oracle(model, task, cfg, np.float32)
    emu = emu_lib.Emu(model, task, cfg, path=2)
    s0, _, _ = o32.env_reset(env._init_q, np.zeros(model.nv))
    eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=0)
    ro = o32.reverse_once(s0, Ybar, sigma, eps, full=True)
    re = emu.rollout_nodes(s0, Ybar, sigma, eps, check_races=False)
    rep = witness_parity(o32, s0, ro["us"], (re["rewss"], re["qss"], re["qdss"], re["xss"]), "unitree_go2_trot", model.nq + 2 * model.nv)
    if rep["witnessed"] == 0:
        assert np.allclose(re["rews"], ro["rews"], rtol=5e-4, atol=5e-4)
