"""The register-resident position / velocity stage (csrc/smooth_quad.h, csrc/smooth_rows.h) is laid out for ONE body tree per
instantiation.  Host side: a model gets that instantiation only if the layout comes out as compiled (cmodel.h: quad_fits /
rows_build inside dims_match); any other model of the same dimensions runs on the generic instantiation -- and gives the same
physics.  CPU tests (host wave emulator = the exact kernel logic; nothing here is the product path)."""
import numpy as np
import pytest

import emu_lib
import oracle as O
from conftest import TOL, perturbed_state, setup_case


def _rollout_err(example, model, task, cfg, env, path=0, H=8):
    o32 = O.Oracle(model, task, cfg, np.float32)
    emu = emu_lib.Emu(model, task, cfg, path=path)
    q, qd = perturbed_state(env, 1)
    s0, _, _ = o32.env_reset(q, qd)
    us = np.random.default_rng(4).uniform(-0.6, 0.6, (6, H + 1, model.nu)).astype(np.float32)
    ro = o32.rollout(s0, us)
    re = emu.rollout(s0, us, check_races=True)
    return emu.sizes()[0], {n: float(np.max(np.abs(a - b) / (TOL[k]["atol"] + TOL[k]["rtol"] * np.abs(a))))
                            for n, k, a, b in zip(("rew", "q", "qd", "x"), ("rewss", "q", "qd", "x"), ro, re)}


@pytest.mark.parametrize("example,inst", [("unitree_go2_trot", 1), ("unitree_h1_jog", 2), ("unitree_h1_loco", 3), ("allegro_reorient", 4)])
def test_specialised_instantiation_runs_the_register_stage_and_matches_the_oracle(example, inst):
    dc, env, model, task, cfg = setup_case(example, 8, 8, per_rollout=True)
    got, err = _rollout_err(example, model, task, cfg, env)
    assert got == inst
    assert max(err.values()) <= 1.0, err


def test_a_go2_shaped_model_with_another_layout_falls_back_to_the_generic_instantiation():
    """Same dimensions, same dof tree, but the foot sites sit on the thighs: not the quadruped layout smooth_quad.h assumes
    (quad_fits) -> generic instantiation (0), same physics (the site positions feed the gait reward: the oracle sees the
    same model)."""
    dc, env, model, task, cfg = setup_case("unitree_go2_trot", 8, 8, per_rollout=True)
    m2 = type(model).from_buffer_copy(model)
    for r in range(4):
        m2.site_bodyid[1 + r] = 3 + 3 * r          # thigh of leg r instead of its calf
    got, err = _rollout_err("unitree_go2_trot", m2, task, cfg, env)
    assert got == 0, got
    assert max(err.values()) <= 1.0, err


def test_an_h1_shaped_model_with_three_geoms_on_a_body_falls_back():
    """rows_build allows two geoms per body lane: a third on the same ankle does not fit the layout -> generic instantiation."""
    dc, env, model, task, cfg = setup_case("unitree_h1_loco", 8, 8, per_rollout=True)
    m2 = type(model).from_buffer_copy(model)
    assert list(m2.geom_bodyid[:5]) == [0, 6, 6, 11, 11]
    m2.geom_bodyid[3] = 6                           # three capsules on the left ankle, one on the right
    emu = emu_lib.Emu(m2, task, cfg)
    assert emu.sizes()[0] == 0


def test_crate_climb_runs_the_quadruped_stage_and_another_site_layout_falls_back():
    """The generic feature set on the Go2's tree (crate climb, Dims::quad_gen): bodies and dofs in registers (smooth_quad.h without
    the fused foot contacts), geom frames / collisions / rows generic.  cmodel.h: quad_tree_fits && quad_gen_fits inside dims_match;
    with the foot sites on the thighs the same model runs on the capacity-dimension instantiation -- same physics."""
    dc, env, model, task, cfg = setup_case("unitree_go2_crate_climb", 8, 8, per_rollout=True)
    got, err = _rollout_err("unitree_go2_crate_climb", model, task, cfg, env)
    assert got == 5, got
    assert max(err.values()) <= 1.0, err
    m2 = type(model).from_buffer_copy(model)
    for r in range(4):
        m2.site_bodyid[1 + r] = 3 + 3 * r
    got2, err2 = _rollout_err("unitree_go2_crate_climb", m2, task, cfg, env)
    assert got2 == 0, got2
    assert max(err2.values()) <= 1.0, err2


def test_push_crate_runs_the_row_layout_and_a_site_on_the_crate_falls_back():
    """The generic feature set on the H1's tree (push crate, Dims::rows_gen): the row layout for the robot, the crate -- a slide
    joint on the world, a tree of its own -- as a one-lane phase (smooth_rows.h: solo_slide_body), geom frames / collisions / rows
    generic.  rows_build(extra_trees) inside dims_match; a site on the crate does not fit -> capacity-dimension instantiation."""
    dc, env, model, task, cfg = setup_case("unitree_h1_push_crate", 8, 8, per_rollout=True)
    got, err = _rollout_err("unitree_h1_push_crate", model, task, cfg, env)
    assert got == 6, got
    assert max(err.values()) <= 1.0, err
    m2 = type(model).from_buffer_copy(model)
    m2.site_bodyid[2] = model.nbody - 1
    assert emu_lib.Emu(m2, task, cfg).sizes()[0] == 0
