"""The contact ARRAY as data: the order and multiplicity of the static contact list can be rebuilt to match
a reference run's array without touching compiler, oracle or kernel; the crate envs find their reward contacts either by geom
identity or at upstream's literal positions, and the two agree wherever the array is laid out as upstream's indices assume."""
import numpy as np
import pytest
import yaml

import oracle as O
from conftest import TOL, with_solver
from dial_mpc_amd import mjcf
from dial_mpc_amd.core.dial_core import load_dial_and_env, make_cfg
from dial_mpc_amd.utils.io_utils import get_example_path


def _env(example, **over):
    d = yaml.safe_load(open(get_example_path(example + ".yaml")))
    d.update(Nsample=8, Hsample=8, **over)
    return load_dial_and_env(d)


@pytest.mark.parametrize("example", ["unitree_go2_crate_climb", "unitree_h1_push_crate"])
def test_reordering_the_contact_array_does_not_change_the_physics(example):
    """A random permutation of the array (given as MuJoCo geom ids per slot, the form a reference run exports): the oracle's
    rollouts agree to summation-order rounding, and the identity lookup follows its contacts to their new positions."""
    dc, _, env0 = _env(example)
    m0 = env0.sys.model
    slots = mjcf.contact_slots(m0, ids="mujoco")
    perm = np.random.default_rng(3).permutation(len(slots))
    # candidates of one pair must keep their relative order (the k-th occurrence takes candidate k): sort each pair's positions
    pos = {}
    for k in perm:
        pos.setdefault(slots[k], []).append(k)
    shuffled = [slots[k] for k in perm]
    dc1, _, env1 = _env(example, contact_slots=[list(p) for p in shuffled])
    m1 = env1.sys.model
    assert int(m1["ncon"]) == int(m0["ncon"]) and int(m1["nefc"]) == int(m0["nefc"])
    assert sorted(mjcf.contact_slots(m1, ids="mujoco")) == sorted(slots) and mjcf.contact_slots(m1, ids="mujoco") == shuffled
    cfg = make_cfg(dc)
    # (solver run to convergence: under the shipped truncated settings the order of the row sums decides knife edges)
    conv = lambda e: with_solver(e.make_model(), ls_rule=0, iterations=50, ls_iterations=50)  # noqa: E731
    o0 = O.Oracle(conv(env0), env0.make_task(), cfg, np.float32)
    o1 = O.Oracle(conv(env1), env1.make_task(), cfg, np.float32)
    s0, _, _ = o0.env_reset(env0._init_q, np.zeros(m0["nv"]))
    s1, _, _ = o1.env_reset(env1._init_q, np.zeros(m0["nv"]))
    us = np.random.default_rng(0).uniform(-0.5, 0.5, (8, 9, int(m0["nu"]))).astype(np.float32)
    r0, r1 = o0.rollout(s0, us), o1.rollout(s1, us)
    for name, a, b in zip(("rewss", "q", "qd", "x"), r0, r1):
        assert np.allclose(a, b, rtol=TOL[name]["rtol"], atol=TOL[name]["atol"]), name
    # the reward's contacts, looked up by identity, name the same geom pairs as before
    if example == "unitree_go2_crate_climb":
        assert [slots[c] for c in env0._crate_contact] == [shuffled[c] for c in env1._crate_contact]
    else:
        assert sorted(slots[c] for c in env0._pc_wanted) == sorted(shuffled[c] for c in env1._pc_wanted)


def test_push_crate_literal_positions_are_the_identity_lookup_in_geom_pair_order():
    """unitree_h1_env.py:474-480, 525-531 read dist[2:4], dist[6:8], contacts 26, 27 and 14 .. 25 by position.  With the
    array in plain geom-pair order -- floor against knee / foot capsules (2 each), torso (4), hands, then the crate against the
    same geoms -- those positions ARE the foot capsules' floor contacts, the hands on the crate and every other part on the
    crate: the literal mode and the identity mode pick the same contacts."""
    _, _, env = _env("unitree_h1_push_crate")
    m = env.sys.model
    slots = mjcf.contact_slots(m, ids="mujoco")
    pair_order = sorted(range(len(slots)), key=lambda c: (slots[c][0], slots[c][1], int(m["con_sub"][c]), int(m["con_kind"][c]) == 2))
    layout = [list(slots[c]) for c in pair_order]
    _, _, e_id = _env("unitree_h1_push_crate", contact_slots=layout)
    _, _, e_lit = _env("unitree_h1_push_crate", contact_slots=layout, contact_lookup="literal")
    assert e_id._pc_foot_contact == [[2, 3], [6, 7]] == e_lit._pc_foot_contact
    assert sorted(e_id._pc_wanted) == [26, 27] == e_lit._pc_wanted
    assert sorted(e_id._pc_unwanted) == list(range(14, 26)) == e_lit._pc_unwanted
    ti, tl = e_id.make_task(), e_lit.make_task()
    assert bytes(ti) == bytes(tl)


def test_crate_climb_literal_positions_follow_the_layout():
    """unitree_go2_env.py:750 reads contacts 16 .. 19.  In THIS compiler's order those are floor contacts (as in
    every MJX ordering checked); a layout that puts the four foot-sphere / crate contacts there makes the literal mode and the
    identity mode agree -- the mechanism a reference run's exported array would go through."""
    _, _, env = _env("unitree_go2_crate_climb")
    m = env.sys.model
    slots = mjcf.contact_slots(m, ids="mujoco")
    feet = list(env._crate_contact)
    assert feet != [16, 17, 18, 19]
    rest = [c for c in range(len(slots)) if c not in feet]
    order = rest[:16] + feet + rest[16:]
    layout = [list(slots[c]) for c in order]
    _, _, e_id = _env("unitree_go2_crate_climb", contact_slots=layout)
    _, _, e_lit = _env("unitree_go2_crate_climb", contact_slots=layout, contact_lookup="literal")
    assert e_id._crate_contact == [16, 17, 18, 19] == e_lit._crate_contact
    assert bytes(e_id.make_task()) == bytes(e_lit.make_task())


def test_extra_candidates_of_a_box_pair_are_parked():
    """A reference array may list a box pair more often than this compiler emits it (newer MJX: 8 box-box points): the extra
    slots get the next con_sub and the geometry parks them (dist = 1, no rows) -- on the oracle and in the kernel logic."""
    import emu_lib
    dc, _, env = _env("unitree_h1_push_crate")
    m = env.sys.model
    slots = mjcf.contact_slots(m, ids="mujoco")
    bb = [c for c in range(len(slots)) if int(m["con_kind"][c]) == 8]
    assert len(bb) == 4
    layout = [list(p) for p in slots] + [list(slots[bb[0]])] * 4          # 8 candidates for the torso / crate pair
    _, _, env8 = _env("unitree_h1_push_crate", contact_slots=layout)
    m8 = env8.sys.model
    assert int(m8["ncon"]) == 32 and int(m8["nefc"]) == int(m["nefc"]) + 16
    assert [int(s) for s in np.asarray(m8["con_sub"])[-4:]] == [4, 5, 6, 7]
    cfg = make_cfg(dc)
    conv = lambda e: with_solver(e.make_model(), ls_rule=0, iterations=50, ls_iterations=50)  # noqa: E731
    o0 = O.Oracle(conv(env), env.make_task(), cfg, np.float32)
    o8 = O.Oracle(conv(env8), env8.make_task(), cfg, np.float32)
    emu8 = emu_lib.Emu(conv(env8), env8.make_task(), cfg, path=1)
    s0, _, _ = o0.env_reset(env._init_q, np.zeros(m["nv"]))
    us = np.random.default_rng(1).uniform(-0.5, 0.5, (4, 9, int(m["nu"]))).astype(np.float32)
    r0, r8, re = o0.rollout(s0, us), o8.rollout(s0, us), emu8.rollout(s0, us)
    for name, a, b, c in zip(("rewss", "q", "qd", "x"), r0, r8, re):
        assert np.allclose(a, b, rtol=TOL[name]["rtol"], atol=TOL[name]["atol"]), name
        assert np.allclose(a, c, rtol=TOL[name]["rtol"], atol=TOL[name]["atol"]), name
