"""Crate climb (UnitreeGo2CrateEnv) on the CPU: compiled scene, the oracle's physics on the crate,
and the kernel body (generic instantiation, host wave emulator) against the fp32 oracle from states that TOUCH the crate
with every kind of geom -- foot and base spheres, calf capsules, the trunk box.

The box narrow phases are restated geometry, not MJX's collision_convex (tests/test_box_collisions.py pins them to brute
force; DESIGN.md section 1 says what that does and does not establish)."""
import numpy as np
import pytest

import emu_lib
import oracle as O
from conftest import TOL, _within, seeded_inputs, setup_case, witness_parity

EX = "unitree_go2_crate_climb"


def _quat(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw / 2), np.sin(yaw / 2)
    return np.array([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy])


def touching_state(env, o64, seed, depth=0.002):
    """A random pose over the crate, lowered until its deepest candidate contact penetrates by `depth`."""
    rng = np.random.default_rng(seed)
    q = np.array(env._init_q, dtype=np.float64)
    q[0:2] = [rng.uniform(0.95, 1.6), rng.uniform(-0.4, 0.4)]
    q[3:7] = _quat(rng.uniform(-0.5, 0.5), rng.uniform(-0.7, 0.5), rng.uniform(-0.6, 0.6))
    q[7:] += rng.uniform(-0.3, 0.3, 12)
    q[2] = 1.2
    nv = env.sys.nv
    for _ in range(40):
        dmin = float(o64.forward_dump(q, np.zeros(nv))["con_dist"].min())
        if abs(dmin + depth) < 1e-5:
            break
        q[2] -= (dmin + depth) * 0.9
    qd = rng.normal(0, 0.3, nv)
    return q, qd


@pytest.fixture(scope="module")
def case():
    dc, env, model, task, cfg = setup_case(EX, 12, 5, per_rollout=True)
    return dc, env, model, task, cfg


def test_compiled_scene(case):
    dc, env, model, task, cfg = case
    md = env.sys.model
    assert (md["nq"], md["nv"], md["nu"], md["nbody"], md["ngeom"], md["ncon"], md["nefc"]) == (19, 18, 12, 15, 17, 52, 220)
    names = md["names"]["geom"]
    kinds = {}
    for c in range(md["ncon"]):
        kinds.setdefault(int(md["con_kind"][c]), []).append(c)
    # 6 spheres + 8 capsules (2 ends) + the trunk box (4 vertices) against the floor; the same 15 geoms against the crate
    assert {k: len(v) for k, v in kinds.items()} == {0: 6, 1: 8, 2: 8, 5: 4, 6: 6, 7: 16, 8: 4}
    floor, box = names.index("floor"), names.index("static_box")
    assert not any({int(md["con_geom1"][c]), int(md["con_geom2"][c])} == {floor, box} for c in range(md["ncon"]))   # both welded to the world
    # the four contacts reward_contact reads: FR, FL, RR, RL foot spheres against the crate
    assert [names[int(md["con_geom1"][c])] for c in env._crate_contact] == ["FR", "FL", "RR", "RL"]
    assert all(int(md["con_geom2"][c]) == box and int(md["con_kind"][c]) == 6 for c in env._crate_contact)
    assert task.kind == 5 and list(task.crate_contact) == env._crate_contact
    assert md["iterations"] == 2 and md["ls_iterations"] == 5 and md["cone"] == 0 and md["eulerdamp"] == 0


def test_robot_robot_contact_falls_back_to_the_capacity_dimension_kernel(case):
    """DimsGo2Crate factorises H = M + J^T D J in the dof TREE's elimination order, which is only valid while every contact has a
    static side (world, or the crate welded to it).  A model with the same counts and one contact between two MOVING bodies
    (two legs: cross-branch fill in H) must not select that instantiation (cmodel.h: dims_match) -- it runs on the generic one."""
    import copy
    dc, env, model, task, cfg = case
    assert emu_lib.Emu(model, task, cfg).sizes()[0] == 5
    m2 = copy.deepcopy(model)
    c = next(c for c in range(m2.ncon) if m2.con_body1[c] == 0)         # a floor contact ...
    other_leg = next(b for b in range(1, m2.nbody) if m2.body_dofnum[b] > 0 and b != m2.con_body2[c] and m2.body_parent[b] != m2.con_body2[c]
                     and m2.body_parent[m2.con_body2[c]] != b and b > 1)
    m2.con_body1[c] = other_leg                                          # ... becomes a contact between two robot bodies
    assert emu_lib.Emu(m2, task, cfg).sizes()[0] == 0


def test_oracle_robot_stands_on_the_crate_and_the_contact_reward_counts_its_feet(case):
    dc, env, model, task, cfg = case
    o64 = O.Oracle(model, task, cfg, np.float64)
    q = np.array(env._init_q, dtype=np.float64)
    q[0:3] = [1.3, 0.0, 0.87]
    s, _, _ = o64.env_reset(q, np.zeros(model.nv))
    # hold the home pose with the PD law: the action that maps to the home joint angles
    jr = np.asarray(env.joint_range)
    act = (2 * (q[7:] - jr[:, 0]) / (jr[:, 1] - jr[:, 0]) - 1.0)
    zs, rews = [], []
    for _ in range(80):
        s_prev = s
        s, xp, xq, _c = o64.env_step(s, act)
        zs.append(s[2])
        rews.append(s[model.nq + 2 * model.nv + 21])
    # the soft PD law (kp = 30) lets the stance sag by ~7 cm, then it rests on the crate's top face (0.6 m)
    assert 0.75 < zs[-1] < 0.88 and abs(zs[-1] - zs[-10]) < 2e-3
    f = o64.forward_dump(s[:model.nq], s[model.nq:model.nq + model.nv], ctrl=np.zeros(model.nu))
    live = set(np.flatnonzero(f["con_dist"] < 0.001))
    md = env.sys.model
    box = md["names"]["geom"].index("static_box")
    assert set(env._crate_contact) <= live and all(int(md["con_geom2"][c]) == box for c in live)   # the feet (and nothing but the crate)
    # reward = -|head - target|^2 - 0.01 |up - z|^2 - 0.3 yaw^2 + 0.02 * 4, from the PRE-integration pose of the step
    from dial_mpc_amd import mjcf
    R = mjcf.quat_to_mat(s_prev[3:7])
    head = s_prev[0:3] + R @ np.array([0.285, 0, 0])
    up = R @ np.array([0.0, 0, 1])
    yaw = np.arctan2(R[1, 0], R[0, 0])
    want = -np.sum((head - np.array([1.45, 0, 0.87])) ** 2) - 0.01 * np.sum((up - [0, 0, 1]) ** 2) - 0.3 * yaw ** 2 + 0.08
    assert abs(rews[-1] - want) < 1e-6, (rews[-1], want)
    # on the floor in front of the crate no foot counts
    s0, _, _ = o64.env_reset(env._init_q, np.zeros(model.nv))
    s1, *_ = o64.env_step(s0, act)
    head0 = s0[0:3] + np.array([0.285, 0, 0])
    assert abs(s1[model.nq + 2 * model.nv + 21] + np.sum((head0 - np.array([1.45, 0, 0.87])) ** 2)) < 1e-6   # (the task constants are fp32)


def test_oracle_contact_forces_carry_the_weight(case):
    dc, env, model, task, cfg = case
    o64 = O.Oracle(model, task, cfg, np.float64)
    md = env.sys.model
    q = np.array(env._init_q, dtype=np.float64)
    q[0:3] = [1.3, 0.0, 0.87]
    s, _, _ = o64.env_reset(q, np.zeros(model.nv))
    jr = np.asarray(env.joint_range)
    act = (2 * (q[7:] - jr[:, 0]) / (jr[:, 1] - jr[:, 0]) - 1.0)
    for _ in range(60):
        s, xp, xq, ctrl = o64.env_step(s, act)
    f = o64.forward_dump(s[:model.nq], s[model.nq:model.nq + model.nv], ctrl=ctrl, warm=s[model.nq + model.nv:model.nq + 2 * model.nv])
    # generalised constraint force on the base's z translation = the robot's weight (quasi-static)
    fz = (f["efc_J"].T @ f["efc_force"])[2]
    weight = 9.81 * float(np.sum(md["body_mass"][1:14]))
    assert abs(fz - weight) < 0.05 * weight, (fz, weight)


@pytest.mark.parametrize("path", [0, 1], ids=["DimsGo2Crate", "DimsMax"])
@pytest.mark.parametrize("seed", range(6))
def test_emulated_generic_kernel_matches_oracle_on_the_crate(case, seed, path):
    """path 0: the scene's own instantiation (generic feature set at compile-time dimensions, what the HIP library picks);
    path 1: the capacity-dimension instantiation (dial_options.force_generic)."""
    dc, env, model, task, cfg = case
    o32, o64 = O.Oracle(model, task, cfg, np.float32), O.Oracle(model, task, cfg, np.float64)
    emu = emu_lib.Emu(model, task, cfg, path=path)
    if path == 1 and seed > 1:
        pytest.skip("two seeds on the capacity-dimension instantiation")
    nv, nu = model.nv, model.nu
    q, qd = touching_state(env, o64, seed)
    live = np.flatnonzero(o64.forward_dump(q, np.zeros(nv))["con_dist"] < 0.001)
    assert live.size >= 1
    s_o, xp_o, xq_o = o32.env_reset(q, qd)
    s_e, xp_e, xq_e = emu.env_reset(q, qd, check_races=(seed == 0))
    nqv = model.nq + nv
    atol = np.full(s_o.shape, 2e-4)
    atol[nqv:nqv + nv] = 2e-4 * max(1.0, float(np.abs(s_o[nqv:nqv + nv]).max()) * 1e-2)
    assert np.all(np.abs(s_o - s_e) <= atol + 2e-4 * np.abs(s_o)), np.abs(s_o - s_e).max()
    assert np.allclose(xp_o, xp_e, atol=1e-6)
    rng = np.random.default_rng(100 + seed)
    us = rng.uniform(-1, 1, (dc.Nsample, dc.Hsample + 1, nu)).astype(np.float32)
    r_e = emu.rollout(s_o, us, check_races=(seed == 0))
    rep = witness_parity(o32, s_o, us, (r_e[0], r_e[1], r_e[2], r_e[3]), EX, model.nq + 2 * nv)
    assert rep["rollouts"] == dc.Nsample


def test_emulated_kernel_touches_with_every_geom_kind(case):
    """Coverage of the states the parity test draws: spheres, capsules and the trunk box all touch the crate in some of them."""
    dc, env, model, task, cfg = case
    o64 = O.Oracle(model, task, cfg, np.float64)
    md = env.sys.model
    seen = set()
    for seed in range(6):
        q, _ = touching_state(env, o64, seed)
        d = o64.forward_dump(q, np.zeros(model.nv))["con_dist"]
        seen |= {int(md["con_kind"][c]) for c in np.flatnonzero(d < 0.001)}
    assert {6, 7} <= seen, seen            # sphere-box and capsule-box
    # the trunk box needs a pose of its own: belly across the crate's front edge, legs stretched back
    q = np.array(env._init_q, dtype=np.float64)
    q[3:7] = _quat(0.0, -0.5, 0.0)
    q[7:] = [0.0, 1.4, -0.9, 0.0, 1.4, -0.9, 0.0, 2.5, -0.9, 0.0, 2.5, -0.9]
    Rp = np.array([[np.cos(-0.5), 0, np.sin(-0.5)], [0, 1, 0], [-np.sin(-0.5), 0, np.cos(-0.5)]])
    q[0:3] = np.array([0.99, 0.0, 0.6]) + Rp @ np.array([0.0, 0.0, 0.057 - 0.003])
    d = o64.forward_dump(q, np.zeros(model.nv))["con_dist"]
    assert any(int(md["con_kind"][c]) == 8 for c in np.flatnonzero(d < 0.001))


def test_emulated_kernel_matches_oracle_with_the_trunk_on_the_crate_edge(case):
    dc, env, model, task, cfg = case
    o32 = O.Oracle(model, task, cfg, np.float32)
    emu = emu_lib.Emu(model, task, cfg)
    nv, nu = model.nv, model.nu
    q = np.array(env._init_q, dtype=np.float64)
    q[3:7] = _quat(0.0, -0.5, 0.0)
    q[7:] = [0.0, 1.4, -0.9, 0.0, 1.4, -0.9, 0.0, 2.5, -0.9, 0.0, 2.5, -0.9]
    Rp = np.array([[np.cos(-0.5), 0, np.sin(-0.5)], [0, 1, 0], [-np.sin(-0.5), 0, np.cos(-0.5)]])
    q[0:3] = np.array([0.99, 0.0, 0.6]) + Rp @ np.array([0.0, 0.0, 0.057 - 0.003])
    s_o, _, _ = o32.env_reset(q, np.zeros(nv))
    rng = np.random.default_rng(7)
    us = rng.uniform(-1, 1, (dc.Nsample, dc.Hsample + 1, nu)).astype(np.float32)
    r_e = emu.rollout(s_o, us, check_races=False)
    witness_parity(o32, s_o, us, (r_e[0], r_e[1], r_e[2], r_e[3]), EX, model.nq + 2 * nv)


def test_emulated_reverse_once_on_the_crate(case):
    dc, env, model, task, cfg = case
    o32, o64 = O.Oracle(model, task, cfg, np.float32), O.Oracle(model, task, cfg, np.float64)
    emu = emu_lib.Emu(model, task, cfg)
    q, qd = touching_state(env, o64, 1)
    s_o, _, _ = o32.env_reset(q, qd)
    eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=0)
    ro = o32.reverse_once(s_o, Ybar, sigma, eps, full=True)
    re = emu.rollout_nodes(s_o, Ybar, sigma, eps, check_races=False)
    rep = witness_parity(o32, s_o, ro["us"], (re["rewss"], re["qss"], re["qdss"], re["xss"]), EX, model.nq + 2 * model.nv)
    if rep["witnessed"] == 0:
        assert _within(re["rews"], ro["rews"], TOL["rewss"]).all()


def test_capped_workspace_and_its_overflow_path_are_bit_identical(case):
    """The GPU rollout kernel sizes its LDS workspace for 16 TOUCHING contacts and runs a sample that touches with more on an
    overflow area in global memory (derived.h: ws_carve / ws_overflow).  Same code, same order of operations: the results
    must be bit-identical to the full-size workspace -- with a cap of 16 (never exceeded in these poses), of 2 (most steps
    overflow) and of 1 (every touching step overflows).  The emulator process reads the cap from the environment, hence
    the subprocesses."""
    import os
    import subprocess
    import sys
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    script = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {here!r})\n"
        "from conftest import setup_case\n"
        "import emu_lib, oracle as O\n"
        "from test_crate_climb import touching_state, EX\n"
        "dc, env, model, task, cfg = setup_case(EX, 12, 5, per_rollout=True)\n"
        "o64 = O.Oracle(model, task, cfg, np.float64); o32 = O.Oracle(model, task, cfg, np.float32)\n"
        "emu = emu_lib.Emu(model, task, cfg)\n"
        "out = []\n"
        "for seed in (0, 3):\n"
        "    q, qd = touching_state(env, o64, seed)\n"
        "    s0, _, _ = o32.env_reset(q, qd)\n"
        "    us = np.random.default_rng(100 + seed).uniform(-1, 1, (12, 6, model.nu)).astype(np.float32)\n"
        "    r = emu.rollout(s0, us, check_races=(seed == 0))\n"
        "    out += [r[0], r[1], r[2]]\n"
        "np.savez(sys.argv[1], *out)\n")
    results = {}
    with tempfile.TemporaryDirectory() as tmp:
        for cap in ("", "16", "2", "1"):
            envv = dict(os.environ)
            envv.pop("DIAL_EMU_CON_CAP", None)
            if cap:
                envv["DIAL_EMU_CON_CAP"] = cap
            path = os.path.join(tmp, f"cap{cap or 'none'}.npz")
            subprocess.check_call([sys.executable, "-c", script, path], env=envv, cwd=here)
            d = np.load(path)
            results[cap] = [d[k] for k in d.files]
    for cap in ("16", "2", "1"):
        for a, b in zip(results[""], results[cap]):
            assert np.array_equal(a, b), f"cap {cap}"
