"""Pins of the CPU oracle: known-answer values and physics invariants.  The oracle has no
reference-generated golden vectors ("parity unpinned" vs JAX) -- these are the pins that exist."""
import numpy as np
import pytest

import oracle as O
from conftest import setup_case


@pytest.fixture(scope="module")
def go2():
    dc, env, model, task, cfg = setup_case("unitree_go2_trot", 64, 8, per_rollout=True)   # fp32-vs-fp64 below is a per-rollout comparison
    return dc, env, model, task, cfg, O.Oracle(model, task, cfg, np.float64)


def test_foot_step_kats(go2):
    """SURVEY B.6; order of the env's feet list is FL, FR, RL, RR with trot phases [0,.5,.5,0]."""
    orc = go2[5]
    table = {0.0: [0, 0.08, 0.08, 0], 0.02: [0, 0.07792095, 0.07792095, 0], 0.10: [0, 0.03323321, 0.03323321, 0],
             0.20: [0.0673003, 0, 0, 0.0673003], 0.26: [0.07947854, 0, 0, 0.07947854]}
    for t, ref in table.items():
        assert np.allclose(orc.foot_step(t), ref, atol=1e-7), t


def test_h1_foot_step_kats():
    dc, env, model, task, cfg = setup_case("unitree_h1_jog", 8, 8)
    orc = O.Oracle(model, task, cfg, np.float64)
    for t, ref in {0.0: [0, 0.2], 0.1: [0.04450419, 0.12469796], 0.2: [0.18019377, 0]}.items():
        assert np.allclose(orc.foot_step(t), ref, atol=1e-7), t


def test_noise_schedule_kat():
    """sigma_control and the second row of the sync-driver factors for unitree_go2_trot."""
    dc = setup_case("unitree_go2_trot", 64, 16)[0]
    sigma = dc.horizon_diffuse_factor ** np.arange(dc.Hnode + 1)[::-1] * dc.sigma_scale
    assert np.allclose(sigma, [0.6561, 0.729, 0.81, 0.9, 1.0])
    assert np.allclose(sigma * dc.traj_diffuse_factor ** 1, [0.32805, 0.3645, 0.405, 0.45, 0.5])


def test_forward_known_answers(go2):
    """SURVEY D 'Go2 known-answer constants' at the home keyframe."""
    dc, env, model, task, cfg, orc = go2
    d = orc.forward_dump(env._init_q, np.zeros(18))
    assert np.allclose(d["qfrc_bias"][:6], [0, 0, 158.984862, 0, 0.116470, 0], atol=1e-5)
    assert np.allclose(d["qfrc_bias"][6:9], [-1.064784, 0.439244, -0.271546], atol=1e-5)
    assert np.allclose(d["con_dist"], -0.013873, atol=1e-6)
    fn = d["efc_force"][12:].reshape(4, 4).sum(1)           # sum of the 4 pyramid edges = normal force
    assert np.allclose(fn, [12.7, 12.7, 12.0, 12.0], atol=0.05)


def test_standing_contact_force_equals_weight(go2):
    """Hold the home joint targets for 150 steps: sum of normal forces -> m g = 158.98 N (SURVEY C.5 probe)."""
    dc, env, model, task, cfg, orc = go2
    jr, home = env.joint_range, env._init_q[7:]
    act = 2 * (home - jr[:, 0]) / (jr[:, 1] - jr[:, 0]) - 1
    state, _, _ = orc.env_reset(env._init_q, np.zeros(18))
    for _ in range(150):
        state, _, _, ctrl = orc.env_step(state, act)
    d = orc.forward_dump(state[:19], state[19:37], ctrl, state[37:55])
    assert abs(d["efc_force"][12:].sum() - 16.206408 * 9.81) < 0.2
    assert np.linalg.norm(state[19:37]) < 0.05
    assert abs(state[2] - 0.168) < 2e-3                      # sags below the env's done height, as probed


def test_free_flight_invariants(go2):
    """No contact (base 1 m up): every body falls with -g, the solver returns the unconstrained solution."""
    dc, env, model, task, cfg, orc = go2
    q = np.array(env._init_q)
    q[2] = 1.0
    d = orc.forward_dump(q, np.zeros(18))
    assert np.all(d["con_dist"] > 0.5) and np.allclose(d["efc_force"], 0)
    g = float(np.float32(9.81))                              # the model blob stores fp32 constants
    assert abs(d["qacc"][2] + g) < 1e-9 and np.allclose(d["qacc"][:2], 0, atol=1e-12)
    assert np.allclose(d["qacc"][6:], 0, atol=1e-9)
    assert np.allclose(d["qacc"], d["qacc_smooth"])          # nothing active => unconstrained solution
    M = d["qM"]
    assert np.allclose(M, M.T) and np.all(np.linalg.eigvalsh(M) > 0)


def test_momentum_and_energy_conservation(go2):
    """Torque-free flight without gravity / damping: linear momentum (M qd)[0:3] is conserved to rounding and
    the kinetic energy to O(dt) (SURVEY C.6b probe), which exercises kinematics, cdof, cdof_dot, CRB and RNE."""
    import copy
    import ctypes
    dc, env, model, task, cfg, _ = go2
    m2, t2 = copy.copy(model), copy.copy(task)
    m2 = type(model).from_buffer_copy(model)
    t2 = type(task).from_buffer_copy(task)
    m2.timestep = 2e-4
    for k in range(3):
        m2.gravity[k] = 0.0
    for i in range(18):
        m2.dof_damping[i] = 0.0
    for a in range(12):
        t2.kp[a] = 0.0
        t2.kd[a] = 0.0
    t2.dt = 2e-4
    orc = O.Oracle(m2, t2, cfg, np.float64)
    rng = np.random.default_rng(7)
    q = np.array(env._init_q)
    q[2] = 2.0
    qd = rng.normal(0, 1.0, 18)
    state, _, _ = orc.env_reset(q, qd)

    def momentum_energy(st):
        d = orc.forward_dump(st[:19], st[19:37])
        v = st[19:37]
        return (d["qM"] @ v)[:3], 0.5 * v @ d["qM"] @ v

    p0, e0 = momentum_energy(state)
    for _ in range(500):
        state, _, _, _ = orc.env_step(state, np.zeros(12))
    p1, e1 = momentum_energy(state)
    assert np.abs(p1 - p0).max() < 1e-4 * max(1.0, np.abs(p0).max())
    assert abs(e1 - e0) / e0 < 2e-2


def test_fp32_matches_fp64(go2):
    dc, env, model, task, cfg, o64 = go2
    o32 = O.Oracle(model, task, cfg, np.float32)
    rng = np.random.default_rng(3)
    us = rng.uniform(-0.6, 0.6, (8, 9, 12))
    s0, _, _ = o64.env_reset(env._init_q, np.zeros(18))
    r64 = o64.rollout(s0, us)
    r32 = o32.rollout(s0.astype(np.float32), us.astype(np.float32))
    # per rollout (measured: 1.4e-6 in the rewards, 3.7e-6 in q over all eight -- no knife edge in this case)
    dr, dq = np.abs(r64[0] - r32[0]).max(1), np.abs(r64[1] - r32[1]).reshape(8, -1).max(1)
    assert dr.max() < 5e-5 and dq.max() < 5e-5, (dr, dq)


def test_reward_lags_action_by_one_step(go2):
    """pose-based reward terms use the PRE-integration forward pass, so the first reward of a
    rollout does not depend on the action at all."""
    dc, env, model, task, cfg, orc = go2
    s0, _, _ = orc.env_reset(env._init_q, np.zeros(18))
    rng = np.random.default_rng(5)
    us = rng.uniform(-1, 1, (4, 9, 12))
    rew = orc.rollout(s0, us)[0]
    assert np.allclose(rew[:, 0], rew[0, 0], atol=1e-12) and np.ptp(rew[:, 1]) > 1e-6


@pytest.mark.parametrize("example,H", [("unitree_h1_jog", 8), ("unitree_h1_loco", 8)])
def test_h1_models_free_flight_and_conservation(example, H):
    """H1 walk / loco (plane-capsule contacts, welded arm links with mesh-inferred inertias): unconstrained free
    fall is -g for every dof, M is SPD, and torque-free flight conserves linear momentum / kinetic energy."""
    dc, env, model, task, cfg = setup_case(example, 8, H)
    nq, nv, nu = model.nq, model.nv, model.nu
    orc = O.Oracle(model, task, cfg, np.float64)
    q = np.array(env._init_q)
    q[2] = 3.0
    d = orc.forward_dump(q, np.zeros(nv))
    assert np.all(d["con_dist"] > 0.5) and np.allclose(d["efc_force"], 0)
    assert abs(d["qacc"][2] + float(np.float32(9.81))) < 1e-9 and np.allclose(np.delete(d["qacc"], 2), 0, atol=1e-8)
    assert np.allclose(d["qM"], d["qM"].T) and np.all(np.linalg.eigvalsh(d["qM"]) > 0)
    total_mass = sum(model.body_mass[b] for b in range(model.nbody))
    assert abs(d["qM"][0, 0] - total_mass) < 1e-4 and abs(d["qM"][2, 2] - total_mass) < 1e-4   # translational block = m I
    m2, t2 = type(model).from_buffer_copy(model), type(task).from_buffer_copy(task)
    m2.timestep = 2e-4
    t2.dt = 2e-4
    for k in range(3):
        m2.gravity[k] = 0.0
    for i in range(nv):
        m2.dof_damping[i] = 0.0
    for a in range(nu):
        t2.kp[a] = 0.0
        t2.kd[a] = 0.0
    o2 = O.Oracle(m2, t2, cfg, np.float64)
    state, _, _ = o2.env_reset(q, np.random.default_rng(11).normal(0, 0.5, nv))

    def momentum_energy(st):
        dd = o2.forward_dump(st[:nq], st[nq:nq + nv])
        v = st[nq:nq + nv]
        return (dd["qM"] @ v)[:3], 0.5 * v @ dd["qM"] @ v

    p0, e0 = momentum_energy(state)
    for _ in range(300):
        state, _, _, _ = o2.env_step(state, np.zeros(nu))
    p1, e1 = momentum_energy(state)
    assert np.abs(p1 - p0).max() < 1e-4 * max(1.0, np.abs(p0).max())
    assert abs(e1 - e0) / e0 < 2e-2


@pytest.mark.parametrize("example,ncon_per_foot", [("unitree_h1_jog", 2), ("unitree_h1_loco", 4)])
def test_h1_contact_impulse_balances_momentum(example, ncon_per_foot):
    """Holding the home pose (the humanoid sways and eventually tips -- only the feet collide): over the first 0.6 s
    the impulse of the contact normal forces minus weight equals the change of the total vertical momentum
    (M qd)[2], and the load is shared evenly by the two feet (left/right mirror symmetry of the model, incl. the
    hull-inferred arm inertias of the loco model).  Exercises the capsule contacts, J^T f and the integrator."""
    dc, env, model, task, cfg = setup_case(example, 8, 8)
    nq, nv, nu, nl = model.nq, model.nv, model.nu, model.nlim
    m2, t2 = type(model).from_buffer_copy(model), type(task).from_buffer_copy(task)
    m2.timestep = t2.dt = 0.005            # the identity holds up to the O(dt) integrator residual
    orc = O.Oracle(m2, t2, cfg, np.float64)
    jr, home = env.joint_range, env._init_q[7:7 + nu]
    act = np.clip(2 * (home - jr[:, 0]) / (jr[:, 1] - jr[:, 0]) - 1, -1, 1)
    state, _, _ = orc.env_reset(env._init_q, np.zeros(nv))
    weight = sum(model.body_mass[b] for b in range(model.nbody)) * float(np.float32(9.81))

    def pz(st):
        return (orc.forward_dump(st[:nq], st[nq:nq + nv])["qM"] @ st[nq:nq + nv])[2]

    p_start, impulse, dt = pz(state), 0.0, m2.timestep
    for t in range(120):
        before = state.copy()
        state, _, _, ctrl = orc.env_step(state, act)
        d = orc.forward_dump(before[:nq], before[nq:nq + nv], ctrl, before[nq + nv:nq + 2 * nv])
        fn = d["efc_force"][nl:].reshape(-1, 4).sum(1)        # normal force per contact in this step
        impulse += dt * (fn.sum() - weight)
        left, right = fn[:ncon_per_foot].sum(), fn[ncon_per_foot:].sum()
        assert abs(left - right) < 0.03 * weight, (t, left, right)
        assert np.all(fn >= -1e-9)
    assert 0.3 * weight < fn.sum() < 3 * weight and state[2] > 0.85          # still on its feet
    assert abs((pz(state) - p_start) - impulse) < 0.01 * weight * 120 * dt
