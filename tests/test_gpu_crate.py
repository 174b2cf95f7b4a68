"""Crate climb on the GPU: libdialhip.so's generic instantiation (52 candidate contacts, 220 constraint rows, box narrow
phases) vs the fp32 oracle -- from poses that touch the crate with spheres, capsules and the trunk box, at the example's
full size (N = 2048, H = 25), under both line-search rules (per rollout under SWAP, distribution level under the default)."""
import numpy as np
import pytest

from conftest import (TOL, agg_tol, distribution_parity, seeded_inputs, setup_case, transition_parity, transition_sample,
                      witness_parity)
from test_crate_climb import EX, _quat, touching_state

pytestmark = pytest.mark.gpu
# transitions of the crate scenes that may stay without a witness: measured on MI355X 0 of 2496 / 2400 from the home pose and a
# perturbed one (profiles/r04_transition_parity.txt), 1 of 2496 from the pose standing ON the crate (a calf capsule within
# micrometres of its radius next to the crate's edge: the contact normal turns by degrees per micrometre, DESIGN.md "crate scenes")
CRATE_UNWITNESSED_TRANSITIONS = 2


def _dev(x):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device="cuda")


def _close(a, b, tol):
    return np.allclose(a, b, rtol=tol["rtol"], atol=tol["atol"])


def _poses(env, o64):
    """home (in front of the crate), standing on the crate, belly across the crate's front edge, six random touching poses"""
    nv = env.sys.nv
    out = [(np.array(env._init_q, dtype=np.float64), np.zeros(nv))]
    q = np.array(env._init_q, dtype=np.float64)
    q[0:3] = [1.3, 0.0, 0.87]
    out.append((q, np.zeros(nv)))
    q = np.array(env._init_q, dtype=np.float64)
    q[3:7] = _quat(0.0, -0.5, 0.0)
    q[7:] = [0.0, 1.4, -0.9, 0.0, 1.4, -0.9, 0.0, 2.5, -0.9, 0.0, 2.5, -0.9]
    Rp = np.array([[np.cos(-0.5), 0, np.sin(-0.5)], [0, 1, 0], [-np.sin(-0.5), 0, np.cos(-0.5)]])
    q[0:3] = np.array([0.99, 0.0, 0.6]) + Rp @ np.array([0.0, 0.0, 0.057 - 0.003])
    out.append((q, np.zeros(nv)))
    out += [touching_state(env, o64, seed) for seed in range(6)]
    return out


def test_crate_context_runs_on_the_generic_instantiation():
    from dial_mpc_amd import _lib
    dc, env, model, task, cfg = setup_case(EX, 64, 8)
    ctx = _lib.Context(model, task, cfg)
    ctx.status()                      # raises on a sticky error
    assert model.nefc == 220 and model.ncon == 52


def test_box_contacts_with_a_wide_margin_are_rejected():
    """The box narrow phases park candidates that are provably more than 1 cm apart with a placeholder distance / frame
    (csrc/box_collide.h); a box contact whose margin reaches that distance would be activated on those values: dial_create refuses it."""
    import copy
    from dial_mpc_amd import _lib
    dc, env, model, task, cfg = setup_case(EX, 64, 8)
    m2 = copy.deepcopy(model)
    c = next(c for c in range(m2.ncon) if m2.con_kind[c] >= 5)
    m2.con_margin[c] = 0.02
    with pytest.raises(_lib.DialHipError, match="margin"):
        _lib.Context(m2, task, cfg)


def test_crate_env_step_and_rollouts_match_oracle():
    import oracle as O
    from dial_mpc_amd import _lib
    H = 8
    dc, env, model, task, cfg = setup_case(EX, 64, H, per_rollout=True)
    o32, o64 = O.Oracle(model, task, cfg, np.float32), O.Oracle(model, task, cfg, np.float64)
    ctx = _lib.Context(model, task, cfg)
    rng = np.random.default_rng(4)
    nqv = model.nq + model.nv
    for q, qd in _poses(env, o64):
        s0, xp_o, xq_o = o32.env_reset(q, qd)
        s_g, xp_g, xq_g = ctx.env_reset(_dev(q), _dev(qd))
        s_g = s_g.cpu().numpy()
        # (the GPU gate of tests/test_gpu_parity.py: qacc_warmstart is a difference of forces, its error scales with the
        #  largest acceleration)
        atol = np.full(s0.shape, 5e-4)
        atol[nqv:nqv + model.nv] = 5e-4 * max(1.0, float(np.abs(s0[nqv:nqv + model.nv]).max()) * 1e-2)
        assert np.all(np.abs(s0 - s_g) <= atol + 2e-4 * np.abs(s0)), np.abs(s0 - s_g).max()
        us = rng.uniform(-1.0, 1.0, (16, H + 1, model.nu)).astype(np.float32)
        r_g = [t.cpu().numpy() for t in ctx.rollout(_dev(s0), _dev(us))]
        witness_parity(o32, s0, us, r_g, EX, model.nq + 2 * model.nv)


@pytest.mark.parametrize("pose", [0, 1, 4])
def test_crate_full_size_oracle_parity(pose):
    """The example's own size (N = 2048, H = 25, Hnode = 5): every per-step reward, q, qd, x.pos of the 2049 rollouts,
    then the product outputs."""
    import oracle as O
    from dial_mpc_amd import _lib
    N, H = 2048, 25
    dc, env, model, task, cfg = setup_case(EX, N, H, per_rollout=True)
    o32, o64 = O.Oracle(model, task, cfg, np.float32), O.Oracle(model, task, cfg, np.float64)
    ctx = _lib.Context(model, task, cfg)
    q, qd = _poses(env, o64)[pose]
    s0, _, _ = o32.env_reset(q, qd)
    eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=pose, Ybar_scale=0.2)
    ro = o32.reverse_once(s0, Ybar, sigma, eps, full=True)
    out = ctx.reverse_once(_dev(s0), _dev(Ybar), _dev(sigma), _dev(eps))
    sc = ctx.debug_scratch()
    got = (sc["rewss"], sc["qss"], sc["qdss"], sc["xss"])
    # 26 steps on 52 candidate contacts with a truncated solver: by the time a rollout meets a knife edge the GPU's state may
    # differ from the oracle's by up to TOL, out of reach of the 64-ulp jitter (restart_ok: see witness_parity); and a capsule
    # sunk to within micrometres of its radius next to one of the crate's edges has a contact normal that turns by degrees
    # per micrometre -- the same pose through the kernel's fp32 code and the oracle's differs beyond TOL after one step
    # (measured: 2-6 of 2049 rollouts, IEEE build and fast-math build alike).  Up to 12 (0.6 %) are allowed, everything else is not.
    rep = witness_parity(o32, s0, ro["us"], got, EX, model.nq + 2 * model.nv, unwitnessed_ok=12, restart_ok=True)
    print(f"{EX} pose {pose}: {rep['outside_tol']} of {rep['rollouts']} rollouts on a knife edge, "
          f"{rep.get('restart_witnessed', 0)} witnessed from the GPU's own state, {rep.get('unwitnessed', 0)} without a witness")
    # product outputs against the oracle's own <= 1 ulp jitter envelope (the knife-edge rollouts carry arbitrary weight)
    prod = {k: out[k].cpu().numpy() for k in ("Ybar", "qbar", "qdbar", "xbar")}
    drep = distribution_parity(o32, s0, ro["us"], sc["Y0s"], got, prod, cfg.temp_sample)
    print(f"   distribution level: GPU {drep['gpu']}\n   jitter envelope: {drep['envelope']}")
    rews_g = out["rews"].cpu().numpy().astype(np.float64)
    logp = (rews_g - rews_g[-1]) / rews_g.std() / float(cfg.temp_sample)
    w_ref = np.exp(logp - logp.max())
    w_ref /= w_ref.sum()
    assert np.allclose(sc["weights"], w_ref, rtol=5e-3, atol=1e-7)
    assert np.allclose(out["Ybar"].cpu().numpy(), np.einsum("n,nka->ka", w_ref, sc["Y0s"].astype(np.float64)), atol=1e-4)


@pytest.mark.parametrize("pose", [0, 1])
def test_crate_default_rule_distribution_parity(pose):
    """The shipped model (line-search rule `_in_bracket`) at the example's size, bounded at the distribution level like the
    other envs (conftest.distribution_parity)."""
    import oracle as O
    from dial_mpc_amd import _lib
    N, H = 2048, 25
    dc, env, model, task, cfg = setup_case(EX, N, H)
    assert model.ls_rule == 1
    o32, o64 = O.Oracle(model, task, cfg, np.float32), O.Oracle(model, task, cfg, np.float64)
    ctx = _lib.Context(model, task, cfg)
    trace_dev = ctx.set_state_trace(N + 1)
    q, qd = _poses(env, o64)[pose]
    s0, _, _ = o32.env_reset(q, qd)
    eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=pose, Ybar_scale=0.2)
    out = ctx.reverse_once(_dev(s0), _dev(Ybar), _dev(sigma), _dev(eps))
    sc = ctx.debug_scratch()
    W = np.array([[cfg.W[t][k] for k in range(dc.Hnode + 1)] for t in range(H + 1)], np.float32)
    us = np.einsum("tk,nka->nta", W, sc["Y0s"]).astype(np.float32)
    prod = {k: out[k].cpu().numpy() for k in ("Ybar", "qbar", "qdbar", "xbar")}
    got = (sc["rewss"], sc["qss"], sc["qdss"], sc["xss"])
    # per transition, deterministic: the oracle restarted from the device's OWN traced state (q, qd, qacc_warmstart, info) after
    # step t reproduces the device's step t + 1 at 1 x TOL; knife edges need a <= 64 ulp witness (conftest.transition_parity)
    trep = transition_parity(o32, s0, us, got, trace_dev.cpu().numpy(), transition_sample(N, 96, pose), model.nq, model.nv,
                             example=EX, unwitnessed_ok=CRATE_UNWITNESSED_TRANSITIONS)
    print(f"{EX} shipped rule, per transition: {trep['transitions']} transitions, direct {100 * trep['direct_share']:.2f} % "
          f"(worst {trep['direct_worst']:.2f} x gate), witnessed {trep['witnessed']} {trep['witness_ulp']}, unwitnessed {trep['unwitnessed']}")
    rep = distribution_parity(o32, s0, us, sc["Y0s"], got, prod, cfg.temp_sample)
    print(f"{EX} pose {pose} default rule: ESS oracle {rep['ess_oracle']:.1f} / GPU {rep['ess_gpu']:.1f}\n"
          f"   GPU vs oracle   {rep['gpu']}\n   jitter envelope {rep['envelope']}")


def test_crate_closed_loop_runs_and_approaches_the_crate():
    """The reference's main loop (dial_core.py:242-268) on the crate example, 50 control ticks at N = 1024: finite plans, no
    sticky error, and the head moves towards its target on the crate (the reward's dominant term)."""
    import torch
    import yaml
    from dial_mpc_amd.core.dial_core import MBDPI, load_dial_and_env
    from dial_mpc_amd.utils.io_utils import get_example_path
    d = yaml.safe_load(open(get_example_path(EX + ".yaml")))
    d["Nsample"] = 1024
    dial_config, env_config, env = load_dial_and_env(d)
    mbdpi = MBDPI(dial_config, env)
    state = env.reset(0)
    Y0 = torch.zeros((dial_config.Hnode + 1, mbdpi.nu), device=mbdpi.device)
    rng, rews, xs = 0, [], []
    for t in range(50):
        state = env.step(state, Y0[0])
        rews.append(float(state.reward))
        xs.append(float(state.pipeline_state.qpos[0]))
        Y0 = mbdpi.shift(Y0)
        n_diffuse = dial_config.Ndiffuse_init if t == 0 else dial_config.Ndiffuse
        for i in range(n_diffuse):
            rng, Y0, info = mbdpi.reverse_once(state, rng, Y0, mbdpi.sigma_control * dial_config.traj_diffuse_factor ** i)
        assert torch.isfinite(Y0).all()
    torch.cuda.synchronize()
    mbdpi.ctx.status()                # raises on a sticky error
    print(f"crate closed loop: base x {xs[0]:.3f} -> {xs[-1]:.3f}, reward {rews[1]:.3f} -> {rews[-1]:.3f}")
    assert np.all(np.isfinite(rews))
    assert xs[-1] > xs[0] + 0.1 and rews[-1] > rews[1]


def test_crate_overflow_path_on_the_gpu_is_bit_identical():
    """The rollout kernel's LDS workspace holds a capped number of touching contacts; a sample with more runs the second compiled
    copy of the constraint code on its overflow area in global memory.  dial_options.con_cap = 1 sends every touching step
    down that path, < 0 switches the cap off (full-size LDS workspace, capacity-dimension kernel only).  Within one kernel
    instantiation all caps must agree bit for bit; the scene's own instantiation (compile-time dimensions, dof-tree
    factorisation) and the capacity-dimension one (dense factorisation) agree to rounding."""
    import oracle as O
    from dial_mpc_amd import _lib
    N, H = 256, 12
    dc, env, model, task, cfg = setup_case(EX, N, H)
    o32, o64 = O.Oracle(model, task, cfg, np.float32), O.Oracle(model, task, cfg, np.float64)
    q, qd = _poses(env, o64)[5]
    s0, _, _ = o32.env_reset(q, qd)
    eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=3, Ybar_scale=0.2)
    outs = {}
    for generic, caps in ((0, (0, 14, 1)), (1, (0, 14, 1, -1))):
        for cap in caps:
            ctx = _lib.Context(model, task, cfg, options=dict(con_cap=cap, force_generic=generic))
            lds = ctx.lib.dial_lds_bytes(ctx.h)
            out = ctx.reverse_once(_dev(s0), _dev(Ybar), _dev(sigma), _dev(eps))
            sc = ctx.debug_scratch()
            outs[(generic, cap)] = (lds, out["Ybar"].cpu().numpy(), out["rews"].cpu().numpy(), sc["qss"].copy(), sc["qdss"].copy())
            del ctx
    assert outs[(0, 1)][0] < outs[(0, 14)][0] < outs[(0, 0)][0]          # three different LDS footprints (nine wavefronts each)
    assert outs[(1, 1)][0] < outs[(1, 14)][0] < outs[(1, -1)][0]
    for generic, caps in ((0, (14, 1)), (1, (14, 1, -1))):
        for cap in caps:
            for a, b in zip(outs[(generic, 0)][1:], outs[(generic, cap)][1:]):
                assert np.array_equal(a, b), f"force_generic={generic} con_cap={cap}"
    # the two instantiations against each other: same physics, different summation / elimination orders -- compared over the
    # FIRST two steps only (under the shipped truncated rule whole rollouts part at the first knife edge)
    same = np.abs(outs[(0, 0)][3][:, :2] - outs[(1, 0)][3][:, :2]).reshape(N + 1, -1).max(1) <= TOL["q"]["atol"]
    assert same.mean() > 0.8, same.mean()


def test_crate_overflow_path_under_the_relay_at_full_size():
    """at the example's N = 2048 the mean trajectory runs as relay pieces, i.e. the grid holds MORE wavefronts
    (N + pieces) than rollouts (N + 1), and every wavefront slot owns an overflow area.  con_cap = 1 makes every touching step
    of every wavefront -- the relay pieces included -- use its area: results must equal the default cap's bit for bit, and
    a batch beyond the context's capacity must be refused instead of overrunning the areas."""
    import torch
    from dial_mpc_amd import _lib
    N, H = 2048, 25
    dc, env, model, task, cfg = setup_case(EX, N, H)
    outs = {}
    for cap in (0, 1):
        ctx = _lib.Context(model, task, cfg, options=dict(con_cap=cap))
        assert ctx.lib.dial_debug_resident_rollouts(ctx.h, N + 1) >= N + 1 + (H + 3) // 3      # everything resident: the relay runs
        s0, _, _ = ctx.env_reset(_dev(env._init_q), _dev(np.zeros(model.nv)))
        eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=5, Ybar_scale=0.2)
        out = ctx.reverse_once(s0, _dev(Ybar), _dev(sigma), _dev(eps))
        torch.cuda.synchronize()
        ctx.status()
        sc = ctx.debug_scratch()
        outs[cap] = (out["Ybar"].cpu().numpy(), out["rews"].cpu().numpy(), sc["qss"].copy(), sc["rewss"].copy())
        if cap == 1:
            us = torch.zeros((N + H + 8, H + 1, model.nu), device="cuda")
            with pytest.raises(_lib.DialHipError, match="overflow areas"):
                ctx.rollout(s0, us)
        del ctx
    assert np.all(np.isfinite(outs[0][1]))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
