"""Push crate on the GPU: libdialhip.so's generic instantiation (28 candidate contacts, a dry-friction row, contacts between
two moving bodies) vs the fp32 oracle, at small size from pushing poses and at the example's full size (N = 2048, H = 24)."""
import numpy as np
import pytest

from conftest import distribution_parity, seeded_inputs, setup_case, transition_parity, transition_sample, witness_parity
from test_push_crate import EX, pushing_state

pytestmark = pytest.mark.gpu
# transitions of the crate scenes that may stay without a witness: measured on MI355X 0 of 2496 / 2400 from the home pose and a
# perturbed one (profiles/r04_transition_parity.txt), 1 of 2496 from the pose standing ON the crate (a calf capsule within
# micrometres of its radius next to the crate's edge: the contact normal turns by degrees per micrometre, DESIGN.md "crate scenes")
CRATE_UNWITNESSED_TRANSITIONS = 2


def _dev(x):
    import torch
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device="cuda")


def _poses(env, o64):
    return [(np.array(env._init_q, dtype=np.float64), np.zeros(env.sys.nv))] + [pushing_state(env, o64, seed) for seed in range(4)]


def test_push_crate_env_reset_and_rollouts_match_oracle():
    import oracle as O
    from dial_mpc_amd import _lib
    H = 8
    dc, env, model, task, cfg = setup_case(EX, 64, H, per_rollout=True)
    o32, o64 = O.Oracle(model, task, cfg, np.float32), O.Oracle(model, task, cfg, np.float64)
    ctx = _lib.Context(model, task, cfg)
    rng = np.random.default_rng(4)
    nqv = model.nq + model.nv
    moved = 0.0
    for q, qd in _poses(env, o64):
        s0, _, _ = o32.env_reset(q, qd)
        s_g, _, _ = ctx.env_reset(_dev(q), _dev(qd))
        s_g = s_g.cpu().numpy()
        atol = np.full(s0.shape, 5e-4)
        atol[nqv:nqv + model.nv] = 5e-4 * max(1.0, float(np.abs(s0[nqv:nqv + model.nv]).max()) * 1e-2)
        assert np.all(np.abs(s0 - s_g) <= atol + 2e-4 * np.abs(s0)), np.abs(s0 - s_g).max()
        us = rng.uniform(-1.0, 1.0, (16, H + 1, model.nu)).astype(np.float32)
        r_g = [t.cpu().numpy() for t in ctx.rollout(_dev(s0), _dev(us))]
        witness_parity(o32, s0, us, r_g, EX, model.nq + 2 * model.nv)
        moved = max(moved, float(np.abs(r_g[2][:, :, 25]).max()))
    assert moved > 1e-3            # the crate slid in some rollout: the friction row left its quadratic zone


@pytest.mark.parametrize("pose", [0, 2])
def test_push_crate_full_size_oracle_parity(pose):
    import oracle as O
    from dial_mpc_amd import _lib
    N, H = 2048, 24
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
    rep = witness_parity(o32, s0, ro["us"], got, EX, model.nq + 2 * model.nv, unwitnessed_ok=12, restart_ok=True)
    print(f"{EX} pose {pose}: {rep['outside_tol']} of {rep['rollouts']} rollouts on a knife edge, "
          f"{rep.get('restart_witnessed', 0)} witnessed from the GPU's own state, {rep.get('unwitnessed', 0)} without a witness")
    prod = {k: out[k].cpu().numpy() for k in ("Ybar", "qbar", "qdbar", "xbar")}
    drep = distribution_parity(o32, s0, ro["us"], sc["Y0s"], got, prod, cfg.temp_sample)
    print(f"   distribution level: GPU {drep['gpu']}\n   jitter envelope: {drep['envelope']}")
    rews_g = out["rews"].cpu().numpy().astype(np.float64)
    logp = (rews_g - rews_g[-1]) / rews_g.std() / float(cfg.temp_sample)
    w_ref = np.exp(logp - logp.max())
    w_ref /= w_ref.sum()
    assert np.allclose(sc["weights"], w_ref, rtol=5e-3, atol=1e-7)
    assert np.allclose(out["Ybar"].cpu().numpy(), np.einsum("n,nka->ka", w_ref, sc["Y0s"].astype(np.float64)), atol=1e-4)


def test_push_crate_default_rule_distribution_parity():
    import oracle as O
    from dial_mpc_amd import _lib
    N, H = 2048, 24
    dc, env, model, task, cfg = setup_case(EX, N, H)
    assert model.ls_rule == 1
    o32, o64 = O.Oracle(model, task, cfg, np.float32), O.Oracle(model, task, cfg, np.float64)
    ctx = _lib.Context(model, task, cfg)
    trace_dev = ctx.set_state_trace(N + 1)
    q, qd = _poses(env, o64)[0]
    s0, _, _ = o32.env_reset(q, qd)
    eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=0, Ybar_scale=0.2)
    out = ctx.reverse_once(_dev(s0), _dev(Ybar), _dev(sigma), _dev(eps))
    sc = ctx.debug_scratch()
    W = np.array([[cfg.W[t][k] for k in range(dc.Hnode + 1)] for t in range(H + 1)], np.float32)
    us = np.einsum("tk,nka->nta", W, sc["Y0s"]).astype(np.float32)
    prod = {k: out[k].cpu().numpy() for k in ("Ybar", "qbar", "qdbar", "xbar")}
    got = (sc["rewss"], sc["qss"], sc["qdss"], sc["xss"])
    # per transition, deterministic: the oracle restarted from the device's OWN traced state (q, qd, qacc_warmstart, info) after
    # step t reproduces the device's step t + 1 at 1 x TOL; knife edges need a <= 64 ulp witness (conftest.transition_parity)
    trep = transition_parity(o32, s0, us, got, trace_dev.cpu().numpy(), transition_sample(N, 96, 0), model.nq, model.nv,
                             example=EX, unwitnessed_ok=CRATE_UNWITNESSED_TRANSITIONS)
    print(f"{EX} shipped rule, per transition: {trep['transitions']} transitions, direct {100 * trep['direct_share']:.2f} % "
          f"(worst {trep['direct_worst']:.2f} x gate), witnessed {trep['witnessed']} {trep['witness_ulp']}, unwitnessed {trep['unwitnessed']}")
    rep = distribution_parity(o32, s0, us, sc["Y0s"], got, prod, cfg.temp_sample)
    print(f"{EX} default rule: ESS oracle {rep['ess_oracle']:.1f} / GPU {rep['ess_gpu']:.1f}\n   GPU vs oracle   {rep['gpu']}\n   jitter envelope {rep['envelope']}")


def test_push_crate_closed_loop_walks_up_to_the_crate_and_moves_it():
    """The reference's main loop on the push-crate example (N = 1024, 100 control ticks): finite plans, no sticky error, the
    robot stays up, and the crate ends up further away than it started (the task: walk forward at 0.8 m/s behind a crate)."""
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
    rng, crate, z = 0, [], []
    for t in range(100):
        state = env.step(state, Y0[0])
        qpos = state.pipeline_state.qpos
        crate.append(float(qpos[26]))
        z.append(float(qpos[2]))
        Y0 = mbdpi.shift(Y0)
        n_diffuse = dial_config.Ndiffuse_init if t == 0 else dial_config.Ndiffuse
        for i in range(n_diffuse):
            rng, Y0, info = mbdpi.reverse_once(state, rng, Y0, mbdpi.sigma_control * dial_config.traj_diffuse_factor ** i)
        assert torch.isfinite(Y0).all()
    torch.cuda.synchronize()
    mbdpi.ctx.status()
    print(f"push crate closed loop: crate x {crate[0]:.3f} -> {crate[-1]:.3f}, pelvis z {z[0]:.3f} -> {min(z):.3f} (min) -> {z[-1]:.3f}")
    assert min(z) > 0.6                      # the robot did not fall
    assert crate[-1] > crate[0] + 0.05       # and pushed the crate forward
