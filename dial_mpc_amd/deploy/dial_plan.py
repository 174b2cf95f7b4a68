"""Asynchronous planner process (``dial-mpc-plan``): the reference's ``MBDPublisher``
(dial_mpc/deploy/dial_plan.py:64-229) on top of the HIP kernels -- SURVEY 8f NEXT row 1.

Protocol (unchanged): six named ``multiprocessing.shared_memory`` segments created by the plant process
(``dial_sim.py:84-123`` / ``dial_real.py:115-154``), float32 views, no locking:
    time_shm (1)  state_shm (nq+nv)  acts_shm (T, nu)  refs_shm (T, nu, 3)  plan_time_shm (1)  tau_shm (T, nu)
(the reference over-allocates every segment by 8x -- ``size = n * 32`` -- which is reproduced so that either
side can create them).  Each loop: read (t, q, qd) -> inject into the planner state (only qpos/qvel and
info.step = int(t / dt) are replaced, :149-155) -> shift the plan by the elapsed time with the spline
re-evaluated at step_nodes + shift_time (:136-139, right-side extrapolation as FITPACK) -> Ndiffuse
``reverse_once`` iterations with the ASYNC noise schedule traj_diffuse_factor**i (no sigma_control, :207-209)
-> publish joint targets, torques, plan time and body-position references.

The plant side (native-MuJoCo simulator, Unitree DDS bridge) is out of scope; the tests drive this module with a
test double that owns the segments and steps the same HIP env (``tests/fake_plant.py``).
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import time
from multiprocessing import shared_memory
from typing import Optional

import numpy as np

from dial_mpc_amd.core import spline
from dial_mpc_amd.core.dial_core import MBDPI, _generator, load_dial_and_env

SEGMENTS = ("time_shm", "state_shm", "acts_shm", "refs_shm", "plan_time_shm", "tau_shm")


def segment_shapes(nq: int, nv: int, nu: int, n_acts: int):
    return {"time_shm": (1,), "state_shm": (nq + nv,), "acts_shm": (n_acts, nu), "refs_shm": (n_acts, nu, 3),
            "plan_time_shm": (1,), "tau_shm": (n_acts, nu)}


def open_segments(nq, nv, nu, n_acts, create: bool, prefix: str = ""):
    """Attach to (or create) the six segments; returns {name: (SharedMemory, float32 ndarray view)}."""
    out = {}
    for name, shape in segment_shapes(nq, nv, nu, n_acts).items():
        size = int(np.prod(shape)) * 32                      # the reference's 8x over-allocation
        shm = shared_memory.SharedMemory(name=prefix + name, create=create, size=size)
        out[name] = (shm, np.ndarray(shape, dtype=np.float32, buffer=shm.buf))
    return out


class MBDPublisher:
    def __init__(self, env, env_config, dial_config, shm_prefix: str = ""):
        import torch
        self.dial_config, self.env, self.env_config = dial_config, env, env_config
        self.mbdpi = MBDPI(dial_config, env)
        self.rng = _generator(dial_config.seed, self.mbdpi.device)
        self.Y = torch.zeros((dial_config.Hnode + 1, self.mbdpi.nu), dtype=torch.float32, device=self.mbdpi.device)
        self.ctrl_dt = env_config.dt
        self.timer_period = env_config.dt
        self.n_acts = dial_config.Hsample + 1
        self.max_bad_plans = 5      # consecutive non-finite plans after which main_loop raises instead of spinning
        mj = env.sys.mj_model
        self.nq, self.nv, self.nu = mj.nq, mj.nv, mj.nu
        self.nx = mj.nq + mj.nv
        self.default_q = np.asarray(mj.keyframe("home").qpos, dtype=np.float32)
        self.default_u = np.zeros(self.nu, dtype=np.float32)   # the Go2/H1 "home" keyframes carry no ctrl
        seg = open_segments(self.nq, self.nv, self.nu, self.n_acts, create=False, prefix=shm_prefix)
        self._seg = seg
        self.acts_shared = seg["acts_shm"][1]
        self.refs_shared = seg["refs_shm"][1]
        self.plan_time_shared = seg["plan_time_shm"][1]
        self.time_shared = seg["time_shm"][1]
        self.state_shared = seg["state_shm"][1]
        self.tau_shared = seg["tau_shm"][1]
        self.acts_shared[:] = self.default_u
        self.refs_shared[:] = 1.0
        self.plan_time_shared[0] = -0.02
        self.time_shared[0] = 0.0
        self.state_shared[: self.default_q.shape[0]] = self.default_q

    # ---- dial_plan.py:136-139: re-evaluate the node spline at step_nodes + shift_time
    def shift_matrix(self, shift_time: float) -> np.ndarray:
        nodes = self.mbdpi.step_nodes_np
        return spline.interp_matrix(nodes, nodes + shift_time)

    def shift(self, Y, shift_time: float):
        import torch
        A = torch.as_tensor(self.shift_matrix(float(shift_time)), dtype=torch.float32, device=Y.device)
        return A @ Y

    # ---- dial_plan.py:141-155
    def init_mjx_state(self, q, qd, t):
        state = self.env.reset(self.rng)
        return self.update_mjx_state(state, q, qd, t)

    def update_mjx_state(self, state, q, qd, t):
        import torch
        dev = state.packed.device
        state.packed[: self.nq] = torch.as_tensor(np.asarray(q, dtype=np.float32), device=dev)
        state.packed[self.nq: self.nq + self.nv] = torch.as_tensor(np.asarray(qd, dtype=np.float32), device=dev)
        state.info["step"] = int(t / self.ctrl_dt)
        return state

    def plan_once(self, state, n_diffuse: int):
        cfg = self.dial_config
        info = None
        for i in range(n_diffuse):      # async schedule: traj_diffuse_factor**i, shape (1,) (dial_plan.py:207-209)
            self.rng, self.Y, info = self.mbdpi.reverse_once(state, self.rng, self.Y,
                                                             np.array([cfg.traj_diffuse_factor ** i], np.float32),
                                                             want_bars=(i == n_diffuse - 1))
        return info

    def main_loop(self, max_ticks: Optional[int] = None, sleep_when_idle: float = 0.0, on_tick=None):
        """on_tick(tick): called after every published plan (tests use it to advance the plant inside ONE loop)."""
        import torch
        last_plan_time = float(self.time_shared[0])
        state = self.init_mjx_state(self.state_shared[: self.nq].copy(), self.state_shared[self.nq:].copy(),
                                    last_plan_time)
        first_time, ticks, latencies, bad_plans = True, 0, [], 0
        while max_ticks is None or ticks < max_ticks:
            t0 = time.time()
            plan_time = float(self.time_shared[0])
            state = self.update_mjx_state(state, self.state_shared[: self.nq].copy(),
                                          self.state_shared[self.nq:].copy(), plan_time)
            shift_time = plan_time - last_plan_time
            if shift_time > self.ctrl_dt + 1e-3:
                print(f"[WRAN] sim overtime {(shift_time - self.ctrl_dt) * 1000:.1f} ms")
            if shift_time > self.ctrl_dt * self.n_acts:
                print(f"[WARN] long time unplanned {shift_time * 1000:.1f} ms, reset control")
                self.Y = torch.zeros_like(self.Y)          # (not Y * 0: NaN * 0 is NaN)
            else:
                self.Y = self.shift(self.Y, shift_time)
            if first_time:
                print("Performing the initial diffusion on DIAL-MPC")
                self.plan_once(state, self.dial_config.Ndiffuse_init)
                first_time = False
            info = self.plan_once(state, self.dial_config.Ndiffuse)
            x_targets = info["xbar"]                                   # (T, nbody-1, 3)
            us = self.mbdpi.node2u_vmap(self.Y)                        # plan -> controls
            us_np = us.cpu().numpy()                                  # (synchronises with the planner's stream)
            xt = x_targets.cpu().numpy()[:, 1:, :3]                    # drop the root body, as the reference does
            # nothing non-finite ever reaches the robot: a launch that gave up (dial_status) raises, a NaN plan (the
            # reference's 0 / 0 when every sample earns the same reward, dial_core.py:126) is dropped -- the plant keeps
            # executing the previous plan, plan_time_shm is not advanced, the plan restarts from zero
            self.mbdpi.ctx.status()
            if not (np.isfinite(us_np).all() and np.isfinite(xt).all()):
                bad_plans += 1
                print(f"[ERROR] non-finite plan ({bad_plans} in a row), not published; control reset")
                if bad_plans >= self.max_bad_plans:
                    raise RuntimeError(f"{bad_plans} consecutive non-finite plans: the planner cannot recover from this state")
                self.Y = torch.zeros_like(self.Y)          # a fresh plan (Y * 0 would keep the NaNs: NaN * 0 = NaN)
                last_plan_time = plan_time
                if on_tick is not None:
                    on_tick(ticks)
                ticks += 1
                if sleep_when_idle:
                    time.sleep(sleep_when_idle)
                continue
            bad_plans = 0
            joint_targets = np.stack([self.env.act2joint(u) for u in us_np])
            ps = state.pipeline_state
            taus = np.stack([self.env.act2tau(u, ps) for u in us_np])
            self.acts_shared[: joint_targets.shape[0], :] = joint_targets
            self.tau_shared[: taus.shape[0], :] = taus
            self.plan_time_shared[0] = plan_time
            n = min(self.refs_shared.shape[1], xt.shape[1])
            self.refs_shared[:, :n, :] = xt[: self.refs_shared.shape[0], :n, :]
            last_plan_time = plan_time
            torch.cuda.synchronize()
            dt_wall = time.time() - t0
            latencies.append(dt_wall)
            if dt_wall > self.ctrl_dt:
                print(f"[WRAN] real overtime {dt_wall * 1000:.1f} ms")
            if on_tick is not None:
                on_tick(ticks)
            ticks += 1
            if sleep_when_idle:
                time.sleep(sleep_when_idle)
        return latencies

    def close(self):
        for shm, _ in self._seg.values():
            shm.close()


def main(args=None):
    import yaml
    from dial_mpc_amd.examples import deploy_examples
    from dial_mpc_amd.utils.io_utils import get_example_path
    parser = argparse.ArgumentParser()
    group = parser.add_mutually_exclusive_group(required=True)
    group.add_argument("--config", type=str, default=None, help="Path to config file")
    group.add_argument("--example", type=str, default=None, help="Example to run")
    group.add_argument("--list-examples", action="store_true", help="List available examples")
    parser.add_argument("--custom-env", type=str, default=None, help="Custom environment to import dynamically")
    args = parser.parse_args(args)
    if args.custom_env is not None:
        sys.path.append(os.getcwd())
        importlib.import_module(args.custom_env)
    if args.list_examples:
        print("Available examples:")
        for example in deploy_examples:
            print(f"  - {example}")
        return
    if args.example is not None:
        if args.example not in deploy_examples:
            print(f"Example {args.example} not found.")
            return
        config_dict = yaml.safe_load(open(get_example_path(args.example + ".yaml"), "r"))
    else:
        config_dict = yaml.safe_load(open(args.config, "r"))
    print("Creating environment")
    dial_config, env_config, env = load_dial_and_env(config_dict)
    pub = MBDPublisher(env, env_config, dial_config)
    try:
        pub.main_loop()
    except KeyboardInterrupt:
        pass
    finally:
        pub.close()


if __name__ == "__main__":
    main()
