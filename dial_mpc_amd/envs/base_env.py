"""``BaseEnv`` -- the reference's env base class surface (dial_mpc/envs/base_env.py:14-66) on top of
the compiled model (`dial_model`), the task description (`dial_task`) and the HIP library.

What the reference inherits from ``brax.envs.base.PipelineEnv`` (``sys``, ``dt``, ``action_size``,
``pipeline_init`` / ``pipeline_step``) is provided here directly; the physics itself runs in
``libdialhip.so`` (``dial_env_step`` / ``dial_env_reset``), never in Python.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any, Dict, Optional

import numpy as np

from dial_mpc_amd import _abi, mjcf
from dial_mpc_amd.config.base_env_config import BaseEnvConfig
from dial_mpc_amd.utils.io_utils import get_model_path

M = _abi.MACROS


def load_model(robot_name: str, model_name: str) -> Dict[str, Any]:
    """Load ``<models>/<robot>/<model_name>``: the MJCF is compiled when it exists, otherwise the
    pre-compiled constants ``<model_name minus .xml>.json`` shipped with the package are used."""
    path = str(get_model_path(robot_name, model_name))
    if os.path.exists(path) and path.endswith(".xml"):
        return mjcf.compile_mjcf(path)
    jpath = os.path.splitext(path)[0] + ".json"
    if not os.path.exists(jpath):
        raise FileNotFoundError(f"neither {path} nor {jpath} exists")
    return mjcf.model_from_json(open(jpath).read())


class _Keyframe(SimpleNamespace):
    pass


class _MjModelView:
    """The few ``mujoco.MjModel`` attributes the reference's drivers read (dial_plan.py:86-89)."""

    def __init__(self, model: Dict[str, Any]):
        self._m = model
        self.nq, self.nv, self.nu = int(model["nq"]), int(model["nv"]), int(model["nu"])
        self.nbody = int(model["nbody"])

    def keyframe(self, name: str):
        return _Keyframe(qpos=np.array(self._m["keyframes"][name], dtype=np.float64))

    def body_id(self, name: str) -> int:
        return self._m["names"]["body"].index(name)

    def site_id(self, name: str) -> int:
        return self._m["names"]["site"].index(name)


class System:
    """Stand-in for ``brax.base.System``: the attributes the reference touches."""

    def __init__(self, model: Dict[str, Any]):
        self.model = model
        self.mj_model = _MjModelView(model)
        self.nq, self.nv, self.nu = int(model["nq"]), int(model["nv"]), int(model["nu"])
        self.jnt_range = np.asarray(model["jnt_range"], dtype=np.float64)
        self.actuator_ctrlrange = np.asarray(model["act_ctrlrange"], dtype=np.float64)
        self.opt = SimpleNamespace(timestep=float(model["timestep"]))

    def tree_replace(self, params: Dict[str, Any]) -> "System":
        model = dict(self.model)
        for k, v in params.items():
            if k == "opt.timestep":
                model["timestep"] = float(v)
            else:
                raise KeyError(f"tree_replace: unsupported key {k!r}")
        return System(model)


class BaseEnv:
    task_kind: int = -1

    def __init__(self, config: BaseEnvConfig):
        assert np.allclose(config.dt % config.timestep, 0.0), "timestep must be divisible by dt"
        self._config = config
        self._n_frames = int(config.dt / config.timestep)
        self.sys = self.make_system(config)
        self.backend = config.backend
        self._debug = config.debug

        # joint limit definitions (base_env.py:22-25)
        self.physical_joint_range = self.sys.jnt_range[1:]
        self.joint_range = self.physical_joint_range
        self.joint_torque_range = self.sys.actuator_ctrlrange

        self._nv = self.sys.nv
        self._nq = self.sys.nq
        self._ctx = None  # HIP context for env.step / env.reset, created on first use
        self._cmd_table = None   # randomize_tasks: (n_cmd, 3) commands (vx, vy, vyaw), see command_table()

    # ---- reference surface
    def make_system(self, config: BaseEnvConfig) -> System:
        raise NotImplementedError

    @property
    def dt(self) -> float:
        return self.sys.opt.timestep * self._n_frames

    @property
    def action_size(self) -> int:
        return self.sys.nu

    def act2joint(self, act):
        """base_env.py:38-50 (host NumPy; the in-kernel copy lives in csrc/rollout_body.h)."""
        act = np.asarray(_to_numpy(act), dtype=np.float32)
        jr = np.asarray(self.joint_range, dtype=np.float32)
        pr = np.asarray(self.physical_joint_range, dtype=np.float32)
        act_normalized = (act * np.float32(self._config.action_scale) + np.float32(1.0)) / np.float32(2.0)
        joint_targets = jr[:, 0] + act_normalized * (jr[:, 1] - jr[:, 0])
        return np.clip(joint_targets, pr[: jr.shape[0], 0], pr[: jr.shape[0], 1])

    def act2tau(self, act, pipeline_state):
        """base_env.py:53-66."""
        joint_target = self.act2joint(act)
        n = joint_target.shape[-1]
        q = np.asarray(_to_numpy(pipeline_state.qpos), dtype=np.float32)[7:][:n]
        qd = np.asarray(_to_numpy(pipeline_state.qvel), dtype=np.float32)[6:][:n]
        kp = np.asarray(self._config.kp, dtype=np.float32)
        kd = np.asarray(self._config.kd, dtype=np.float32)
        tau = kp * (joint_target - q) - kd * qd
        tr = np.asarray(self.joint_torque_range, dtype=np.float32)
        return np.clip(tau, tr[:, 0], tr[:, 1])

    # ---- task description consumed by the kernels / the oracle
    def task_dict(self) -> Dict[str, Any]:
        cfg = self._config
        nu = self.sys.nu
        if cfg.leg_control not in ("torque", "position"):
            raise ValueError("Invalid leg control type.")
        jr = np.asarray(self.joint_range, dtype=np.float64)
        return dict(
            kind=self.task_kind, n_frames=self._n_frames,
            position_control=int(cfg.leg_control == "position"),
            dt=self.dt, action_scale=cfg.action_scale,
            kp=np.broadcast_to(np.asarray(cfg.kp, dtype=np.float64), (nu,)),
            kd=np.broadcast_to(np.asarray(cfg.kd, dtype=np.float64), (nu,)),
            joint_range=jr, phys_range=np.asarray(self.physical_joint_range)[:nu],
            tau_range=np.asarray(self.joint_torque_range),
            **self._randomize_dict(),
        )

    def _randomize_dict(self) -> Dict[str, Any]:
        if not getattr(self._config, "randomize_tasks", False):
            return dict(randomize_tasks=0, n_cmd=0)
        tab = self.command_table()
        full = np.zeros((int(M["DIAL_MAX_CMD"]), 3))
        full[: tab.shape[0]] = tab
        return dict(randomize_tasks=1, n_cmd=int(tab.shape[0]), cmd_table=full)

    # ---- randomize_tasks (unitree_go2_env.py:142-155, :298-325; unitree_h1_env.py:199-212, :885-902)
    # sample_command's ranges: lin_vel_x, lin_vel_y, ang_vel_yaw
    COMMAND_RANGES = ((-1.5, 1.5), (-0.5, 0.5), (-1.5, 1.5))

    def command_table(self) -> np.ndarray:
        """The commands `sample_command` hands out, as DATA (include/dial_mpc.h: dial_task.cmd_table): entry e holds
        for the step 500 e.  Drawn once per env from numpy's PCG64 seeded with the run's seed (``config.seed``, set by
        ``load_dial_and_env`` from DialConfig.seed; 0 otherwise) -- the reference draws from its JAX key chain, whose
        stream is version dependent; ``set_command_table`` installs exported reference values instead."""
        if self._cmd_table is None:
            rng = np.random.default_rng(int(getattr(self._config, "seed", 0)))
            n = int(M["DIAL_MAX_CMD"])
            self._cmd_table = np.stack([rng.uniform(lo, hi, n) for lo, hi in self.COMMAND_RANGES], axis=1)
        return self._cmd_table

    def set_command_table(self, table) -> None:
        table = np.asarray(table, dtype=np.float64).reshape(-1, 3)
        assert 1 <= table.shape[0] <= int(M["DIAL_MAX_CMD"])
        self._cmd_table = table
        self._ctx = None                       # contexts bake the task in

    def model_dict(self) -> Dict[str, Any]:
        return self.sys.model

    def make_task(self) -> "_abi.DialTask":
        return _abi.fill(_abi.DialTask(), self.task_dict())

    def make_model(self) -> "_abi.DialModel":
        return _abi.make_model(self.sys.model)

    # ---- HIP-backed reset / step (device tensors in, device tensors out)
    def bind_device(self, device: Optional[int]):
        """Run env.reset / env.step on this GPU (MBDPI binds the env to its own device)."""
        if self._ctx is not None and device is not None and self._ctx.device != int(device):
            self._ctx = None
        self._device = device

    def _context(self):
        if self._ctx is None:
            from dial_mpc_amd import _lib
            self._ctx = _lib.Context(self.make_model(), self.make_task(), None, getattr(self, "_device", None))
        return self._ctx

    def reset(self, rng=None):
        from dial_mpc_amd.envs.state import State
        ctx = self._context()
        return State.from_reset(self, ctx, self._init_q, np.zeros(self._nv))

    def step(self, state, action):
        ctx = self._context()
        return state.stepped(ctx, action)

    def pipeline_init(self, q, qd):
        from dial_mpc_amd.envs.state import State
        return State.from_reset(self, self._context(), q, qd).pipeline_state


def _to_numpy(x):
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy()
    return np.asarray(x)
