"""Unitree H1 walk/jog environment: config and task description with the reference's constants
(dial_mpc/envs/unitree_h1_env.py:25-179) and the H1 loco environment (:570-858, legs + torso only,
arms welded, two capsules per foot).  push_crate is a NEXT row."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Union

import numpy as np

from dial_mpc_amd import _abi
from dial_mpc_amd.envs.base_env import BaseEnv, BaseEnvConfig, System, load_model

TASK_H1_WALK = _abi.MACROS["DIAL_TASK_H1_WALK"]
TASK_H1_LOCO = _abi.MACROS["DIAL_TASK_H1_LOCO"]

_KP = [200.0, 200.0, 200.0, 200.0, 60.0, 200.0, 200.0, 200.0, 200.0, 60.0, 200.0,
       60.0, 60.0, 60.0, 60.0, 60.0, 60.0, 60.0, 60.0]
_KD = [5.0, 5.0, 5.0, 5.0, 1.5, 5.0, 5.0, 5.0, 5.0, 1.5, 5.0,
       1.5, 1.5, 1.5, 1.5, 1.5, 1.5, 1.5, 1.5]


@dataclass
class UnitreeH1WalkEnvConfig(BaseEnvConfig):
    kp: Union[float, Any] = field(default_factory=lambda: np.array(_KP))
    kd: Union[float, Any] = field(default_factory=lambda: np.array(_KD))
    default_vx: float = 1.0
    default_vy: float = 0.0
    default_vyaw: float = 0.0
    ramp_up_time: float = 2.0
    gait: str = "jog"


class UnitreeH1WalkEnv(BaseEnv):
    task_kind = TASK_H1_WALK

    def __init__(self, config: UnitreeH1WalkEnvConfig):
        super().__init__(config)
        self._pelvis_idx = self.sys.mj_model.body_id("pelvis")
        self._torso_idx = self.sys.mj_model.body_id("torso_link")
        self._feet_site_id = np.array([self.sys.mj_model.site_id("left_foot"),
                                       self.sys.mj_model.site_id("right_foot")])
        self._gait = config.gait
        self._gait_phase = {"stand": np.zeros(2), "slow_walk": np.array([0.0, 0.5]),
                            "walk": np.array([0.0, 0.5]), "jog": np.array([0.0, 0.5])}
        self._gait_params = {"stand": np.array([1.0, 1.0, 0.0]), "slow_walk": np.array([0.6, 0.8, 0.15]),
                             "walk": np.array([0.5, 1.0, 0.15]), "jog": np.array([0.3, 2, 0.2])}
        self._init_q = self.sys.mj_model.keyframe("home").qpos
        self._default_pose = self._init_q[7:]
        self.joint_range = np.array(  # sampling range, unitree_h1_env.py:121-147
            [[-0.3, 0.3], [-0.3, 0.3], [-1.0, 1.0], [0.0, 1.74], [-0.6, 0.4],
             [-0.3, 0.3], [-0.3, 0.3], [-1.0, 1.0], [0.0, 1.74], [-0.6, 0.4],
             [-0.5, 0.5],
             [-0.78, 0.78], [-0.3, 0.3], [-0.3, 0.3], [-0.3, 0.3],
             [-0.78, 0.78], [-0.3, 0.3], [-0.3, 0.3], [-0.3, 0.3]])
        self._init_pos_tar = np.array([0.0, 0.0, 1.3])  # :163
        self._done_height = 0.18  # :307

    def make_system(self, config: UnitreeH1WalkEnvConfig) -> System:
        model = load_model("unitree_h1", "mjx_scene_h1_walk.xml")
        return System(model).tree_replace({"opt.timestep": config.timestep})

    def task_dict(self) -> Dict[str, Any]:
        d = super().task_dict()
        cfg = self._config
        duty, cadence, amp = self._gait_params[self._gait]
        d.update(
            torso_x=self._torso_idx - 1, upright_x=0, nfeet=2, feet_site=self._feet_site_id,
            foot_radius=0.0, gait_duty=duty, gait_cadence=cadence, gait_amp=amp,
            gait_phase=self._gait_phase[self._gait],
            cmd_vel=[cfg.default_vx, cfg.default_vy, 0.0], cmd_ang_vel=[0.0, 0.0, cfg.default_vyaw],
            ramp_up_time=cfg.ramp_up_time, done_height=self._done_height,
            init_pos_tar=self._init_pos_tar, n_stage=0, jump_dt=1.0,
        )
        return d


@dataclass
class UnitreeH1LocoEnvConfig(BaseEnvConfig):
    """unitree_h1_env.py:570-606 (11 actuators: two 5-joint legs and the torso yaw)."""
    kp: Union[float, Any] = field(default_factory=lambda: np.array(_KP[:11]))
    kd: Union[float, Any] = field(default_factory=lambda: np.array(_KD[:11]))
    default_vx: float = 1.0
    default_vy: float = 0.0
    default_vyaw: float = 0.0
    ramp_up_time: float = 2.0
    gait: str = "jog"


class UnitreeH1LocoEnv(BaseEnv):
    """unitree_h1_env.py:609-858.  Differences from the walk env that reach the kernel: four contacts
    per foot in the gait term, all three angular-velocity components, a foot-level term from the foot
    site frames, an energy term that uses the post-step joint velocity, and its own reward weights."""
    task_kind = TASK_H1_LOCO

    def __init__(self, config: UnitreeH1LocoEnvConfig):
        super().__init__(config)
        self._pelvis_idx = self.sys.mj_model.body_id("pelvis")
        self._torso_idx = self.sys.mj_model.body_id("torso_link")
        self._left_foot_idx = self.sys.mj_model.site_id("left_foot")
        self._right_foot_idx = self.sys.mj_model.site_id("right_foot")
        self._feet_site_id = np.array([self._left_foot_idx, self._right_foot_idx])
        self._gait = config.gait
        self._gait_phase = {"stand": np.zeros(2), "slow_walk": np.array([0.0, 0.5]),
                            "walk": np.array([0.0, 0.5]), "jog": np.array([0.0, 0.5])}
        self._gait_params = {"stand": np.array([1.0, 1.0, 0.0]), "slow_walk": np.array([0.6, 0.8, 0.15]),
                             "walk": np.array([0.5, 1.5, 0.10]), "jog": np.array([0.3, 2.0, 0.2])}  # :639-645
        self._init_q = self.sys.mj_model.keyframe("home").qpos
        self._default_pose = self._init_q[7:]
        self.joint_range = np.array(  # sampling range, :651-668
            [[-0.2, 0.2], [-0.2, 0.2], [-0.6, 0.6], [0.0, 1.5], [-0.6, 0.4],
             [-0.2, 0.2], [-0.2, 0.2], [-0.6, 0.6], [0.0, 1.5], [-0.6, 0.4],
             [-0.5, 0.5]])
        self._init_pos_tar = np.array([0.0, 0.0, 1.3])  # :685
        self._done_height = 0.18  # :836

    def make_system(self, config: UnitreeH1LocoEnvConfig) -> System:
        model = load_model("unitree_h1", "mjx_scene_h1_loco.xml")
        return System(model).tree_replace({"opt.timestep": config.timestep})

    def task_dict(self) -> Dict[str, Any]:
        d = super().task_dict()
        cfg = self._config
        duty, cadence, amp = self._gait_params[self._gait]
        d.update(
            torso_x=self._torso_idx - 1, upright_x=0, nfeet=2, feet_site=self._feet_site_id,
            foot_radius=0.0, gait_duty=duty, gait_cadence=cadence, gait_amp=amp,
            gait_phase=self._gait_phase[self._gait],
            cmd_vel=[cfg.default_vx, cfg.default_vy, 0.0], cmd_ang_vel=[0.0, 0.0, cfg.default_vyaw],
            ramp_up_time=cfg.ramp_up_time, done_height=self._done_height,
            init_pos_tar=self._init_pos_tar, n_stage=0, jump_dt=1.0,
        )
        return d
