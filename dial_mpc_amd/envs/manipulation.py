"""Allegro in-hand reorientation: config and task description with the reference's constants
(dial_mpc/envs/manipulation.py:23-117, models/wonik_allegro/scene_left.xml + left_hand.xml).

``reset`` / ``step`` execute in libdialhip.so (task kind DIAL_TASK_ALLEGRO); the reward and the ``act2joint``
override are restated in csrc/rollout_body.h (product) and, independently, in the CPU checker.

Model facts the kernels rely on (elliptic cones, impratio 10, Euler damping ON, position actuators, 4 physics
sub-steps per control step): 1 free object (sphere, condim 6, priority 1) + a welded palm with four 4-hinge
fingers; 19 potential contacts -- 8 plane-capsule, 6 capsule-capsule (condim 3), plane-sphere and 4
sphere-capsule (condim 6) -- i.e. 72 contact rows + 16 joint-limit rows.  The box collision geoms of the hand are
commented out in the reference's MJCF, so no box narrow phase is involved."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Union

import numpy as np

from dial_mpc_amd import _abi
from dial_mpc_amd.envs.base_env import BaseEnv, BaseEnvConfig, System, load_model

TASK_ALLEGRO = _abi.MACROS["DIAL_TASK_ALLEGRO"]


@dataclass
class AllegroReorientEnvConfig(BaseEnvConfig):
    kp: Union[float, Any] = 1.0
    kd: Union[float, Any] = 0.1


class AllegroReorientEnv(BaseEnv):
    task_kind = TASK_ALLEGRO

    def __init__(self, config: AllegroReorientEnvConfig):
        super().__init__(config)
        if config.leg_control != "position":
            raise NotImplementedError("AllegroReorientEnv: only leg_control='position' (manipulation.py:69-72)")
        self._object_body_idx = self.sys.mj_model.body_id("object")
        self._init_q = self.sys.mj_model.keyframe("in_hand_reorient").qpos
        self._init_ang_vel_tar = np.array([0.0, 0.0, 0.5])   # manipulation.py:54
        self._init_pos_tar = np.array([0.0, 0.0, 0.13])      # :55

    def make_system(self, config: AllegroReorientEnvConfig) -> System:
        model = load_model("wonik_allegro", "scene_left.xml")
        return System(model).tree_replace({"opt.timestep": config.timestep})

    def act2joint(self, act):
        """manipulation.py:102-115: the keyframe pose is ADDED to the lower range before scaling, then clipped.
        (The actuator order ff, mf, rf, th differs from the joint order rf, mf, ff, th; like upstream, target a is
        computed from joint a's range and handed to actuator a.)"""
        from dial_mpc_amd.envs.base_env import _to_numpy
        act = np.asarray(_to_numpy(act), dtype=np.float32)
        jr = np.asarray(self.joint_range, dtype=np.float32)
        pr = np.asarray(self.physical_joint_range, dtype=np.float32)
        init = np.asarray(self._init_q[7:], dtype=np.float32)
        act_normalized = (act * np.float32(self._config.action_scale) + np.float32(1.0)) / np.float32(2.0)
        joint_targets = jr[:, 0] + init + act_normalized * (jr[:, 1] - jr[:, 0])
        return np.clip(joint_targets, pr[:, 0], pr[:, 1])

    def task_dict(self) -> Dict[str, Any]:
        d = super().task_dict()
        d.update(
            torso_x=self._object_body_idx - 1, upright_x=0, nfeet=0, feet_site=np.zeros(4, dtype=np.int64),
            foot_radius=0.0, gait_duty=1.0, gait_cadence=1.0, gait_amp=0.0, gait_phase=np.zeros(4),
            cmd_vel=[0.0, 0.0, 0.0], cmd_ang_vel=[0.0, 0.0, 0.0], ramp_up_time=1.0, done_height=0.0,
            init_pos_tar=self._init_pos_tar, init_ang_vel_tar=self._init_ang_vel_tar,
            joint_offset=np.asarray(self._init_q[7:], dtype=np.float64), n_stage=0, jump_dt=1.0,
        )
        return d
