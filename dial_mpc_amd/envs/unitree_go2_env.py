"""Unitree Go2 environments: config dataclasses and task descriptions with the reference's
constants (dial_mpc/envs/unitree_go2_env.py:25-124 walk/trot, :319-401,559-592 seq_jump).

``reset`` / ``step`` execute in libdialhip.so; the reward formulas are in csrc/rollout_body.h
(product); the CPU checker restates them independently.  Crate climb: :649-803 (generic kernel instantiation)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Union

import numpy as np

from dial_mpc_amd import _abi, mjcf
from dial_mpc_amd.envs.base_env import BaseEnv, BaseEnvConfig, System, load_model

TASK_GO2_WALK = _abi.MACROS["DIAL_TASK_GO2_WALK"]
TASK_GO2_SEQ_JUMP = _abi.MACROS["DIAL_TASK_GO2_SEQ_JUMP"]
TASK_GO2_CRATE = _abi.MACROS["DIAL_TASK_GO2_CRATE"]


@dataclass
class UnitreeGo2EnvConfig(BaseEnvConfig):
    kp: Union[float, Any] = 30.0
    kd: Union[float, Any] = 0.0
    default_vx: float = 1.0
    default_vy: float = 0.0
    default_vyaw: float = 0.0
    ramp_up_time: float = 2.0
    gait: str = "trot"


class UnitreeGo2Env(BaseEnv):
    task_kind = TASK_GO2_WALK

    def __init__(self, config: UnitreeGo2EnvConfig):
        super().__init__(config)
        self._foot_radius = 0.0175
        self._gait = config.gait
        self._gait_phase = {  # unitree_go2_env.py:43-49
            "stand": np.zeros(4),
            "walk": np.array([0.0, 0.5, 0.75, 0.25]),
            "trot": np.array([0.0, 0.5, 0.5, 0.0]),
            "canter": np.array([0.0, 0.33, 0.33, 0.66]),
            "gallop": np.array([0.0, 0.05, 0.4, 0.35]),
        }
        self._gait_params = {  # ratio, cadence, amplitude (:50-57)
            "stand": np.array([1.0, 1.0, 0.0]),
            "walk": np.array([0.75, 1.0, 0.08]),
            "trot": np.array([0.45, 2, 0.08]),
            "canter": np.array([0.4, 4, 0.06]),
            "gallop": np.array([0.3, 3.5, 0.10]),
        }
        self._torso_idx = self.sys.mj_model.body_id("base")
        self._init_q = self.sys.mj_model.keyframe("home").qpos
        self._default_pose = self._init_q[7:]
        self.joint_range = np.array(  # sampling range (:66-81)
            [[-0.5, 0.5], [0.4, 1.4], [-2.3, -0.85],
             [-0.5, 0.5], [0.4, 1.4], [-2.3, -0.85],
             [-0.5, 0.5], [0.4, 1.4], [-2.3, -1.3],
             [-0.5, 0.5], [0.4, 1.4], [-2.3, -1.3]])
        feet_site = ["FL_foot", "FR_foot", "RL_foot", "RR_foot"]  # :82-87 (note: FL first)
        self._feet_site_id = np.array([self.sys.mj_model.site_id(f) for f in feet_site])
        self._init_pos_tar = np.array([0.282, 0.0, 0.3])  # :108
        self._done_height = 0.18  # :247

    def make_system(self, config: UnitreeGo2EnvConfig) -> System:
        model = load_model("unitree_go2", "mjx_scene_force.xml")
        return System(model).tree_replace({"opt.timestep": config.timestep})

    def task_dict(self) -> Dict[str, Any]:
        d = super().task_dict()
        cfg = self._config
        duty, cadence, amp = self._gait_params[self._gait]
        d.update(
            torso_x=self._torso_idx - 1, upright_x=0, nfeet=4, feet_site=self._feet_site_id,
            foot_radius=self._foot_radius, gait_duty=duty, gait_cadence=cadence, gait_amp=amp,
            gait_phase=self._gait_phase[self._gait],
            cmd_vel=[cfg.default_vx, cfg.default_vy, 0.0], cmd_ang_vel=[0.0, 0.0, cfg.default_vyaw],
            ramp_up_time=cfg.ramp_up_time, done_height=self._done_height,
            init_pos_tar=self._init_pos_tar, n_stage=0, jump_dt=1.0,
        )
        return d


@dataclass
class UnitreeGo2SeqJumpEnvConfig(UnitreeGo2EnvConfig):
    jump_dt: float = 1.0
    contact_targets: Any = None
    contact_target_radius: Any = None
    pose_target_sequence: Any = None
    yaw_target_sequence: Any = None


def _euler_to_quat_deg(v):
    """brax.math.euler_to_quat (degrees, intrinsic x-y'-z'')."""
    c1, c2, c3 = np.cos(np.asarray(v) * np.pi / 360)
    s1, s2, s3 = np.sin(np.asarray(v) * np.pi / 360)
    return np.array([c1 * c2 * c3 - s1 * s2 * s3, s1 * c2 * c3 + c1 * s2 * s3,
                     c1 * s2 * c3 - s1 * c2 * s3, c1 * c2 * s3 + s1 * s2 * c3])


def _quat_to_3x3(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


class UnitreeGo2SeqJumpEnv(UnitreeGo2Env):
    task_kind = TASK_GO2_SEQ_JUMP

    def __init__(self, config: UnitreeGo2SeqJumpEnvConfig = None):
        config = config if config is not None else UnitreeGo2SeqJumpEnvConfig()
        super().__init__(config)
        if getattr(config, "randomize_tasks", False):
            # unitree_go2_env.py:383-392, 594-629: reset() replaces the configured sequence by `sample_command(rng)` -- a
            # 10-jump random walk of the body position (+-0.65 m in x, y per jump) and of the heading (+-0.5 rad), turned
            # into foot targets by generate_jumping_sequence.  Drawn here, once, from the run's seed (see
            # BaseEnv.command_table for why the draw is data and not JAX's stream); the kernels only see the tables.
            rng = np.random.default_rng(int(getattr(config, "seed", 0)))
            com_pos = np.zeros((11, 3))
            com_pos[:, 2] = 0.27
            com_pos[1:, :2] = np.cumsum(rng.uniform(-0.65, 0.65, (10, 2)), axis=0)
            com_yaw = np.concatenate([[0.0], np.cumsum(rng.uniform(-0.5, 0.5, 10))])
            (self._contact_targets, self._contact_target_radius, self._pose_target_sequence,
             self._yaw_target_sequence) = UnitreeGo2SeqJumpEnv.generate_jumping_sequence(com_pos, com_yaw, 0.1)
        elif config.contact_targets is None or config.contact_target_radius is None:
            (self._contact_targets, self._contact_target_radius, self._pose_target_sequence,
             self._yaw_target_sequence) = UnitreeGo2SeqJumpEnv.generate_jumping_sequence(
                np.asarray(config.pose_target_sequence, dtype=np.float64),
                np.asarray(config.yaw_target_sequence, dtype=np.float64), 0.1)
        else:
            self._contact_targets = np.asarray(config.contact_targets, dtype=np.float64)
            self._contact_target_radius = np.asarray(config.contact_target_radius, dtype=np.float64)
            self._pose_target_sequence = np.asarray(config.pose_target_sequence, dtype=np.float64)
            self._yaw_target_sequence = np.asarray(config.yaw_target_sequence, dtype=np.float64)
        self.joint_range = np.array(  # unitree_go2_env.py:346-361
            [[-0.5, 0.5], [0.4, 2.0], [-2.3, -1.3],
             [-0.5, 0.5], [0.4, 2.0], [-2.3, -1.3],
             [-0.5, 0.5], [0.4, 1.4], [-2.3, -1.3],
             [-0.5, 0.5], [0.4, 1.4], [-2.3, -1.3]])
        self._init_pos_tar = np.array([0.0, 0.0, 0.27])  # :369
        self._done_height = 0.1  # :504

    @staticmethod
    def generate_jumping_sequence(com_pos, com_heading, foot_place_radius: float):
        """unitree_go2_env.py:559-592 (foot order FR, FL, RR, RL = contact order)."""
        com_pos = np.asarray(com_pos, dtype=np.float64)
        n_steps = com_pos.shape[0]
        assert n_steps == len(com_heading)
        contact_target_radius = np.full((n_steps, 4), foot_place_radius)
        contact_targets = []
        for i in range(n_steps):
            contact_target = np.repeat(com_pos[i][None], 4, axis=0)
            offsets = np.array([[0.2, -0.135, 0.0], [0.2, 0.135, 0.0],
                                [-0.2, -0.135, 0.0], [-0.2, 0.135, 0.0]])
            R = _quat_to_3x3(_euler_to_quat_deg(np.array([0.0, 0.0, com_heading[i] * 180 / np.pi])))
            contact_targets.append(contact_target + offsets @ R.T)
        return (np.array(contact_targets), contact_target_radius, np.array(com_pos),
                np.array(com_heading, dtype=np.float64))

    def _randomize_dict(self) -> Dict[str, Any]:
        return dict(randomize_tasks=0, n_cmd=0)     # seq-jump's env.step never redraws a command (:403-521)

    def task_dict(self) -> Dict[str, Any]:
        d = super().task_dict()
        S = self._contact_targets.shape[0]
        d.update(n_stage=S, jump_dt=self._config.jump_dt, contact_targets=self._contact_targets,
                 contact_radius=self._contact_target_radius, pose_targets=self._pose_target_sequence,
                 yaw_targets=self._yaw_target_sequence)
        return d


@dataclass
class UnitreeGo2CrateEnvConfig(UnitreeGo2EnvConfig):
    # Not upstream keys (both optional): the order of MJX's contact ARRAY as data, for comparisons against a reference run.
    # contact_slots: [[geom1, geom2], ...] MuJoCo geom ids per array position, as tools/export_reference_vectors.py records
    #   them (mjcf.reorder_contacts rebuilds the model's contact list in that order / multiplicity);
    # contact_lookup: "identity" (default) finds the reward's contacts by geom identity in whatever order the list has;
    #   "literal" uses upstream's hard-coded positions (unitree_go2_env.py:750) -- meaningful once contact_slots reproduces
    #   the array of the MJX release upstream ran.
    contact_slots: Any = None
    contact_lookup: str = "identity"


class UnitreeGo2CrateEnv(UnitreeGo2Env):
    """unitree_go2_env.py:653-803: the Go2 with its collision model (trunk box, calf capsules, foot spheres) in front of
    a 0.6 m crate.  Reward = head position towards (1.45, 0, 0.87) + upright + yaw + 0.02 per foot standing on the crate."""
    task_kind = TASK_GO2_CRATE

    def __init__(self, config: UnitreeGo2CrateEnvConfig = None):
        super().__init__(config if config is not None else UnitreeGo2CrateEnvConfig())
        self.joint_range = np.array(  # :656-671
            [[-0.25, 0.25], [-1.0, 1.4], [-2.7, -1.0],
             [-0.25, 0.25], [-1.0, 1.4], [-2.7, -1.0],
             [-0.25, 0.25], [0.0, 1.8], [-2.7, -1.0],
             [-0.25, 0.25], [0.0, 1.8], [-2.7, -1.0]])
        self._init_pos_tar = np.array([1.45, 0.0, 0.87])  # reset(), :797-803 (vel_tar = ang_vel_tar = yaw_tar = 0)
        # reward_contact (:741-766) reads contact.pos[contact_indices[i]] with contact_indices = [16, 17, 18, 19]: positions
        # in the contact array of the MJX release upstream ran, which depend on that release's grouping of geom pairs.  The
        # condition that follows (a point on the crate's top face) and the commented-out index formulas above it say what is
        # meant: the contacts of the four FOOT spheres with the crate.  They are looked up here by geom identity, in upstream's
        # loop order i = 0..3 = geom order FR, FL, RR, RL.
        m = self.sys.model
        names = m["names"]["geom"]
        box = names.index("static_box")
        self._crate_contact = []
        for foot in ("FR", "FL", "RR", "RL"):
            g = names.index(foot)
            hits = [c for c in range(int(m["ncon"])) if int(m["con_geom1"][c]) == g and int(m["con_geom2"][c]) == box]
            assert len(hits) == 1, f"foot geom {foot!r} has no contact with the crate in the compiled model"
            self._crate_contact.append(hits[0])
        self._crate_contact_identity = list(self._crate_contact)
        if getattr(self._config, "contact_lookup", "identity") == "literal":
            self._crate_contact = [16, 17, 18, 19]                           # :750, verbatim
        elif getattr(self._config, "contact_lookup", "identity") != "identity":
            raise ValueError("contact_lookup must be 'identity' or 'literal'")
        self._crate_region = np.array([1.0, 1.6, -0.45, 0.45, 0.59, 0.61])   # :753-760
        self._head_vec = np.array([0.285, 0.0, 0.0])                         # :717

    def make_system(self, config: UnitreeGo2EnvConfig) -> System:
        model = load_model("unitree_go2", "mjx_scene_force_crate.xml")
        if getattr(config, "contact_slots", None) is not None:
            model = mjcf.reorder_contacts(model, [tuple(p) for p in config.contact_slots], ids="mujoco")
        return System(model).tree_replace({"opt.timestep": config.timestep})

    def _randomize_dict(self) -> Dict[str, Any]:
        return dict(randomize_tasks=0, n_cmd=0)     # this env's step has no command to redraw (:679-795)

    def task_dict(self) -> Dict[str, Any]:
        d = super().task_dict()
        d.update(crate_contact=np.array(self._crate_contact), crate_region=self._crate_region, head_vec=self._head_vec)
        return d
