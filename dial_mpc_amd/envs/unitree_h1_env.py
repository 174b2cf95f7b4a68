"""Unitree H1 walk/jog environment: config and task description with the reference's constants
(dial_mpc/envs/unitree_h1_env.py:25-179) and the H1 loco environment (:570-858, legs + torso only,
arms welded, two capsules per foot) and the push-crate environment (:378-567, generic kernel instantiation)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Union

import numpy as np

from dial_mpc_amd import _abi, mjcf
from dial_mpc_amd.envs.base_env import BaseEnv, BaseEnvConfig, System, load_model

TASK_H1_WALK = _abi.MACROS["DIAL_TASK_H1_WALK"]
TASK_H1_LOCO = _abi.MACROS["DIAL_TASK_H1_LOCO"]
TASK_H1_PUSH_CRATE = _abi.MACROS["DIAL_TASK_H1_PUSH_CRATE"]

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


@dataclass
class UnitreeH1PushCrateEnvConfig(UnitreeH1WalkEnvConfig):
    # optional, not upstream keys -- see UnitreeGo2CrateEnvConfig: the contact ARRAY's order as data, and whether the reward's
    # contacts are found by geom identity or at upstream's literal positions (unitree_h1_env.py:474-480, 525-531)
    contact_slots: Any = None
    contact_lookup: str = "identity"


class UnitreeH1PushCrateEnv(UnitreeH1WalkEnv):
    """unitree_h1_env.py:382-567: the H1 (knee and foot capsules, torso box, hand spheres) behind a 1.2 m crate of 30 kg on a slide
    joint with 50 N of dry friction (`frictionloss`).  The walk env's reward with other weights, the feet heights taken from
    the foot capsules' floor contacts, and a contact term: +1 per hand on the crate (below 1.1 m), -1 per other robot part
    that touches it."""
    task_kind = TASK_H1_PUSH_CRATE

    def __init__(self, config: UnitreeH1PushCrateEnvConfig = None):
        super().__init__(config if config is not None else UnitreeH1PushCrateEnvConfig())
        self.physical_joint_range = self.physical_joint_range[:-1]      # :385 (the last joint is the crate's)
        self._init_pos_tar = np.array([0.0, 0.0, 1.2])                  # reset(), :399
        # Upstream reads contact.dist / contact.pos by POSITION in its MJX release's contact array: z_feet from [2:4] and [6:8],
        # wanted_contacts = [26, 27], unwanted_contacts = 14 .. 25 (:474-480, 525-531).  In geom-pair order -- floor against
        # left knee, left foot, right knee, right foot (2 each), torso (4), hands (1 each), then the crate against the same
        # eight geoms -- those positions are: the floor contacts of the two FOOT capsules; the two HAND spheres against the
        # crate; every other robot geom against the crate.  Looked up here by geom identity in this compiler's list.
        m = self.sys.model
        gname, bname = m["names"]["geom"], m["names"]["body"]
        floor, box = gname.index("floor"), gname.index("static_box")
        gbody = [bname[int(b)] for b in m["geom_bodyid"]]
        con = [(int(m["con_geom1"][c]), int(m["con_geom2"][c])) for c in range(int(m["ncon"]))]
        self._pc_foot_contact = []
        for foot in ("left_ankle_link", "right_ankle_link"):
            hits = [c for c, (g1, g2) in enumerate(con) if g1 == floor and gbody[g2] == foot]
            assert len(hits) == 2, f"{foot}: expected the two floor contacts of one foot capsule, found {hits}"
            self._pc_foot_contact.append(hits)
        hands = ("left_elbow_link", "right_elbow_link")
        self._pc_wanted = [c for c, (g1, g2) in enumerate(con) if g2 == box and gbody[g1] in hands]
        self._pc_unwanted = [c for c, (g1, g2) in enumerate(con) if g2 == box and gbody[g1] not in hands]
        # (12 with this compiler's candidate counts; a contact array rebuilt from a reference run may list a pair more often)
        assert len(self._pc_wanted) == 2 and 12 <= len(self._pc_unwanted) <= 16, (self._pc_wanted, self._pc_unwanted)
        self._pc_identity = ([list(h) for h in self._pc_foot_contact], list(self._pc_wanted), list(self._pc_unwanted))
        if getattr(self._config, "contact_lookup", "identity") == "literal":   # upstream's positions, verbatim
            self._pc_foot_contact, self._pc_wanted, self._pc_unwanted = [[2, 3], [6, 7]], [26, 27], list(range(14, 26))
        elif getattr(self._config, "contact_lookup", "identity") != "identity":
            raise ValueError("contact_lookup must be 'identity' or 'literal'")

    def make_system(self, config: UnitreeH1WalkEnvConfig) -> System:
        model = load_model("unitree_h1", "mjx_scene_h1_push_crate.xml")
        if getattr(config, "contact_slots", None) is not None:
            model = mjcf.reorder_contacts(model, [tuple(p) for p in config.contact_slots], ids="mujoco")
        return System(model).tree_replace({"opt.timestep": config.timestep})

    def task_dict(self) -> Dict[str, Any]:
        d = super().task_dict()
        unw = np.zeros(16, dtype=np.int64)
        unw[: len(self._pc_unwanted)] = self._pc_unwanted
        d.update(pc_foot_contact=np.array(self._pc_foot_contact), pc_wanted=np.array(self._pc_wanted),
                 pc_n_unwanted=len(self._pc_unwanted), pc_unwanted=unw, pc_wanted_zmax=1.1)
        return d
