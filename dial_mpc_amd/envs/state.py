"""Environment state: a packed float32 device tensor [qpos | qvel | qacc_warmstart | info] plus the
brax-``State``-shaped views the reference's drivers read (dial_core.py:39,122-124,313-315;
dial_plan.py:150-154)."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from dial_mpc_amd import _abi

M = _abi.MACROS


class _Info:
    """dict-like view of the info slots of the packed state (``state.info["step"]`` etc.)."""
    _SLOTS = {"step": (M["DIAL_INFO_STEP"], 1), "pos_tar": (M["DIAL_INFO_POS_TAR"], 3),
              "vel_tar": (M["DIAL_INFO_VEL_TAR"], 3), "ang_vel_tar": (M["DIAL_INFO_ANG_VEL_TAR"], 3),
              "yaw_tar": (M["DIAL_INFO_YAW_TAR"], 1), "last_contact": (M["DIAL_INFO_LAST_CONTACT"], 4),
              "feet_air_time": (M["DIAL_INFO_AIR_TIME"], 4), "contact_stage": (M["DIAL_INFO_STAGE"], 1),
              "last_ctrl": (M["DIAL_INFO_LAST_CTRL"], M["DIAL_MAX_U"])}

    def __init__(self, packed, base):
        self._packed, self._base = packed, base

    def __getitem__(self, key):
        off, n = self._SLOTS[key]
        v = self._packed[self._base + off:self._base + off + n]
        return v[0] if n == 1 else v

    def __setitem__(self, key, value):
        off, n = self._SLOTS[key]
        if n == 1:
            self._packed[self._base + off] = float(value)
        else:
            import torch
            self._packed[self._base + off:self._base + off + n] = torch.as_tensor(
                value, dtype=self._packed.dtype, device=self._packed.device)

    def keys(self):
        return self._SLOTS.keys()


class State:
    def __init__(self, env, packed, xpos, xquat, ctrl):
        self.env = env
        self.packed = packed
        nq, nv = env.sys.nq, env.sys.nv
        self._base = nq + 2 * nv
        q, qd = packed[:nq], packed[nq:nq + nv]
        self.pipeline_state = SimpleNamespace(q=q, qd=qd, qpos=q, qvel=qd, ctrl=ctrl,
                                              x=SimpleNamespace(pos=xpos, rot=xquat),
                                              qacc_warmstart=packed[nq + nv:nq + 2 * nv])
        self.info = _Info(packed, self._base)
        self.obs = None
        self.metrics = {}

    @property
    def reward(self):
        return self.packed[self._base + M["DIAL_INFO_REWARD"]]

    @property
    def done(self):
        return self.packed[self._base + M["DIAL_INFO_DONE"]]

    @classmethod
    def from_reset(cls, env, ctx, qpos, qvel):
        import torch
        q = torch.as_tensor(np.asarray(qpos, dtype=np.float32), device=ctx.torch_device)
        qd = torch.as_tensor(np.asarray(qvel, dtype=np.float32), device=ctx.torch_device)
        packed, xpos, xquat = ctx.env_reset(q, qd)
        ctrl = torch.zeros(env.sys.nu, dtype=torch.float32, device=ctx.torch_device)
        return cls(env, packed, xpos, xquat, ctrl)

    def stepped(self, ctx, action):
        import torch
        act = torch.as_tensor(action, dtype=torch.float32, device=ctx.torch_device).contiguous()
        packed, xpos, xquat, ctrl = ctx.env_step(self.packed, act)
        return State(self.env, packed, xpos, xquat, ctrl)

    def replace(self, **kw):
        """Minimal ``state.replace(pipeline_state=...)`` support for the deploy-style state injection
        (dial_plan.py:149-155): qpos / qvel of the given pipeline_state are written into the packed state."""
        import torch
        new = State(self.env, self.packed.clone(), self.pipeline_state.x.pos, self.pipeline_state.x.rot,
                    self.pipeline_state.ctrl)
        ps = kw.get("pipeline_state")
        if ps is not None:
            nq, nv = self.env.sys.nq, self.env.sys.nv
            new.packed[:nq] = torch.as_tensor(ps.qpos, dtype=torch.float32, device=new.packed.device)
            new.packed[nq:nq + nv] = torch.as_tensor(ps.qvel, dtype=torch.float32, device=new.packed.device)
        return new
