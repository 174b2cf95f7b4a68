"""ctypes wrapper of the host wave emulator (tests/wave_emu/emu.cpp).  TEST INFRASTRUCTURE ONLY:
it compiles the kernel body with -DDIAL_EMU so that the exact kernel logic can be compared with the
oracle (and race-checked) without a GPU.  Nothing under dial_mpc_amd/ loads it."""
import ctypes
import os
import subprocess

import numpy as np

from dial_mpc_amd import _abi

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wave_emu")
_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dial_mpc_amd", "csrc")
_SO = os.path.join(_HERE, "libwave_emu.so")


def build(defines=(), tag=""):
    """defines / tag: a VARIANT of the emulator (e.g. ("-DDIAL_NO_FACTOR_REUSE",), "_noreuse") next to the default one."""
    so = _SO if not tag else _SO.replace(".so", f"{tag}.so")
    srcs = [os.path.join(_HERE, "emu.cpp")] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
    srcs.append(_abi.HEADER)
    if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(f) for f in srcs):
        return so
    tmp = f"{so}.{os.getpid()}.tmp"     # atomic: parallel test workers may all find the library stale at once
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-fno-strict-aliasing",
                           "-ffp-contract=off", *defines, "-o", tmp, os.path.join(_HERE, "emu.cpp")])
    os.replace(tmp, so)
    return so


class Emu:
    def __init__(self, model, task, cfg=None, path=0, defines=(), tag=""):
        """path 0: dimension-specialised instantiation when the model matches one (like the HIP library);
        path 1: force the generic instantiation.  defines / tag: a variant build of the emulator (see build)."""
        self.path = int(path)
        self.lib = ctypes.CDLL(build(defines, tag))
        self.model, self.task, self.cfg = model, task, cfg
        self.nq, self.nv, self.nu, self.nbody = model.nq, model.nv, model.nu, model.nbody
        self.nx = (model.nbody - 1) * 3
        self.state_size = _abi.state_size(model.nq, model.nv)

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(ctypes.c_void_p)

    @staticmethod
    def _a(x):
        return np.ascontiguousarray(np.asarray(x, dtype=np.float32))

    def env_reset(self, qpos, qvel, check_races=True):
        state = np.zeros(self.state_size, np.float32)
        xpos = np.zeros((self.nbody - 1, 3), np.float32)
        xquat = np.zeros((self.nbody - 1, 4), np.float32)
        rc = self.lib.emu_env_reset(ctypes.byref(self.model), ctypes.byref(self.task), self._p(self._a(qpos)),
                                    self._p(self._a(qvel)), self._p(state), self._p(xpos), self._p(xquat),
                                    int(check_races), self.path)
        assert rc == 0, f"emu_env_reset: rc={rc} (races or error)"
        return state, xpos, xquat

    def env_step(self, state, action, check_races=True):
        state = self._a(state).copy()
        xpos = np.zeros((self.nbody - 1, 3), np.float32)
        xquat = np.zeros((self.nbody - 1, 4), np.float32)
        ctrl = np.zeros(self.nu, np.float32)
        rc = self.lib.emu_env_step(ctypes.byref(self.model), ctypes.byref(self.task), self._p(state),
                                   self._p(self._a(action)), self._p(xpos), self._p(xquat), self._p(ctrl),
                                   int(check_races), self.path)
        assert rc == 0, f"emu_env_step: rc={rc} (races or error)"
        return state, xpos, xquat, ctrl

    def rollout(self, state, us, check_races=False, trace=False):
        """trace=True: also returns the packed state after every env.step, [B, T, state_size] (RolloutIO::trace)."""
        us = self._a(us)
        B, T = us.shape[:2]
        tr = np.zeros((B, T, self.state_size), np.float32) if trace else None
        rewss = np.zeros((B, T), np.float32)
        rews = np.zeros(B, np.float32)
        qss = np.zeros((B, T, self.nq), np.float32)
        qdss = np.zeros((B, T, self.nv), np.float32)
        xss = np.zeros((B, T, self.nx), np.float32)
        rc = self.lib.emu_rollout(ctypes.byref(self.model), ctypes.byref(self.task),
                                  ctypes.byref(self.cfg) if self.cfg is not None else None,
                                  self._p(self._a(state)), self._p(us), None, None, None, 0, 0, B, T, 0, None,
                                  self._p(rewss), self._p(rews), self._p(qss), self._p(qdss), self._p(xss),
                                  int(check_races), self.path, self._p(tr))
        assert rc == 0, f"emu_rollout: rc={rc} (races or error)"
        if trace:
            return rewss, qss, qdss, xss, rews, tr
        return rewss, qss, qdss, xss, rews

    def rollout_nodes(self, state, Ybar, noise_scale, eps, check_races=False):
        cfg = self.cfg
        N, Hn1, T = cfg.Nsample, cfg.Hnode + 1, cfg.Hsample + 1
        B = N + 1
        eps, Ybar = self._a(eps), self._a(Ybar)
        ns = self._a(noise_scale).reshape(-1)
        Y0s = np.zeros((B, Hn1, self.nu), np.float32)
        rewss = np.zeros((B, T), np.float32)
        rews = np.zeros(B, np.float32)
        qss = np.zeros((B, T, self.nq), np.float32)
        qdss = np.zeros((B, T, self.nv), np.float32)
        xss = np.zeros((B, T, self.nx), np.float32)
        rc = self.lib.emu_rollout(ctypes.byref(self.model), ctypes.byref(self.task), ctypes.byref(cfg),
                                  self._p(self._a(state)), None, self._p(eps), self._p(Ybar), self._p(ns),
                                  int(ns.size), N, B, T, Hn1, self._p(Y0s), self._p(rewss), self._p(rews),
                                  self._p(qss), self._p(qdss), self._p(xss), int(check_races), self.path, None)
        assert rc == 0, f"emu_rollout: rc={rc} (races or error)"
        return dict(Y0s=Y0s, rewss=rewss, rews=rews, qss=qss, qdss=qdss, xss=xss)

    def box_contact(self, kind, sub, g1, g2):
        """The kernel's box narrow phase on one pair; g = (pos[3], quat[4], size[3]).  Returns (dist, pos, frame[3, 3])."""
        pack = lambda g: self._a(np.concatenate([np.asarray(x, float).ravel() for x in g]))
        a, b = pack(g1), pack(g2)
        dist, pos, frame = np.zeros(1, np.float32), np.zeros(3, np.float32), np.zeros(9, np.float32)
        rc = self.lib.emu_box_contact(int(kind), int(sub), self._p(a), self._p(b), self._p(dist), self._p(pos), self._p(frame))
        assert rc == 0
        return float(dist[0]), pos.astype(np.float64), frame.reshape(3, 3).astype(np.float64)

    def sizes(self):
        """(which instantiation: 0 generic / 1 Go2 / 2 H1, sizeof(CModel), workspace words)."""
        a, b = ctypes.c_int(), ctypes.c_int()
        k = self.lib.emu_sizes(ctypes.byref(self.model), ctypes.byref(a), ctypes.byref(b))
        return k, a.value, b.value
