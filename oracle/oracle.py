"""ctypes wrapper of the CPU oracle (oracle/dial_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under dial_mpc_amd/ does (tests/test_no_oracle_in_product.py enforces it).

PARITY STATUS: parity unpinned against the JAX reference (see the header of dial_oracle.c).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from dial_mpc_amd import _abi  # noqa: E402  (struct mirrors only; no product code path)


def build(force: bool = False) -> None:
    need = force or any(not os.path.exists(os.path.join(_HERE, f"liboracle_{s}.so")) or
                        os.path.getmtime(os.path.join(_HERE, f"liboracle_{s}.so")) <
                        max(os.path.getmtime(os.path.join(_HERE, "dial_oracle.c")),
                            os.path.getmtime(_abi.HEADER))
                        for s in ("f32", "f64"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])


class Oracle:
    """One flavour (float32 / float64) of the oracle bound to a (model, task, cfg)."""

    def __init__(self, model: "_abi.DialModel", task: "_abi.DialTask", cfg: Optional["_abi.DialCfg"],
                 dtype=np.float64, native: bool = False):
        """native=True (fp32 only; bench.py's cpu_baseline): the -O3 -march=native build, compiled on this host."""
        build()
        self.dtype = np.dtype(dtype)
        name = "liboracle_f32.so" if self.dtype == np.float32 else "liboracle_f64.so"
        if native:
            assert self.dtype == np.float32
            subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
            name = os.path.join("_native", name)
        self.lib = ctypes.CDLL(os.path.join(_HERE, name))
        assert self.lib.oracle_real_bytes() == self.dtype.itemsize
        self.model, self.task, self.cfg = model, task, cfg
        self.nq, self.nv, self.nu, self.nbody = model.nq, model.nv, model.nu, model.nbody
        self.nx = (model.nbody - 1) * 3
        self.state_size = _abi.state_size(model.nq, model.nv)

    def _p(self, a):
        return None if a is None else a.ctypes.data_as(ctypes.c_void_p)

    def _a(self, x):
        return np.ascontiguousarray(np.asarray(x, dtype=self.dtype))

    def env_reset(self, qpos, qvel):
        state = np.zeros(self.state_size, self.dtype)
        xpos = np.zeros((self.nbody - 1, 3), self.dtype)
        xquat = np.zeros((self.nbody - 1, 4), self.dtype)
        self.lib.oracle_env_reset(ctypes.byref(self.model), ctypes.byref(self.task), self._p(self._a(qpos)),
                                  self._p(self._a(qvel)), self._p(state), self._p(xpos), self._p(xquat))
        return state, xpos, xquat

    def env_step(self, state, action):
        state = self._a(state).copy()
        xpos = np.zeros((self.nbody - 1, 3), self.dtype)
        xquat = np.zeros((self.nbody - 1, 4), self.dtype)
        ctrl = np.zeros(self.nu, self.dtype)
        self.lib.oracle_env_step(ctypes.byref(self.model), ctypes.byref(self.task), self._p(state),
                                 self._p(self._a(action)), self._p(xpos), self._p(xquat), self._p(ctrl))
        return state, xpos, xquat, ctrl

    def rollout(self, state, us):
        us = self._a(us)
        B, T = us.shape[0], us.shape[1]
        rewss = np.zeros((B, T), self.dtype)
        qss = np.zeros((B, T, self.nq), self.dtype)
        qdss = np.zeros((B, T, self.nv), self.dtype)
        xss = np.zeros((B, T, self.nx), self.dtype)
        self.lib.oracle_rollout(ctypes.byref(self.model), ctypes.byref(self.task), self._p(self._a(state)),
                                self._p(us), B, T, self._p(rewss), self._p(qss), self._p(qdss), self._p(xss))
        return rewss, qss, qdss, xss

    def rollout_jitter(self, state, us, noise_seed: int, noise_mag: float = 1.0):
        """`rollout` with qpos / qvel / qacc_warmstart of every rollout jittered by <= noise_mag ulp (fp32) before every
        step: one member of the jitter ensemble of the distribution-level gates (conftest.jitter_envelope)."""
        us = self._a(us)
        B, T = us.shape[0], us.shape[1]
        rewss = np.zeros((B, T), self.dtype)
        qss = np.zeros((B, T, self.nq), self.dtype)
        qdss = np.zeros((B, T, self.nv), self.dtype)
        xss = np.zeros((B, T, self.nx), self.dtype)
        self.lib.oracle_rollout_jitter(ctypes.byref(self.model), ctypes.byref(self.task), self._p(self._a(state)),
                                       self._p(us), B, T, self._p(rewss), self._p(qss), self._p(qdss), self._p(xss),
                                       ctypes.c_ulonglong(int(noise_seed)), ctypes.c_double(float(noise_mag)))
        return rewss, qss, qdss, xss

    def rollout_trace(self, state, us, noise_seed: int = 0, noise_mag: float = 0.0):
        """One rollout (us [T,nu]) with the solver's decision trace per step:
        [use_warm, niter, nactive_start, nactive_end, ls_iters_total, improved_mask, ncontact_rows_on, nlimit_rows_on].
        noise_mag > 0: qpos / qvel / qacc_warmstart are jittered by <= noise_mag ulp (fp32) before every step."""
        us = self._a(us)
        T = us.shape[0]
        trace = np.zeros((T, 8), np.int32)
        rewss = np.zeros(T, self.dtype)
        qss = np.zeros((T, self.nq), self.dtype)
        qdss = np.zeros((T, self.nv), self.dtype)
        self.lib.oracle_rollout_trace(ctypes.byref(self.model), ctypes.byref(self.task), self._p(self._a(state)),
                                      self._p(us), T, trace.ctypes.data_as(ctypes.c_void_p), self._p(rewss),
                                      self._p(qss), self._p(qdss), ctypes.c_ulonglong(int(noise_seed)),
                                      ctypes.c_double(float(noise_mag)))
        return trace, rewss, qss, qdss

    def reverse_once(self, state, Ybar, noise_scale, eps, full: bool = False):
        cfg = self.cfg
        N, Hn1, T = cfg.Nsample, cfg.Hnode + 1, cfg.Hsample + 1
        ns_arr = self._a(noise_scale).reshape(-1)
        eps = self._a(eps)
        assert eps.shape == (N, Hn1, self.nu)
        Yo = np.zeros((Hn1, self.nu), self.dtype)
        rews = np.zeros(N + 1, self.dtype)
        qbar = np.zeros((T, self.nq), self.dtype)
        qdbar = np.zeros((T, self.nv), self.dtype)
        xbar = np.zeros((T, self.nx), self.dtype)
        us = np.zeros((N + 1, T, self.nu), self.dtype) if full else None
        rewss = np.zeros((N + 1, T), self.dtype) if full else None
        w = np.zeros(N + 1, self.dtype) if full else None
        self.lib.oracle_reverse_once(ctypes.byref(self.model), ctypes.byref(self.task), ctypes.byref(cfg),
                                     self._p(self._a(state)), self._p(self._a(Ybar)), self._p(ns_arr),
                                     int(ns_arr.size), self._p(eps), self._p(Yo), self._p(rews), self._p(qbar),
                                     self._p(qdbar), self._p(xbar), self._p(us), self._p(rewss), self._p(w))
        out = dict(Ybar=Yo, rews=rews, qbar=qbar, qdbar=qdbar, xbar=xbar)
        if full:
            out.update(us=us, rewss=rewss, weights=w)
        return out

    def shift(self, Y):
        Y = self._a(Y).copy()
        self.lib.oracle_shift(ctypes.byref(self.model), ctypes.byref(self.cfg), self._p(Y))
        return Y

    def forward_dump(self, qpos, qvel, ctrl=None, warm=None) -> Dict[str, np.ndarray]:
        m = self.model
        nv, ne, nc, nb = m.nv, m.nefc, m.ncon, m.nbody
        o = dict(qM=np.zeros((nv, nv)), qfrc_bias=np.zeros(nv), qacc_smooth=np.zeros(nv), qacc=np.zeros(nv),
                 efc_force=np.zeros(ne), con_dist=np.zeros(nc), con_pos=np.zeros((nc, 3)),
                 subtree_com=np.zeros((nb, 3)), site_xpos=np.zeros((m.nsite, 3)), cvel=np.zeros((nb, 6)),
                 xpos=np.zeros((nb, 3)), xquat=np.zeros((nb, 4)), efc_J=np.zeros((ne, nv)),
                 efc_aref=np.zeros(ne), efc_D=np.zeros(ne))
        o = {k: v.astype(self.dtype) for k, v in o.items()}
        niter = ctypes.c_int(0)
        self.lib.oracle_forward_dump(
            ctypes.byref(m), self._p(self._a(qpos)), self._p(self._a(qvel)),
            self._p(self._a(ctrl)) if ctrl is not None else None,
            self._p(self._a(warm)) if warm is not None else None,
            *[self._p(o[k]) for k in ("qM", "qfrc_bias", "qacc_smooth", "qacc", "efc_force", "con_dist", "con_pos",
                                      "subtree_com", "site_xpos", "cvel", "xpos", "xquat", "efc_J", "efc_aref",
                                      "efc_D")],
            ctypes.byref(niter))
        o["niter"] = niter.value
        return o

    def box_contact(self, kind: int, sub: int, g1, g2):
        """One candidate contact of a box pair; g = (pos[3], mat[3, 3], size[3]).  Returns (dist, pos, frame[3, 3])."""
        pack = lambda g: self._a(np.concatenate([np.asarray(g[0], float).ravel(), np.asarray(g[1], float).ravel(),
                                                 np.asarray(g[2], float).ravel()]))
        a, b = pack(g1), pack(g2)
        dist, pos, frame = np.zeros(1, self.dtype), np.zeros(3, self.dtype), np.zeros(9, self.dtype)
        rc = self.lib.oracle_box_contact(int(kind), int(sub), self._p(a), self._p(b), self._p(dist), self._p(pos), self._p(frame))
        if rc:
            raise ValueError(f"oracle_box_contact: kind {kind}")
        return float(dist[0]), pos.astype(np.float64), frame.reshape(3, 3).astype(np.float64)

    def foot_step(self, time: float) -> np.ndarray:
        h = np.zeros(self.task.nfeet, self.dtype)
        t = ctypes.c_float(time) if self.dtype == np.float32 else ctypes.c_double(time)
        self.lib.oracle_foot_step(ctypes.byref(self.task), t, self._p(h))
        return h
