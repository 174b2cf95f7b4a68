"""Consumes golden vectors exported from the JAX reference (tools/export_reference_vectors.py, recipe pinned in
tools/reference_env.txt) when a maintainer has placed them under tests/golden/reference/.

Absent files => parity vs JAX stays UNVERIFIED and the first test below is reported as XFAIL (not as a pass): the
suite cannot be read as "green against the reference" while no reference vector exists.  With vectors present, the
CPU leg checks the oracle and the `-m gpu` leg checks the HIP path through the C ABI, on the exported inputs."""
import glob
import os

import numpy as np
import pytest

REF_DIR = os.path.join(os.path.dirname(__file__), "golden", "reference")
FILES = sorted(glob.glob(os.path.join(REF_DIR, "*.npz")))


def test_reference_vectors_present():
    if not FILES:
        pytest.xfail("golden vectors from the JAX reference are absent -- parity vs the reference is unpinned "
                     "(DESIGN.md section 2); produce them with tools/export_reference_vectors.py")


def _case(path):
    from conftest import setup_case
    g = np.load(path)
    N, Hn1, nu = g["eps"].shape
    H = g["us"].shape[1] - 1
    example = os.path.basename(path).split("__")[0]
    return g, example, setup_case(example, N, H)


def _check(example, got, g):
    """got: dict with rewss [B,T], qss, qdss, xss, Ybar -- against the exported reference arrays."""
    from conftest import TOL, agg_tol
    B, T = g["rewss"].shape
    for name, key in (("rewss", "rewss"), ("q", "qss"), ("qd", "qdss"), ("x", "xss")):
        ref = np.asarray(g[key]).reshape(B, T, -1) if key != "rewss" else np.asarray(g[key])
        val = np.asarray(got[key]).reshape(ref.shape)
        ok = np.abs(val - ref) <= TOL[name]["atol"] + TOL[name]["rtol"] * np.abs(ref)
        per_rollout = ok.reshape(B, -1).all(1)
        # knife-edge rollouts (discrete solver decisions flipped by rounding) are bounded, not waved through silently
        assert per_rollout.mean() >= 0.97, (example, name, float(per_rollout.mean()), float(np.abs(val - ref).max()))
    assert np.allclose(got["Ybar"], g["Ybar"], **agg_tol(example, "Ybar"))


@pytest.mark.skipif(not FILES, reason="golden vectors absent -- parity vs JAX unverified")
@pytest.mark.parametrize("path", FILES)
def test_oracle_against_reference_vectors(path):
    import oracle as O
    g, example, (dc, env, model, task, cfg) = _case(path)
    orc = O.Oracle(model, task, cfg, np.float32)
    state, _, _ = orc.env_reset(g["qpos"], g["qvel"])
    r = orc.reverse_once(state, g["Ybar_in"], g["noise_scale"], g["eps"], full=True)
    assert np.allclose(r["us"], g["us"], atol=2e-6)                      # spline (jax_cosmo) == FITPACK matrices
    ro = orc.rollout(state, r["us"])
    _check(example, dict(rewss=ro[0], qss=ro[1], qdss=ro[2], xss=ro[3], Ybar=r["Ybar"]), g)


@pytest.mark.gpu
@pytest.mark.skipif(not FILES, reason="golden vectors absent -- parity vs JAX unverified")
@pytest.mark.parametrize("path", FILES)
def test_hip_against_reference_vectors(path):
    import torch
    from dial_mpc_amd import _lib
    g, example, (dc, env, model, task, cfg) = _case(path)
    ctx = _lib.Context(model, task, cfg)
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device="cuda")  # noqa: E731
    state, _, _ = ctx.env_reset(dev(g["qpos"]), dev(g["qvel"]))
    out = ctx.reverse_once(state, dev(g["Ybar_in"]), dev(g["noise_scale"]), dev(g["eps"]))
    sc = ctx.debug_scratch()
    _check(example, dict(rewss=sc["rewss"], qss=sc["qss"], qdss=sc["qdss"], xss=sc["xss"],
                         Ybar=out["Ybar"].cpu().numpy()), g)
