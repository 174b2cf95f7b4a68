"""Quadratic interpolating spline used by ``MBDPI.node2u`` / ``u2node``
(dial_mpc/core/dial_core.py:92-101, jax_cosmo ``InterpolatedUnivariateSpline(k=2)``).

The reference fits a fresh spline per (sample, action-dim); since both abscissa sets are fixed, the
map is a constant matrix.  This module builds those matrices in fp64 with the FITPACK rule for
even-degree interpolating splines (interior knots at the midpoints of the interior data intervals,
SURVEY C.1); tests pin it against ``scipy.interpolate.InterpolatedUnivariateSpline``.
"""
from __future__ import annotations

import numpy as np

K = 2


def _knots(x: np.ndarray) -> np.ndarray:
    m = len(x)
    if m < K + 1:
        raise ValueError("quadratic interpolating spline needs at least 3 points")
    interior = 0.5 * (x[1:m - 2] + x[2:m - 1])
    return np.concatenate([[x[0]] * (K + 1), interior, [x[-1]] * (K + 1)])


def _basis_row(t: np.ndarray, n: int, xq: float) -> np.ndarray:
    """All n quadratic B-spline basis values at xq; outside the data range the end polynomial
    piece is extrapolated (FITPACK splev ext=0)."""
    # interval index l with t[l] <= xq < t[l+1], clamped to the valid span [K, n-1]
    l = int(np.searchsorted(t, xq, side="right") - 1)
    l = min(max(l, K), n - 1)
    N = np.zeros(K + 1)
    N[0] = 1.0
    for d in range(1, K + 1):  # de Boor-Cox on the fixed span l
        saved = 0.0
        for r in range(d):
            tr = t[l + r + 1]
            tl = t[l + 1 - d + r]
            tmp = N[r] / (tr - tl)
            N[r] = saved + (tr - xq) * tmp
            saved = (xq - tl) * tmp
        N[d] = saved
    row = np.zeros(n)
    row[l - K:l + 1] = N
    return row


def interp_matrix(x_data: np.ndarray, x_query: np.ndarray) -> np.ndarray:
    """Matrix A with  spline(x_data, y)(x_query) == A @ y  for every y."""
    x_data = np.asarray(x_data, dtype=np.float64)
    x_query = np.asarray(x_query, dtype=np.float64)
    n = len(x_data)
    t = _knots(x_data)
    C = np.stack([_basis_row(t, n, xv) for xv in x_data])  # collocation
    B = np.stack([_basis_row(t, n, xv) for xv in x_query])
    return B @ np.linalg.inv(C)


def node2u_matrix(Hsample: int, Hnode: int, ctrl_dt: float = 0.02) -> np.ndarray:
    step_us = np.linspace(0, ctrl_dt * Hsample, Hsample + 1)
    step_nodes = np.linspace(0, ctrl_dt * Hsample, Hnode + 1)
    return interp_matrix(step_nodes, step_us)


def u2node_matrix(Hsample: int, Hnode: int, ctrl_dt: float = 0.02) -> np.ndarray:
    step_us = np.linspace(0, ctrl_dt * Hsample, Hsample + 1)
    step_nodes = np.linspace(0, ctrl_dt * Hsample, Hnode + 1)
    return interp_matrix(step_us, step_nodes)
