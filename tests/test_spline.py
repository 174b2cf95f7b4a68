"""node2u / u2node matrices pinned against SciPy FITPACK (jax_cosmo's k=2 spline == FITPACK)."""
import numpy as np
import pytest
from scipy.interpolate import InterpolatedUnivariateSpline as IUS

from dial_mpc_amd.core import spline


@pytest.mark.parametrize("Hs,Hn", [(16, 4), (20, 5), (25, 5), (24, 6), (20, 4), (8, 4), (16, 5)])
def test_matrices_match_fitpack(Hs, Hn):
    su, sn = np.linspace(0, 0.02 * Hs, Hs + 1), np.linspace(0, 0.02 * Hs, Hn + 1)
    W, V = spline.node2u_matrix(Hs, Hn), spline.u2node_matrix(Hs, Hn)
    Wr = np.stack([IUS(sn, e, k=2)(su) for e in np.eye(Hn + 1)], 1)
    Vr = np.stack([IUS(su, e, k=2)(sn) for e in np.eye(Hs + 1)], 1)
    assert np.abs(W - Wr).max() < 1e-13 and np.abs(V - Vr).max() < 1e-13
    assert np.allclose(W.sum(1), 1) and np.allclose(V.sum(1), 1)          # partition of unity
    if Hs % Hn == 0:                                                      # interpolation => sub-sampling
        assert np.allclose(V, np.eye(Hs + 1)[:: Hs // Hn], atol=1e-12)
        assert np.allclose(W[:: Hs // Hn], np.eye(Hn + 1), atol=1e-12)
        assert np.allclose(V @ W, np.eye(Hn + 1), atol=1e-10)             # u2node(node2u(Y)) == Y


def test_extrapolation_matches_fitpack():
    """dial_plan.py:136-139 evaluates the spline at step_nodes + shift_time (beyond the last node)."""
    sn = np.linspace(0, 0.32, 5)
    for shift in (0.0, 0.013, 0.02, 0.05):
        A = spline.interp_matrix(sn, sn + shift)
        Ar = np.stack([IUS(sn, e, k=2)(sn + shift) for e in np.eye(5)], 1)
        assert np.abs(A - Ar).max() < 1e-12


def test_known_answer_row():
    W = spline.node2u_matrix(16, 4)
    assert np.allclose(W[1], [0.64017857, 0.48839286, -0.15, 0.02410714, -0.00267857], atol=1e-8)
