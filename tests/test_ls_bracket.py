"""The line search's bracket update on integer slope keys (csrc/ls_bracket.h: ls_update_lazy, the code every solver of the kernels runs;
here compiled into the host emulator) against a plain restatement of MJX's two rules (mujoco.mjx._src.solver._linesearch: `swap` of
<= 3.1.3, `_in_bracket` of >= 3.1.4, cf. oracle/dial_oracle.c) -- the kernel's form of `_in_bracket` is min / max arithmetic on the keys,
which must take the same decisions for EVERY combination of keys, ties and zeros included."""
import ctypes
import itertools

import numpy as np
import pytest

import emu_lib


def _reference(rule_swap, arr):
    """Vectorised over the rows (lo, hi, k_lo_next, k_hi_next, k_mid) of `arr`: MJX's rules with plain comparisons."""
    lo, hi, kl, kh, km = (arr[:, k].astype(np.int64) for k in range(5))
    lo_sel, hi_sel = np.full(lo.shape, -1), np.full(lo.shape, -1)
    moved = np.zeros(lo.shape, bool)

    def take(x, sel, c, y, lane):
        return np.where(c, y, x), np.where(c, lane, sel), moved | c

    if rule_swap:
        lo, lo_sel, moved = take(lo, lo_sel, (lo > 0) | (lo < kl), kl, 0)
        lo, lo_sel, moved = take(lo, lo_sel, (km < 0) & (lo < km), km, 2)
        hi, hi_sel, moved = take(hi, hi_sel, (hi < 0) | (hi > kh), kh, 1)
        hi, hi_sel, moved = take(hi, hi_sel, (km > 0) & (hi > km), km, 2)
    else:
        in_bracket = lambda x, y: ((x < y) & (y < 0)) | ((x > y) & (y > 0))   # noqa: E731
        for y, lane in ((kl, 0), (km, 2), (kh, 1)):
            lo, lo_sel, moved = take(lo, lo_sel, in_bracket(lo, y), y, lane)
        for y, lane in ((kh, 1), (km, 2), (kl, 0)):
            hi, hi_sel, moved = take(hi, hi_sel, in_bracket(hi, y), y, lane)
    return np.stack([lo, hi, lo_sel, hi_sel, moved.astype(np.int64)], 1).astype(np.int32)


def _fkey(x):
    b = np.float32(x + np.float32(0)).view(np.int32)
    return int(b ^ ((b >> 31) & 0x7fffffff))


@pytest.mark.parametrize("rule_swap", [0, 1, 2])   # 2: `_in_bracket` in the boolean form (the capacity-dimension kernel)
def test_bracket_update_takes_the_reference_decisions(rule_swap):
    lib = ctypes.CDLL(emu_lib.build())
    edge = [_fkey(v) for v in (0.0, -0.0, 1e-45, -1e-45, 1.0, -1.0, 1.0000001, -1.0000001, 3e38, -3e38, np.inf, -np.inf)]
    cases = list(itertools.product(edge, repeat=5))                                  # every combination of the edge keys (incl. all ties)
    rng = np.random.default_rng(5)
    small = rng.integers(-4, 5, size=(200000, 5))                                    # dense ties around zero
    wide = rng.standard_normal((200000, 5)).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, size=(200000, 1))
    wide_keys = np.vectorize(_fkey)(wide)
    arr = np.concatenate([np.array(cases, np.int64), small, wide_keys]).astype(np.int32)
    out = np.zeros_like(arr)
    rc = lib.emu_ls_update(rule_swap, arr.shape[0], arr.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    ref = _reference(rule_swap == 1, arr)
    bad = np.nonzero((ref != out).any(1))[0]
    assert bad.size == 0, (arr[bad[:5]], ref[bad[:5]], out[bad[:5]])


def test_range_compare_form_of_the_done_test():
    """`lo.d0 in (kng, 0) or hi.d0 in (0, kg)` as two unsigned range compares (ls_gate / ls_converged_lo / _hi: what the kernels' loops
    branch on) against the four-compare form, over edge keys (gtol = 0 included: both intervals empty) and random ones."""
    lib = ctypes.CDLL(emu_lib.build())
    edge = [_fkey(v) for v in (0.0, 1e-45, -1e-45, 1e-12, -1e-12, 1.0, -1.0, np.inf, -np.inf)] + [1, -1, 2, -2, 2 ** 31 - 1, -2 ** 31 + 1]
    gt = [0.0, 1e-45, 1e-12, 3e-7, 1.0, 3e38]
    cases = [(a, b, _fkey(g), _fkey(-g)) for a in edge for b in edge for g in gt]
    rng = np.random.default_rng(9)
    for _ in range(20000):
        g = float(np.float32(10.0 ** rng.uniform(-40, 3)))
        x = (rng.standard_normal(2).astype(np.float32) * np.float32(10.0 ** rng.uniform(-42, 4))).tolist()
        cases.append((_fkey(x[0]), _fkey(x[1]), _fkey(g), _fkey(-g)))
    arr = np.array(cases, np.int64).astype(np.int32)
    out = np.zeros(arr.shape[0], np.int32)
    assert lib.emu_ls_converged(arr.shape[0], arr.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.all((out == 0) | (out == 3)), arr[(out == 1) | (out == 2)][:5]
    assert (out == 3).sum() > 100 and (out == 0).sum() > 100
