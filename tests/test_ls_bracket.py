"""The line search's bracket update on integer slope keys (csrc/ls_bracket.h: ls_update_lazy, the code every solver of the kernels runs;
here compiled into the host emulator) against a plain restatement of MJX's two rules (mujoco.mjx._src.solver._linesearch: `swap` of
<= 3.1.3, `_in_bracket` of >= 3.1.4, cf. oracle/dial_oracle.c) -- the kernel's form of `_in_bracket` is min / max arithmetic on the keys,
which must take the same decisions for EVERY combination of keys, ties and zeros included."""
import ctypes
import itertools

import numpy as np
import pytest

import emu_lib


def _reference(rule_swap, lo, hi, k_lo_next, k_hi_next, k_mid):
    lo_sel = hi_sel = -1
    if rule_swap:
        moved = []
        c = lo > 0 or lo < k_lo_next
        lo, lo_sel = (k_lo_next, 0) if c else (lo, lo_sel); moved.append(c)
        c = k_mid < 0 and lo < k_mid
        lo, lo_sel = (k_mid, 2) if c else (lo, lo_sel); moved.append(c)
        c = hi < 0 or hi > k_hi_next
        hi, hi_sel = (k_hi_next, 1) if c else (hi, hi_sel); moved.append(c)
        c = k_mid > 0 and hi > k_mid
        hi, hi_sel = (k_mid, 2) if c else (hi, hi_sel); moved.append(c)
        return lo, hi, lo_sel, hi_sel, int(any(moved))
    in_bracket = lambda x, y: (x < y and y < 0) or (x > y and y > 0)   # noqa: E731
    moved = False
    for y, lane in ((k_lo_next, 0), (k_mid, 2), (k_hi_next, 1)):
        if in_bracket(lo, y):
            lo, lo_sel, moved = y, lane, True
    for y, lane in ((k_hi_next, 1), (k_mid, 2), (k_lo_next, 0)):
        if in_bracket(hi, y):
            hi, hi_sel, moved = y, lane, True
    return lo, hi, lo_sel, hi_sel, int(moved)


def _fkey(x):
    b = np.float32(x + np.float32(0)).view(np.int32)
    return int(b ^ ((b >> 31) & 0x7fffffff))


@pytest.mark.parametrize("rule_swap", [0, 1, 2])   # 2: `_in_bracket` in the boolean form (the capacity-dimension kernel)
def test_bracket_update_takes_the_reference_decisions(rule_swap):
    lib = ctypes.CDLL(emu_lib.build())
    edge = [_fkey(v) for v in (0.0, -0.0, 1e-45, -1e-45, 1.0, -1.0, 1.0000001, -1.0000001, np.inf, -np.inf)]
    cases = list(itertools.product(edge, repeat=5))                                  # every combination of the edge keys (incl. all ties)
    rng = np.random.default_rng(5)
    small = rng.integers(-4, 5, size=(30000, 5))                                     # dense ties around zero
    wide = rng.standard_normal((30000, 5)).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, size=(30000, 1))
    wide_keys = np.vectorize(_fkey)(wide)
    arr = np.concatenate([np.array(cases, np.int64), small, wide_keys]).astype(np.int32)
    out = np.zeros_like(arr)
    rc = lib.emu_ls_update(rule_swap, arr.shape[0], arr.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    ref = np.array([_reference(rule_swap == 1, *map(int, row)) for row in arr], np.int32)
    bad = np.nonzero((ref != out).any(1))[0]
    assert bad.size == 0, (arr[bad[:5]], ref[bad[:5]], out[bad[:5]])
