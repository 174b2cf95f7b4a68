"""C ABI: the header parses, the ctypes mirrors agree with the compiled structs, the HIP library loads on a
CPU-only box and exports every symbol include/dial_mpc.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from dial_mpc_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", open(_abi.HEADER).read(), flags=re.S)
    text = text.split("product C ABI")[-1] if "product C ABI" in open(_abi.HEADER).read() else text
    return sorted(set(re.findall(r"\b(dial_[a-z_]+)\s*\(", text)) - {"dial_state_size"})


def test_header_structs_match_oracle_build():
    import oracle as O
    O.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle_f64.so"))
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.oracle_abi_sizes(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    assert (a.value, b.value, c.value) == (ctypes.sizeof(_abi.DialModel), ctypes.sizeof(_abi.DialTask),
                                           ctypes.sizeof(_abi.DialCfg))


def test_hip_library_loads_and_exports_every_declared_symbol():
    _lib.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert set(_lib.EXPORTED) <= set(declared)
    for name in declared:
        assert hasattr(lib, name), f"libdialhip.so does not export {name}"
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.dial_abi_sizes(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
    assert (a.value, b.value, c.value) == (ctypes.sizeof(_abi.DialModel), ctypes.sizeof(_abi.DialTask),
                                           ctypes.sizeof(_abi.DialCfg))


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from conftest import setup_case
    dc, env, model, task, cfg = setup_case("unitree_go2_trot", 8, 8)
    with pytest.raises(_lib.DialHipError):
        _lib.Context(model, task, cfg)
    with pytest.raises(_lib.DialHipError):
        env.reset(0)


def test_fill_rejects_models_exceeding_capacity():
    with pytest.raises(ValueError):
        _abi.fill(_abi.DialModel(), dict(qpos0=list(range(1000))))
