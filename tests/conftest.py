import os
import sys

import numpy as np
import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- tolerances of the fp32 parity gate (HIP vs fp32 oracle on identical inputs).  Measured agreement
# is ~1e-5; the gate leaves a margin for rollouts that amplify rounding differences.
TOL = dict(rewss=dict(rtol=2e-3, atol=2e-3), q=dict(rtol=0, atol=1e-3), qd=dict(rtol=2e-3, atol=2e-2),
           x=dict(rtol=0, atol=1e-3), weights=dict(rtol=2e-2, atol=2e-4), Ybar=dict(rtol=0, atol=2e-3),
           bar=dict(rtol=0, atol=5e-3))


def setup_case(example: str, N: int, H: int, Hnode=None):
    """(dial_config, env, model, task, cfg) for an example YAML with N / H overridden (BASELINE configs)."""
    from dial_mpc_amd.core.dial_core import load_dial_and_env, make_cfg
    from dial_mpc_amd.utils.io_utils import get_example_path
    d = yaml.safe_load(open(get_example_path(example + ".yaml")))
    d["Nsample"], d["Hsample"] = N, H
    if Hnode is not None:
        d["Hnode"] = Hnode
    dc, ec, env = load_dial_and_env(d)
    return dc, env, env.make_model(), env.make_task(), make_cfg(dc)


def seeded_inputs(dc, nu, seed=0, Ybar_scale=0.0):
    """The 'documented host generator' of SURVEY 8d: eps ~ N(0,1) fp32 from numpy PCG64(seed)."""
    rng = np.random.default_rng(seed)
    eps = rng.standard_normal((dc.Nsample, dc.Hnode + 1, nu)).astype(np.float32)
    sigma = (dc.horizon_diffuse_factor ** np.arange(dc.Hnode + 1)[::-1] * dc.sigma_scale).astype(np.float32)
    Ybar = (Ybar_scale * rng.uniform(-1, 1, (dc.Hnode + 1, nu))).astype(np.float32)
    return eps, sigma, Ybar


from dial_mpc_amd.utils.synthetic import perturbed_state  # noqa: E402,F401


CASES = [("unitree_go2_trot", 64, 8), ("unitree_go2_seq_jump", 48, 16), ("unitree_h1_jog", 32, 16),
         ("unitree_h1_loco", 32, 20)]
