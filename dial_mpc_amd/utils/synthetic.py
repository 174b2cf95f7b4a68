"""Synthetic benchmark / test inputs: perturbed initial states."""
import numpy as np


def perturbed_state(env, seed):
    """Home pose + U(-0.1,0.1) rad on the joints, base z in U(0.25,0.35) m (Go2 only), qd ~ N(0, 0.5)."""
    rng = np.random.default_rng(seed)
    q = np.array(env._init_q, dtype=np.float64)
    q[7:] += rng.uniform(-0.1, 0.1, q.shape[0] - 7)
    if q.shape[0] == 19:
        q[2] = rng.uniform(0.25, 0.35)
    qd = rng.normal(0, 0.5, env.sys.nv)
    return q, qd
