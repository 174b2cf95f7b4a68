#!/usr/bin/env python3
"""Soak of the two crate examples (generic kernel instantiation): the reference's closed loop (dial_core.py:242-268) for
hundreds of control ticks each at the examples' own settings; every plan must be finite, the context's sticky status clean,
and the number of rollout samples that ever needed the overflow workspace is reported by proxy (the LDS footprint).
python tools/soak_crate.py [ticks]   (needs an MI355X)"""
import sys
import time

import numpy as np
import torch
import yaml

sys.path[:0] = ["."]
from dial_mpc_amd.core.dial_core import MBDPI, load_dial_and_env  # noqa: E402
from dial_mpc_amd.utils.io_utils import get_example_path  # noqa: E402

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for ex in ("unitree_go2_crate_climb", "unitree_h1_push_crate"):
    d = yaml.safe_load(open(get_example_path(ex + ".yaml")))
    dial_config, env_config, env = load_dial_and_env(d)
    mbdpi = MBDPI(dial_config, env)
    state = env.reset(0)
    Y0 = torch.zeros((dial_config.Hnode + 1, mbdpi.nu), device=mbdpi.device)
    rng, rews, track, t0 = 0, [], [], time.time()
    for t in range(ticks):
        state = env.step(state, Y0[0])
        rews.append(float(state.reward))
        q = state.pipeline_state.qpos
        track.append((float(q[0]), float(q[2]), float(q[-1])))
        Y0 = mbdpi.shift(Y0)
        for i in range(dial_config.Ndiffuse_init if t == 0 else dial_config.Ndiffuse):
            rng, Y0, info = mbdpi.reverse_once(state, rng, Y0, mbdpi.sigma_control * dial_config.traj_diffuse_factor ** i)
        if not bool(torch.isfinite(Y0).all()):
            print(ex, "NON-FINITE plan at tick", t)
            break
    torch.cuda.synchronize()
    mbdpi.ctx.status()
    tr = np.array(track)
    print(f"{ex}: {len(rews)} ticks, N={dial_config.Nsample}, wall {time.time() - t0:.1f} s ({1e3 * (time.time() - t0) / len(rews):.1f} ms per tick incl. host), "
          f"rewards finite: {bool(np.all(np.isfinite(rews)))}, reward first/last {rews[1]:.3f} / {rews[-1]:.3f}, "
          f"base x {tr[0, 0]:.2f} -> {tr[-1, 0]:.2f}, base z min {tr[:, 1].min():.2f} max {tr[:, 1].max():.2f}, last qpos entry {tr[0, 2]:.2f} -> {tr[-1, 2]:.2f}, "
          f"LDS per wavefront {mbdpi.ctx.lib.dial_lds_bytes(mbdpi.ctx.h)} B")
