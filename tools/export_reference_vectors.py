#!/usr/bin/env python3
"""Export golden vectors from the JAX reference (run on a machine that HAS jax, brax, mujoco, jax_cosmo and
the reference checkout on PYTHONPATH; it cannot run in the build container).  Pinned environment + the exact
command sequence: tools/reference_env.txt.

    python tools/export_reference_vectors.py --example unitree_go2_trot --nsample 64 --hsample 8 --out ref.npz

It monkey-patches `jax.random.normal` so that `MBDPI.reverse_once` consumes a fixed NumPy-generated `eps`
(the same generator as tests/conftest.py: seeded_inputs), then dumps state, eps, Ybar_in, noise_scale, rewss,
qss, qdss, xss, Ybar, weights.  Place the file under tests/golden/reference/ — tests/test_reference_vectors.py
compares the oracle (CPU leg) and the HIP path (`-m gpu` leg) against it with the fp32 tolerances of conftest.TOL
and otherwise reports an XFAIL "golden vectors absent -- parity vs JAX unpinned".
"""
import argparse

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--example", default="unitree_go2_trot")
    ap.add_argument("--nsample", type=int, default=64)
    ap.add_argument("--hsample", type=int, default=8)
    ap.add_argument("--out", default="reference_vectors.npz")
    args = ap.parse_args()

    import jax
    import jax.numpy as jnp
    import yaml
    import brax.envs as brax_envs
    import dial_mpc.envs as dial_envs
    from dial_mpc.core.dial_config import DialConfig
    from dial_mpc.core.dial_core import MBDPI
    from dial_mpc.utils.io_utils import get_example_path, load_dataclass_from_dict

    cfg = yaml.safe_load(open(get_example_path(args.example + ".yaml")))
    cfg["Nsample"], cfg["Hsample"] = args.nsample, args.hsample
    dial_config = load_dataclass_from_dict(DialConfig, cfg)
    env_config = load_dataclass_from_dict(dial_envs.get_config(dial_config.env_name), cfg, convert_list_to_array=True)
    env = brax_envs.get_environment(dial_config.env_name, config=env_config)
    mbdpi = MBDPI(dial_config, env)
    state = jax.jit(env.reset)(jax.random.PRNGKey(0))

    rng = np.random.default_rng(0)
    eps = rng.standard_normal((dial_config.Nsample, dial_config.Hnode + 1, mbdpi.nu)).astype(np.float32)
    sigma = np.asarray(mbdpi.sigma_control, dtype=np.float32)
    Ybar = (0.2 * rng.uniform(-1, 1, (dial_config.Hnode + 1, mbdpi.nu))).astype(np.float32)
    jax.random.normal = lambda key, shape, dtype=jnp.float32: jnp.asarray(eps)      # fixed noise

    Y0s = jnp.clip(jnp.concatenate([(jnp.asarray(eps) * sigma[None, :, None] + Ybar).at[:, 0].set(Ybar[0]),
                                    Ybar[None]], 0), -1, 1)
    us = mbdpi.node2u_vvmap(Y0s)
    rewss, ps = mbdpi.rollout_us_vmap(state, us)
    _, Ybar_out, info = mbdpi.reverse_once(state, jax.random.PRNGKey(1), jnp.asarray(Ybar), jnp.asarray(sigma))
    ps0 = state.pipeline_state
    import importlib.metadata as md
    versions = {p: md.version(p) for p in ("jax", "jaxlib", "mujoco", "mujoco-mjx", "brax", "jax-cosmo", "numpy")}
    # The contact ARRAY of the reference's MJX release, for the two crate envs whose rewards read it by position
    # (unitree_go2_env.py:750, unitree_h1_env.py:476-478, 525-526): geom ids per slot, so that the lookup by geom identity on
    # our side (dial_task.crate_contact / pc_*; DESIGN.md section 1 "crate scenes") can be checked against the real order.
    extra = {}
    con = getattr(ps0, "contact", None)
    if con is not None and getattr(con, "geom", None) is not None:
        extra.update(contact_geom=np.asarray(con.geom), contact_dist=np.asarray(con.dist), contact_pos=np.asarray(con.pos),
                     geom_names=np.array([env.sys.mj_model.geom(i).name for i in range(env.sys.mj_model.ngeom)]))
    np.savez_compressed(args.out, **extra, versions=np.array(repr(versions)), qpos=np.asarray(ps0.qpos), qvel=np.asarray(ps0.qvel),
                        qacc_warmstart=np.asarray(ps0.qacc_warmstart), eps=eps, noise_scale=sigma, Ybar_in=Ybar,
                        us=np.asarray(us), rewss=np.asarray(rewss), qss=np.asarray(ps.q), qdss=np.asarray(ps.qd),
                        xss=np.asarray(ps.x.pos), Ybar=np.asarray(Ybar_out), rews=np.asarray(info["rews"]),
                        qbar=np.asarray(info["qbar"]), xbar=np.asarray(info["xbar"]))
    print("wrote", args.out)


if __name__ == "__main__":
    main()
