"""``MBDPI`` -- the DIAL-MPC planner with the reference's Python surface (dial_mpc/core/dial_core.py:51-172)
on top of the HIP kernels, plus the synchronous-simulation driver ``main`` (dial_core.py:175-329).

Differences that are deliberate and documented:
* ``rng`` is a ``torch.Generator`` on the GPU (or an int seed) instead of a JAX key; the noise ``eps`` can
  also be passed explicitly (``eps=``) so that "identical noise draws" means *the same array*
  (JAX's threefry stream is version dependent).
* samples shard over the ranks of ``torch.distributed`` when it is initialised.
* the Brax HTML render / Flask server at the end of the reference's ``main`` are out of scope.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import time
from typing import Any, Dict, Optional

import numpy as np

from dial_mpc_amd import _abi, _lib
from dial_mpc_amd.core import spline
from dial_mpc_amd.core.dial_config import DialConfig


def make_cfg(args: DialConfig) -> "_abi.DialCfg":
    W = spline.node2u_matrix(args.Hsample, args.Hnode)
    V = spline.u2node_matrix(args.Hsample, args.Hnode)
    return _abi.fill(_abi.DialCfg(), dict(Nsample=args.Nsample, Hsample=args.Hsample, Hnode=args.Hnode,
                                          temp_sample=args.temp_sample, W=W, V=V))


def softmax_update(weights, Y0s, sigma, mu_0t):
    """dial_core.py:45-48 (kept for API parity; the kernels compute the same einsum in K4b)."""
    import torch
    return torch.einsum("n,nij->ij", weights, Y0s), sigma


class MBDPI:
    def __init__(self, args: DialConfig, env, device: Optional[int] = None, kernel_rng: bool = False,
                 force_sharded: bool = False, options: Optional[dict] = None):
        """kernel_rng=True: the noise is generated inside the rollout kernel (Philox keyed by args.seed and a call
        counter) instead of by torch.randn -- the production setting; parity runs pass `eps` explicitly.
        force_sharded (measurement hook): run the sharded code path, collectives included, on a 1-rank process group.
        options: `dial_options` fields for the context (launch-shape / measurement switches, include/dial_mpc.h)."""
        import torch
        self.kernel_rng = bool(kernel_rng)
        self._rng_counter = 0
        self._plan = None    # preallocated buffers of the sharded iteration (core/sharding.py)
        self._force_sharded = bool(force_sharded)
        self.args = args
        self.env = env
        self.nu = env.action_size
        self.update_fn = {"mppi": softmax_update}[args.update_method]  # KeyError for anything else, as upstream

        sigma_control = args.horizon_diffuse_factor ** np.arange(args.Hnode + 1)[::-1]  # dial_core.py:66-70
        sigma_control = sigma_control * args.sigma_scale
        self.ctrl_dt = 0.02  # hard-coded upstream (dial_core.py:74)
        self.step_us_np = np.linspace(0, self.ctrl_dt * args.Hsample, args.Hsample + 1)
        self.step_nodes_np = np.linspace(0, self.ctrl_dt * args.Hsample, args.Hnode + 1)
        self.node_dt = self.ctrl_dt * args.Hsample / args.Hnode

        self.cfg = make_cfg(args)
        # sample sharding over torch.distributed ranks (one process per GPU)
        self.rank, self.world = 0, 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.rank, self.world = dist.get_rank(), dist.get_world_size()
        except Exception:
            pass
        from dial_mpc_amd.core.sharding import partition
        self._per, self.n_begin, self.n_local = partition(args.Nsample, self.rank, self.world)
        # a rank's rollout scratch is sized by its own shard, not by the global sample count
        self.ctx = _lib.Context(env.make_model(), env.make_task(), self.cfg, device,
                                n_local_cap=None if self.world == 1 else self._per, options=options)
        if hasattr(env, "bind_device"):
            env.bind_device(self.ctx.device)
        dev = self.ctx.torch_device
        self.device = dev
        self.sigma_control = torch.as_tensor(sigma_control.copy(), dtype=torch.float32, device=dev)
        self.step_us = torch.as_tensor(self.step_us_np, dtype=torch.float32, device=dev)
        self.step_nodes = torch.as_tensor(self.step_nodes_np, dtype=torch.float32, device=dev)
        self.W = torch.as_tensor(spline.node2u_matrix(args.Hsample, args.Hnode), dtype=torch.float32, device=dev)
        self.V = torch.as_tensor(spline.u2node_matrix(args.Hsample, args.Hnode), dtype=torch.float32, device=dev)

    # ---- spline maps (constant matrices; dial_core.py:82-101)
    def node2u(self, nodes):
        return self.W @ nodes

    def u2node(self, us):
        return self.V @ us

    def node2u_vmap(self, Y):          # (Hnode+1, nu) -> (Hsample+1, nu)
        return self.W @ Y

    def u2node_vmap(self, u):
        return self.V @ u

    def node2u_vvmap(self, Ys):        # (B, Hnode+1, nu) -> (B, Hsample+1, nu)
        import torch
        return torch.einsum("tk,bka->bta", self.W, Ys)

    def u2node_vvmap(self, us):
        import torch
        return torch.einsum("kt,bta->bka", self.V, us)

    def rollout_us(self, state, us):
        rewss, qss, qdss, xss = self.ctx.rollout(_packed(state), us[None].contiguous())
        return rewss[0], dict(q=qss[0], qd=qdss[0], x_pos=xss[0].reshape(us.shape[0], -1, 3))

    def rollout_us_vmap(self, state, us):
        rewss, qss, qdss, xss = self.ctx.rollout(_packed(state), us.contiguous())
        B, T = us.shape[:2]
        return rewss, dict(q=qss, qd=qdss, x_pos=xss.reshape(B, T, -1, 3))

    # ---- one annealing iteration (dial_core.py:103-145)
    def sample_eps(self, rng):
        import torch
        gen = _generator(rng, self.device)
        eps = torch.randn((self.args.Nsample, self.args.Hnode + 1, self.nu), generator=gen, device=self.device,
                          dtype=torch.float32)
        return gen, eps

    def reverse_once(self, state, rng, Ybar_i, noise_scale, eps=None, want_bars: bool = True):
        """want_bars=False skips qbar/qdbar/xbar (None in info): the rollouts then do not write their per-step states and K4b
        sums the candidate nodes only; in a sharded run this also drops the all-reduce, leaving one collective per annealing
        iteration (core/sharding.py).  The drivers ask for the bars on the last iteration of a plan only, as upstream reads
        them (dial_core.py:262-264, dial_plan.py:214-215)."""
        import torch
        packed = _packed(state)
        if eps is None and self.kernel_rng:
            Yb = torch.as_tensor(Ybar_i, dtype=torch.float32, device=self.device).contiguous()
            nsc = torch.as_tensor(noise_scale, dtype=torch.float32, device=self.device).reshape(-1).contiguous()
            T, nb1 = self.args.Hsample + 1, self.ctx.nbody - 1
            counter = self._rng_counter
            self._rng_counter += 1
            if self.world == 1 and not self._force_sharded:
                out = self.ctx.reverse_once_rng(packed, Yb, nsc, int(self.args.seed), counter, want_bars=want_bars)
                Ybar, rews, qbar, qdbar, xbar = out["Ybar"], out["rews"], out["qbar"], out["qdbar"], out["xbar"]
            else:   # sharded: every rank draws its own shard's noise (and, for the mean action, everybody's) in-kernel
                Ybar, rews, qbar, qdbar, xbar = self._reverse_once_sharded(packed, Yb, nsc, None, want_bars,
                                                                           rng=(int(self.args.seed), counter))
            return rng, Ybar, {"rews": rews, "qbar": qbar, "qdbar": qdbar,
                               "xbar": xbar.reshape(T, nb1, 3) if xbar is not None else None, "new_noise_scale": nsc}
        if eps is None:
            rng, eps = self.sample_eps(rng)
        Ybar_i = torch.as_tensor(Ybar_i, dtype=torch.float32, device=self.device).contiguous()
        noise_scale = torch.as_tensor(noise_scale, dtype=torch.float32, device=self.device).reshape(-1).contiguous()
        T, nb1 = self.args.Hsample + 1, self.ctx.nbody - 1
        if self.world == 1 and not self._force_sharded:
            out = self.ctx.reverse_once(packed, Ybar_i, noise_scale, eps.contiguous(), want_bars=want_bars)
            Ybar, rews = out["Ybar"], out["rews"]
            qbar, qdbar, xbar = out["qbar"], out["qdbar"], out["xbar"]
        else:
            Ybar, rews, qbar, qdbar, xbar = self._reverse_once_sharded(packed, Ybar_i, noise_scale, eps, want_bars)
        info = {"rews": rews, "qbar": qbar, "qdbar": qdbar, "xbar": xbar.reshape(T, nb1, 3) if xbar is not None else None,
                "new_noise_scale": noise_scale}
        return rng, Ybar, info

    def _reverse_once_sharded(self, packed, Ybar_i, noise_scale, eps, want_bars=True, rng=None):
        import torch.distributed as dist
        from dial_mpc_amd.core.sharding import ShardPlan, sharded_reverse_once
        if self._plan is None:
            self._plan = ShardPlan(self.ctx, self.rank, self.world, self.args.Nsample, self.args.Hsample + 1,
                                   self.args.Hnode + 1)
        return sharded_reverse_once(self.ctx, dist, self.rank, self.world, self.args.Nsample, self.args.Hsample + 1,
                                    self.args.Hnode + 1, packed, Ybar_i, noise_scale,
                                    eps.contiguous() if eps is not None else None, want_bars, plan=self._plan, rng=rng)

    # ---- receding-horizon shift (dial_core.py:160-172)
    def shift(self, Y):
        import torch
        Y = torch.as_tensor(Y, dtype=torch.float32, device=self.device).contiguous()
        return self.ctx.shift(Y)

    def shift_Y_from_u(self, u, n_step):
        import torch
        u = torch.roll(u, -n_step, dims=0)
        u[-n_step:] = 0
        return self.u2node_vmap(u)


def _packed(state):
    return state.packed if hasattr(state, "packed") else state


def _generator(rng, device):
    import torch
    if isinstance(rng, torch.Generator):
        return rng
    gen = torch.Generator(device=device)
    gen.manual_seed(int(rng) if rng is not None else 0)
    return gen


def load_dial_and_env(config_dict: Dict[str, Any]):
    import dial_mpc_amd.envs as dial_envs
    from dial_mpc_amd.utils.io_utils import load_dataclass_from_dict
    dial_config = load_dataclass_from_dict(DialConfig, config_dict)
    env_config_type = dial_envs.get_config(dial_config.env_name)
    env_config = load_dataclass_from_dict(env_config_type, config_dict, convert_list_to_array=True)
    env_config.seed = int(dial_config.seed)      # randomize_tasks draws from the run's seed (envs/base_env.py)
    env = dial_envs.get_environment(dial_config.env_name, config=env_config)
    return dial_config, env_config, env


def main():
    """Synchronous simulation driver: the body of the reference's ``main`` (dial_core.py:175-329)."""
    import torch
    import yaml
    from dial_mpc_amd.examples import examples
    from dial_mpc_amd.utils.io_utils import get_example_path

    parser = argparse.ArgumentParser()
    group = parser.add_mutually_exclusive_group(required=True)
    group.add_argument("--config", type=str, default=None)
    group.add_argument("--example", type=str, default=None)
    group.add_argument("--list-examples", action="store_true")
    parser.add_argument("--custom-env", type=str, default=None, help="Custom environment to import dynamically")
    parser.add_argument("--n-steps", type=int, default=None, help="override n_steps from the YAML")
    args = parser.parse_args()

    if args.list_examples:
        print("Examples:")
        for example in examples:
            print(f"  {example}")
        return
    if args.custom_env is not None:
        sys.path.append(os.getcwd())
        importlib.import_module(args.custom_env)
    if args.example is not None:
        config_dict = yaml.safe_load(open(get_example_path(args.example + ".yaml")))
    else:
        config_dict = yaml.safe_load(open(args.config))
    dial_config, env_config, env = load_dial_and_env(config_dict)
    if args.n_steps is not None:
        dial_config.n_steps = args.n_steps
    print("Creating environment")
    mbdpi = MBDPI(dial_config, env)
    rng = _generator(dial_config.seed, mbdpi.device)
    state = env.reset(rng)
    Y0 = torch.zeros((dial_config.Hnode + 1, mbdpi.nu), dtype=torch.float32, device=mbdpi.device)

    rews, rews_plan, rollout, infos = [], [], [], []
    plan_ms = []
    for t in range(dial_config.n_steps):
        state = env.step(state, Y0[0])                      # dial_core.py:245
        rollout.append(state)
        rews.append(float(state.reward))
        Y0 = mbdpi.shift(Y0)                                # :251
        n_diffuse = dial_config.Ndiffuse_init if t == 0 else dial_config.Ndiffuse
        t0 = time.time()
        factors = mbdpi.sigma_control[None, :] * (dial_config.traj_diffuse_factor **
                                                  torch.arange(n_diffuse, device=mbdpi.device))[:, None]
        info = None
        for i in range(n_diffuse):                          # lax.scan(reverse_scan) :262-264
            rng, Y0, info = mbdpi.reverse_once(state, rng, Y0, factors[i], want_bars=(i == n_diffuse - 1))
        torch.cuda.synchronize()
        mbdpi.ctx.status()                                   # raises if an asynchronous launch of this tick gave up
        plan_ms.append((time.time() - t0) * 1e3)
        rews_plan.append(float(info["rews"].mean()))         # :266 mean over all N+1 sample rewards of the last iteration
        infos.append(info)
        if t % 20 == 0:
            print(f"step {t:4d}  rew {rews[-1]: .3e}  plan {plan_ms[-1]:.2f} ms")
    print(f"mean reward = {np.mean(rews):.2e}")
    if len(plan_ms) > 1:
        print(f"plan latency p50 = {np.percentile(plan_ms[1:], 50):.3f} ms (tick = 20 ms)")

    os.makedirs(dial_config.output_dir, exist_ok=True)
    timestamp = time.strftime("%Y%m%d-%H%M%S")
    states_arr, pred_arr = result_arrays(rollout, infos)
    np.save(os.path.join(dial_config.output_dir, f"{timestamp}_states"), states_arr)
    np.save(os.path.join(dial_config.output_dir, f"{timestamp}_predictions"), pred_arr)


def result_arrays(rollout, infos):
    """The two artefacts of the reference's ``main`` (dial_core.py:305-323):
    ``states``      (n_steps, 1 + nq + nv + nu): [step index | qpos | qvel | ctrl] of every executed state;
    ``predictions`` (n_steps, Hsample+1, nbody-1, 3): the weighted-mean body positions ``xbar`` of the LAST annealing
    iteration of every tick -- upstream ``infos[i]["xbar"]`` comes out of ``lax.scan`` with a leading diffusion axis
    and ``[-1]`` selects that iteration; here ``infos[i]`` already is the last iteration's dict."""
    data, xdata = [], []
    for i, st in enumerate(rollout):
        ps = st.pipeline_state
        data.append(np.concatenate([[i], _np(ps.qpos), _np(ps.qvel), _np(ps.ctrl)]))
        xdata.append(_np(infos[i]["xbar"]))
    return np.array(data), np.array(xdata)


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


if __name__ == "__main__":
    main()
