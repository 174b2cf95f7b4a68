"""world_size-2 and -4 gloo tests (CPU) of the sample-sharding / collective logic of reverse_once.

The compute backend is a stand-in context (wave emulator for the rollouts + NumPy for the K4 algebra) --
the production backend is dial_mpc_amd._lib.Context on a GPU; what is under test here is the partition,
the all-gather layout, the identical-weights property and the all-reduce of the packed partial sums:
the sharded result must equal the unsharded one."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import seeded_inputs, setup_case


class FakeCtx:
    """Implements shard_rollout / shard_reduce / packed_size with the emulator + NumPy on CPU tensors."""

    def __init__(self, model, task, cfg):
        import emu_lib
        self.emu = emu_lib.Emu(model, task, cfg)
        self.cfg = cfg
        self.torch_device = torch.device("cpu")
        self.nq, self.nv, self.nu, self.nx = model.nq, model.nv, model.nu, (model.nbody - 1) * 3
        self.last = None

    def packed_size(self):
        T, Hn1 = self.cfg.Hsample + 1, self.cfg.Hnode + 1
        return Hn1 * self.nu + T * (self.nq + self.nv + self.nx)

    def shard_rollout(self, state, Ybar, noise_scale, eps_local, n_local, with_mean, rews_out):
        import ctypes
        cfg = self.cfg
        Hn1, T, B = cfg.Hnode + 1, cfg.Hsample + 1, n_local + 1
        e = self.emu
        Y0s = np.zeros((B, Hn1, self.nu), np.float32)
        rewss = np.zeros((B, T), np.float32)
        rews = np.zeros(B, np.float32)
        qss = np.zeros((B, T, self.nq), np.float32)
        qdss = np.zeros((B, T, self.nv), np.float32)
        xss = np.zeros((B, T, self.nx), np.float32)
        ns = noise_scale.numpy().astype(np.float32)
        rc = e.lib.emu_rollout(ctypes.byref(e.model), ctypes.byref(e.task), ctypes.byref(cfg),
                               e._p(e._a(state.numpy())), None, e._p(e._a(eps_local.numpy())),
                               e._p(e._a(Ybar.numpy())), e._p(ns), int(ns.size), n_local, B, T, Hn1, e._p(Y0s),
                               e._p(rewss), e._p(rews), e._p(qss), e._p(qdss), e._p(xss), 0, 0, None)
        assert rc == 0
        rews_out[:B].copy_(torch.from_numpy(rews))   # the kernel writes n_local + 1 entries of the (per + 1)-sized send buffer
        self.last = (Y0s, qss, qdss, xss)

    def shard_pack_rewards(self, gathered, world, per, n_total, rews_all):
        g = gathered.numpy().reshape(world, per + 1)
        rews_all.copy_(torch.from_numpy(np.concatenate([g[:, :per].reshape(-1)[:n_total], g[0, per:per + 1]])))

    # the fused entry points of round 6 (phase B straight from the all-gather's receive buffer)
    def shard_ybar_gathered(self, gathered, world, per, n_total, eps_all, Ybar, noise_scale, rews_all, Ybar_out):
        self.shard_pack_rewards(gathered, world, per, n_total, rews_all)
        self.shard_ybar(rews_all, n_total, eps_all, Ybar, noise_scale, Ybar_out)

    def shard_reduce_gathered(self, gathered, world, per, n_total, n_begin, n_local, include_mean, rews_all, packed_out):
        self.shard_pack_rewards(gathered, world, per, n_total, rews_all)
        self.shard_reduce(rews_all, n_total, n_begin, n_local, include_mean, packed_out)

    def shard_ybar(self, rews_all, n_total, eps_all, Ybar, noise_scale, Ybar_out):
        r = rews_all.numpy().astype(np.float32)
        logp = (r - r[-1]) / r.std() / np.float32(self.cfg.temp_sample)
        w = np.exp(logp - logp.max())
        w = (w / w.sum()).astype(np.float32)
        ns, Yb = noise_scale.numpy(), Ybar.numpy()
        Y0s = eps_all.numpy() * (ns[None, :, None] if ns.size > 1 else ns[0]) + Yb
        Y0s[:, 0] = Yb[0]
        Y0s = np.clip(np.concatenate([Y0s, Yb[None]], 0), -1, 1)
        Ybar_out.copy_(torch.from_numpy(np.einsum("n,nka->ka", w, Y0s).astype(np.float32)))

    def shard_reduce(self, rews_all, n_total, n_begin, n_local, include_mean, packed_out):
        r = rews_all.numpy().astype(np.float32)
        logp = (r - r[-1]) / r.std() / np.float32(self.cfg.temp_sample)
        w = np.exp(logp - logp.max())
        w = (w / w.sum()).astype(np.float32)
        wl = np.concatenate([w[n_begin:n_begin + n_local], [w[n_total] if include_mean else 0.0]]).astype(np.float32)
        parts = [np.einsum("n,nc->c", wl, a.reshape(a.shape[0], -1)) for a in self.last]
        packed_out.copy_(torch.from_numpy(np.concatenate(parts).astype(np.float32)))


def _worker(rank, world, port, N, H, ret):
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dial_mpc_amd.core.sharding import sharded_reverse_once
        import oracle as O
        dc, env, model, task, cfg = setup_case("unitree_go2_trot", N, H, per_rollout=True)
        ctx = FakeCtx(model, task, cfg)
        o32 = O.Oracle(model, task, cfg, np.float32)
        s0, _, _ = o32.env_reset(env._init_q, np.zeros(18))
        eps, sigma, Ybar = seeded_inputs(dc, 12, seed=0)
        from dial_mpc_amd.core.sharding import ShardPlan
        plan = ShardPlan(ctx, rank, world, N, H + 1, dc.Hnode + 1)        # one set of buffers, reused like the driver does
        args = (ctx, dist, rank, world, N, H + 1, dc.Hnode + 1, torch.from_numpy(s0), torch.from_numpy(Ybar),
                torch.from_numpy(sigma), torch.from_numpy(eps))
        # the driver's pattern: want_bars only on the last annealing iteration of a plan
        one = sharded_reverse_once(*args, want_bars=False, plan=plan)     # single-collective variant
        assert one[2] is None and one[3] is None and one[4] is None
        Yb_single = one[0].numpy().copy()
        out = sharded_reverse_once(*args, want_bars=True, plan=plan)
        ret[rank] = [o.numpy().copy() for o in out] + [Yb_single]
        # callers keep `info` dicts across ticks (dial_core.main appends them and reads xbar at the very end): what a call
        # returned must survive the next call on the same plan buffers
        args2 = args[:8] + (out[0],) + args[9:]                           # next iteration starts from the new mean plan
        out2 = sharded_reverse_once(*args2, want_bars=True, plan=plan)
        for a, b in zip(out, ret[rank][:5]):
            assert np.array_equal(a.numpy(), b), "an earlier call's result was overwritten by the next call"
        assert not np.array_equal(out2[4].numpy(), out[4].numpy())        # the second plan differs from the first
        one2 = sharded_reverse_once(*args2, want_bars=False, plan=plan)
        assert np.array_equal(one[0].numpy(), Yb_single) and one2[0].data_ptr() != one[0].data_ptr()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,world", [(64, 2), (37, 2), (64, 4), (37, 4)])   # 37: ragged last shard
def test_sharded_reverse_once_equals_unsharded(N, world):
    import oracle as O
    H = 8
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, N, H, ret), nprocs=world, join=True)
    dc, env, model, task, cfg = setup_case("unitree_go2_trot", N, H, per_rollout=True)
    o32 = O.Oracle(model, task, cfg, np.float32)
    s0, _, _ = o32.env_reset(env._init_q, np.zeros(18))
    eps, sigma, Ybar = seeded_inputs(dc, 12, seed=0)
    ref = o32.reverse_once(s0, Ybar, sigma, eps)
    for rank in range(world):
        Yb, rews, qbar, qdbar, xbar, Yb_single = ret[rank]
        assert np.allclose(Yb_single, ref["Ybar"], atol=2e-3) and np.allclose(Yb_single, Yb, atol=1e-5)
        assert np.allclose(rews, ref["rews"], atol=1e-3)
        assert np.allclose(Yb, ref["Ybar"], atol=2e-3) and np.allclose(qbar, ref["qbar"], atol=5e-3)
        assert np.allclose(xbar, ref["xbar"].reshape(xbar.shape), atol=5e-3)
    with pytest.raises(ValueError):
        from dial_mpc_amd.core.sharding import ShardPlan
        ShardPlan(type("C", (), dict(torch_device=torch.device("cpu"), nu=12, packed_size=lambda self: 8))(), 0, 8, 5, 9, 5)
    for rank in range(1, world):                        # every rank holds bit-identical results
        for a, b in zip(ret[0], ret[rank]):
            assert np.array_equal(a, b)
