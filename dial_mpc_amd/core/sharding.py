"""Sample sharding of ``reverse_once`` over ``torch.distributed`` ranks (one process per GPU).

The reference has no multi-device code.  The N noisy samples are partitioned contiguously by rank; every
rank additionally rolls out the mean trajectory (the appended sample, dial_core.py:114) so ``rew_Ybar_i`` is
available everywhere.  The softmax couples all samples through the global std / max, hence:

  1. all-gather of the per-sample mean rewards (4*N/world bytes per rank) -> every rank forms the SAME N+1
     weights with the same fixed-order reduction (bit-identical Ybar on all ranks, otherwise plans diverge);
  2a. want_bars=False (every annealing iteration but the last): the weighted mean action is formed LOCALLY on
     every rank from the full noise array (all candidate nodes are regenerated, 8e option (a)) -- the all-gather
     is the ONLY collective of the iteration;
  2b. want_bars=True (last iteration of a plan, whose qbar/qdbar/xbar the drivers read): all-reduce (sum) of
     the packed partial weighted sums [Ybar | qbar | qdbar | xbar] (5.4 KB for Go2).

Both messages are KB-sized, i.e. latency-bound on xGMI.  The compute backend is passed in as ``ctx``
(``dial_mpc_amd._lib.Context`` in production) so that the partition / collective logic is testable with
world_size-2 gloo on CPU against a stand-in context.
"""
from __future__ import annotations


def partition(N: int, rank: int, world: int):
    per = (N + world - 1) // world
    n_begin = min(rank * per, N)
    return per, n_begin, min(per, N - n_begin)


def sharded_reverse_once(ctx, dist, rank: int, world: int, N: int, T: int, Hn1: int, packed_state, Ybar_i,
                         noise_scale, eps, want_bars: bool = True):
    import torch
    dev = ctx.torch_device
    per, n_begin, n_local = partition(N, rank, world)
    eps_local = eps[n_begin:n_begin + n_local].contiguous()
    tmp = torch.empty(n_local + 1, dtype=torch.float32, device=dev)
    ctx.shard_rollout(packed_state, Ybar_i, noise_scale, eps_local, n_local, True, tmp)
    rews_local = torch.zeros(per + 1, dtype=torch.float32, device=dev)
    rews_local[:n_local] = tmp[:n_local]
    rews_local[per] = tmp[n_local]
    flat = torch.empty(world * (per + 1), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(flat, rews_local)
    gathered = flat.reshape(world, per + 1)
    rews_all = torch.cat([gathered[:, :per].reshape(-1)[:N], gathered[0, per:per + 1]]).contiguous()
    if not want_bars:
        Ybar = torch.empty((Hn1, ctx.nu), dtype=torch.float32, device=dev)
        ctx.shard_ybar(rews_all, N, eps.contiguous(), Ybar_i, noise_scale, Ybar)
        return Ybar, rews_all, None, None, None
    packed_out = torch.empty(ctx.packed_size(), dtype=torch.float32, device=dev)
    ctx.shard_reduce(rews_all, N, n_begin, n_local, rank == 0, packed_out)
    dist.all_reduce(packed_out, op=dist.ReduceOp.SUM)
    nq, nv, nx, nu = ctx.nq, ctx.nv, ctx.nx, ctx.nu
    o = 0
    Ybar = packed_out[o:o + Hn1 * nu].reshape(Hn1, nu)
    o += Hn1 * nu
    qbar = packed_out[o:o + T * nq].reshape(T, nq)
    o += T * nq
    qdbar = packed_out[o:o + T * nv].reshape(T, nv)
    o += T * nv
    xbar = packed_out[o:o + T * nx].reshape(T, nx)
    return Ybar, rews_all, qbar, qdbar, xbar
