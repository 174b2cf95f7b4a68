"""Sample sharding of ``reverse_once`` over ``torch.distributed`` ranks (one process per GPU).

The reference has no multi-device code.  The N noisy samples are partitioned contiguously by rank; every
rank additionally rolls out the mean trajectory (the appended sample, dial_core.py:114) so ``rew_Ybar_i`` is
available everywhere.  The softmax couples all samples through the global std / max, hence per annealing iteration:

  1. ONE all-gather of the per-sample mean rewards (4 (N/world + 1) bytes per rank; the rollout kernel writes them
     straight into the send buffer) -> every rank forms the SAME N+1 weights with the same fixed-order reduction
     (bit-identical Ybar on all ranks, otherwise the plans diverge);
  2a. want_bars=False (every annealing iteration but the last): the weighted mean action is formed LOCALLY on every
     rank -- all candidate nodes are rebuilt from the noise, which with the in-kernel Philox generator is regenerated
     from (seed, iteration, sample index), so no noise array exists anywhere -- the all-gather is the ONLY collective;
  2b. want_bars=True (last iteration of a plan, whose qbar/qdbar/xbar the drivers read): all-reduce (sum) of the
     packed partial weighted sums [Ybar | qbar | qdbar | xbar] (5.4 KB for Go2).

Both messages are KB-sized, i.e. latency-bound on xGMI.  The collective staging buffers are allocated once (``ShardPlan``), results are fresh tensors; an iteration
is 1 rollout launch + the all-gather + 2 K4 launches (+ the all-reduce): the weights kernel reads the all-gather's receive buffer
directly (rounds 2-5: a packing launch), the weighted sums finish in the launch that forms them (rounds 1-5: a second launch).  The compute backend is
passed in as ``ctx`` (``dial_mpc_amd._lib.Context`` in production) so that the partition / collective logic is
testable with gloo on CPU against a stand-in context.
"""
from __future__ import annotations


def partition(N: int, rank: int, world: int):
    per = (N + world - 1) // world
    n_begin = min(rank * per, N)
    return per, n_begin, min(per, N - n_begin)


class ShardPlan:
    """Preallocated buffers of one rank's sharded iteration."""

    def __init__(self, ctx, rank: int, world: int, N: int, T: int, Hn1: int):
        import torch
        self.ctx, self.rank, self.world, self.N, self.T, self.Hn1 = ctx, rank, world, N, T, Hn1
        self.per, self.n_begin, self.n_local = partition(N, rank, world)
        # invariant the packing kernel relies on: rank 0's shard is always FULL (n_local == per), so its copy of the
        # mean-trajectory reward sits in slot `per` of its send buffer.  Later ranks may hold a ragged or an EMPTY shard
        # (N = 5 over 4 ranks: 2 + 2 + 1 + 0); such a rank rolls out the mean trajectory only.  Slots [n_local + 1, per]
        # of a ragged rank's send buffer are never read (the packing kernel reads the first N noisy entries + rank 0's
        # slot `per`).  Fewer samples than ranks is refused outright.
        if N < world:
            raise ValueError(f"sample sharding needs Nsample >= world size (Nsample = {N}, world = {world})")
        dev = ctx.torch_device
        f32 = dict(dtype=torch.float32, device=dev)
        self.send = torch.zeros(self.per + 1, **f32)               # [n_local noisy rewards ... | slot `per`: unused unless full]
        self.gathered = torch.empty(world * (self.per + 1), **f32)
        self.rews_all = torch.empty(N + 1, **f32)
        self.Ybar = torch.empty((Hn1, ctx.nu), **f32)
        self.packed = torch.empty(ctx.packed_size(), **f32)


def sharded_reverse_once(ctx, dist, rank: int, world: int, N: int, T: int, Hn1: int, packed_state, Ybar_i,
                         noise_scale, eps, want_bars: bool = True, plan: ShardPlan = None, rng=None):
    """eps: the GLOBAL noise array [N, Hn1, nu] (parity runs), or None with rng = (seed, counter): in-kernel noise."""
    if plan is None:
        plan = ShardPlan(ctx, rank, world, N, T, Hn1)
    per, n_begin, n_local = plan.per, plan.n_begin, plan.n_local
    # phase A: this rank's rollouts; rewards land in the all-gather send buffer ([0, n_local) noisy, [n_local] mean;
    # rank 0 always holds a full shard, so its mean reward sits in slot `per`, where dial_shard_pack_rewards reads it)
    mode = 1 if want_bars else 3   # bit 0: with the mean trajectory; bit 1 (DIAL_SHARD_LEAN): no per-step states / nodes are materialised
    if eps is None:
        seed, counter = rng
        ctx.shard_rollout_rng(packed_state, Ybar_i, noise_scale, seed, counter, n_begin, n_local, mode, plan.send)
    else:
        ctx.shard_rollout(packed_state, Ybar_i, noise_scale, eps[n_begin:n_begin + n_local], n_local, mode, plan.send)
    dist.all_gather_into_tensor(plan.gathered, plan.send)
    # Results go into FRESH tensors (allocator bookkeeping, no kernel): callers keep `info` dicts across ticks
    # (dial_core.main reads xbar of every tick at the very end), so nothing handed out may alias a buffer the next call
    # writes.  Only the collective's own staging buffers (send / gathered) are reused.
    import torch
    rews_all = torch.empty_like(plan.rews_all)
    # phase B reads the all-gather's receive buffer directly: the weights kernel puts the rewards in order on the way (-> rews_all)
    if not want_bars:
        Ybar = torch.empty_like(plan.Ybar)
        if eps is None:
            ctx.shard_ybar_gathered_rng(plan.gathered, world, per, N, seed, counter, Ybar_i, noise_scale, rews_all, Ybar)
        else:
            ctx.shard_ybar_gathered(plan.gathered, world, per, N, eps, Ybar_i, noise_scale, rews_all, Ybar)
        return Ybar, rews_all, None, None, None
    packed = torch.empty_like(plan.packed)
    ctx.shard_reduce_gathered(plan.gathered, world, per, N, n_begin, n_local, rank == 0, rews_all, packed)
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    nq, nv, nx, nu = ctx.nq, ctx.nv, ctx.nx, ctx.nu
    o = 0
    Ybar = packed[o:o + Hn1 * nu].reshape(Hn1, nu)
    o += Hn1 * nu
    qbar = packed[o:o + T * nq].reshape(T, nq)
    o += T * nq
    qdbar = packed[o:o + T * nv].reshape(T, nv)
    o += T * nv
    xbar = packed[o:o + T * nx].reshape(T, nx)
    return Ybar, rews_all, qbar, qdbar, xbar
