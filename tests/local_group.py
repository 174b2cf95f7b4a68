"""A `torch.distributed`-shaped collective group whose ranks are THREADS of one process sharing one GPU.

Test infrastructure: RCCL refuses two ranks on one device, and a round's GPU box has exactly one MI355X -- so this is
how the HIP kernels of the sharded path (`dial_shard_rollout[_rng]` with n_begin > 0, `pack_rewards_kernel` with
world > 1 and ragged shards, `dial_shard_ybar[_rng]` over a gathered layout, `dial_shard_reduce` on a strict sub-range)
execute at world 2 and 4 through the PRODUCTION code `core.sharding.sharded_reverse_once`, one `dial_create_sharded`
context per pseudo-rank.  Every thread enqueues on the device's default stream, so "the collective" is a device-side
copy / sum ordered behind the producers; the barriers only order the host threads."""
import threading

import torch


class ReduceOp:
    SUM = "sum"
    MAX = "max"


class LocalGroup:
    def __init__(self, world: int):
        self.world = world
        self._barrier = threading.Barrier(world)
        self._slots = [None] * world
        self._tls = threading.local()
        self.ReduceOp = ReduceOp

    def bind(self, rank: int):
        self._tls.rank = rank

    def _exchange(self, t):
        r = self._tls.rank
        self._slots[r] = t
        self._barrier.wait()
        parts = list(self._slots)
        self._barrier.wait()          # nobody overwrites a slot before everyone has read it
        return parts

    def all_gather_into_tensor(self, out, inp):
        parts = self._exchange(inp)
        torch.cat([p.reshape(-1) for p in parts], out=out.view(-1))
        self._barrier.wait()          # the inputs stay untouched until every rank has enqueued its copy

    def all_reduce(self, t, op=ReduceOp.SUM):
        parts = self._exchange(t)
        acc = parts[0].clone()
        for p in parts[1:]:           # fixed rank order on every rank: bit-identical results
            acc = acc + p if op == ReduceOp.SUM else torch.maximum(acc, p)
        self._barrier.wait()          # everyone has read everyone's input ...
        t.copy_(acc)                  # ... before anyone overwrites its own
        self._barrier.wait()

    def run(self, fn):
        """fn(rank) on `world` threads; returns the per-rank results, re-raises the first exception."""
        out, err = [None] * self.world, []

        def body(r):
            try:
                self.bind(r)
                out[r] = fn(r)
            except BaseException as e:      # noqa: BLE001
                err.append(e)
                self._barrier.abort()

        th = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if err:
            real = [e for e in err if not isinstance(e, threading.BrokenBarrierError)]
            raise (real or err)[0]
        return out
