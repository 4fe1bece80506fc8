"""Sharding of a batch of independent problems over the GPUs of one node, and the single exchange
of the path: one gather of the converged results to rank 0 (RCCL over xGMI when the process group is
"nccl"; "gloo" in the CPU tests).  Problems are independent, so there is no
collective inside a solve -- each rank owns a block of the batch, contiguous or interleaved.
"""
import torch
import torch.distributed as dist


def shard_bounds(batch, rank, world):
    """Contiguous block [lo, hi) of a global batch owned by `rank`: ceil(batch/world) problems per rank."""
    per = -(-batch // world)
    lo = min(batch, rank * per)
    return lo, min(batch, lo + per)


def shard_size(batch, rank, world, interleaved=False):
    """problems owned by `rank`"""
    if interleaved:
        return max(0, -(-(batch - rank) // world))
    lo, hi = shard_bounds(batch, rank, world)
    return hi - lo


def shard_indices(batch, rank, world, interleaved=False):
    """global problem ids owned by `rank`.  interleaved: problem k -> rank k mod world, which spreads any trend of
    the iteration counts along the batch (sorted condition numbers, say) evenly over the GPUs."""
    if interleaved:
        return torch.arange(rank, batch, world) if rank < batch else torch.arange(0)
    lo, hi = shard_bounds(batch, rank, world)
    return torch.arange(lo, hi)


class Gatherer:
    """The one exchange of the job, with every buffer allocated up front (nothing is allocated inside a timed step).

    shapes: {name: (per-problem shape tuple, dtype)}.  Every rank contributes ceil(batch/world) rows per array --
    ragged shards are padded here, not by the caller -- and rank `dst` receives them in one [world, per, ...] buffer
    per array whose slices are the gather list (the blocks land in place; no concatenation pass).
    """

    def __init__(self, batch, shapes, device, dst=0, group=None, interleaved=False):
        self.batch, self.dst, self.group, self.interleaved = batch, dst, group, interleaved
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.per = -(-batch // self.world)
        self.mine = shard_size(batch, self.rank, self.world, interleaved)
        # RCCL ("nccl") moves device tensors directly over xGMI; gloo (CPU tests, rehearsals) needs host tensors
        self.stage = dist.get_backend(group) == "gloo"
        self.device = device
        buf_dev = torch.device("cpu") if self.stage else device
        self.send, self.recv, self.pending = {}, {}, []
        for name, (shape, dtype) in shapes.items():
            self.send[name] = torch.zeros((self.per,) + tuple(shape), dtype=dtype, device=buf_dev)
            if self.rank == dst:
                self.recv[name] = torch.empty((self.world, self.per) + tuple(shape), dtype=dtype, device=buf_dev)

    def gather(self, results, assemble=True, overlap=False):
        """results: {name: tensor [shard, ...]} of this rank.  Returns {name: [batch, ...]} in global problem order on
        dst (views of the receive buffers where the layout allows: copy what must outlive the next call), None
        elsewhere.  assemble=False: only move the data (rank dst then holds it as self.recv[name][rank, row]); the
        interleaved order needs one reordering copy to become a flat [batch, ...] array, which a caller that only
        wants the exchange done -- bench.py's timed step -- can skip.

        overlap=True (implies assemble=False): the collectives are only ENQUEUED (async_op) -- RCCL runs them on its own
        stream, so the next batch's solve overlaps this batch's exchange.  The results were first copied into the send
        buffers, so the caller may overwrite them at once; the buffers themselves are reused only after the previous
        exchange has finished (the wait below), and `finish()` waits for the one in flight."""
        self.finish()  # the send / receive buffers are free again
        for name in sorted(self.send):
            t = results[name]
            assert t.shape[0] == self.mine, (name, t.shape, self.mine)
            self.send[name][: self.mine].copy_(t)  # device -> (staged) send buffer; the pad rows stay zero
            w = dist.gather(self.send[name], list(self.recv[name].unbind(0)) if self.rank == self.dst else None,
                            dst=self.dst, group=self.group, async_op=overlap)
            if overlap:
                self.pending.append(w)
        if overlap or self.rank != self.dst or not assemble:
            return None
        return self.assembled()

    def finish(self):
        """wait for the exchange enqueued by gather(..., overlap=True): the current stream (and, for gloo, the host)
        continues only when rank dst holds the data"""
        for w in self.pending:
            w.wait()
        self.pending = []

    def assembled(self):
        """{name: [batch, ...]} in global problem order from the receive buffers (rank dst; after finish())"""
        out = {}
        for name, whole in self.recv.items():
            tail = tuple(whole.shape[2:])
            if self.interleaved:  # (rank r, row i) is problem i*world + r
                flat = whole.transpose(0, 1).reshape((self.world * self.per,) + tail)
            else:
                flat = whole.reshape((self.world * self.per,) + tail)
            out[name] = flat[: self.batch]
        return out


def gather_results(results, dst=0, group=None, batch=None, interleaved=False):
    """One-shot form: gather a dict of per-rank tensors (x*, f*, iterations, status ...) to rank `dst`.
    `batch` = global number of problems (default: equal shards, world * rows); ragged shards are padded inside.
    Returns {name: tensor [batch, ...]} on dst (on the device of the inputs), None elsewhere."""
    world = dist.get_world_size(group)
    some = next(iter(results.values()))
    if batch is None:
        batch = world * some.shape[0]
    shapes = {k: (tuple(v.shape[1:]), v.dtype) for k, v in results.items()}
    G = Gatherer(batch, shapes, some.device, dst=dst, group=group, interleaved=interleaved)
    out = G.gather(results)
    if out is None:
        return None
    return {k: v.to(some.device).clone() for k, v in out.items()}
