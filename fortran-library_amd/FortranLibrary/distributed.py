"""Sharding of a batch of independent problems over the GPUs of one node, and the single exchange
of the path: one gather of the converged results to rank 0 (RCCL over xGMI when the process group is
"nccl"; "gloo" in the CPU tests).  Problems are independent, so there is no
collective inside a solve -- each rank owns a contiguous block of the batch.
"""
import torch
import torch.distributed as dist


def shard_bounds(batch, rank, world):
    """Contiguous block [lo, hi) of a global batch owned by `rank`: ceil(batch/world) problems per rank."""
    per = -(-batch // world)
    lo = min(batch, rank * per)
    return lo, min(batch, lo + per)


def gather_results(results, dst=0, group=None):
    """Gather a dict of equally shaped per-rank tensors (x*, f*, iterations, status ...) to rank `dst`.

    Returns {name: tensor concatenated over ranks along dim 0} on dst, None elsewhere.  One
    torch.distributed.gather per array -- the only communication of the whole job.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    # RCCL ("nccl") moves device tensors directly over xGMI; gloo (CPU tests, single-GPU rehearsals) needs host tensors
    stage = dist.get_backend(group) == "gloo"
    out = {} if rank == dst else None
    for name in sorted(results):
        t = results[name].contiguous()
        dev = t.device
        if stage and t.is_cuda:
            t = t.cpu()
        # one [world, ...] buffer whose slices are the gather list: the ranks' blocks land in place, no concatenation pass
        whole = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device) if rank == dst else None
        dist.gather(t, list(whole.unbind(0)) if rank == dst else None, dst=dst, group=group)
        if rank == dst:
            out[name] = whole.reshape((world * t.shape[0],) + tuple(t.shape[1:])).to(dev)
    return out
