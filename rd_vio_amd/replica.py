"""Replica harness for the multi-GPU bench: the hot path does not shard within a stream, so
N GPUs = N independent replicas.  This module holds the only distributed logic there is: rendezvous, a barrier
on both sides of the timed region and the max-over-ranks reduction of the elapsed time."""
import os
import time


def init_distributed(backend=None, device_id=None):
    """-> (rank, local_rank, world, dist_module_or_None); reads the torchrun environment."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return rank, local_rank, world, None
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if not dist.is_initialized():
        kw = {}
        if device_id is not None:
            kw["device_id"] = device_id
        dist.init_process_group(backend=backend or "gloo", rank=rank, world_size=world, **kw)
    return rank, local_rank, world, dist


def timed_region(step, steps, sync, dist=None, device=None, first_index=0, per_rank=False, block=None):
    """barrier + sync, EXACTLY `steps` calls of step(k) -- or ONE call of block(first_index, steps), which must perform
    exactly `steps` steps (a native loop) --, sync + barrier; returns the MAX elapsed seconds over ranks (with per_rank:
    also every rank's own time between the barriers' release and its own last sync, rank order)."""
    import torch

    sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    if block is not None:
        block(first_index, steps)
    else:
        for k in range(steps):
            step(first_index + k)
    sync()
    own = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    own_all = [own]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if per_rank:
            mine = torch.tensor([own], dtype=torch.float64, device=device if device is not None else "cpu")
            every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(every, mine)
            own_all = [float(x.item()) for x in every]
    return (elapsed, own_all) if per_rank else elapsed


def aggregate_rate(world, steps, elapsed):
    """whole-job throughput: every rank processed `steps` units in the (max) elapsed time"""
    return world * steps / elapsed
