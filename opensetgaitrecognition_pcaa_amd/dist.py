"""Data-parallel helpers (one process per GPU, torch.distributed: backend "nccl" is RCCL on
ROCm; "gloo" on CPU for the logic tests).  The PCAA path shards by sequence; the only
cross-rank exchanges are the gradient all-reduce (one flat buffer per optimiser) and, with
SyncBN, the fp64 BatchNorm statistics."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """torchrun-style environment (RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world, group)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, **kwargs)
    return rank, world, dist.group.WORLD


def shard_rows(t, rank, world):
    """This rank's contiguous slice of a per-sequence tensor drawn for the GLOBAL batch (the
    reference's host RNG draws z0 / alphas are made identically on every rank and sliced, so
    the union over ranks is bit-identical to the single-process global batch)."""
    n = t.shape[0]
    if n % world:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return t[rank * per:(rank + 1) * per]


def allreduce_sum_(t, group):
    if group is not None and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return t
