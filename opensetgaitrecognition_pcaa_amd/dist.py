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


def zero_tail_multiple(world, chunks, align):
    """The sharded decoder optimizer (ZeRO-1, train.PCAATrainer dp_zero) cuts the decoder region of the flat buffer into
    ``chunks`` pieces, each reduce-scattered into ``world`` slices of whole ``align``-float (256-B) lines: the region is
    zero-padded to a multiple of this many floats."""
    return world * chunks * align


def zero_slices(dec_start, total, world, chunks, rank):
    """[(chunk_lo, chunk_hi, mine_lo, mine_hi)] in flat-buffer element coordinates: chunk c of the decoder region
    [dec_start, total) and rank ``rank``'s slice of it (what reduce_scatter_tensor leaves on this rank, what its Adam
    updates, what all_gather_into_tensor puts back).  The region's length must be a multiple of world * chunks."""
    n_all = total - dec_start
    if n_all % (world * chunks):
        raise ValueError(f"decoder region of {n_all} floats does not split into {chunks} chunks x {world} ranks")
    n = n_all // chunks
    per = n // world
    return [(dec_start + c * n, dec_start + (c + 1) * n, dec_start + c * n + rank * per, dec_start + c * n + (rank + 1) * per)
            for c in range(chunks)]
