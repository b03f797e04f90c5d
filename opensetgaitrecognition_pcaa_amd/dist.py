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


# ----------------------------------------------------------------------------------------------------------------------
# The step's exchanges behind one small interface, so that a one-GPU box can run ONE RANK'S PROGRAM OF A WORLD OF W
# (round 6: no multi-GPU node has been available in any round).
# ----------------------------------------------------------------------------------------------------------------------
class GroupExchange:
    """The real thing: torch.distributed collectives on ``group`` (RCCL on the GPU, gloo in the CPU tests)."""

    emulated = False

    def __init__(self, group):
        self.group = group
        self.world = dist.get_world_size(group)
        # RCCL collectives are stream operations and can be recorded into a hipGraph; gloo's run on the host
        self.capturable = dist.get_backend(group) == "nccl"

    def all_reduce(self, t, async_op=False, tag=None):
        return dist.all_reduce(t, group=self.group, async_op=async_op)

    def all_gather_into_tensor(self, dst, src, async_op=False, tag=None):
        return dist.all_gather_into_tensor(dst, src, group=self.group, async_op=async_op)


class _StreamWork:
    """What an emulated collective returns: ``wait()`` makes the CURRENT stream wait for it (like a c10d work's)."""

    __slots__ = ("event",)

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class _NoWork:
    def wait(self):
        pass


class EmulatedExchange:
    """One rank of a world of ``world`` ranks on ONE GPU, without peers: every collective becomes a device operation
    that moves the bytes the rank's memory system would see and leaves the values the collective would leave IF EVERY
    RANK HELD THIS RANK'S DATA (or, for the gathered operands, the peers' rows the caller staged with ``set_peers``):

    * ``all_reduce(t)``: t *= world in place (reads and writes t once: what a ring all-reduce reads and writes per rank up
      to the factor 2 (w-1)/w), the sum of ``world`` identical shards;
    * ``all_gather_into_tensor(dst, src)``: own rows into slot 0, the other ``world - 1`` slots from the staged peers of
      that ``tag`` (``set_peers``; shape [world - 1, *src.shape]) or, without any, copies of ``src``: (world - 1) x the
      bytes arrive in ``dst`` (contiguous, world x src), as from the wire.

    Like RCCL's, the operations run on a stream of their own that picks up behind the issuing stream; the returned work's
    ``wait()`` orders the waiting stream behind them (under a hipGraph capture: on the issuing stream, see _issue).  What is NOT emulated: the wire itself (latency, link bandwidth,
    other ranks' skew) -- bench.py's ``scale_projection`` adds that from stated link figures -- and SyncBN."""

    emulated = True
    capturable = True

    def __init__(self, world, device):
        if world < 1:
            raise ValueError("EmulatedExchange: world must be >= 1")
        self.world = int(world)
        self.group = None
        self.stream = torch.cuda.Stream(device=device)
        self.peers = {}
        self.bytes_moved = 0

    def set_peers(self, tag, rows):
        """``rows`` [world - 1, *shape of the gathered source]: what the other ranks contribute to the gathers tagged ``tag``"""
        self.peers[tag] = rows

    def _issue(self, fn):
        cur = torch.cuda.current_stream()
        if torch.cuda.is_current_stream_capturing():
            # Under a hipGraph capture the operation runs on the issuing stream itself and the work is a no-op: every
            # consumer in PCAATrainer._step already orders itself behind the issuing stream (the side stream waits for
            # the main and weight-gradient streams before it waits for the works).  The event pair of the eager form --
            # record on the issuer, wait on the exchange stream, record there, wait on the consumer -- crashed
            # hipStreamEndCapture on ROCm 7.2 (tools/lab/dp_graph_probe.py: with the events, with or without the
            # operations: segfault; without the events: captured and replayed); RCCL's own collectives capture fine.
            fn()
            return _NoWork()
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            fn()
            done = torch.cuda.Event()
            done.record(self.stream)
        return _StreamWork(done)

    def all_reduce(self, t, async_op=False, tag=None):
        if not torch.cuda.is_current_stream_capturing():
            t.record_stream(self.stream)
        work = self._issue(lambda: t.mul_(self.world))
        self.bytes_moved += 2 * t.numel() * t.element_size()
        if async_op:
            return work
        work.wait()
        return None

    def all_gather_into_tensor(self, dst, src, async_op=False, tag=None):
        if dst.numel() != src.numel() * self.world:
            raise ValueError(f"EmulatedExchange.all_gather_into_tensor: dst {tuple(dst.shape)} is not {self.world} x src "
                             f"{tuple(src.shape)}")
        slots = dst.view((self.world,) + tuple(src.shape))
        peers = self.peers.get(tag)
        if peers is not None and tuple(peers.shape) != (self.world - 1,) + tuple(src.shape):
            raise ValueError(f"EmulatedExchange: peers of {tag!r} are {tuple(peers.shape)}, want "
                             f"{(self.world - 1,) + tuple(src.shape)}")

        def fn():
            slots[0].copy_(src)
            if self.world > 1:
                slots[1:].copy_(peers if peers is not None else src.unsqueeze(0).expand_as(slots[1:]))
        if not torch.cuda.is_current_stream_capturing():
            dst.record_stream(self.stream)
            src.record_stream(self.stream)
        work = self._issue(fn)
        self.bytes_moved += 2 * dst.numel() * dst.element_size()
        if async_op:
            return work
        work.wait()
        return None
