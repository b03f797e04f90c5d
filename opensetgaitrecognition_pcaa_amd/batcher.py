"""Real-data batch source for the PCAA loops: the split's crops packed once into one contiguous
point-major fp32 store, resident in HBM, batches assembled on the device.

The reference feeds its loop with ``DataLoader(MSRadarDataset(split), batch_size, shuffle=True,
drop_last=True, num_workers=0)`` (``PCAA_ablation.py:794-800``): per sample one ``np.load`` of a
``[T,N,C]`` float64 file, a cast, a permute, a Python-level stack and a host-to-device copy
(``datasets.py:466-479``) -- ~1 ms of host work per 61 KB crop, two orders of magnitude short of
the ~10^4 sequences/s the HIP train step consumes.  Here:

* :func:`pack_split` reads the crop files of a dataset ONCE and writes ``crops.f32``
  (``[M,T,N,C]`` fp32, the cast of ``__getitem__``), ``labels.i64`` and ``manifest.json``
  (file order, shape) next to them;
* :class:`PackedCrops` memory-maps that store; ``to_device`` uploads it to HBM through two pinned
  staging buffers on a copy stream (a whole split is 0.6-6 GB: it simply lives in the 288 GB);
* :class:`DeviceBatcher` iterates batches: the epoch's order comes from the same draws
  ``DataLoader(shuffle=True)`` makes from torch's global RNG (so a seeded run sees the reference's
  batches), each batch is one ``pcaa_gather_rows`` launch (+ one for the labels) and is handed to
  the trainer as the zero-copy ``[B,C,T,N]`` view of point-major ``[B,T,N,C]`` storage.
"""
import json
import os

import numpy as np
import torch

from . import ops

MANIFEST = "manifest.json"


def pack_split(dataset, out_dir, chunk=256):
    """Pack every item of ``dataset`` (``MSRadarDataset`` or anything with ``filenames``, ``labels`` and
    ``dataset_dir``; or a sequence of ``([C,T,N] tensor, label)`` items) into ``out_dir``.  Returns the
    manifest dict."""
    os.makedirs(out_dir, exist_ok=True)
    n = len(dataset)
    if n == 0:
        raise ValueError("pack_split: empty dataset")
    first, _ = dataset[0]
    C, T, N = first.shape
    crops = np.lib.format.open_memmap(os.path.join(out_dir, "crops.f32.npy"), mode="w+", dtype=np.float32,
                                      shape=(n, T, N, C))
    labels = np.empty(n, dtype=np.int64)
    for i0 in range(0, n, chunk):
        for i in range(i0, min(n, i0 + chunk)):
            x, y = dataset[i]
            if tuple(x.shape) != (C, T, N):
                raise ValueError(f"pack_split: item {i} has shape {tuple(x.shape)}, expected {(C, T, N)}")
            # items are [C,T,N] views of point-major data: store point-major [T,N,C]
            crops[i] = x.permute(1, 2, 0).contiguous().numpy()
            labels[i] = int(y)
    crops.flush()
    np.save(os.path.join(out_dir, "labels.i64.npy"), labels)
    manifest = {"n": n, "T": T, "N": N, "C": C, "dtype": "float32", "layout": "[M,T,N,C] point-major",
                "filenames": list(getattr(dataset, "filenames", [])),
                "original_labels": [int(v) for v in getattr(dataset, "original_labels", [])],
                "source_signature": source_signature(dataset)}
    with open(os.path.join(out_dir, MANIFEST), "w") as f:
        json.dump(manifest, f)
    return manifest


def source_signature(dataset):
    """[number of files, total bytes, newest mtime in ns] of a file-backed dataset's crops: a regenerated split
    (same file names, other contents -- e.g. generate_splits with another NMAX) must not be served from a
    stale packed store."""
    d, names = getattr(dataset, "dataset_dir", None), getattr(dataset, "filenames", None)
    if d is None or names is None:
        return None
    size, newest = 0, 0
    for f in names:
        st = os.stat(os.path.join(d, f))
        size += st.st_size
        newest = max(newest, st.st_mtime_ns)
    return [len(names), size, newest]


class PackedCrops:
    """Memory-mapped packed split (see :func:`pack_split`)."""

    def __init__(self, directory):
        with open(os.path.join(directory, MANIFEST)) as f:
            self.manifest = json.load(f)
        m = self.manifest
        self.crops = np.load(os.path.join(directory, "crops.f32.npy"), mmap_mode="r")
        self.labels = np.load(os.path.join(directory, "labels.i64.npy"))
        if self.crops.shape != (m["n"], m["T"], m["N"], m["C"]) or self.labels.shape != (m["n"],):
            raise ValueError("PackedCrops: store does not match its manifest")

    def __len__(self):
        return self.manifest["n"]

    def to_device(self, device="cuda", chunk_bytes=64 << 20):
        """Upload the store: two pinned staging buffers, copies on a side stream, the host fills one
        buffer while the other is in flight.  Returns (crops [M,T,N,C] fp32, labels [M] int64) on ``device``."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("PackedCrops.to_device: the batcher assembles batches on the HIP device")
        M = len(self)
        row = int(np.prod(self.crops.shape[1:]))
        dst = torch.empty((M, row), dtype=torch.float32, device=device)
        rows_per = max(1, chunk_bytes // (row * 4))
        stage = [torch.empty((rows_per, row), dtype=torch.float32).pin_memory() for _ in range(2)]
        done = [None, None]
        copy = torch.cuda.Stream(device=device)
        flat = self.crops.reshape(M, row)
        for k, r0 in enumerate(range(0, M, rows_per)):
            r1 = min(M, r0 + rows_per)
            b = k & 1
            if done[b] is not None:
                done[b].synchronize()                  # the copy that last used this buffer has left it
            stage[b][:r1 - r0].copy_(torch.from_numpy(np.array(flat[r0:r1])))     # np.array: a writable copy of the mmap slice
            with torch.cuda.stream(copy):
                dst[r0:r1].copy_(stage[b][:r1 - r0], non_blocking=True)
                done[b] = torch.cuda.Event()
                done[b].record(copy)
        torch.cuda.current_stream(device).wait_stream(copy)
        copy.synchronize()
        m = self.manifest
        return dst.view(M, m["T"], m["N"], m["C"]), torch.from_numpy(self.labels).to(device)


def dataloader_epoch_order(n, shuffle=True):
    """Index order of one epoch of ``DataLoader(dataset, shuffle=shuffle, num_workers=0)`` under torch's
    GLOBAL RNG, consuming exactly the draws the loader consumes: its iterator first draws a base seed
    (one int64, unused without workers -- but it moves the global generator, which the loop's
    ``torch.rand`` alphas come from), then, when shuffling, ``RandomSampler`` seeds a fresh generator
    with a second draw and takes ``randperm(n)``."""
    torch.empty((), dtype=torch.int64).random_()            # _BaseDataLoaderIter._base_seed
    if not shuffle:
        return torch.arange(n)
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g)


class DeviceBatcher:
    """Iterable of ``(pcs [B,C,T,N] fp32 view, labels [B] int64)`` device batches over a store resident
    in HBM.  ``shuffle=True`` reproduces the batch composition of the reference's DataLoader under the
    same global torch seed; ``drop_last`` as there."""

    def __init__(self, crops_dev, labels_dev, batch_size, shuffle=True, drop_last=True, rank=0, world=1, group=None):
        """``world`` > 1 (data parallel): ``batch_size`` is the GLOBAL batch; rank 0's epoch order is broadcast
        over ``group`` and every rank assembles rows [rank*B/world, (rank+1)*B/world) of each global batch."""
        if not crops_dev.is_cuda or crops_dev.dim() != 4 or crops_dev.dtype != torch.float32:
            raise RuntimeError("DeviceBatcher: crops must be a [M,T,N,C] float32 tensor on the HIP device")
        if labels_dev.shape != (crops_dev.shape[0],) or labels_dev.dtype != torch.int64:
            raise ValueError("DeviceBatcher: labels must be [M] int64")
        self.crops, self.labels = crops_dev.contiguous(), labels_dev.contiguous()
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self.rank, self.world, self.group = int(rank), int(world), group
        if self.batch_size % self.world:
            raise ValueError(f"DeviceBatcher: global batch {batch_size} is not divisible by world size {world}")
        self.err = torch.zeros(1, dtype=torch.int32, device=crops_dev.device)
        # labels ride through the same row gather: pad each to one 16-byte row
        self._lab_rows = torch.zeros((labels_dev.numel(), 2), dtype=torch.int64, device=crops_dev.device)
        self._lab_rows[:, 0] = self.labels

    def __len__(self):
        n = self.crops.shape[0]
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.crops.shape[0]
        order = dataloader_epoch_order(n, self.shuffle).to(self.crops.device, non_blocking=True)
        if self.world > 1:
            import torch.distributed as dist
            dist.broadcast(order, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                           group=self.group)            # one order for all ranks (rank 0's RNG)
        nb = len(self)
        per = self.batch_size // self.world
        for k in range(nb):
            idx = order[k * self.batch_size:(k + 1) * self.batch_size]
            if self.world > 1:
                if idx.numel() < self.batch_size:
                    break                               # a ragged last global batch cannot be sharded evenly
                idx = idx[self.rank * per:(self.rank + 1) * per]
            yield self.batch(idx)

    def batch(self, idx):
        """The batch of the given device index vector."""
        pm = ops.gather_rows(self.crops, idx, err_flag=self.err)                 # [B,T,N,C]
        lab = ops.gather_rows(self._lab_rows, idx, err_flag=self.err)[:, 0].contiguous()
        return pm.permute(0, 3, 1, 2), lab

    def check(self):
        """Raise if any batch so far used an index outside the store (one host sync)."""
        if int(self.err.item()):
            raise IndexError("DeviceBatcher: a batch index was outside the packed store")


def batcher_for(dataset, batch_size, device, shuffle, drop_last=True, cache_dir=None, rank=0, world=1, group=None):
    """A :class:`DeviceBatcher` over ``dataset``: in-memory point-major datasets (``SyntheticGaitDataset``:
    ``.pcs [M,T,N,C]``, ``.labels``) go to the device as they are; file-backed ones (``MSRadarDataset``) are
    packed once into ``<dataset_dir>_packed`` (re-packed when the file list changed) and uploaded."""
    if hasattr(dataset, "pcs") and torch.is_tensor(dataset.pcs):
        crops = dataset.pcs.to(device).float().contiguous()
        labels = torch.as_tensor(dataset.labels).to(torch.int64).to(device)
        return DeviceBatcher(crops, labels, batch_size, shuffle, drop_last, rank, world, group)
    cache_dir = cache_dir or str(dataset.dataset_dir).rstrip("/\\") + "_packed"
    fresh = False
    if os.path.exists(os.path.join(cache_dir, MANIFEST)):
        with open(os.path.join(cache_dir, MANIFEST)) as f:
            man = json.load(f)
            fresh = (man.get("filenames") == list(dataset.filenames)
                     and man.get("source_signature") == source_signature(dataset))
    if not fresh and rank == 0:
        pack_split(dataset, cache_dir)
    if world > 1:
        import torch.distributed as dist
        dist.barrier(group=group)                        # the other ranks wait for rank 0's pack
    crops, labels = PackedCrops(cache_dir).to_device(device)
    return DeviceBatcher(crops, labels, batch_size, shuffle, drop_last, rank, world, group)

