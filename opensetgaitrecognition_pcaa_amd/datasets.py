"""Batch sources for the PCAA loops.

``MSRadarDataset`` keeps the item contract of the reference's dataset
(``datasets.py:381-482``): crop files ``crop{i}_subj{s}_{scenario}_track{t}.npy``
holding float64 ``[T, N, C]``; ``__getitem__`` -> (``[C,T,N]`` float32, int64
label re-indexed densely over the subjects present).  Difference on purpose:
the file list is sorted (the reference uses raw ``os.listdir`` order, which is
file-system dependent).  Dataset *generation* from raw radar tracks is out of
scope.

``SyntheticGaitDataset`` produces mmGait10-shaped crops from a seed; items are
permuted views of point-major ``[T,N,C]`` storage.
"""
import os

import numpy as np
import torch

from . import constants
from .constants import SCENARIO, SPLIT


def filename2crop(filename):
    return int(filename.split("_")[0][4:])


def filename2subj(filename):
    return int(filename.split("_")[1][4:])


def filename2track(filename):
    return filename.split("_")[-1][5:].split(".")[0]


def filename2scenario(filename):
    return "_".join(filename.split("_")[2:-1])


class MSRadarDataset(torch.utils.data.Dataset):
    def __init__(self, split: SPLIT, scenarios=None, sequential=False, subsample_factor=1.0):
        scenarios = constants.TRAIN_SCENARIOS if scenarios is None else scenarios
        self.dataset_dir = os.path.join(constants.GEN_DATA_PATH, split.value)
        self.sequential = sequential
        names = sorted(os.listdir(self.dataset_dir))
        if sequential:
            # group by (subject, track), crops in temporal order
            names.sort(key=lambda f: (filename2subj(f), filename2track(f), filename2crop(f)))
        wanted = {s.value for s in scenarios}
        names = [f for f in names if filename2scenario(f) in wanted]
        if subsample_factor < 1.0:
            names = list(np.random.choice(names, int(len(names) * subsample_factor), replace=False))
        self.filenames = names
        self.original_labels = [filename2subj(f) for f in names]
        dense = {c: i for i, c in enumerate(sorted(set(self.original_labels)))}
        self.labels = np.array([dense[j] for j in self.original_labels], dtype=np.int64)

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, idx):
        arr = np.load(os.path.join(self.dataset_dir, self.filenames[idx]), allow_pickle=True)
        pc_seq = torch.from_numpy(arr).to(torch.float)      # [T, N, C] point-major
        return pc_seq.permute(2, 0, 1), torch.tensor(self.labels[idx]).type(torch.LongTensor)


class SyntheticGaitDataset(torch.utils.data.Dataset):
    """``n_items`` random crops [T,N,C] (per-frame centred), labels uniform in [0,K)."""

    def __init__(self, n_items, K, N=None, C=None, T=None, seed=0):
        from .synthetic import synthetic_labels, synthetic_pcs
        N = constants.NMAX if N is None else N
        C = constants.NFEATURES if C is None else C
        T = constants.NSTEPS if T is None else T
        self.pcs = synthetic_pcs(n_items, T, N, C, seed=seed)          # [M,T,N,C]
        self.labels = synthetic_labels(n_items, K, seed=seed + 1)

    def __len__(self):
        return self.pcs.shape[0]

    def __getitem__(self, idx):
        return self.pcs[idx].permute(2, 0, 1), self.labels[idx]
