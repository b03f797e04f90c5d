"""Batch sources for the PCAA loops.

``MSRadarDataset`` keeps the item contract of the reference's dataset
(``datasets.py:381-482``): crop files ``crop{i}_subj{s}_{scenario}_track{t}.npy``
holding float64 ``[T, N, C]``; ``__getitem__`` -> (``[C,T,N]`` float32, int64
label re-indexed densely over the subjects present).  Difference on purpose:
the file list is sorted (the reference uses raw ``os.listdir`` order, which is
file-system dependent).

Dataset *generation* from raw radar tracks: ``crop_with_step``,
``process_track`` and ``generate_splits`` restate ``datasets.py:16-25, 79-161, 183-379`` with the
SAME calls to numpy's global RNG in the same order, so that under one ``np.random.seed`` the crops are
bit-identical to the reference's (golden: ``tests/golden/datagen.npz``).  Built differently: frames go
into one preallocated array (the reference re-concatenates the whole sequence per frame: O(frames^2))
and the repeat-padding is one fancy-index instead of a Python loop per point.

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


def crop_with_step(sequence, crop_len, step):
    """Sliding windows ``sequence[i:i+crop_len]`` for ``i in arange(len - crop_len, step=step)``
    (reference ``datasets.py:16-25``: the window starting at ``len - crop_len`` itself is NOT taken)."""
    idxs = np.arange(len(sequence) - crop_len, step=step)
    if len(idxs) == 0:
        return np.array([])
    return np.stack([sequence[i:i + crop_len] for i in idxs])


def process_track(track, standardize_point_cloud=True, divide_by_std=False, force_pc_subsampling=0,
                  nmax=None, nfeatures=None):
    """Raw track (path of a pickle, or the loaded list of frame dicts) -> ``[n_frames, nmax, nfeatures]``
    float64 (reference ``MSRadarDataset.process_track``, ``datasets.py:79-161``): powers to dB
    (``10 log10(p + 1e-8)``), features ``[x, y, z, doppler, power_dB][:nfeatures]``, repeat-pad with
    ``np.random.choice(card, nmax - card)`` or subsample with ``np.random.choice(card, nmax,
    replace=False)`` (numpy's GLOBAL generator, as there), per-frame centring (and optional division by
    ``std + 1e-8``).  Reference quirk kept: with ``force_pc_subsampling = k`` the frame keeps a random
    ORDER of its FIRST k points (the cardinality is overwritten before ``default_rng(0).choice``, ``:107-115``)."""
    import pickle
    nmax = constants.NMAX if nmax is None else nmax
    nf = constants.NFEATURES if nfeatures is None else nfeatures
    frames = track
    if isinstance(track, (str, os.PathLike)):
        with open(track, "rb") as f:
            frames = pickle.load(f)
    sub_rng = np.random.default_rng(0)
    out = np.empty((len(frames), nmax, nf), dtype=np.float64)
    for fi, frame in enumerate(frames):
        card = int(frame["cardinality"][0])
        elements = frame["elements"]
        zs = frame["z_coord"][:, np.newaxis]
        dop = frame["dopplers"][:, np.newaxis]
        pw = frame["powers"][:, np.newaxis]
        if 0 < force_pc_subsampling < card:
            card = force_pc_subsampling
            keep = sub_rng.choice(card, force_pc_subsampling, replace=False)
            elements, zs, dop, pw = elements[keep], zs[keep], dop[keep], pw[keep]
        pw = 10 * np.log10(pw + 1e-8)
        arr = np.concatenate([elements, zs, dop, pw], axis=1)[:, :nf]
        if card < nmax:
            pick = np.random.choice(card, nmax - card)
            final = np.concatenate([arr, arr[pick]], axis=0)
        else:
            pick = np.random.choice(card, nmax, replace=False)
            final = arr[pick, :]
        if standardize_point_cloud:
            mean = final.mean(axis=0)
            std = final.std(axis=0)
            final = final - mean
            if divide_by_std:
                final = final / (std + 1e-8)
        out[fi] = final
    return out


LABEL_DICT = {i: f"target{i}" for i in range(10)}        # reference datasets.py:51-62


def generate_splits(train_classes=(), train_ratio=0.8, valid_ratio=0.1, test_ratio=0.1, seed=0,
                    force_pc_subsampling=0, nmax_points=None, listdir=None, verbose=True):
    """Train/valid/test/unseen crop files under ``constants.GEN_DATA_PATH`` from the raw tracks under
    ``constants.DATA_PATH/target<i>/<scenario>/pc_tr*.obj`` (reference ``generate_splits``,
    ``datasets.py:183-379``): per subject and scenario the tracks are split with sklearn's
    ``train_test_split(random_state=seed)`` twice, every track goes through ``process_track``
    (centred, not divided by std) and ``crop_with_step(NSTEPS, CROP_STEP)``, crops are saved as
    ``crop{i}_subj{s}_{scenario}_track{t}.npy`` (float64 ``[T,N,C]``).  ``listdir`` (default: sorted
    ``os.listdir``) fixes the traversal order and with it the consumption of numpy's global RNG; the
    reference uses raw ``os.listdir`` (file-system order) and asks for ENTER first (``safe_mode``)."""
    from sklearn.model_selection import train_test_split
    ls = listdir or (lambda d: sorted(os.listdir(d)))
    nmax_points = constants.NMAX if nmax_points is None else nmax_points
    assert train_ratio + valid_ratio + test_ratio == 1.0
    dirs = {k: os.path.join(constants.GEN_DATA_PATH, k) for k in ("train", "valid", "test", "unseen")}
    for d in dirs.values():
        os.makedirs(d, exist_ok=True)
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
    train_classes = list(train_classes)
    unseen_classes = np.setdiff1d(list(LABEL_DICT.keys()), train_classes).tolist()
    if not train_classes:
        train_classes = list(LABEL_DICT.keys())

    def emit(pc_file, subj, scenario, target_dir):
        arr = process_track(pc_file, standardize_point_cloud=True, divide_by_std=False,
                            force_pc_subsampling=force_pc_subsampling, nmax=nmax_points)
        crops = crop_with_step(arr, crop_len=constants.NSTEPS, step=constants.CROP_STEP)
        track_index = pc_file.split("/")[-1][5:].split(".")[0]
        for ci in range(len(crops)):
            np.save(os.path.join(target_dir, f"crop{ci}_subj{subj}_{scenario}_track{track_index}.npy"), crops[ci])

    for subj in train_classes:
        subject_dir = os.path.join(constants.DATA_PATH, LABEL_DICT[subj])
        for scenario in ls(subject_dir):
            tracks = ls(os.path.join(subject_dir, scenario))
            if not all(t[:2] == "pc" for t in tracks):
                raise ValueError(f"invalid file in {os.path.join(subject_dir, scenario)}")
            tr, vt = train_test_split(tracks, train_size=train_ratio, random_state=seed)
            va, te = train_test_split(vt, train_size=valid_ratio / (valid_ratio + test_ratio), random_state=seed)
            for group, key in ((tr, "train"), (va, "valid"), (te, "test")):
                for t in group:
                    emit(os.path.join(subject_dir, scenario, t), subj, scenario, dirs[key])
    for subj in unseen_classes:
        subject_dir = os.path.join(constants.DATA_PATH, LABEL_DICT[subj])
        for scenario in ls(subject_dir):
            for t in ls(os.path.join(subject_dir, scenario)):
                emit(os.path.join(subject_dir, scenario, t), subj, scenario, dirs["unseen"])
    stats = {k: len(os.listdir(d)) for k, d in dirs.items()}
    if verbose:
        from .utils import openness
        print(f"-> sizes {stats}; training classes {train_classes}; unseen {unseen_classes}; "
              f"openness {openness(len(train_classes), len(LABEL_DICT)) * 100:.3f}%")
    return stats


class MSRadarDataset(torch.utils.data.Dataset):
    label_dict = LABEL_DICT
    process_track = staticmethod(process_track)
    generate_splits = staticmethod(generate_splits)

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
