"""Dataset generation from raw radar tracks against tests/golden/datagen.npz, which
tests/golden/make_golden_datagen.py produced by importing the reference's datasets.py: float64 byte-for-byte
equality under the same numpy global seed (same RNG calls in the same order)."""
import json
import os
import pickle

import numpy as np

from opensetgaitrecognition_pcaa_amd import constants, datasets, synthetic as syn

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "datagen.npz"))
META = json.loads(str(G["meta"]))


def test_process_track_and_crops_bit_exact(tmp_path):
    for tag in ("plain", "force10", "divstd"):
        c = META[tag]
        frames = syn.synthetic_raw_track(c["seed"], c["n_frames"])
        np.random.seed(c["np_seed"])
        arr = datasets.process_track(frames, standardize_point_cloud=True, divide_by_std=c["div"],
                                     force_pc_subsampling=c["force"], nmax=c["nmax"], nfeatures=4)
        assert arr.dtype == np.float64 and arr.shape == G[f"{tag}.track"].shape
        assert np.array_equal(arr, G[f"{tag}.track"]), tag
        crops = datasets.crop_with_step(arr, constants.NSTEPS, constants.CROP_STEP)
        assert np.array_equal(crops, G[f"{tag}.crops"]), tag
    # from a pickle on disk, as generate_splits feeds it
    p = tmp_path / "pc_tr0.obj"
    with open(p, "wb") as f:
        pickle.dump(syn.synthetic_raw_track(META["plain"]["seed"], META["plain"]["n_frames"]), f)
    np.random.seed(META["plain"]["np_seed"])
    assert np.array_equal(datasets.MSRadarDataset.process_track(str(p), nmax=META["plain"]["nmax"], nfeatures=4),
                          G["plain.track"])


def test_crop_with_step_edges():
    x = np.arange(36 * 2).reshape(36, 2)
    c = datasets.crop_with_step(x, 30, 6)              # starts 0 only: arange(6, step=6)
    assert c.shape == (1, 30, 2) and np.array_equal(c[0], x[:30])
    assert datasets.crop_with_step(x[:30], 30, 6).shape == (0,)          # len == crop_len: no crop (as the reference)
    assert datasets.crop_with_step(x[:20], 30, 6).shape == (0,)
    assert datasets.crop_with_step(np.arange(43), 30, 6).shape == (3, 30)    # starts 0, 6, 12


def test_generate_splits_reproduces_the_references_files(tmp_path, monkeypatch):
    m = META["splits"]
    data, gen = tmp_path / "raw", tmp_path / "gen"
    for subj, scen, name, seed, nfr in m["layout"]:
        d = data / f"target{subj}" / scen
        d.mkdir(parents=True, exist_ok=True)
        with open(d / name, "wb") as f:
            pickle.dump(syn.synthetic_raw_track(seed, nfr), f)
    monkeypatch.setattr(constants, "DATA_PATH", str(data))
    monkeypatch.setattr(constants, "GEN_DATA_PATH", str(gen))
    monkeypatch.setattr(constants, "NFEATURES", m["nfeatures"])
    np.random.seed(m["np_seed"])
    stats = datasets.generate_splits(train_classes=m["train_classes"], seed=m["seed"], nmax_points=m["nmax"],
                                     verbose=False)
    for split in ("train", "valid", "test", "unseen"):
        names = sorted(os.listdir(gen / split))
        assert names == m["files"][split], split
        assert stats[split] == len(names)
        ref = G[f"splits.{split}.sums"]
        for i, n in enumerate(names):
            a = np.load(gen / split / n)
            assert a.dtype == np.float64 and a.shape == (constants.NSTEPS, m["nmax"], m["nfeatures"])
            got = np.array([a.sum(), np.abs(a).sum(), a[0, 0, 0], a[-1, -1, -1]])
            assert np.array_equal(got, ref[i]), (split, n)
    # the generated train split feeds the dataset / batch contract
    ds = datasets.MSRadarDataset(constants.SPLIT.TRAIN)
    x, y = ds[0]
    assert tuple(x.shape) == (m["nfeatures"], constants.NSTEPS, m["nmax"]) and int(y) in range(len(m["train_classes"]))
