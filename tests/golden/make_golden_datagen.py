#!/usr/bin/env python3
"""Golden for the dataset-generation row, made by IMPORTING THE REFERENCE's datasets.py in the
build container:  python tests/golden/make_golden_datagen.py  ->  tests/golden/datagen.npz

Inputs are synthetic raw tracks regenerated on either side from
``opensetgaitrecognition_pcaa_amd.synthetic.synthetic_raw_track(seed, n_frames)``; stored are the reference's
outputs: ``process_track`` arrays (plain, forced subsampling, divided by std), ``crop_with_step`` results and the
file list + per-file checksums of one ``generate_splits`` run.  ``os.listdir`` is patched to return sorted
lists during that run so that the traversal order (and with it the consumption of numpy's global RNG) is
defined; the reference itself uses the file system's order."""
import builtins
import json
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PCAA_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
from opensetgaitrecognition_pcaa_amd import synthetic as syn  # noqa: E402

for name in ("wandb", "umap"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, REF)
import constants as rconst  # noqa: E402
import datasets as rdatasets  # noqa: E402  (the reference's datasets.py, not the HF package)

assert rdatasets.__file__.startswith(REF), rdatasets.__file__


def main():
    out = {}
    tmp = tempfile.mkdtemp()
    # --- process_track / crop_with_step on single tracks
    cases = [("plain", dict(seed=3, n_frames=47, nmax=24, force=0, div=False, np_seed=7)),
             ("force10", dict(seed=4, n_frames=40, nmax=16, force=10, div=False, np_seed=8)),
             ("divstd", dict(seed=5, n_frames=36, nmax=32, force=0, div=True, np_seed=9))]
    meta = {}
    for tag, c in cases:
        frames = syn.synthetic_raw_track(c["seed"], c["n_frames"])
        path = os.path.join(tmp, f"pc_tr{tag}.obj")
        with open(path, "wb") as f:
            pickle.dump(frames, f)
        np.random.seed(c["np_seed"])
        arr = rdatasets.MSRadarDataset.process_track(path, standardize_point_cloud=True, divide_by_std=c["div"],
                                                     force_pc_subsampling=c["force"], nmax=c["nmax"])
        crops = rdatasets.crop_with_step(arr, crop_len=rconst.NSTEPS, step=rconst.CROP_STEP)
        out[f"{tag}.track"] = arr
        out[f"{tag}.crops"] = crops
        meta[tag] = c
    # --- one generate_splits run on a tiny raw dataset: 10 subjects x 3 scenarios x 10 tracks
    data = os.path.join(tmp, "raw")
    gen = os.path.join(tmp, "gen")
    layout = []
    for subj in range(10):
        for si, scen in enumerate(("free_walk", "hands_in_pockets", "smartphone")):
            d = os.path.join(data, f"target{subj}", scen)
            os.makedirs(d)
            for t in range(10):
                seed, nfr = 1000 + subj * 100 + si * 10 + t, 38 + ((subj + t) % 3) * 6
                with open(os.path.join(d, f"pc_tr{t}{si}.obj"), "wb") as f:
                    pickle.dump(syn.synthetic_raw_track(seed, nfr), f)
                layout.append([subj, scen, f"pc_tr{t}{si}.obj", seed, nfr])
    rconst.DATA_PATH, rconst.GEN_DATA_PATH = data, gen
    real_listdir = os.listdir
    os.listdir = lambda d: sorted(real_listdir(d))
    real_print = builtins.print
    builtins.print = lambda *a, **k: None
    try:
        np.random.seed(11)
        rdatasets.MSRadarDataset.generate_splits(train_classes=[0, 2, 5, 7], seed=0, safe_mode=False, nmax_points=16)
    finally:
        os.listdir = real_listdir
        builtins.print = real_print
    files = {}
    for split in ("train", "valid", "test", "unseen"):
        names = sorted(real_listdir(os.path.join(gen, split)))
        sums = []
        for n in names:
            a = np.load(os.path.join(gen, split, n))
            sums.append([float(a.sum()), float(np.abs(a).sum()), float(a[0, 0, 0]), float(a[-1, -1, -1])])
        files[split] = names
        out[f"splits.{split}.sums"] = np.array(sums, dtype=np.float64)
        real_print(split, len(names))
    meta["splits"] = {"layout": layout, "files": files, "train_classes": [0, 2, 5, 7], "seed": 0, "np_seed": 11,
                      "nmax": 16, "nfeatures": int(rconst.NFEATURES)}
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "datagen.npz"), **out)
    print("datagen.npz", os.path.getsize(os.path.join(HERE, "datagen.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
