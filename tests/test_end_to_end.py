"""The user's whole flow on the GPU, with the reference's call surface end to end: raw radar tracks ->
generate_splits -> train_variant4 (packed-store batcher, HIP train step, checkpoints of the reference's layout)
-> CGAAE_inference (eval encoder, fp64 likelihoods, Youden threshold, k-window votes, result files)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from opensetgaitrecognition_pcaa_amd import constants, datasets, inference, synthetic as syn
from opensetgaitrecognition_pcaa_amd import functional as F_hip
from opensetgaitrecognition_pcaa_amd.train import train_pointsubsampling, train_variant1, train_variant4

pytestmark = pytest.mark.gpu


def _raw_dataset(root, n_tracks=10, n_frames=60):
    for subj in range(10):
        for si, scen in enumerate(("free_walk", "hands_in_pockets", "smartphone")):
            d = root / f"target{subj}" / scen
            d.mkdir(parents=True)
            for t in range(n_tracks):
                with open(d / f"pc_tr{t}{si}.obj", "wb") as f:
                    pickle.dump(syn.synthetic_raw_track(5000 + subj * 100 + si * 10 + t, n_frames, max_points=24), f)


@pytest.mark.timeout(600)
def test_raw_tracks_to_open_set_predictions(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    _raw_dataset(tmp_path / "raw")
    monkeypatch.setattr(constants, "DATA_PATH", str(tmp_path / "raw"))
    monkeypatch.setattr(constants, "GEN_DATA_PATH", str(tmp_path / "gen"))
    monkeypatch.setattr(constants, "NFEATURES", 4)
    classes = [0, 1, 2, 3, 4, 5]
    np.random.seed(3)
    stats = datasets.generate_splits(train_classes=classes, seed=0, nmax_points=16, verbose=False)
    assert stats["train"] > 100 and stats["unseen"] > 50
    prev = F_hip.get_precision()
    F_hip.set_precision("bf16")
    try:
        _flow(classes)
    finally:
        F_hip.set_precision(prev)


def _flow(classes):
    cfg = dict(constants.CONFIG)
    cfg.update(MODEL_NAME="e2e_V4", TRAIN_CLASSES=classes, NMAX=16, BATCH_SIZE=16, EPOCHS=1, CHECKPOINT_FREQUENCY=1,
               SUBSAMPLE_FACTOR=1.0, NOTES="")
    np.random.seed(0); torch.manual_seed(0)
    trainer, hist = train_variant4(cfg, wandb_mode="disabled")
    assert len(hist) == 1 and all(np.isfinite(v) for v in hist[0].values())
    for sfx in ("_E", "_G", "_D", "_GPH", "_DPH"):
        assert os.path.exists(f"models/e2e_V4/e2e_V4{sfx}.pt")
    assert os.path.exists("models/e2e_V4/discriminator_means.pt") and os.path.exists("models/e2e_V4/config.pkl")
    assert os.path.exists(os.path.join(constants.GEN_DATA_PATH, "train_packed", "manifest.json"))
    out = inference.CGAAE_inference(["e2e_V4"], ks=[2, 4], generate_dataset=False)
    for k in (2, 4):
        with open(f"models/e2e_V4/naive_seq_log_{k}.json") as f:
            m = json.load(f)
        assert m["n_steps"] == k and 0.0 <= m["accuracy"] <= 1.0 and 0.0 <= m["f1_macro"] <= 1.0
        preds = np.load(f"models/e2e_V4/final_preds_{k}.npy")
        labels = np.load(f"models/e2e_V4/final_labels_{k}.npy")
        assert preds.shape == labels.shape and len(preds) > 10
        assert set(np.unique(labels)) <= set(range(len(classes) + 1)) and len(classes) in labels   # unknown class present
        assert preds.min() >= 0 and preds.max() <= len(classes)
        assert set(out[k]) == {"f1_micro", "f1_macro", "f1_weighted"}
    # the reference's own call form of the procedure (inference_PCAA.py:117-125: folders in, (out_log, preds, labels) out)
    enc, means = inference.CGAAE_inference_setup("e2e_V4", 32, False, generate_dataset=False, device=torch.device("cuda"))
    log2, p2, l2 = inference.naive_sequential_procedure(2, enc, means, "figures/e2e_V4", "models/e2e_V4",
                                                         scenarios_list=constants.TRAIN_SCENARIOS, seed=0,
                                                         unseen_valid_ratio=0.2, force_pc_subsampling=0)
    assert np.array_equal(p2, np.load("models/e2e_V4/final_preds_2.npy")) and np.array_equal(l2, np.load("models/e2e_V4/final_labels_2.npy"))
    assert log2["n_steps"] == 2 and os.path.isdir("figures/e2e_V4")
    # variant 1 through its loop: learner checkpoint and the train-mode centroids file
    cfg1 = dict(cfg); cfg1["MODEL_NAME"] = "e2e_V1"
    trainer1, hist1 = train_variant1(cfg1, wandb_mode="disabled")
    assert os.path.exists("models/e2e_V1/e2e_V1_ML.pt")
    cent = torch.load("models/e2e_V1/discriminator_means.pt", map_location="cpu")
    assert tuple(cent.shape) == (len(classes), 32) and torch.isfinite(cent).all()
    # the point-subsampling study (train_pointsubsampling.py), cut down to one class count, one subset, two NMAX
    base = dict(cfg); base.update(EPOCHS=1, BATCH_SIZE=16)
    res = train_pointsubsampling(n_training_classes=(4,), n_points_subs=(12, 20), n_tests=1, ks=(2,), config=base,
                                 model_name_base="sweep_V4_")
    assert set(res) == {"sweep_V4_12.4.1", "sweep_V4_20.4.1"}
    for name, log in res.items():
        assert 0.0 <= log[2]["f1_macro"] <= 1.0 and os.path.exists(f"models/{name}/final_preds_2.npy")
