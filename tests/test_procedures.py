"""The ASSEMBLED procedures against the reference's own functions (tests/golden/make_golden_procedures.py ran the
reference's ``train_variant4`` for two seeded epochs and its ``naive_sequential_procedure`` for k in {1,2,4,6} on a
synthetic raw dataset): init draw order, epoch order, per-step host RNG draws, the 8 logged scalars, the best-valid
checkpoint rule, checkpoint contents; sequential ordering, held-out unseen subjects, straddling-window skip,
unseen-test filter, votes."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from helpers import is_pre_bn_bias, load_golden, make_encoder
from opensetgaitrecognition_pcaa_amd import constants, datasets, synthetic as syn

G, META = load_golden("procedures")


def _raw_and_splits(tmp_path, monkeypatch):
    """The golden's raw dataset (10 subjects x 3 scenarios x 10 tracks) and its splits, regenerated here."""
    data, gen = tmp_path / "raw", tmp_path / "gen"
    for subj in range(10):
        for si, scen in enumerate(("free_walk", "hands_in_pockets", "smartphone")):
            d = data / f"target{subj}" / scen
            d.mkdir(parents=True, exist_ok=True)
            for t in range(10):
                with open(d / f"pc_tr{t}{si}.obj", "wb") as f:
                    pickle.dump(syn.synthetic_raw_track(1000 + subj * 100 + si * 10 + t, 38 + ((subj + t) % 3) * 6), f)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(constants, "DATA_PATH", str(data))
    monkeypatch.setattr(constants, "GEN_DATA_PATH", str(gen))
    monkeypatch.setattr(constants, "NFEATURES", 4)
    np.random.seed(META["np_seed_splits"])
    datasets.generate_splits(train_classes=META["train_classes"], seed=0, nmax_points=META["nmax"], verbose=False)


def test_sequential_order_matches_the_reference(tmp_path, monkeypatch):
    """CPU: MSRadarDataset(sequential=True) lists the crops subject by subject, track by track, in temporal order --
    the order the reference's k-windows are cut over (its tracks iterate a set; the golden pins the sorted order)."""
    _raw_and_splits(tmp_path, monkeypatch)
    for split, key in ((constants.SPLIT.TEST, "infer.test_files"), (constants.SPLIT.UNSEEN, "infer.unseen_files")):
        ds = datasets.MSRadarDataset(split, sequential=True)
        assert list(ds.filenames) == json.loads(str(G[key])), split


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_train_variant4_two_epochs_vs_reference(tmp_path, monkeypatch):
    from opensetgaitrecognition_pcaa_amd import functional as F_hip
    from opensetgaitrecognition_pcaa_amd.train import train_variant4
    _raw_and_splits(tmp_path, monkeypatch)
    F_hip.set_precision("fp32")
    cfg = dict(constants.CONFIG)
    cfg.update(MODEL_NAME="proc_V4", TRAIN_CLASSES=META["train_classes"], NMAX=META["nmax"], BATCH_SIZE=META["batch"],
               EPOCHS=META["epochs"], CHECKPOINT_FREQUENCY=1, SUBSAMPLE_FACTOR=1.0, SUPERVISION_FREQUENCY=1, NOTES="")
    saved_after = []

    def log_fn(record):
        saved_after.append(sorted(os.listdir("models/proc_V4")))
    np.random.seed(META["seed"])
    torch.manual_seed(META["seed"])
    trainer, hist = train_variant4(cfg, wandb_mode="disabled", log_fn=log_fn)
    keys = json.loads(str(G["train.record_keys"]))
    ref = G["train.records"]
    assert len(hist) == META["epochs"] and sorted(hist[0]) == keys
    n_train = (291 // META["batch"]) * META["batch"]
    for e, rec in enumerate(hist):
        for j, k in enumerate(keys):
            got, want = rec[k], ref[e, j]
            if k == "Train Accuracy":
                assert abs(got - want) <= 4.0 / n_train + 1e-9, (e, k, got, want)
            elif k == "Valid Accuracy":
                assert abs(got - want) <= 2.0 / 32 + 1e-9, (e, k, got, want)
            else:
                # epoch 0 starts from the SAME initial weights (same draws from torch's RNG, same construction order)
                # and sees the same batches and host draws; per-step differences are Adam's rounding-noise-signed
                # steps (docs/LAB_LOG.md section 2), so the epoch means agree far inside 1e-2; the gate widens with the epoch
                tol = 1e-2 * (e + 1)
                assert abs(got - want) <= tol * abs(want), (e, k, got, want)
    # best-valid rule (:1073-1076): a checkpoint after an epoch iff its valid accuracy beats the best so far
    best, expect = 0.0, []
    for rec in hist:
        expect.append(rec["Valid Accuracy"] > best)
        best = max(best, rec["Valid Accuracy"])
    assert expect[0], "epoch 0 must checkpoint (valid accuracy > 0)"
    if [int(x) for x in expect] == list(G["train.saved_after_epoch"]):
        assert sorted(os.listdir("models/proc_V4")) == json.loads(str(G["train.files"]))
    # the checkpoint written after epoch 0 (the reference's last save too, if its decisions were the same)
    if list(G["train.saved_after_epoch"]) == [1, 0] and [int(x) for x in expect] == [1, 0]:
        for sfx in ("E", "G", "D", "GPH", "DPH"):
            sd = torch.load(f"models/proc_V4/proc_V4_{sfx}.pt", map_location="cpu")
            want = json.loads(str(G[f"train.ckpt.{sfx}"]))
            assert list(sd) == list(want), sfx
            for name, (s_ref, n_ref) in want.items():
                v = sd[name].double()
                if name.endswith("num_batches_tracked"):
                    assert float(v) == s_ref
                    continue
                if is_pre_bn_bias(name) or name.endswith("running_mean"):
                    continue      # the reference random-walks these by +-lr on a rounding-noise gradient (docs/LAB_LOG.md section 2)
                assert abs(float(v.norm()) - n_ref) <= 2e-3 * max(n_ref, 1e-3), (sfx, name, float(v.norm()), n_ref)
    means = torch.load("models/proc_V4/discriminator_means.pt", map_location="cpu")
    assert np.array_equal(means.numpy(), G["train.means"])


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_naive_sequential_procedure_vs_reference(tmp_path, monkeypatch):
    from opensetgaitrecognition_pcaa_amd import functional as F_hip, inference
    _raw_and_splits(tmp_path, monkeypatch)
    F_hip.set_precision("fp32")
    K = len(META["train_classes"])
    enc = make_encoder(K, META["nmax"], 4, True, seed=META["enc_fill_seed"]).to("cuda").eval()
    means = torch.from_numpy(G["infer.means"]).to("cuda")
    known_pcs, known_labels = inference._sequential_split_on_device(constants.SPLIT.TEST, constants.TRAIN_SCENARIOS, "cuda")
    unseen_pcs, unseen_labels = inference._sequential_split_on_device(constants.SPLIT.UNSEEN, constants.TRAIN_SCENARIOS, "cuda")
    from sklearn.metrics import f1_score
    for k in (1, 2, 4, 6):
        preds, labels, thr = inference.naive_sequential_procedure(k, enc, means, known_pcs, known_labels, unseen_pcs,
                                                                  unseen_labels, seed=0, unseen_valid_ratio=0.2)
        assert np.array_equal(labels.astype(np.int64), G[f"infer.k{k}.labels"]), k
        ref = G[f"infer.k{k}.preds"]
        # the votes compare float64 likelihoods of fp32 embeddings with a threshold that IS one of those likelihoods:
        # an embedding that differs in its last bits can move a vote only at that boundary
        assert preds.shape == ref.shape and (preds != ref).mean() <= 0.02, (k, (preds != ref).mean())
        m = G[f"infer.k{k}.metrics"]
        got = [np.equal(labels, preds).mean(), f1_score(labels, preds, average="micro"),
               f1_score(labels, preds, average="macro"), f1_score(labels, preds, average="weighted")]
        assert np.allclose(got, m, atol=0.03), (k, got, m)
