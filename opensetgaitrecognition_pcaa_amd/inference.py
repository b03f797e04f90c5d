"""Open-set inference of the PCAA path on the HIP device (reference
``inference_PCAA.py``: ``CGAAE_inference_setup`` :34-114, ``naive_sequential_procedure``
:117-347, ``CGAAE_inference`` :382-469).

The reference evaluates the encoder one crop at a time (batch 1) and scores each embedding
with scipy on the host.  Here the eval-mode encoder runs batched (BatchNorm uses running
statistics, so every sequence is independent: the batched result equals the per-crop one),
the float64 mixture likelihood and the k-window vote run as HIP kernels, and only the ROC /
Youden threshold (a sort over a few thousand scores) stays on the host.  F1 / confusion
matrix plotting is reporting and out of scope.
"""
import ctypes
import os
import pickle

import numpy as np
import torch

from . import _lib, constants
from . import functional as F_hip
from . import ops
from ._lib import check
from .models import CGEncoder


def joint_likelihood(sup_fv: torch.Tensor, means: torch.Tensor) -> torch.Tensor:
    """[B,32] fp32 embeddings, [K,32] fp32 centroids -> [B] float64 likelihoods."""
    ops._chk(sup_fv, "joint_likelihood.x", torch.float32, 2)
    ops._chk(means, "joint_likelihood.means", torch.float32, 2)
    B, D = sup_fv.shape
    K = means.shape[0]
    if means.shape[1] != D:
        raise ValueError("joint_likelihood: dimension mismatch")
    out = torch.empty(B, dtype=torch.float64, device=sup_fv.device)
    check(_lib.load().pcaa_joint_likelihood(ops._p(sup_fv), ops._p(means), B, K, D, ops._p(out), ops._s()),
          "pcaa_joint_likelihood")
    return out


def k_vote(lik: torch.Tensor, preds: torch.Tensor, threshold: float, k: int, n_labels: int,
           n_classes: int = None) -> torch.Tensor:
    """Windows of k consecutive crops (trailing partial window dropped, like DataLoader
    drop_last=True) -> [n_windows] int64 open-set predictions.  ``n_labels`` is the "unknown" id (the number
    of distinct labels in the known test split); the majority is taken over the encoder's ``n_classes``
    outputs (``np.argmax(np.bincount(preds))``, inference_PCAA.py:265-266) -- a test split that lacks a trained
    class must not drop the votes for it.  Default ``n_classes = n_labels``."""
    ops._chk(lik, "k_vote.lik", torch.float64, 1)
    ops._chk(preds, "k_vote.preds", torch.int64, 1)
    n_classes = int(n_labels if n_classes is None else n_classes)
    nwin = lik.numel() // k
    out = torch.empty(nwin, dtype=torch.int64, device=lik.device)
    if nwin:
        check(_lib.load().pcaa_kvote(ops._p(lik), ops._p(preds), ctypes.c_double(float(threshold)), int(k),
                                     int(n_labels), n_classes, nwin, ops._p(out), ops._s()), "pcaa_kvote")
    return out


def youden_threshold(known_mask: np.ndarray, scores: np.ndarray) -> float:
    """``thresholds[argmax(tpr - fpr)]`` of ``sklearn.metrics.roc_curve`` (default
    ``drop_intermediate=True``) as used at inference_PCAA.py:230-231: scores sorted
    descending (stable), one candidate per distinct score, collinear points dropped,
    (0,0) with threshold +inf prepended.  Host numpy: a sort of a few thousand float64."""
    y = np.asarray(known_mask, dtype=np.float64)
    s = np.asarray(scores, dtype=np.float64)
    order = np.argsort(s, kind="mergesort")[::-1]
    s, y = s[order], y[order]
    idx = np.r_[np.where(np.diff(s))[0], y.size - 1]
    tps = np.cumsum(y)[idx]
    fps = 1 + idx - tps
    thr = s[idx]
    if len(fps) > 2:
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    tps, fps, thr = np.r_[0, tps], np.r_[0, fps], np.r_[np.inf, thr]
    return float(thr[np.argmax(tps / tps[-1] - fps / fps[-1])])


class OpenSetScorer:
    """Encoder + centroids -> predictions, likelihoods, threshold, k-window votes."""

    def __init__(self, encoder: CGEncoder, discriminator_means: torch.Tensor, batch_size: int = 1024):
        self.encoder = encoder.eval()
        dev = next(encoder.parameters()).device
        self.means = discriminator_means.float().to(dev).contiguous()
        self.batch_size = batch_size
        self.threshold = None

    @torch.no_grad()
    def embed(self, pcs: torch.Tensor):
        """pcs [M,C,T,N] on the device -> (preds [M] int64, sup_fv [M,32], likelihood [M] f64)."""
        preds, fvs = [], []
        for i in range(0, pcs.shape[0], self.batch_size):
            logits, sup_fv, _ = F_hip.encoder_forward(self.encoder, pcs[i:i + self.batch_size], False)
            _, _, p = ops.cross_entropy(logits, None, want_loss=False, want_preds=True)
            preds.append(p)
            fvs.append(sup_fv)
        preds, fvs = torch.cat(preds), torch.cat(fvs)
        return preds, fvs, joint_likelihood(fvs.contiguous(), self.means)

    def fit_threshold(self, known_lik: torch.Tensor, unseen_valid_lik: torch.Tensor) -> float:
        """ROC-optimal (Youden J) separation of known-test vs held-out-unseen likelihoods
        (inference_PCAA.py:225-231: unseen first with label 0, known with label 1)."""
        scores = np.concatenate([unseen_valid_lik.cpu().numpy(), known_lik.cpu().numpy()])
        labels = np.concatenate([np.zeros(unseen_valid_lik.numel()), np.ones(known_lik.numel())])
        self.threshold = youden_threshold(labels, scores)
        return self.threshold

    def vote(self, lik: torch.Tensor, preds: torch.Tensor, k: int, n_labels: int) -> torch.Tensor:
        if self.threshold is None:
            raise RuntimeError("fit_threshold() first")
        return k_vote(lik, preds, self.threshold, k, n_labels, n_classes=self.encoder.MLP_sup2[0].weight.shape[0])


def _variation_uses_head(variation, model_name=""):
    """Projection head on the encoder for variants 1, 2 and 4 (inference_PCAA.py:75-84); ``variation`` may be
    the reference's VARIATION enum, its name ("V1"...), True/False, or be inferred from the model name's
    ``...V4`` suffix (:402-414)."""
    if variation is True:
        return True
    name = getattr(variation, "name", variation) if variation else model_name.split(".")[0][-2:]
    return str(name).upper() in ("V1", "V2", "V4")


def CGAAE_inference_setup(model_name, loaders_batch_size=1, variation=False, generate_dataset=True,
                          force_pc_subsampling=0, device=None):
    """Load ``models/<name>/config.pkl``, ``<name>_E.pt`` and ``discriminator_means.pt``
    (checkpoint format of the reference) -> (encoder.eval(), means on the device); with
    ``generate_dataset`` the splits are regenerated from the raw tracks first for the model's
    TRAIN_CLASSES / NMAX (inference_PCAA.py:66-72)."""
    folder = os.path.join("models", model_name)
    with open(os.path.join(folder, "config.pkl"), "rb") as f:
        config = pickle.load(f)
    if generate_dataset:
        from .datasets import generate_splits
        generate_splits(train_classes=config["TRAIN_CLASSES"], seed=0, force_pc_subsampling=force_pc_subsampling,
                        nmax_points=config["NMAX"], verbose=False)
    dev = torch.device(device or constants.DEVICE)
    enc = CGEncoder(n_out_labels=len(config["TRAIN_CLASSES"]),
                    use_projection_head=_variation_uses_head(variation, model_name),
                    nmax_points=config["NMAX"]).to(dev).float()
    enc.load_state_dict(torch.load(os.path.join(folder, f"{model_name}_E.pt"), map_location=dev))
    means = torch.load(os.path.join(folder, "discriminator_means.pt"), map_location=dev)
    return enc.eval(), means


def naive_sequential_procedure(k, encoder, discriminator_means, *args, **kwargs):
    """Two call forms.

    The REFERENCE's (inference_PCAA.py:117-125), taken when the fourth argument is a path:
    ``naive_sequential_procedure(k, encoder, discriminator_means, figures_folder, model_folder,
    scenarios_list=constants.TRAIN_SCENARIOS, seed=0, unseen_valid_ratio=0.2, force_pc_subsampling=0)``
    -> ``(out_log, final_preds, final_labels)``: the sequentially ordered TEST / UNSEEN splits are read from the generated
    dataset (packed once into an HBM-resident store), the procedure runs on the device, and ``naive_seq_log_{k}*.json`` is
    written into ``model_folder`` under the reference's three file names.  Not reproduced: the confusion-matrix PNG in
    ``figures_folder`` (plotting is out of scope; the folder is only created).

    The in-memory form the drivers and tests use: ``naive_sequential_procedure_tensors`` below (crops already on the device)
    -> ``(final_preds, final_labels, threshold)``."""
    if args and isinstance(args[0], (str, os.PathLike)) or "figures_folder" in kwargs:
        return _naive_sequential_procedure_files(k, encoder, discriminator_means, *args, **kwargs)
    return naive_sequential_procedure_tensors(k, encoder, discriminator_means, *args, **kwargs)


def _naive_sequential_procedure_files(k, encoder, discriminator_means, figures_folder, model_folder,
                                      scenarios_list=None, seed=0, unseen_valid_ratio=0.2, force_pc_subsampling=0):
    import json
    from sklearn.metrics import f1_score
    from .constants import SPLIT
    scenarios_list = constants.TRAIN_SCENARIOS if scenarios_list is None else scenarios_list
    default_scen = list(scenarios_list) == list(constants.TRAIN_SCENARIOS)
    dev = next(encoder.parameters()).device
    os.makedirs(figures_folder, exist_ok=True)
    known_pcs, known_labels = _sequential_split_on_device(SPLIT.TEST, scenarios_list, dev)
    unseen_pcs, unseen_labels = _sequential_split_on_device(SPLIT.UNSEEN, scenarios_list, dev)
    preds, labels, _ = naive_sequential_procedure_tensors(k, encoder, discriminator_means, known_pcs, known_labels, unseen_pcs,
                                                          unseen_labels, seed=seed, unseen_valid_ratio=unseen_valid_ratio)
    labels = labels.astype(int)
    out_log = {"n_steps": k, "accuracy": float(np.equal(labels, preds).sum() / max(len(labels), 1)),
               "f1_micro": float(f1_score(labels, preds, average="micro")),
               "f1_macro": float(f1_score(labels, preds, average="macro")),
               "f1_weighted": float(f1_score(labels, preds, average="weighted"))}
    if force_pc_subsampling and default_scen:
        name = f"naive_seq_log_{k}_subsampled{force_pc_subsampling}.json"
    elif not force_pc_subsampling and not default_scen:
        name = f"naive_seq_log_{k}_scenarios" + "_".join(sc.value for sc in scenarios_list) + ".json"
    else:
        name = f"naive_seq_log_{k}.json"
    with open(os.path.join(model_folder, name), "w") as f:
        json.dump(out_log, f)
    return out_log, preds, labels


def naive_sequential_procedure_tensors(k, encoder, discriminator_means, known_pcs, known_labels, unseen_pcs,
                                       unseen_labels, seed=0, unseen_valid_ratio=0.2, batch_size=1024):
    """The reference's procedure on in-memory, temporally ordered crops: (1) likelihoods of
    known-test and unseen crops, 20 % of the unseen SUBJECTS (rng seed 0) held out to pick the
    threshold; (2) k-window votes on the known test set and on the remaining unseen subjects.
    Returns (open-set predictions, open-set labels, threshold)."""
    rng = np.random.default_rng(seed)
    scorer = OpenSetScorer(encoder, discriminator_means, batch_size)
    n_labels = int(len(np.unique(known_labels.cpu().numpy())))
    u_lab = unseen_labels.cpu().numpy()
    subjects = np.unique(u_lab)
    val_subjects = rng.choice(subjects, size=int(np.ceil(unseen_valid_ratio * len(subjects))), replace=False)
    val_mask = np.isin(u_lab, val_subjects)
    k_preds, _, k_lik = scorer.embed(known_pcs)
    u_preds, _, u_lik = scorer.embed(unseen_pcs)
    vm = torch.from_numpy(val_mask).to(u_lik.device)
    thr = scorer.fit_threshold(k_lik, u_lik[vm])
    preds, labels = [], []

    def windows(lik, pr, lab, unknown):
        lab_np = lab.cpu().numpy()
        n = (len(lab_np) // k) * k
        votes = scorer.vote(lik[:n].contiguous(), pr[:n].contiguous(), k, n_labels).cpu().numpy()
        for w in range(n // k):
            seg = lab_np[w * k:(w + 1) * k]
            if len(np.unique(seg)) != 1:
                continue                       # windows straddling two subjects are skipped (:243-245)
            if unknown and seg[0] in val_subjects:
                continue                       # validation subjects only chose the threshold (:286)
            preds.append(int(votes[w]))
            labels.append(n_labels if unknown else int(seg[0]))

    # windows are cut over the WHOLE sequential set, as the reference's DataLoader(batch_size=k) does
    windows(k_lik, k_preds, known_labels, False)
    windows(u_lik, u_preds, unseen_labels, True)
    return np.asarray(preds), np.asarray(labels), thr


def _sequential_split_on_device(split, scenarios_list, device):
    """A split's crops in the sequential order of ``MSRadarDataset(sequential=True)`` as device tensors
    ([M,C,T,N] view of the packed point-major store, labels [M])."""
    from .batcher import PackedCrops, pack_split
    from .datasets import MSRadarDataset
    ds = MSRadarDataset(split, scenarios=scenarios_list, sequential=True)
    cache = str(ds.dataset_dir).rstrip("/") + "_packed_seq"
    pack_split(ds, cache)
    crops, labels = PackedCrops(cache).to_device(device)
    return crops.permute(0, 3, 1, 2), labels


def CGAAE_inference(model_names, ks, force_pc_subsampling=0, scenarios_list=None, variation=False,
                    generate_dataset=True, device=None):
    """Open-set evaluation driver with the reference's call surface and output files
    (inference_PCAA.py:382-469): for every model and k, the naive sequential procedure on the sequentially
    ordered test / unseen splits; writes ``naive_seq_log_{k}*.json`` (accuracy, F1 micro / macro / weighted),
    ``final_preds_{k}*.npy`` / ``final_labels_{k}*.npy`` and ``naive_seq_log_subsampled{n}.json`` under
    ``models/<name>/``.  Not reproduced: the confusion-matrix PNG (plotting).  The splits are regenerated once per
    call (the reference regenerates them for every (model, k) with identical arguments)."""
    import json
    from sklearn.metrics import f1_score
    from .constants import SPLIT
    scenarios_list = constants.TRAIN_SCENARIOS if scenarios_list is None else scenarios_list
    default_scen = list(scenarios_list) == list(constants.TRAIN_SCENARIOS)
    if force_pc_subsampling and not default_scen:
        raise ValueError("force_pc_subsampling and scenarios_list cannot be both different from default")
    dev = torch.device(device or constants.DEVICE)
    if force_pc_subsampling and default_scen:
        suffix = f"_subsampled{force_pc_subsampling}"
    elif not force_pc_subsampling and not default_scen:
        suffix = "_scenarios" + "_".join(sc.value for sc in scenarios_list)
    else:
        suffix = ""
    out_log = {}
    generated = not generate_dataset
    for model_name in model_names:
        folder = os.path.join("models", model_name)
        os.makedirs(os.path.join("figures", model_name), exist_ok=True)
        enc, means = CGAAE_inference_setup(model_name, 32, variation, generate_dataset=not generated,
                                           force_pc_subsampling=force_pc_subsampling, device=dev)
        generated = True
        known_pcs, known_labels = _sequential_split_on_device(SPLIT.TEST, scenarios_list, dev)
        unseen_pcs, unseen_labels = _sequential_split_on_device(SPLIT.UNSEEN, scenarios_list, dev)
        for k in ks:
            preds, labels, thr = naive_sequential_procedure(k, enc, means, known_pcs, known_labels, unseen_pcs,
                                                            unseen_labels, seed=0, unseen_valid_ratio=0.2)
            labels = labels.astype(int)
            metrics = {"n_steps": k, "accuracy": float(np.equal(labels, preds).sum() / max(len(labels), 1)),
                       "f1_micro": float(f1_score(labels, preds, average="micro")),
                       "f1_macro": float(f1_score(labels, preds, average="macro")),
                       "f1_weighted": float(f1_score(labels, preds, average="weighted"))}
            with open(os.path.join(folder, f"naive_seq_log_{k}{suffix}.json"), "w") as f:
                json.dump(metrics, f)
            np.save(os.path.join(folder, f"final_preds_{k}{suffix}.npy"), preds)
            np.save(os.path.join(folder, f"final_labels_{k}{suffix}.npy"), labels)
            out_log[k] = {m: metrics[m] for m in ("f1_micro", "f1_macro", "f1_weighted")}
        with open(os.path.join(folder, f"naive_seq_log_subsampled{force_pc_subsampling}.json"), "w") as f:
            json.dump(out_log, f)
    return out_log

