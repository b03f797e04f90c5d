"""OR-CED baseline with the reference's call surface: ``train_ORCED(config)``
(``train_ORCED.py:21-280``), ``compute_prob`` / ``ORCED_ensemble_ood_detection`` / ``ORCED_inference_setup`` /
``ORCED_inference`` (``inference_ORCED.py:18-456``).

What runs where: the PointNet / temporal-conv trunk of ``ORCEDEncoder``, ``ORCEDDecoder``, the sequence Chamfer loss,
``GaussianMeanLearner`` and Adam are the HIP kernels of the PCAA path (through the drop-in modules' autograd
Functions and a :class:`~.train.FlatBuffer`); since round 3 so are the [B,32]-sized pieces: the three Linear heads with
the reparametrisation (``pcaa_orced_heads_fwd / _bwd``), cross-entropy (``pcaa_cross_entropy``) and the KL term
(``pcaa_orced_kl``).  What stays on torch device ops: the normal draw itself (the reference's ``torch.randn_like``) and
the triplet term (below: parity unpinned).  The open-set test is host numpy / scipy in float64, as in the reference.

The triplet term restates ``pytorch_metric_learning==1.6.0`` (requirements.txt; NOT installed here, not vendored in
the reference): ``miners.MultiSimilarityMiner()`` (epsilon 0.1 on cosine similarity) feeding
``losses.TripletMarginLoss(margin)`` (Euclidean distance of L2-normalised embeddings, all (anchor, positive,
negative) combinations of the mined pairs, mean over the non-zero losses).  **Parity unpinned** for these two
functions: there is nothing in this container to check the restatement against.
"""
import itertools
import os
import pickle

import numpy as np
import torch
import torch.nn.functional as TF

from . import constants
from . import functional as F_hip
from .models import GaussianMeanLearner, ORCEDDecoder, ORCEDEncoder
from .train import FlatBuffer, _NullRun, _wandb
from .utils import CG_kl_divergence, SeqChamferLoss, save_model


# ---------------------------------------------------------------------------------------------------------
# pytorch_metric_learning 1.6.0 restated (published algorithms; parity unpinned)
# ---------------------------------------------------------------------------------------------------------
def multi_similarity_miner(embeddings, labels, epsilon=0.1):
    """``miners.MultiSimilarityMiner(epsilon=0.1)`` with its default ``CosineSimilarity`` distance: for every anchor,
    the positives whose similarity is below (hardest negative's similarity + epsilon) and the negatives whose
    similarity is above (hardest positive's similarity - epsilon).  Returns (a1, p, a2, n) index tensors."""
    e = TF.normalize(embeddings, p=2, dim=1)
    mat = e @ e.t()
    same = labels.unsqueeze(1) == labels.unsqueeze(0)
    eye = torch.eye(len(labels), dtype=torch.bool, device=labels.device)
    pos_mask, neg_mask = same & ~eye, ~same
    if not pos_mask.any() or not neg_mask.any():
        z = torch.zeros(0, dtype=torch.long, device=labels.device)
        return z, z.clone(), z.clone(), z.clone()
    inf = torch.finfo(mat.dtype).max
    mat_pos = mat.masked_fill(~pos_mask, inf)        # non-positives (negatives, the diagonal) out of the way: +inf
    mat_neg = mat.masked_fill(~neg_mask, -inf)       # non-negatives: -inf
    pos_sorted, pos_idx = torch.sort(mat_pos, dim=1)
    neg_sorted, neg_idx = torch.sort(mat_neg, dim=1)
    hard_pos = torch.where(pos_sorted - epsilon < neg_sorted[:, -1].unsqueeze(1))
    hard_neg = torch.where(neg_sorted + epsilon > pos_sorted[:, 0].unsqueeze(1))
    a1, p = hard_pos[0], pos_idx[hard_pos[0], hard_pos[1]]
    a2, n = hard_neg[0], neg_idx[hard_neg[0], hard_neg[1]]
    # the sort carried the masked entries along: keep real positives / negatives only
    keep_p, keep_n = pos_mask[a1, p], neg_mask[a2, n]
    return a1[keep_p], p[keep_p], a2[keep_n], n[keep_n]


def triplet_margin_loss(embeddings, labels, pairs, margin):
    """``losses.TripletMarginLoss(margin)`` on mined pairs: triplets = every (positive pair, negative pair) with the
    same anchor; ``relu(d(a,p) - d(a,n) + margin)`` with d the Euclidean distance of the L2-normalised embeddings;
    ``AvgNonZeroReducer`` (mean over the losses > 0, zero if there is none)."""
    a1, p, a2, n = pairs
    if a1.numel() == 0 or a2.numel() == 0:
        return embeddings.sum() * 0.0
    pi, ni = torch.where(a1.unsqueeze(1) == a2.unsqueeze(0))
    if pi.numel() == 0:
        return embeddings.sum() * 0.0
    a, pp, nn = a1[pi], p[pi], n[ni]
    e = TF.normalize(embeddings, p=2, dim=1)
    d = torch.cdist(e, e, p=2)
    loss = torch.relu(d[a, pp] - d[a, nn] + margin)
    nz = loss > 0
    return loss[nz].mean() if nz.any() else embeddings.sum() * 0.0


def _predicted(logits):
    """argmax(softmax(logits)) with the first index on ties (train_ORCED.py:181) on the PCAA step's kernel"""
    from . import ops
    return ops.cross_entropy(logits.detach().contiguous().float(), None, want_loss=False, want_preds=True)[2]


# ---------------------------------------------------------------------------------------------------------
# training loop (train_ORCED.py:21-280)
# ---------------------------------------------------------------------------------------------------------
def orced_losses(encoder, decoder, mean_learner, pcs, gt_labels, config, kl_multiplier, chamfer=None):
    """The loss terms of one OR-CED step (train_ORCED.py:143-176), weighted.  Returns a dict of scalar tensors
    (``tot`` carries the graph) plus the predicted labels."""
    K = len(config["TRAIN_CLASSES"])
    chamfer = chamfer or SeqChamferLoss()
    logits, sup_fvs, vae_mu, vae_logvar = encoder(pcs)
    rec_pcs = decoder(sup_fvs)
    mu_gts = mean_learner(TF.one_hot(gt_labels, num_classes=K).float())
    rec = config["REC_W"] * chamfer(rec_pcs, pcs)
    sup = config["CE_W"] * F_hip.cross_entropy_loss(logits, gt_labels)
    nfv = TF.normalize(sup_fvs, p=2, dim=1)
    trip = config["TRIPLET_W"] * triplet_margin_loss(nfv, gt_labels, multi_similarity_miner(nfv, gt_labels),
                                                     config["TRIPLET_MARGIN"])
    kl = config["KL_W"] * CG_kl_divergence(vae_mu, vae_logvar, mu_gts) * kl_multiplier
    preds = _predicted(logits)
    return {"rec": rec, "sup": sup, "trip": trip, "kl": kl, "tot": rec + sup + trip + kl, "preds": preds}


def train_ORCED(config=None, dataset_factory=None, log_fn=None, device=None):
    """OR-CED training with the reference's call surface; returns (modules dict, per-epoch records).  One Adam over
    encoder + decoder + mean learner with betas ``(B1, B1)`` (train_ORCED.py:96-101; the repeated B1 is the
    reference's), KL weight ramped as ``epoch / EPOCHS``, best-valid checkpoints ``_E/_G/_ML.pt``."""
    from .constants import SPLIT
    from .datasets import MSRadarDataset
    config = constants.CONFIG if config is None else config
    dev = torch.device(device or constants.DEVICE)
    os.makedirs(f"models/{config['MODEL_NAME']}", exist_ok=True)
    with open(os.path.join("models", config["MODEL_NAME"], "config.pkl"), "wb") as f:
        pickle.dump(config, f)
    K = len(config["TRAIN_CLASSES"])
    nmax = config.get("NMAX", constants.NMAX)
    encoder = ORCEDEncoder(n_out_labels=K, nmax_points=nmax).to(dev).float()
    decoder = ORCEDDecoder(nmax_points=nmax).to(dev).float()
    mean_learner = GaussianMeanLearner(n_in_labels=K).to(dev).float()
    make = dataset_factory or (lambda split: MSRadarDataset(split, subsample_factor=config["SUBSAMPLE_FACTOR"]))
    loader = lambda ds, shuffle: torch.utils.data.DataLoader(ds, batch_size=config["BATCH_SIZE"], drop_last=True,
                                                             shuffle=shuffle, num_workers=0)
    loader_train, loader_valid = loader(make(SPLIT.TRAIN), True), loader(make(SPLIT.VALID), False)
    # one flat fp32 buffer for the optimizer: parameters re-pointed into it, autograd accumulates into its gradient
    # views, one fused Adam launch per step (the decoder's unused bn1..4 get no gradient: left out, as Adam skips them)
    named = [("E." + n, p) for n, p in encoder.named_parameters()]
    named += [("G." + n, p) for n, p in decoder.named_parameters() if n.startswith("dense")]
    named += [("ML." + n, p) for n, p in mean_learner.named_parameters()]
    flat = FlatBuffer(named, dev)
    for (name, p) in named:
        p.grad = flat.grad_views[name]
    chamfer = SeqChamferLoss()
    wb = _wandb()
    run = _NullRun()
    if wb is not None and hasattr(wb, "init"):
        wb.login()
        run = wb.init(project=constants.WANDB_PROJECT, config=config, name=config["MODEL_NAME"], notes=config["NOTES"],
                      reinit=True, mode=constants.WANDB_MODE)
    best_valid_accuracy, history = 0, []
    for epoch in range(config["EPOCHS"]):
        kl_multiplier = epoch / config["EPOCHS"]
        encoder.train(); decoder.train(); mean_learner.train()
        acc = {k: [] for k in ("rec", "sup", "trip", "kl", "tot")}
        ys, y_hats = [], []
        for pcs, gt_labels in loader_train:
            pcs, gt_labels = pcs.to(dev), gt_labels.to(dev)
            out = orced_losses(encoder, decoder, mean_learner, pcs, gt_labels, config, kl_multiplier, chamfer)
            for k in acc:
                acc[k].append(out[k].detach())
            y_hats.append(out["preds"]); ys.append(gt_labels)
            out["tot"].backward()
            flat.adam(config["LR"], config["B1"], config["B1"])
            flat.g.zero_()
        encoder.eval(); decoder.eval(); mean_learner.eval()
        v_rec, v_ce, v_hat, v_y = [], [], [], []
        with torch.no_grad():
            for pcs, gt_labels in loader_valid:
                pcs, gt_labels = pcs.to(dev), gt_labels.to(dev)
                logits, sup_fv, _, _ = encoder(pcs)
                v_rec.append(config["REC_W"] * chamfer(decoder(sup_fv), pcs))
                v_ce.append(config["CE_W"] * TF.cross_entropy(logits, gt_labels))
                v_hat.append(_predicted(logits)); v_y.append(gt_labels)
        mean = lambda xs: float(torch.stack(xs).double().mean().item()) if xs else float("nan")
        record = {
            "Reconstruction Loss Train": mean(acc["rec"]), "Reconstruction Loss Valid": mean(v_rec),
            "Cross Entropy Loss Train": mean(acc["sup"]), "Cross Entropy Loss Valid": mean(v_ce),
            "Triplet Loss": mean(acc["trip"]), "KL Loss": mean(acc["kl"]), "Total Loss Train": mean(acc["tot"]),
            "Train Accuracy": float((torch.cat(ys) == torch.cat(y_hats)).double().mean().item()),
            "Valid Accuracy": float((torch.cat(v_y) == torch.cat(v_hat)).double().mean().item()) if v_y else 0.0,
        }
        history.append(record)
        if log_fn is not None:
            log_fn(record)
        elif wb is not None and hasattr(wb, "log"):
            wb.log(record)
        print(f"[Epoch {epoch}/{config['EPOCHS']}] " + " ".join(f"[{k}: {v:.4f}]" for k, v in record.items()))
        if epoch % config["CHECKPOINT_FREQUENCY"] == 0 and record["Valid Accuracy"] > best_valid_accuracy:
            best_valid_accuracy = record["Valid Accuracy"]
            base = os.path.join("models", config["MODEL_NAME"], config["MODEL_NAME"])
            save_model(encoder, base + "_E.pt")
            save_model(decoder, base + "_G.pt")
            save_model(mean_learner, base + "_ML.pt")
    run.finish()
    return {"E": encoder, "G": decoder, "ML": mean_learner}, history


# ---------------------------------------------------------------------------------------------------------
# open-set test (inference_ORCED.py:18-132)
# ---------------------------------------------------------------------------------------------------------
def compute_prob(mean, cov, z_test):
    """``mvn.cdf(b) - mvn.cdf(a)`` with ``a, b = mean -+ |z_test - mean|`` (inference_ORCED.py:18-45), float64.
    The reference calls scipy's ``multivariate_normal(mean, cov).cdf``; every call site passes a DIAGONAL ``cov``
    (``np.diag(stds_z[k])`` -- the class's standard deviations used as variances, :107), for which the orthant
    probability factorises exactly: cdf(v) = prod_d Phi((v_d - mean_d) / sqrt(cov_dd)).  That closed form is what is
    evaluated here (scipy integrates the same quantity numerically, to ~1e-5 absolute, with a randomised rule)."""
    from scipy.special import ndtr
    mean = np.asarray(mean, dtype=np.float64)
    cov = np.asarray(cov, dtype=np.float64)
    z = np.atleast_2d(np.asarray(z_test, dtype=np.float64))
    if cov.ndim == 2:
        if np.count_nonzero(cov - np.diag(np.diag(cov))):
            raise NotImplementedError("compute_prob: only the diagonal covariances the procedure uses are supported")
        var = np.diag(cov)
    else:
        var = cov
    sd = np.sqrt(var)
    dev = np.abs(z - mean)
    p = np.prod(ndtr(dev / sd), axis=1) - np.prod(ndtr(-dev / sd), axis=1)
    return p if np.ndim(z_test) > 1 else p[0]


def ORCED_ensemble_ood_detection(rec_err_tr, f_vecs_tr, thresholds_g, gt_labels, pred_labels, x_test_prediction,
                                 z_test, re_test):
    """The ensemble out-of-distribution rule (inference_ORCED.py:48-132): per class k, the latent test rejects a
    sample when ``compute_prob`` around the class's training mean exceeds ``thresholds_g`` for EVERY class; the
    reconstruction test rejects when the error exceeds mean + 2 std of the predicted class's training errors;
    rejected by either -> label ``n_classes`` (unknown).  Host numpy float64; ``x_test_prediction`` a tensor."""
    n_classes = len(np.unique(gt_labels))
    correct = gt_labels == pred_labels
    means_re, std_re, means_z, stds_z, thr_re = [], [], [], [], []
    for k in range(n_classes):
        means_re.append(np.mean(rec_err_tr[gt_labels == k]))
        std_re.append(np.std(rec_err_tr[gt_labels == k]))
        sel = f_vecs_tr[correct][gt_labels[correct] == k]
        means_z.append(np.mean(sel, axis=0))
        stds_z.append(np.std(sel, axis=0))
        thr_re.append(means_re[k] + 2 * std_re[k])
    p_z_ks = np.array([compute_prob(means_z[k], np.diag(stds_z[k]), z_test) for k in range(n_classes)])
    p_zs_mask = np.less(1 - p_z_ks, 1 - thresholds_g)
    latent_bools = np.sum(p_zs_mask, axis=0) == n_classes
    pred = x_test_prediction.detach().cpu() if torch.is_tensor(x_test_prediction) else torch.as_tensor(x_test_prediction)
    rec_err_bools = re_test > np.array([thr_re[j] for j in pred.tolist()])
    out = torch.clone(pred)
    out[torch.from_numpy(np.logical_or(latent_bools, rec_err_bools))] = n_classes
    return out


def ORCED_inference_setup(model_name, loaders_batch_size, generate_dataset=True, device=None):
    """(encoder, decoder, mean_learner, cluster_means, train / test / unseen loaders) from the checkpoints of
    ``train_ORCED`` (inference_ORCED.py:135-254)."""
    from .constants import SPLIT
    from .datasets import MSRadarDataset, generate_splits
    folder = os.path.join("models", model_name)
    with open(os.path.join(folder, "config.pkl"), "rb") as f:
        config = pickle.load(f)
    nmax = config.get("NMAX", constants.NMAX)
    if generate_dataset:
        generate_splits(train_classes=config["TRAIN_CLASSES"], seed=0, nmax_points=nmax, verbose=False)
    dev = torch.device(device or constants.DEVICE)
    K = len(config["TRAIN_CLASSES"])
    encoder = ORCEDEncoder(n_out_labels=K, nmax_points=nmax).to(dev).float()
    decoder = ORCEDDecoder(nmax_points=nmax).to(dev).float()
    mean_learner = GaussianMeanLearner(n_in_labels=K).to(dev).float()
    base = os.path.join(folder, config["MODEL_NAME"])
    encoder.load_state_dict(torch.load(base + "_E.pt", map_location=dev))
    decoder.load_state_dict(torch.load(base + "_G.pt", map_location=dev))
    mean_learner.load_state_dict(torch.load(base + "_ML.pt", map_location=dev))
    encoder.eval(); decoder.eval(); mean_learner.eval()
    mk = lambda split, drop: torch.utils.data.DataLoader(
        MSRadarDataset(split, subsample_factor=config["SUBSAMPLE_FACTOR"]), batch_size=loaders_batch_size,
        drop_last=drop, shuffle=False, num_workers=0)
    with torch.no_grad():
        cluster_means = mean_learner(TF.one_hot(torch.arange(0, K), num_classes=K).float().to(dev))
    return (encoder, decoder, mean_learner, cluster_means, mk(SPLIT.TRAIN, True), mk(SPLIT.TEST, False),
            mk(SPLIT.UNSEEN, False))


def ORCED_inference(model_names, generate_dataset=True, device=None):
    """Open-set evaluation of OR-CED models (inference_ORCED.py:257-456): training-set statistics, then the ensemble
    rule on the test split and on the unseen split minus its first subject (the reference leaves one unseen subject
    out for a like-for-like comparison with the PCAA procedure).  Writes ``ensemble_ood_final_preds_fixed.npy`` /
    ``..._labels_fixed.npy`` under ``figures/<name>/``; returns {name: metrics}.  Not reproduced: the PNG."""
    from sklearn.metrics import f1_score
    chamfer = SeqChamferLoss()
    dev = torch.device(device or constants.DEVICE)
    results = {}
    for model_name in model_names:
        figures = os.path.join("figures", model_name)
        os.makedirs(figures, exist_ok=True)
        encoder, decoder, _, _, dl_train, dl_test, dl_unseen = ORCED_inference_setup(model_name, 64, generate_dataset, dev)

        def run(pcs):
            logits, sup_fvs, _, _ = encoder(pcs)
            rec_err = chamfer(decoder(sup_fvs), pcs, avg_out=False)
            return _predicted(logits), sup_fvs.cpu().numpy(), rec_err.cpu().numpy()

        fv, re, pl, gl = [], [], [], []
        with torch.no_grad():
            for pcs, gt in dl_train:
                p, f, r = run(pcs.to(dev))
                fv.append(f); re.append(r); pl.append(p.cpu().numpy()); gl.append(gt.numpy())
        rec_err_tr, f_vecs_tr = np.concatenate(re), np.concatenate(fv)
        gt_labels, pred_labels = np.concatenate(gl), np.concatenate(pl)
        n_labels = len(np.unique(gt_labels))
        test_out, test_lab, unseen_out = [], [], []
        with torch.no_grad():
            for pcs, gt in dl_test:
                p, f, r = run(pcs.to(dev))
                test_lab.append(gt.numpy())
                test_out.append(ORCED_ensemble_ood_detection(rec_err_tr, f_vecs_tr, 0.95, gt_labels, pred_labels, p, f, r))
            leave_out = None
            for pcs, gt in dl_unseen:
                if leave_out is None:
                    leave_out = gt[0].item()
                p, f, r = run(pcs.to(dev))            # (the draw of eps happens for every batch, as in the reference)
                if gt[0].item() != leave_out:
                    unseen_out.append(ORCED_ensemble_ood_detection(rec_err_tr, f_vecs_tr, 0.95, gt_labels,
                                                                   pred_labels, p, f, r))
        test_out = np.concatenate(test_out)
        unseen_out = np.concatenate(unseen_out) if unseen_out else np.zeros(0, dtype=np.int64)
        final_preds = np.concatenate([test_out, unseen_out])
        final_labels = np.concatenate([np.concatenate(test_lab), [n_labels] * len(unseen_out)])
        results[model_name] = {
            "accuracy": float(np.equal(final_labels, final_preds).sum() / len(final_labels)),
            "f1_micro": float(f1_score(final_labels, final_preds, average="micro")),
            "f1_macro": float(f1_score(final_labels, final_preds, average="macro")),
            "f1_weighted": float(f1_score(final_labels, final_preds, average="weighted"))}
        np.save(os.path.join(figures, "ensemble_ood_final_preds_fixed.npy"), final_preds)
        np.save(os.path.join(figures, "ensemble_ood_final_labels_fixed.npy"), final_labels)
    return results
