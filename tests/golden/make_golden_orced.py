#!/usr/bin/env python3
"""Golden for the OR-CED baseline, made by importing the reference's models / utils / inference_ORCED:

* one train step of the OR-CED loss WITHOUT its triplet term (``rec + sup + kl``, train_ORCED.py:143-176 with
  TRIPLET_W = 0; the triplet miner / loss come from pytorch_metric_learning, which is not installed -- that term is
  "parity unpinned") on the reference's ORCEDEncoder / ORCEDDecoder / GaussianMeanLearner with torch.optim.Adam
  betas (B1, B1): outputs, loss terms, gradient records, post-step parameters.  The reparametrisation draw
  ``torch.randn_like`` is replaced by a recorded tensor on both sides (CPU and device generators differ);
* ``CG_kl_divergence`` values; ``compute_prob`` (scipy's mvn.cdf) and ``ORCED_ensemble_ood_detection`` on fixed inputs.

    python tests/golden/make_golden_orced.py  ->  tests/golden/orced.npz"""
import itertools
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

rmodels, rutils, rconst, syn, T = mg.rmodels, mg.rutils, mg.rconst, mg.syn, mg.T
np_, grads_record, tensor_record = mg.np_, mg.grads_record, mg.tensor_record
import inference_ORCED as rinf  # noqa: E402


def main():
    B, N, C, K = 6, 32, 4, 4
    mg.set_nfeatures(C)
    rconst.NMAX, rconst.DEC_MLP_SIZE = N, T * N * C
    import importlib
    importlib.reload(rmodels)                       # ORCEDEncoder / ORCEDDecoder read NMAX / DEC_MLP_SIZE at class scope
    cfg = dict(LR=1e-4, B1=0.9, REC_W=1.0, CE_W=1.0, KL_W=1.0)
    enc = rmodels.ORCEDEncoder(n_out_labels=K).float()
    dec = rmodels.ORCEDDecoder(nmax_points=N).float()
    gml = rmodels.GaussianMeanLearner(n_in_labels=K).float()
    seeds = [80, 81, 82]
    for m, s in zip((enc, dec, gml), seeds):
        syn.deterministic_fill_(m, seed=s)
    opt = torch.optim.Adam(itertools.chain(enc.parameters(), dec.parameters(), gml.parameters()), lr=cfg["LR"],
                           betas=(cfg["B1"], cfg["B1"]))
    chamfer = rutils.SeqChamferLoss()
    ce = torch.nn.CrossEntropyLoss()
    eps = torch.from_numpy(np.random.default_rng(17).standard_normal((B, 32)).astype(np.float32))
    pcs = syn.synthetic_pcs(B, T, N, C, seed=140).permute(0, 3, 1, 2).contiguous()
    gt = syn.synthetic_labels(B, K, seed=240)
    kl_mult = 0.5
    rec = {"meta": np.array(json.dumps(dict(B=B, N=N, C=C, K=K, fill_seeds=seeds, pcs_seed=140, gt_seed=240, eps_seed=17,
                                            kl_multiplier=kl_mult, **cfg))), "eps": np_(eps)}
    enc.train(); dec.train(); gml.train()
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t, *a, **k: eps.clone()
    try:
        logits, sup_fvs, mu, logvar = enc(pcs)
    finally:
        torch.randn_like = real_randn_like
    rec_pcs = dec(sup_fvs)
    mu_gts = gml(torch.nn.functional.one_hot(gt, num_classes=K).float())
    l_rec = cfg["REC_W"] * chamfer(rec_pcs, pcs)
    l_sup = cfg["CE_W"] * ce(logits, gt)
    l_kl = cfg["KL_W"] * rutils.CG_kl_divergence(mu, logvar, mu_gts) * kl_mult
    tot = l_rec + l_sup + l_kl
    tot.backward()
    for k, v in (("logits", logits), ("sup_fvs", sup_fvs), ("mu", mu), ("logvar", logvar), ("mu_gts", mu_gts)):
        rec[f"out.{k}"] = np_(v)
    rec["losses"] = np.array([l_rec.item(), l_sup.item(), l_kl.item(), tot.item()], dtype=np.float64)
    grads_record("grad.E.", enc.named_parameters(), rec)
    grads_record("grad.G.", dec.named_parameters(), rec)
    grads_record("grad.ML.", gml.named_parameters(), rec)
    opt.step()
    for nm, m in (("E", enc), ("G", dec), ("ML", gml)):
        tensor_record(f"param.{nm}.", m.state_dict(), rec)
    # ---- KL on plain inputs
    rng = np.random.default_rng(3)
    a, b, c = (torch.from_numpy(rng.standard_normal((5, 32)).astype(np.float32)) for _ in range(3))
    rec["kl.mu"], rec["kl.logvar"], rec["kl.mu_k"] = np_(a), np_(b * 0.3), np_(c)
    rec["kl.value"] = np.float64(rutils.CG_kl_divergence(a, b * 0.3, c).item())
    # ---- the box test and the ensemble rule (inference_ORCED.py:18-132)
    D, ntr, nte = 32, 96, 24
    f_tr = rng.standard_normal((ntr, D)) * 0.8
    gl = rng.integers(0, 3, ntr)
    f_tr += np.eye(3)[gl] @ (rng.standard_normal((3, D)) * 2.0)
    pl = gl.copy(); pl[::7] = (pl[::7] + 1) % 3
    re_tr = np.abs(rng.standard_normal(ntr)) + 1.0
    z_te = rng.standard_normal((nte, D)) * 1.2 + np.eye(3)[rng.integers(0, 3, nte)] @ (rng.standard_normal((3, D)) * 2.0)
    z_te[:8] = f_tr[:8] * 0.05 + f_tr[gl == 0].mean(0) * 0.95     # some points close to a class mean
    re_te = np.abs(rng.standard_normal(nte)) * 1.5 + 1.0
    pred_te = torch.from_numpy(rng.integers(0, 3, nte))
    mean0, std0 = f_tr[gl == 0].mean(0), f_tr[gl == 0].std(0)
    rec["box.mean"], rec["box.cov_diag"], rec["box.z"] = mean0, std0, z_te
    rec["box.p"] = np.array([rinf.compute_prob(mean0, np.diag(std0), z) for z in z_te])
    out = rinf.ORCED_ensemble_ood_detection(re_tr, f_tr, 0.95, gl, pl, pred_te, z_te, re_te)
    for k, v in (("f_tr", f_tr), ("gl", gl), ("pl", pl), ("re_tr", re_tr), ("z_te", z_te), ("re_te", re_te)):
        rec[f"ood.{k}"] = np.asarray(v)
    rec["ood.pred_te"], rec["ood.out"] = pred_te.numpy(), out.numpy()
    np.savez_compressed(os.path.join(HERE, "orced.npz"), **rec)
    print("orced.npz", os.path.getsize(os.path.join(HERE, "orced.npz")) // 1024, "KiB;", "box p", rec["box.p"][:10], "ood", rec["ood.out"])


if __name__ == "__main__":
    main()
