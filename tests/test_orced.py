"""OR-CED baseline against the reference-generated golden (tests/golden/make_golden_orced.py): one train
step of rec + sup + kl on the reference's ORCEDEncoder / ORCEDDecoder / GaussianMeanLearner (reparametrisation draw
injected), CG_kl_divergence, the float64 box test and the ensemble rule.  The triplet term restates
pytorch_metric_learning 1.6.0, which is not installed: PARITY UNPINNED, checked for its defining properties only."""
import itertools
import json

import numpy as np
import pytest
import torch

from helpers import check_against_record, is_pre_bn_bias, load_golden
from opensetgaitrecognition_pcaa_amd import constants, models, orced, synthetic as syn
from opensetgaitrecognition_pcaa_amd.utils import CG_kl_divergence

G, META = load_golden("orced")
T = constants.NSTEPS


def test_kl_divergence_oracle_vs_reference():
    from oracle import pcaa_oracle as O
    got = O.cg_kl_divergence(torch.from_numpy(G["kl.mu"]), torch.from_numpy(G["kl.logvar"]), torch.from_numpy(G["kl.mu_k"]))
    assert abs(got.item() - float(G["kl.value"])) <= 1e-6 * abs(float(G["kl.value"]))
    with pytest.raises(RuntimeError):          # the product has no CPU path (round 4: the torch branch is gone)
        CG_kl_divergence(torch.from_numpy(G["kl.mu"]), torch.from_numpy(G["kl.logvar"]), torch.from_numpy(G["kl.mu_k"]))


@pytest.mark.gpu
def test_kl_divergence_hip_vs_reference():
    dev = "cuda:0"
    got = CG_kl_divergence(torch.from_numpy(G["kl.mu"]).to(dev), torch.from_numpy(G["kl.logvar"]).to(dev),
                           torch.from_numpy(G["kl.mu_k"]).to(dev))
    assert abs(got.item() - float(G["kl.value"])) <= 1e-6 * abs(float(G["kl.value"]))


def test_box_probability_vs_scipy_and_ensemble_rule_vs_reference():
    p = orced.compute_prob(G["box.mean"], np.diag(G["box.cov_diag"]), G["box.z"])
    ref = G["box.p"]
    # scipy integrates the orthant probabilities numerically (abseps = releps = 1e-5); the closed form is exact
    assert np.all(np.abs(p - ref) <= 2e-5 + 2e-3 * np.abs(ref)), np.abs(p - ref).max()
    one = orced.compute_prob(G["box.mean"], np.diag(G["box.cov_diag"]), G["box.z"][3])
    assert np.isscalar(one) or np.ndim(one) == 0
    out = orced.ORCED_ensemble_ood_detection(G["ood.re_tr"], G["ood.f_tr"], 0.95, G["ood.gl"], G["ood.pl"],
                                             torch.from_numpy(G["ood.pred_te"]), G["ood.z_te"], G["ood.re_te"])
    assert np.array_equal(out.numpy(), G["ood.out"])
    with pytest.raises(NotImplementedError):
        orced.compute_prob(np.zeros(2), np.array([[1.0, 0.5], [0.5, 1.0]]), np.zeros(2))


def test_triplet_restatement_properties():
    """Parity unpinned (pytorch_metric_learning is absent): the defining properties of the two published algorithms."""
    e = torch.tensor([[1.0, 0.0], [0.9, 0.1], [0.0, 1.0], [0.1, 0.9], [0.7, 0.7]])
    lab = torch.tensor([0, 0, 1, 1, 0])
    a1, p, a2, n = orced.multi_similarity_miner(e, lab, epsilon=0.1)
    assert (lab[a1] == lab[p]).all() and (a1 != p).all() and (lab[a2] != lab[n]).all()
    en = torch.nn.functional.normalize(e, dim=1)
    sim = en @ en.t()
    for a, pp in zip(a1.tolist(), p.tolist()):         # a mined positive is no more similar than the hardest negative + eps
        assert sim[a, pp] - 0.1 < sim[a][lab != lab[a]].max()
    for a, nn in zip(a2.tolist(), n.tolist()):         # a mined negative is no less similar than the hardest positive - eps
        pos = (lab == lab[a]) & (torch.arange(5) != a)
        assert sim[a, nn] + 0.1 > sim[a][pos].min()
    loss = orced.triplet_margin_loss(en, lab, (a1, p, a2, n), margin=0.5)
    assert loss.item() > 0
    # well separated classes: nothing to mine, zero loss, still differentiable
    e2 = torch.tensor([[1.0, 0.0], [1.0, 0.01], [-1.0, 0.0], [-1.0, 0.01]], requires_grad=True)
    l2 = orced.triplet_margin_loss(e2, torch.tensor([0, 0, 1, 1]),
                                   orced.multi_similarity_miner(e2, torch.tensor([0, 0, 1, 1])), margin=0.5)
    assert l2.item() == 0.0
    l2.backward()


@pytest.mark.gpu
def test_orced_train_step_vs_reference_golden(monkeypatch):
    from opensetgaitrecognition_pcaa_amd import functional as F_hip
    from opensetgaitrecognition_pcaa_amd.train import FlatBuffer
    F_hip.set_precision("fp32")
    m = META
    B, N, C, K = m["B"], m["N"], m["C"], m["K"]
    constants.NFEATURES = C
    dev = "cuda"
    enc = models.ORCEDEncoder(K, nmax_points=N).float()
    dec = models.ORCEDDecoder(nmax_points=N).float()
    gml = models.GaussianMeanLearner(K).float()
    for mod, seed in zip((enc, dec, gml), m["fill_seeds"]):
        syn.deterministic_fill_(mod, seed)
        mod.to(dev).train()
    named = [("E." + n, p) for n, p in enc.named_parameters()]
    named += [("G." + n, p) for n, p in dec.named_parameters() if n.startswith("dense")]
    named += [("ML." + n, p) for n, p in gml.named_parameters()]
    flat = FlatBuffer(named, dev)
    for name, p in named:
        p.grad = flat.grad_views[name]
    eps = torch.from_numpy(G["eps"]).to(dev)
    monkeypatch.setattr(torch, "randn_like", lambda t, *a, **k: eps.clone())
    pcs = syn.synthetic_pcs(B, T, N, C, seed=m["pcs_seed"]).to(dev).permute(0, 3, 1, 2)
    gt = syn.synthetic_labels(B, K, seed=m["gt_seed"]).to(dev)
    cfg = dict(TRAIN_CLASSES=list(range(K)), REC_W=m["REC_W"], CE_W=m["CE_W"], KL_W=m["KL_W"], TRIPLET_W=0.0,
               TRIPLET_MARGIN=0.5)
    out = orced.orced_losses(enc, dec, gml, pcs, gt, cfg, m["kl_multiplier"])
    monkeypatch.undo()
    got = np.array([out["rec"].item(), out["sup"].item(), out["kl"].item(), out["tot"].item()])
    assert np.allclose(got, G["losses"], rtol=1e-4, atol=1e-6), (got, G["losses"])
    out["tot"].backward()
    wscale = max(float(np.abs(G[k]).max()) for k in G.files if k.startswith("grad.E.") and k.endswith("weight::full"))
    for name, p in named:
        nm, pname = name.split(".", 1)
        if is_pre_bn_bias(pname) or (nm == "ML" and pname in ("model.0.bias", "model.3.bias", "model.6.bias")):
            assert float(p.grad.abs().max()) <= 1e-4 * wscale + 1e-4, name
            continue
        check_against_record(G, f"grad.{nm}.", pname, p.grad, 5e-4, scale_floor=1e-3)
    flat.adam(m["LR"], m["B1"], m["B1"])
    torch.cuda.synchronize()
    for nm, mod in (("E", enc), ("G", dec), ("ML", gml)):
        for name, v in mod.state_dict().items():
            key = f"param.{nm}.{name}::full"
            if key not in G.files or not v.dtype.is_floating_point or is_pre_bn_bias(name) or \
                    (nm == "ML" and name in ("model.0.bias", "model.3.bias", "model.6.bias")):
                continue
            err = np.abs(v.cpu().numpy().astype(np.float64) - G[key])
            scale = max(float(np.abs(G[key]).max()), 5.0 if name.endswith("running_mean") else 0.0)
            assert err.max() <= 5e-5 * scale + 2.0e-4 * 1.001, (nm, name, err.max())


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_orced_loop_and_inference_run(tmp_path, monkeypatch):
    """train_ORCED -> checkpoints -> ORCED_inference on a small synthetic split (with the triplet term on)."""
    import pickle
    from opensetgaitrecognition_pcaa_amd import datasets
    monkeypatch.chdir(tmp_path)
    for subj in range(10):
        for si, scen in enumerate(("free_walk", "hands_in_pockets", "smartphone")):
            d = tmp_path / "raw" / f"target{subj}" / scen
            d.mkdir(parents=True)
            for t in range(10):
                with open(d / f"pc_tr{t}{si}.obj", "wb") as f:
                    pickle.dump(syn.synthetic_raw_track(7000 + subj * 100 + si * 10 + t, 48, max_points=24), f)
    monkeypatch.setattr(constants, "DATA_PATH", str(tmp_path / "raw"))
    monkeypatch.setattr(constants, "GEN_DATA_PATH", str(tmp_path / "gen"))
    monkeypatch.setattr(constants, "NFEATURES", 4)
    classes = [0, 1, 2, 3]
    np.random.seed(3)
    datasets.generate_splits(train_classes=classes, seed=0, nmax_points=16, verbose=False)
    cfg = dict(constants.CONFIG)
    cfg.update(MODEL_NAME="ORCED_t", TRAIN_CLASSES=classes, NMAX=16, BATCH_SIZE=16, EPOCHS=2, CHECKPOINT_FREQUENCY=1,
               SUBSAMPLE_FACTOR=1.0, NOTES="", TRIPLET_W=1, CE_W=1, REC_W=1, KL_W=1, TRIPLET_MARGIN=0.5)
    torch.manual_seed(0)
    mods, hist = orced.train_ORCED(cfg)
    assert len(hist) == 2 and all(np.isfinite(v) for r in hist for v in r.values())
    assert hist[0]["KL Loss"] == 0.0 and hist[1]["KL Loss"] > 0.0          # the KL weight ramps as epoch / EPOCHS
    assert hist[1]["Reconstruction Loss Train"] < hist[0]["Reconstruction Loss Train"]
    import os
    for sfx in ("_E", "_G", "_ML"):
        assert os.path.exists(f"models/ORCED_t/ORCED_t{sfx}.pt")
    res = orced.ORCED_inference(["ORCED_t"], generate_dataset=False)
    r = res["ORCED_t"]
    assert 0.0 <= r["accuracy"] <= 1.0 and 0.0 <= r["f1_macro"] <= 1.0
    preds = np.load("figures/ORCED_t/ensemble_ood_final_preds_fixed.npy")
    labels = np.load("figures/ORCED_t/ensemble_ood_final_labels_fixed.npy")
    assert preds.shape == labels.shape and len(classes) in labels and preds.max() <= len(classes)


@pytest.mark.gpu
def test_orced_heads_and_kl_kernels_vs_torch_autograd():
    """pcaa_orced_heads_fwd / _bwd and pcaa_orced_kl against the same expressions in torch (fp64 autograd): the three
    Linear heads, the reparametrisation, and CG_kl_divergence with all upstream gradients present."""
    from opensetgaitrecognition_pcaa_amd import functional as F_hip
    dev, B, K, D, L = "cuda", 11, 6, 512, 32
    g = torch.Generator(device=dev).manual_seed(4)
    r = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)
    x4, eps = r(B, D), r(B, L)
    Wmu, bmu, Wlv, blv, Wc, bc = r(L, D, sc=0.05), r(L, sc=0.1), r(L, D, sc=0.02), r(L, sc=0.1), r(K, L, sc=0.2), r(K, sc=0.1)
    mk = r(B, L)
    leaves = [t.clone().requires_grad_(True) for t in (x4, Wmu, bmu, Wlv, blv, Wc, bc, mk)]
    logits, sup, mu, lv = F_hip._OrcedHeadsFn.apply(leaves[0], eps, *leaves[1:7])
    kl = F_hip.cg_kl_divergence(mu, lv, leaves[7])
    wl, ws = r(B, K), r(B, L)
    tot = (logits * wl).sum() + (sup * ws).sum() + 0.7 * kl + F_hip.cross_entropy_loss(logits, torch.arange(B, device=dev) % K)
    tot.backward()
    # fp64 reference
    ref = [t.detach().double().clone().requires_grad_(True) for t in (x4, Wmu, bmu, Wlv, blv, Wc, bc, mk)]
    rmu = ref[0] @ ref[1].t() + ref[2]
    rlv = ref[0] @ ref[3].t() + ref[4]
    rsup = rmu + eps.double() * torch.exp(0.5 * rlv)
    rlogits = rsup @ ref[5].t() + ref[6]
    rkl = torch.mean(-0.5 * torch.sum(1 + rlv - (rmu - ref[7]) ** 2 - torch.exp(rlv), dim=1))
    rtot = (rlogits * wl.double()).sum() + (rsup * ws.double()).sum() + 0.7 * rkl + \
        torch.nn.functional.cross_entropy(rlogits, torch.arange(B, device=dev) % K)
    rtot.backward()
    for a, b, nm in ((logits, rlogits, "logits"), (sup, rsup, "sup_fv"), (mu, rmu, "mu"), (lv, rlv, "logvar"), (kl, rkl, "kl")):
        assert torch.allclose(a.double(), b, rtol=1e-5, atol=1e-5), nm
    for a, b, nm in zip(leaves, ref, ("x4", "Wmu", "bmu", "Wlv", "blv", "Wc", "bc", "mu_k")):
        err = (a.grad.double() - b.grad).abs().max().item()
        assert err <= 1e-4 * b.grad.abs().max().item() + 1e-6, (nm, err)
