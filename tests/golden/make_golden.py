#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE
(/root/reference) and running it on CPU with deterministic inputs/weights.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

What is stored are inputs' seeds and the reference's outputs (or checksums of
large outputs) -- never reference source, never weights.  Weights and inputs
are regenerated on either side from ``opensetgaitrecognition_pcaa_amd.synthetic``.
"""
import itertools
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PCAA_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from opensetgaitrecognition_pcaa_amd import synthetic as syn  # noqa: E402

# the reference's utils.py imports wandb/umap at module scope but never uses
# them on this path (utils.py:5,7)
for name in ("wandb", "umap"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, REF)
import constants as rconst  # noqa: E402
import models as rmodels  # noqa: E402
import utils as rutils  # noqa: E402

rconst.DEVICE = "cpu"
torch.set_num_threads(8)
T = rconst.NSTEPS


def np_(t):
    return t.detach().cpu().numpy().copy()


def grads_record(prefix, named_params, out):
    """full gradient for small tensors, checksum for large ones."""
    for k, p in named_params:
        g = p.grad
        if g is None:
            out[f"{prefix}{k}::none"] = np.zeros(0)
            continue
        if g.numel() <= 1 << 15:
            out[f"{prefix}{k}::full"] = np_(g)
        else:
            cs = syn.checksum(g)
            out[f"{prefix}{k}::sum"] = np.float64(cs["sum"])
            out[f"{prefix}{k}::l2"] = np.float64(cs["l2"])
            out[f"{prefix}{k}::samples"] = cs["samples"]


def tensor_record(prefix, sd, out):
    for k, v in sd.items():
        if v.numel() <= 1 << 15:
            out[f"{prefix}{k}::full"] = np_(v)
        else:
            cs = syn.checksum(v)
            out[f"{prefix}{k}::sum"] = np.float64(cs["sum"])
            out[f"{prefix}{k}::l2"] = np.float64(cs["l2"])
            out[f"{prefix}{k}::samples"] = cs["samples"]


def set_nfeatures(C):
    rconst.NFEATURES = C


# ---------------------------------------------------------------- encoder
def encoder_case(tag, B, N, C, K, head, out):
    set_nfeatures(C)
    enc = rmodels.CGEncoder(K, nmax_points=N, use_projection_head=head).float()
    syn.deterministic_fill_(enc, seed=0)
    x = syn.synthetic_pcs(B, T, N, C, seed=1234).permute(0, 3, 1, 2).contiguous()
    meta = dict(B=B, N=N, C=C, K=K, head=int(head), fill_seed=0, pcs_seed=1234)
    rec = {"meta": np.array(json.dumps(meta))}
    # eval forward
    enc.eval()
    with torch.no_grad():
        oc, fv = enc(x)
        x2 = torch.squeeze(enc.glob_avg_pool1(enc.pc_block(x)), dim=-1)
    rec["eval_out_classes"] = np_(oc)
    rec["eval_sup_fv"] = np_(fv)
    rec["eval_x2_samples"] = syn.checksum(x2, 64)["samples"]
    rec["eval_x2_sum"] = np.float64(syn.checksum(x2)["sum"])
    # two train-mode steps with a fixed linear probe loss, grads at step 1
    enc.train()
    rng = np.random.default_rng(77)
    r1 = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32))
    r2 = torch.from_numpy(rng.standard_normal((B, 32)).astype(np.float32))
    xg = x.clone().requires_grad_(True)
    oc, fv = enc(xg)
    loss = (oc * r1).sum() + (fv * r2).sum()
    loss.backward()
    rec["train_out_classes"] = np_(oc)
    rec["train_sup_fv"] = np_(fv)
    rec["train_loss"] = np.float64(loss.item())
    rec["train_dx_sum"] = np.float64(xg.grad.double().sum().item())
    rec["train_dx_l2"] = np.float64(xg.grad.double().norm().item())
    rec["train_dx_samples"] = syn.checksum(xg.grad, 64)["samples"]
    grads_record("grad.", enc.named_parameters(), rec)
    tensor_record("bn1.", {k: v for k, v in enc.state_dict().items() if "running" in k or "num_batches" in k}, rec)
    with torch.no_grad():
        oc2, fv2 = enc(x)
    rec["train2_sup_fv"] = np_(fv2)
    tensor_record("bn2.", {k: v for k, v in enc.state_dict().items() if "running" in k or "num_batches" in k}, rec)
    out[tag] = rec


# ---------------------------------------------------------------- decoder
def decoder_case(tag, B, N, C, in_dim, out):
    set_nfeatures(C)
    dec = rmodels.CGDecoder(input_dim=in_dim, nmax_points=N).float()
    syn.deterministic_fill_(dec, seed=1)
    rng = np.random.default_rng(5)
    z = torch.from_numpy(rng.standard_normal((B, in_dim)).astype(np.float32)).requires_grad_(True)
    r = torch.from_numpy(rng.standard_normal((B, C, T, N)).astype(np.float32))
    y = dec(z)
    (y * r).sum().backward()
    rec = {"meta": np.array(json.dumps(dict(B=B, N=N, C=C, in_dim=in_dim, fill_seed=1, z_seed=5)))}
    rec["out_sum"] = np.float64(y.double().sum().item())
    rec["out_l2"] = np.float64(y.double().norm().item())
    rec["out_samples"] = syn.checksum(y, 64)["samples"]
    rec["dz"] = np_(z.grad)
    grads_record("grad.", dec.named_parameters(), rec)
    rec["state_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in dec.state_dict().items()}))
    out[tag] = rec


# ---------------------------------------------------------------- chamfer
def chamfer_case(tag, B, N, C, out):
    set_nfeatures(C)
    loss_fn = rutils.SeqChamferLoss()
    gts = syn.synthetic_pcs(B, T, N, C, seed=21).permute(0, 3, 1, 2).contiguous()
    preds = (syn.synthetic_pcs(B, T, N, C, seed=22) * 0.7 + 0.1).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    l = loss_fn(preds, gts)
    l.backward()
    with torch.no_grad():
        lv = loss_fn(preds, gts, avg_out=False)
    out[tag] = {
        "meta": np.array(json.dumps(dict(B=B, N=N, C=C, gts_seed=21, preds_seed=22))),
        "loss": np.float64(l.item()), "loss_per_seq": np_(lv), "dpreds": np_(preds.grad),
    }


# ---------------------------------------------------------------- discriminator + D-step
def disc_case(tag, B, K, out):
    set_nfeatures(4)
    disc = rmodels.CGDiscriminator(K).float()
    syn.deterministic_fill_(disc, seed=2)
    rng = np.random.default_rng(9)
    fv = torch.from_numpy(rng.standard_normal((B, 32)).astype(np.float32))
    gt = syn.synthetic_labels(B, K, seed=1235)
    oh = torch.nn.functional.one_hot(gt, K).float()
    z = (syn.synthetic_z0(B, 32, seed=1236) + 3.0 * oh @ torch.from_numpy(rng.standard_normal((K, 32)).astype(np.float32)))
    z = z.clone().requires_grad_(True)
    alphas = syn.synthetic_alphas(B, seed=1237)
    real = disc(z, oh)
    fake = disc(fv, oh)
    interp = z + alphas.repeat(1, 32) * (fv - z)
    di = disc(interp, oh)
    g = torch.autograd.grad(di, interp, torch.ones_like(di), create_graph=True, retain_graph=True, only_inputs=True)[0]
    slopes = torch.sqrt(torch.sum(g ** 2, dim=1) + 1e-12)
    gp = ((slopes - 1) ** 2).mean()
    d_loss = fake.mean() - real.mean() + 15 * gp
    d_loss.backward()
    rec = {"meta": np.array(json.dumps(dict(B=B, K=K, fill_seed=2, rng_seed=9))),
           "fv": np_(fv), "z": np_(z), "alphas": np_(alphas), "gt": np_(gt),
           "real": np_(real), "fake": np_(fake), "interp_grad": np_(g),
           "gp": np.float64(gp.item()), "d_loss": np.float64(d_loss.item())}
    grads_record("grad.", disc.named_parameters(), rec)
    out[tag] = rec


# ---------------------------------------------------------------- full V4 step
def v4_case(tag, B, N, C, K, steps, out):
    """Loop body of the reference's train_variant4 (PCAA_ablation.py:882-1021,
    proj_head_on_discriminator=False) driven with the reference's own modules,
    loss class and torch.optim.Adam; the two host RNG draws are injected."""
    set_nfeatures(C)
    rconst.BATCH_SIZE = B
    cfg = dict(LR=1e-4, B1=0.9, B2=0.99, GP_WEIGHT=15, ADV_WEIGHT=1, SUP_LATENT_DIM=32)
    enc = rmodels.CGEncoder(K, use_projection_head=True, nmax_points=N).float()
    dec = rmodels.CGDecoder(input_dim=64, nmax_points=N).float()
    disc = rmodels.CGDiscriminator(K).float()
    gph = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ELU()).float()
    dph = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.ELU()).float()
    for i, m in enumerate((enc, dec, disc, gph, dph)):
        syn.deterministic_fill_(m, seed=10 + i)
    chamfer = rutils.SeqChamferLoss()
    ce = torch.nn.CrossEntropyLoss()
    opt_g = torch.optim.Adam(itertools.chain(enc.parameters(), gph.parameters(), dec.parameters()),
                             lr=cfg["LR"], betas=(cfg["B1"], cfg["B2"]))
    opt_d = torch.optim.Adam(itertools.chain(dph.parameters(), disc.parameters()),
                             lr=cfg["LR"], betas=(cfg["B1"], cfg["B2"]))
    means = rutils.sample_distant_points(dimension=32, n=K, min_dist=10, sphere_radius=10).float()
    rec = {"meta": np.array(json.dumps(dict(B=B, N=N, C=C, K=K, steps=steps, fill_seeds=[10, 11, 12, 13, 14],
                                            pcs_seed0=100, gt_seed0=200, z0_seed0=300, alpha_seed0=400))),
           "means": np_(means)}
    enc.train(); dec.train(); disc.train()
    for s in range(steps):
        pcs = syn.synthetic_pcs(B, T, N, C, seed=100 + s).permute(0, 3, 1, 2).contiguous()
        gt = syn.synthetic_labels(B, K, seed=200 + s)
        z0 = syn.synthetic_z0(B, 32, seed=300 + s)
        alphas = syn.synthetic_alphas(B, seed=400 + s)
        out_labels, sup_fvs = enc(pcs)
        with torch.no_grad():
            preds = torch.argmax(torch.nn.Softmax(dim=1)(out_labels), dim=1)
        opt_d.zero_grad()
        oh = torch.nn.functional.one_hot(gt, num_classes=K).float()
        mus = torch.matmul(oh.unsqueeze(1), means.unsqueeze(0)).squeeze()
        z = (z0 + mus).clone()
        z.requires_grad = True
        fake_in = sup_fvs.detach()
        real = disc(z, oh)
        fake = disc(fake_in, oh)
        interp = z + alphas.repeat(1, 32) * (fake_in - z)
        di = disc(interp, oh)
        g = torch.autograd.grad(outputs=di, inputs=interp, grad_outputs=torch.ones_like(di),
                                create_graph=True, retain_graph=True, only_inputs=True)[0]
        slopes = torch.sqrt(torch.sum(g ** 2, dim=1) + 1e-12)
        gp = ((slopes - 1) ** 2).mean()
        d_loss = torch.mean(fake) - torch.mean(real) + cfg["GP_WEIGHT"] * gp
        d_loss.backward()
        if s == 0:
            grads_record("s0.dgrad.", disc.named_parameters(), rec)
        opt_d.step()
        opt_d.zero_grad()
        opt_g.zero_grad()
        rec_pcs = dec(gph(sup_fvs))
        rec_loss = chamfer(rec_pcs, pcs)
        synth = disc(sup_fvs, oh)
        loss_g = -torch.mean(synth) * cfg["ADV_WEIGHT"]
        sup_loss = ce(out_labels, gt)
        tot = rec_loss + loss_g + sup_loss
        tot.backward()
        if s == 0:
            grads_record("s0.ggrad.E.", enc.named_parameters(), rec)
            grads_record("s0.ggrad.GPH.", gph.named_parameters(), rec)
            grads_record("s0.ggrad.G.", dec.named_parameters(), rec)
        opt_g.step()
        rec[f"s{s}.losses"] = np.array([d_loss.item(), gp.item(), rec_loss.item(), loss_g.item(),
                                        sup_loss.item(), tot.item()], dtype=np.float64)
        rec[f"s{s}.preds"] = np_(preds)
        rec[f"s{s}.out_labels"] = np_(out_labels)
        rec[f"s{s}.sup_fvs"] = np_(sup_fvs)
        if s in (0, steps - 1):
            for nm, m in (("E", enc), ("G", dec), ("D", disc), ("GPH", gph), ("DPH", dph)):
                tensor_record(f"s{s}.param.{nm}.", m.state_dict(), rec)
    out[tag] = rec


# ---------------------------------------------------------------- misc
def misc_case(out):
    rec = {}
    for K in (2, 4, 6, 8):
        rec[f"means_K{K}"] = np_(rutils.sample_distant_points(dimension=32, n=K, min_dist=10, sphere_radius=10))
    set_nfeatures(4)
    gml = rmodels.GaussianMeanLearner(6).float()
    syn.deterministic_fill_(gml, seed=3)
    oh = torch.nn.functional.one_hot(syn.synthetic_labels(12, 6, seed=1), 6).float()
    gml.train()
    rec["gml_train_out"] = np_(gml(oh))
    gml.eval()
    with torch.no_grad():
        rec["gml_eval_out"] = np_(gml(oh))
    # state_dict manifest (Appendix B)
    man = {}
    enc = rmodels.CGEncoder(8, nmax_points=32, use_projection_head=True)
    dec = rmodels.CGDecoder(input_dim=64, nmax_points=32)
    disc = rmodels.CGDiscriminator(8)
    for nm, m in (("E", enc), ("G", dec), ("D", disc)):
        man[nm] = {k: [list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()}
    rec["manifest_N32_C4_K8"] = np.array(json.dumps(man))
    out["misc"] = rec


def inference_case(out):
    """joint_likelihood (scipy), ROC/Youden threshold (sklearn) and the k-window
    vote exactly as inference_PCAA.py:129-136, 225-231, 263-271 evaluate them."""
    from scipy.stats import multivariate_normal
    from sklearn.metrics import roc_curve
    rng = np.random.default_rng(123)
    K = 6
    means = np_(rutils.sample_distant_points(dimension=32, n=K, min_dist=10, sphere_radius=10).float())
    n_known, n_unseen = 240, 120
    lab = rng.integers(0, K, n_known)
    known = (means[lab] + rng.standard_normal((n_known, 32)) * 1.1).astype(np.float32)
    unseen = (rng.standard_normal((n_unseen, 32)) * 4.0 + means[rng.integers(0, K, n_unseen)] * 0.6).astype(np.float32)

    def jl(x):
        lk = 0
        for mean in means:
            lk += multivariate_normal(mean=mean, cov=np.eye(32)).pdf(x)
        return lk / means.shape[0]

    lk_known = np.array([jl(v[None]) for v in known]).reshape(-1)
    lk_unseen = np.array([jl(v[None]) for v in unseen]).reshape(-1)
    scores = np.concatenate([lk_unseen, lk_known])
    y = np.concatenate([np.zeros_like(lk_unseen), np.ones_like(lk_known)])
    fpr, tpr, thr = roc_curve(y, scores)
    best = thr[np.argmax(tpr - fpr)]
    rec = {"means": means, "known": known, "unseen": unseen, "lk_known": lk_known, "lk_unseen": lk_unseen,
           "threshold": np.float64(best)}
    preds = rng.integers(0, K, n_known)
    for k in (1, 2, 4, 6):
        votes = []
        for w in range(n_known // k):
            lks = lk_known[w * k:(w + 1) * k]
            pr = preds[w * k:(w + 1) * k]
            if np.sum(np.array(lks) > best) > k / 2:
                votes.append(np.argmax(np.bincount(pr)))
            else:
                votes.append(K)
        rec[f"votes_k{k}"] = np.array(votes, dtype=np.int64)
    rec["preds"] = preds.astype(np.int64)
    out["inference"] = rec


def main():
    out = {}
    encoder_case("enc_cfg1_B4_N128_C5_K8", 4, 128, 5, 8, True, out)
    encoder_case("enc_B2_N32_C4_K4", 2, 32, 4, 4, True, out)
    encoder_case("enc_B3_N150_C4_K6_nohead", 3, 150, 4, 6, False, out)
    decoder_case("dec_B2_N32_C4", 2, 32, 4, 64, out)
    decoder_case("dec_B3_N50_C5_in32", 3, 50, 5, 32, out)
    chamfer_case("chamfer_B2_N32_C4", 2, 32, 4, out)
    chamfer_case("chamfer_B2_N150_C5", 2, 150, 5, out)
    disc_case("disc_B6_K4", 6, 4, out)
    disc_case("disc_B16_K8", 16, 8, out)
    v4_case("v4_B6_N32_C4_K4", 6, 32, 4, 4, 3, out)
    misc_case(out)
    inference_case(out)
    for tag, rec in out.items():
        path = os.path.join(HERE, tag + ".npz")
        np.savez_compressed(path, **rec)
        print(f"{tag}: {os.path.getsize(path) / 1024:.1f} KiB, {len(rec)} arrays")
    with open(os.path.join(HERE, "PROVENANCE.json"), "w") as f:
        json.dump({"torch": torch.__version__, "numpy": np.__version__, "threads": torch.get_num_threads(),
                   "reference": "rmazzier/OpenSetGaitRecognition_PCAA @ 2025-03-21", "script": "tests/golden/make_golden.py"}, f, indent=1)


if __name__ == "__main__":
    main()
