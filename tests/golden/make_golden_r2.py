#!/usr/bin/env python3
"""Round-2 goldens, made in the build container by driving the REFERENCE's own modules, loss classes and
torch.optim.Adam through the reference's loop bodies (the two host RNG draws injected, deterministic fills):

* ``v4_supfreq2_B6_N32_C4_K4.npz`` -- train_variant4's loop body (PCAA_ablation.py:882-1021) with
  SUPERVISION_FREQUENCY = 2 for four steps: on the unsupervised steps ``zero_grad`` leaves the gradients of
  MLP_head / MLP_sup2 at None, Adam skips them (no update, no moment decay, their step count does not move).
* ``v3_B6_N32_C4_K4.npz`` -- train_variant3's loop body (PCAA_ablation.py:514-655): encoder without projection
  head + critic, no decoder, optimizer_G betas (B1, B1) (:455).

    python tests/golden/make_golden_r2.py
"""
import itertools
import json
import os
import sys

import numpy as np
import torch
from torch.autograd import Variable

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference's models / utils / constants)

rmodels, rutils, rconst, syn, T = mg.rmodels, mg.rutils, mg.rconst, mg.syn, mg.T
np_, grads_record, tensor_record = mg.np_, mg.grads_record, mg.tensor_record


def _d_step(disc, opt_d, sup_fvs, gt, K, means, z0, alphas, cfg):
    """:897-980 (identical in variant 3, :557-612)."""
    opt_d.zero_grad()
    disc.zero_grad()
    oh = torch.nn.functional.one_hot(gt, num_classes=K).float()
    mus = torch.matmul(oh.unsqueeze(1), means.unsqueeze(0)).squeeze()
    z = Variable(z0 + mus)
    z.requires_grad = True
    real = disc(z, oh)
    fake = disc(sup_fvs.detach(), oh)
    a = alphas.repeat(1, cfg["SUP_LATENT_DIM"])
    interp = z + a * (sup_fvs.detach() - z)
    di = disc(interp, oh)
    g = torch.autograd.grad(outputs=di, inputs=interp, grad_outputs=torch.ones_like(di), create_graph=True,
                            retain_graph=True, only_inputs=True)[0]
    slopes = torch.sqrt(torch.sum(g ** 2, dim=1) + 1e-12)
    gp = ((slopes - 1) ** 2).mean()
    d_loss = torch.mean(fake) - torch.mean(real) + cfg["GP_WEIGHT"] * gp
    d_loss.backward()
    opt_d.step()
    opt_d.zero_grad()
    disc.zero_grad()
    return oh, d_loss, gp


def v4_supfreq_case(tag, B, N, C, K, steps, freq, out):
    mg.set_nfeatures(C)
    rconst.BATCH_SIZE = B
    cfg = dict(LR=1e-4, B1=0.9, B2=0.99, GP_WEIGHT=15, ADV_WEIGHT=1, SUP_LATENT_DIM=32, SUPERVISION_FREQUENCY=freq)
    enc = rmodels.CGEncoder(K, use_projection_head=True, nmax_points=N).float()
    dec = rmodels.CGDecoder(input_dim=64, nmax_points=N).float()
    disc = rmodels.CGDiscriminator(K).float()
    gph = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ELU()).float()
    dph = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.ELU()).float()
    seeds = [60, 61, 62, 63, 64]
    for m, sd in zip((enc, dec, disc, gph, dph), seeds):
        syn.deterministic_fill_(m, seed=sd)
    chamfer = rutils.SeqChamferLoss()
    ce = torch.nn.CrossEntropyLoss()
    opt_g = torch.optim.Adam(itertools.chain(enc.parameters(), gph.parameters(), dec.parameters()),
                             lr=cfg["LR"], betas=(cfg["B1"], cfg["B2"]))
    opt_d = torch.optim.Adam(itertools.chain(dph.parameters(), disc.parameters()),
                             lr=cfg["LR"], betas=(cfg["B1"], cfg["B2"]))
    means = rutils.sample_distant_points(dimension=32, n=K, min_dist=10, sphere_radius=10).float()
    rec = {"meta": np.array(json.dumps(dict(B=B, N=N, C=C, K=K, steps=steps, freq=freq, fill_seeds=seeds,
                                            pcs_seed0=120, gt_seed0=220, z0_seed0=320, alpha_seed0=420))),
           "means": np_(means)}
    enc.train(); dec.train(); disc.train()
    for i in range(steps):
        pcs = syn.synthetic_pcs(B, T, N, C, seed=120 + i).permute(0, 3, 1, 2).contiguous()
        gt = syn.synthetic_labels(B, K, seed=220 + i)
        z0 = syn.synthetic_z0(B, 32, seed=320 + i)
        alphas = syn.synthetic_alphas(B, seed=420 + i)
        out_labels, sup_fvs = enc(pcs)
        with torch.no_grad():
            preds = torch.argmax(torch.nn.Softmax(dim=1)(out_labels), dim=1)
        oh, d_loss, gp = _d_step(disc, opt_d, sup_fvs, gt, K, means, z0, alphas, cfg)
        opt_g.zero_grad(); enc.zero_grad(); dec.zero_grad(); gph.zero_grad()
        rec_loss = chamfer(dec(gph(sup_fvs)), pcs)
        loss_g = -torch.mean(disc(sup_fvs, oh)) * cfg["ADV_WEIGHT"]
        supervised = i % cfg["SUPERVISION_FREQUENCY"] == 0
        if supervised:
            sup_loss = ce(out_labels, gt)
            tot = rec_loss + loss_g + sup_loss
        else:
            tot = rec_loss + loss_g
        tot.backward()
        if i == 1:
            grads_record("s1.ggrad.E.", enc.named_parameters(), rec)       # MLP_head / MLP_sup2: ::none
        opt_g.step()
        rec[f"s{i}.losses"] = np.array([d_loss.item(), gp.item(), rec_loss.item(), loss_g.item(),
                                        sup_loss.item(), tot.item()], dtype=np.float64)   # sup_loss: last supervised
        rec[f"s{i}.supervised"] = np.array(int(supervised))
        rec[f"s{i}.preds"] = np_(preds)
        rec[f"s{i}.sup_fvs"] = np_(sup_fvs)
        for nm in ("MLP_sup1.0.weight", "MLP_head.0.weight", "MLP_head.0.bias", "MLP_sup2.0.weight",
                   "MLP_sup2.0.bias"):
            rec[f"s{i}.param.E.{nm}::full"] = np_(enc.state_dict()[nm])
    st = opt_g.state_dict()["state"]
    names = [n for n, _ in itertools.chain(enc.named_parameters(), gph.named_parameters(), dec.named_parameters())]
    rec["adam_steps"] = np.array(json.dumps({names[k]: int(v["step"]) for k, v in st.items()}))
    for nm, m in (("E", enc), ("G", dec), ("D", disc), ("GPH", gph)):
        tensor_record(f"s{steps - 1}.param.{nm}.", m.state_dict(), rec)
    out[tag] = rec


def v3_case(tag, B, N, C, K, steps, out):
    mg.set_nfeatures(C)
    rconst.BATCH_SIZE = B
    cfg = dict(LR=1e-4, B1=0.9, B2=0.99, GP_WEIGHT=15, ADV_WEIGHT=1, SUP_LATENT_DIM=32, SUPERVISION_FREQUENCY=1)
    # the reference builds this encoder with the default nmax_points (:404-409); N is passed here so that the
    # case stays small -- the arithmetic is the same
    enc = rmodels.CGEncoder(n_out_labels=K, use_projection_head=False, nmax_points=N).float()
    disc = rmodels.CGDiscriminator(K).float()
    seeds = [70, 72]
    syn.deterministic_fill_(enc, seed=seeds[0])
    syn.deterministic_fill_(disc, seed=seeds[1])
    ce = torch.nn.CrossEntropyLoss()
    opt_g = torch.optim.Adam(itertools.chain(enc.parameters()), lr=cfg["LR"], betas=(cfg["B1"], cfg["B1"]))   # :452-456
    opt_d = torch.optim.Adam(itertools.chain(disc.parameters()), lr=cfg["LR"], betas=(cfg["B1"], cfg["B2"]))
    means = rutils.sample_distant_points(dimension=32, n=K, min_dist=10, sphere_radius=10).float()
    rec = {"meta": np.array(json.dumps(dict(B=B, N=N, C=C, K=K, steps=steps, fill_seeds=seeds, pcs_seed0=130,
                                            gt_seed0=230, z0_seed0=330, alpha_seed0=430))),
           "means": np_(means)}
    enc.train(); disc.train()
    for i in range(steps):
        pcs = syn.synthetic_pcs(B, T, N, C, seed=130 + i).permute(0, 3, 1, 2).contiguous()
        gt = syn.synthetic_labels(B, K, seed=230 + i)
        z0 = syn.synthetic_z0(B, 32, seed=330 + i)
        alphas = syn.synthetic_alphas(B, seed=430 + i)
        out_labels, sup_fvs = enc(pcs)
        with torch.no_grad():
            preds = torch.argmax(torch.nn.Softmax(dim=1)(out_labels), dim=1)
        oh, d_loss, gp = _d_step(disc, opt_d, sup_fvs, gt, K, means, z0, alphas, cfg)
        opt_g.zero_grad(); enc.zero_grad()
        loss_g = -torch.mean(disc(sup_fvs, oh)) * cfg["ADV_WEIGHT"]
        sup_loss = ce(out_labels, gt)
        tot = loss_g + sup_loss
        tot.backward()
        if i == 0:
            grads_record("s0.ggrad.E.", enc.named_parameters(), rec)
        opt_g.step()
        rec[f"s{i}.losses"] = np.array([d_loss.item(), gp.item(), loss_g.item(), sup_loss.item(), tot.item()],
                                       dtype=np.float64)
        rec[f"s{i}.preds"] = np_(preds)
        rec[f"s{i}.sup_fvs"] = np_(sup_fvs)
        if i in (0, steps - 1):
            for nm, m in (("E", enc), ("D", disc)):
                tensor_record(f"s{i}.param.{nm}.", m.state_dict(), rec)
    out[tag] = rec


def main():
    out = {}
    v4_supfreq_case("v4_supfreq2_B6_N32_C4_K4", 6, 32, 4, 4, 4, 2, out)
    v3_case("v3_B6_N32_C4_K4", 6, 32, 4, 4, 3, out)
    for tag, rec in out.items():
        path = os.path.join(HERE, tag + ".npz")
        np.savez_compressed(path, **rec)
        print(f"{tag}: {os.path.getsize(path) / 1024:.1f} KiB, {len(rec)} arrays")


if __name__ == "__main__":
    main()
