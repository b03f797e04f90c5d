#!/usr/bin/env python3
"""Golden for ablation variant 1 (learned prior centroids), made by running the loop body of the
reference's train_variant1 (PCAA_ablation.py:145-283) with the REFERENCE's own modules, loss class and
torch.optim.Adam in the build container; the two host RNG draws are injected.

    python tests/golden/make_golden_v1.py  ->  tests/golden/v1_B6_N32_C4_K4.npz"""
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


def main():
    B, N, C, K, steps = 6, 32, 4, 4, 3
    mg.set_nfeatures(C)
    rconst.BATCH_SIZE = B
    cfg = dict(LR=1e-4, B1=0.9, B2=0.99, GP_WEIGHT=15, ADV_WEIGHT=1, SUP_LATENT_DIM=32)
    enc = rmodels.CGEncoder(K, use_projection_head=True, nmax_points=N).float()
    dec = rmodels.CGDecoder(input_dim=64, nmax_points=N).float()
    disc = rmodels.CGDiscriminator(K).float()
    gph = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ELU()).float()
    gml = rmodels.GaussianMeanLearner(K).float()
    seeds = [40, 41, 42, 43, 44]
    for m, sd in zip((enc, dec, disc, gph, gml), seeds):
        syn.deterministic_fill_(m, seed=sd)
    chamfer = rutils.SeqChamferLoss()
    ce = torch.nn.CrossEntropyLoss()
    opt_g = torch.optim.Adam(itertools.chain(enc.parameters(), gph.parameters(), dec.parameters()),
                             lr=cfg["LR"], betas=(cfg["B1"], cfg["B2"]))
    opt_d = torch.optim.Adam(itertools.chain(gml.parameters(), disc.parameters()),
                             lr=cfg["LR"], betas=(cfg["B1"], cfg["B2"]))
    rec = {"meta": np.array(json.dumps(dict(B=B, N=N, C=C, K=K, steps=steps, fill_seeds=seeds, pcs_seed0=110,
                                            gt_seed0=210, z0_seed0=310, alpha_seed0=410)))}
    enc.train(); dec.train(); disc.train(); gml.train(); gph.train()
    for s in range(steps):
        pcs = syn.synthetic_pcs(B, T, N, C, seed=110 + s).permute(0, 3, 1, 2).contiguous()
        gt = syn.synthetic_labels(B, K, seed=210 + s)
        z0 = syn.synthetic_z0(B, 32, seed=310 + s)
        alphas = syn.synthetic_alphas(B, seed=410 + s)
        out_labels, sup_fvs = enc(pcs)
        with torch.no_grad():
            preds = torch.argmax(torch.nn.Softmax(dim=1)(out_labels), dim=1)
        opt_d.zero_grad()
        oh = torch.nn.functional.one_hot(gt, num_classes=K).float()
        mus = gml(oh)
        z = Variable(z0 + mus)
        z.requires_grad = True
        real = disc(z, oh)
        fake = disc(sup_fvs.detach(), oh)
        a = alphas.repeat(1, 32)
        interp = z + a * (sup_fvs.detach() - z)
        di = disc(interp, oh)
        g = torch.autograd.grad(outputs=di, inputs=interp, grad_outputs=torch.ones_like(di),
                                create_graph=True, retain_graph=True, only_inputs=True)[0]
        slopes = torch.sqrt(torch.sum(g ** 2, dim=1) + 1e-12)
        gp = ((slopes - 1) ** 2).mean()
        d_loss = torch.mean(fake) - torch.mean(real) + cfg["GP_WEIGHT"] * gp
        d_loss.backward()
        if s == 0:
            mg.grads_record("s0.dgrad.D.", disc.named_parameters(), rec)
            mg.grads_record("s0.dgrad.ML.", gml.named_parameters(), rec)
        opt_d.step()
        opt_d.zero_grad()
        disc.zero_grad()
        opt_g.zero_grad()
        rec_pcs = dec(gph(sup_fvs))
        rec_loss = chamfer(rec_pcs, pcs)
        synth = disc(sup_fvs, oh)
        loss_g = -torch.mean(synth) * cfg["ADV_WEIGHT"]
        sup_loss = ce(out_labels, gt)
        tot = rec_loss + loss_g + sup_loss
        tot.backward()
        opt_g.step()
        rec[f"s{s}.losses"] = np.array([d_loss.item(), gp.item(), rec_loss.item(), loss_g.item(), sup_loss.item(),
                                        tot.item()], dtype=np.float64)
        rec[f"s{s}.preds"] = mg.np_(preds)
        rec[f"s{s}.sup_fvs"] = mg.np_(sup_fvs)
        rec[f"s{s}.mus"] = mg.np_(mus)
        if s in (0, steps - 1):
            for nm, m in (("D", disc), ("ML", gml)):
                mg.tensor_record(f"s{s}.param.{nm}.", m.state_dict(), rec)
    # the learned centroids the checkpoint stores (:367-375: mean_learner is still in train mode there)
    with torch.no_grad():
        rec["centroids_train_mode"] = mg.np_(gml(torch.nn.functional.one_hot(torch.arange(0, K), num_classes=K).float()))
    path = os.path.join(HERE, "v1_B6_N32_C4_K4.npz")
    np.savez_compressed(path, **rec)
    print(os.path.basename(path), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
