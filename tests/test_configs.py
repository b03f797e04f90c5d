"""BASELINE.json configs 3-5 as parity / property cases (GPU): the point-subsampling sweep
(train_pointsubsampling.py path: N in {32,64,128,256}), the reference's own N=150 / K sweep,
and the B=1024 open-set inference batch."""
import numpy as np
import pytest
import torch

from helpers import T, load_golden, make_encoder, sd_clone
from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip, inference, synthetic as syn
from opensetgaitrecognition_pcaa_amd.train import PCAATrainer
from oracle import pcaa_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _trainer(B, N, C, K, precision, seed0=40):
    constants.NFEATURES = C
    cfg = dict(constants.CONFIG)
    cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B)
    tr = PCAATrainer(cfg, precision=precision)
    mods = (tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head, tr.discriminator_projection_head)
    for i, m in enumerate(mods):
        syn.deterministic_fill_(m, seed0 + i)
    return tr, mods


@pytest.mark.parametrize("N,C,K", [(32, 4, 2), (64, 4, 4), (150, 4, 6), (256, 4, 8), (128, 5, 8)])
def test_point_sweep_train_step_vs_oracle_fp32(N, C, K):
    """one fp32 train step from identical state, HIP vs CPU oracle, for every N of the sweep
    (and the reference's default N=150, and C=5)."""
    B = 4
    tr, mods = _trainer(B, N, C, K, "fp32")
    means = O.sample_distant_points(32, K, 10, 10).float()
    tr.set_prior_means(means)
    st = O.V4State(*({k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for m in mods),
                   means, C, T, N, K)
    tr.finalize()
    tr.train()
    pcs = syn.synthetic_pcs(B, T, N, C, seed=900 + N)
    gt = syn.synthetic_labels(B, K, seed=901)
    z0 = syn.synthetic_z0(B, 32, seed=902)
    al = syn.synthetic_alphas(B, seed=903)
    cfg = dict(LR=1e-4, B1=0.9, B2=0.99, GP_WEIGHT=15, ADV_WEIGHT=1)
    ref = O.v4_train_step(st, pcs.permute(0, 3, 1, 2), gt, z0, al, cfg)
    out = tr.step(pcs.to(DEV).permute(0, 3, 1, 2), gt.to(DEV), z0.to(DEV), al.to(DEV))
    for k in ("d_loss", "gp", "rec_loss", "loss_g", "sup_loss", "tot_loss"):
        assert abs(out[k].item() - ref[k].item()) <= 1e-4 * abs(ref[k].item()) + 1e-5, (k, out[k].item(), ref[k].item())
    assert torch.equal(out["preds"].cpu(), ref["preds"])
    err = (out["sup_fvs"].cpu() - ref["sup_fvs"]).abs().max().item()
    assert err <= 1e-4 * ref["sup_fvs"].abs().max().item()


@pytest.mark.parametrize("N,B", [(32, 64), (64, 16), (128, 64), (150, 16), (256, 8)])
def test_point_sweep_bf16_mode_runs_and_tracks_fp32(N, B):
    """bf16 throughput mode (B=64, N=128 is the bench shape): finite, and within the stated
    bf16 tolerance of the fp32 mode from the same state."""
    C, K = 4, 8
    pcs = syn.synthetic_pcs(B, T, N, C, seed=77).to(DEV).permute(0, 3, 1, 2)
    gt = syn.synthetic_labels(B, K, seed=78).to(DEV)
    z0 = syn.synthetic_z0(B, 32, seed=79).to(DEV)
    al = syn.synthetic_alphas(B, seed=80).to(DEV)
    outs = {}
    for prec in ("fp32", "bf16"):
        tr, _ = _trainer(B, N, C, K, prec)
        tr.set_prior_means(O.sample_distant_points(32, K, 10, 10))
        tr.finalize()
        tr.train()
        outs[prec] = tr.step(pcs, gt, z0, al)
        del tr
        torch.cuda.empty_cache()
    a, b = outs["bf16"], outs["fp32"]
    for k in ("d_loss", "rec_loss", "sup_loss", "tot_loss"):
        assert np.isfinite(a[k].item())
        assert abs(a[k].item() - b[k].item()) <= 3e-2 * abs(b[k].item()) + 3e-2, (k, a[k].item(), b[k].item())
    scale = b["sup_fvs"].abs().max().item()
    assert (a["sup_fvs"] - b["sup_fvs"]).abs().max().item() <= 6e-2 * scale
    agree = (a["preds"] == b["preds"]).float().mean().item()
    assert agree >= 0.9, f"bf16 argmax agreement {agree}"


def test_inference_batch_1024_config5():
    """config 5: B=1024 sequences through the eval-mode encoder + likelihood + k=6 vote; the batched
    result equals chunked evaluation (no cross-sample coupling), labels bit-exact."""
    N, C, K = 128, 4, 8
    enc = make_encoder(K, N, C, True, seed=0).to(DEV).eval()
    means = torch.from_numpy(load_golden("misc")[0]["means_K8"]).float()
    pcs = syn.synthetic_pcs(1024, T, N, C, seed=5).to(DEV).permute(0, 3, 1, 2)
    big = inference.OpenSetScorer(enc, means, batch_size=1024)
    small = inference.OpenSetScorer(enc, means, batch_size=96)
    p1, f1, l1 = big.embed(pcs)
    p2, f2, l2 = small.embed(pcs)
    assert torch.equal(p1, p2)
    assert torch.allclose(f1, f2, rtol=1e-5, atol=1e-6)
    lab = torch.arange(1024) // 64
    big.threshold = float(l1.median().item())
    votes = big.vote(l1, p1, 6, K)
    assert votes.numel() == 1024 // 6 and int(votes.max()) <= K
    # oracle likelihood on the device embeddings
    ref = O.joint_likelihood(f1.cpu().numpy(), means.numpy())
    assert np.allclose(l1.cpu().numpy(), ref, rtol=1e-10, atol=0)
    # the vote is checked on the device's own likelihoods: the threshold chosen above IS one of
    # them, and the oracle's value of that element differs in the last ulp
    ref_votes = O.k_vote(l1.cpu().numpy(), p1.cpu().numpy(), big.threshold, 6, K)
    assert np.array_equal(votes.cpu().numpy(), ref_votes)
