"""The decoder's bf16 weight images (round 4, ABI 14): single-process bf16 steps stream them instead of the fp32 matrices.
The kernels are bit-for-bit the kernels on the fp32 stream (tests/test_hip_ops.py::test_skinny_w16_*); a whole step is not
reproducible to the bit even against itself (fp64 atomics in the BatchNorm statistics arrive in any order: two runs of
the same three steps differ by ~1e-5 relative), so here: the steps agree to that noise; an image always EQUALS the rounded
current weight; it is built once and then kept by the fused update; a weight written from outside (load_state_dict, an
in-place edit) is noticed, also by the graphed step."""
import os

import pytest
import torch

from helpers import T
from opensetgaitrecognition_pcaa_amd import constants, synthetic as syn
from opensetgaitrecognition_pcaa_amd.train import PCAATrainer

pytestmark = pytest.mark.gpu
B, N, C, K = 64, 32, 4, 8
LOSS_KEYS = ("d_loss", "gp", "rec_loss", "loss_g", "sup_loss", "tot_loss")


def _cfg():
    cfg = dict(constants.CONFIG)
    cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B, LR=1e-4, B1=0.9, B2=0.99, GP_WEIGHT=15,
               ADV_WEIGHT=1, SUP_LATENT_DIM=32)
    return cfg


def _trainer():
    constants.NFEATURES = C
    tr = PCAATrainer(_cfg(), precision="bf16")
    mods = (tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head, tr.discriminator_projection_head)
    for mod, seed in zip(mods, range(5)):
        syn.deterministic_fill_(mod, seed)
    g = torch.Generator().manual_seed(7)
    tr.set_prior_means(10.0 * torch.nn.functional.normalize(torch.randn(K, 32, generator=g), dim=1))
    tr.finalize()
    tr.train()
    return tr


def _inputs(seed):
    pcs, gt, z0, al = (syn.synthetic_pcs(B, T, N, C, seed=seed), syn.synthetic_labels(B, K, seed=seed + 1),
                       syn.synthetic_z0(B, 32, seed=seed + 2), syn.synthetic_alphas(B, seed=seed + 3))
    return pcs.to("cuda").permute(0, 3, 1, 2), gt.to("cuda"), z0.to("cuda"), al.to("cuda")


def _run(steps, images, graphed=False, poke_at=None):
    os.environ["PCAA_DEC_W16"] = "1" if images else "0"
    try:
        torch.manual_seed(0)
        tr = _trainer()
        losses = []
        for i in range(steps):
            if poke_at is not None and i == poke_at:
                with torch.no_grad():                     # a weight written from outside, through torch
                    tr.decoder.dense_layers()[3].weight.mul_(1.01)
            out = (tr.step_graphed if graphed else tr.step)(*_inputs(100 + 10 * i))
            losses.append({k: float(out[k]) for k in LOSS_KEYS})
        torch.cuda.synchronize()
        w = [l.weight.detach().clone() for l in tr.decoder.dense_layers()]
        return tr, losses, w
    finally:
        os.environ.pop("PCAA_DEC_W16", None)


def _same_to_run_noise(l1, l0, w1, w0):
    for a, b in zip(l1, l0):
        for k in LOSS_KEYS:
            assert abs(a[k] - b[k]) <= 1e-3 * max(abs(b[k]), 1.0), (k, a[k], b[k])      # run-to-run noise grows with the step count
    # Adam moves a weight by at most ~LR per step whatever the gradient's size: an element whose gradient is at the noise
    # level can go the other way in one of the two runs
    for a, b in zip(w1, w0):
        assert (a - b).abs().max().item() <= 2 * 1e-4 * len(l1)


def _images_are_current(tr):
    assert tr._dec_w16, "no image in use"
    for i, (img, ver) in tr._dec_w16.items():
        assert ver is not None and torch.equal(img, tr._dec_fused[i][2].bfloat16()), f"layer {i}: image != rounded weight"


def test_steps_with_images_match_the_steps_without():
    tr1, l1, w1 = _run(4, True)
    tr0, l0, w0 = _run(4, False)
    assert l1[0] == l0[0], "the first step starts from identical state: identical kernels, identical result"
    _same_to_run_noise(l1, l0, w1, w0)
    _images_are_current(tr1)
    n_img = len(tr1._dec_w16)
    assert n_img >= 3 and tr1.w16_casts == n_img, "one cast per layer, then the fused update keeps the image"
    assert tr0.w16_casts == 0 and not tr0._dec_w16


@pytest.mark.parametrize("graphed", [False, True])
def test_a_weight_written_from_outside_is_noticed(graphed):
    steps, poke = (6, 4) if graphed else (4, 2)            # graphed: two eager warm-up calls, capture, replay, poke, replay
    tr1, l1, w1 = _run(steps, True, graphed=graphed, poke_at=poke)
    tr0, l0, w0 = _run(steps, False, graphed=graphed, poke_at=poke)
    _same_to_run_noise(l1, l0, w1, w0)
    _images_are_current(tr1)
    assert tr1.w16_casts == len(tr1._dec_w16) + 1, "exactly the poked layer was rebuilt"
