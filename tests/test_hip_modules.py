"""Module- and step-level parity of the HIP path (drop-in modules, fused
trainer) against the golden fixtures captured from the reference and against
the CPU oracle on the same seeded inputs.  GPU only; tolerance for fp32 mode is
the north-star's 1e-4 relative, argmax labels bit-exact."""
import json

import numpy as np
import pytest
import torch

from helpers import (T, check_against_record, is_pre_bn_bias, load_golden, make_decoder, make_disc,
                     make_encoder, make_head, sd_clone)
from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip, models, ops, synthetic as syn
from opensetgaitrecognition_pcaa_amd.train import PCAATrainer
from opensetgaitrecognition_pcaa_amd.utils import SeqChamferLoss
from oracle import pcaa_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def _close(a, ref, tol=TOL, floor=1e-6, what=""):
    a = a.detach().float().cpu().double()
    ref = torch.as_tensor(ref).detach().cpu().double()
    err = (a - ref).abs().max().item()
    den = max(ref.abs().max().item(), floor)
    assert err <= tol * den, f"{what}: abs err {err:.3e}, scale {den:.3e}, rel {err / den:.3e}"


@pytest.mark.parametrize("tag", ["enc_cfg1_B4_N128_C5_K8", "enc_B2_N32_C4_K4", "enc_B3_N150_C4_K6_nohead"])
def test_encoder_vs_golden(tag):
    F_hip.set_precision("fp32")
    g, m = load_golden(tag)
    B, N, C, K, head = m["B"], m["N"], m["C"], m["K"], bool(m["head"])
    enc = make_encoder(K, N, C, head, seed=m["fill_seed"]).to(DEV)
    xpm = syn.synthetic_pcs(B, T, N, C, seed=m["pcs_seed"]).to(DEV)          # point-major storage
    x = xpm.permute(0, 3, 1, 2)                                              # zero-copy [B,C,T,N] view
    enc.eval()
    with torch.no_grad():
        oc, fv = enc(x)
    _close(oc, g["eval_out_classes"], what="eval out_classes")
    _close(fv, g["eval_sup_fv"], what="eval sup_fv")
    # standard channel-major contiguous input (what the reference's DataLoader yields) gives the same
    with torch.no_grad():
        oc2, fv2 = enc(x.contiguous())
    # (split-K fp32 atomics in the small head GEMMs make the last bits order-dependent)
    _close(oc2, oc, 1e-5, what="channel-major input")
    _close(fv2, fv, 1e-5, what="channel-major input")

    enc.train()
    rng = np.random.default_rng(77)
    r1 = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32)).to(DEV)
    r2 = torch.from_numpy(rng.standard_normal((B, 32)).astype(np.float32)).to(DEV)
    xg = x.detach().clone().requires_grad_(True)
    oc, fv = enc(xg)
    loss = (oc * r1).sum() + (fv * r2).sum()
    loss.backward()
    _close(oc, g["train_out_classes"], what="train out_classes")
    _close(fv, g["train_sup_fv"], what="train sup_fv")
    assert abs(loss.item() - float(g["train_loss"])) <= TOL * abs(float(g["train_loss"])) + 1e-5
    assert abs(xg.grad.double().norm().item() - float(g["train_dx_l2"])) <= 2e-4 * float(g["train_dx_l2"])
    wscale = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("grad.") and k.endswith("weight::full"))
    for name, p in enc.named_parameters():
        if is_pre_bn_bias(name):
            assert p.grad is not None and float(p.grad.abs().max()) <= 1e-4 * wscale + 1e-4, name
            continue
        check_against_record(g, "grad.", name, p.grad, 3e-4)
    sd = enc.state_dict()
    for name, v in sd.items():
        if "running" in name or "num_batches" in name:
            check_against_record(g, "bn1.", name, v, 2e-5)
    with torch.no_grad():
        _, fv2 = enc(x)
    _close(fv2, g["train2_sup_fv"], what="second train-mode forward")
    for name, v in enc.state_dict().items():
        if "running" in name or "num_batches" in name:
            check_against_record(g, "bn2.", name, v, 2e-5)


@pytest.mark.parametrize("tag", ["dec_B2_N32_C4", "dec_B3_N50_C5_in32"])
def test_decoder_vs_golden(tag):
    g, m = load_golden(tag)
    B, N, C, in_dim = m["B"], m["N"], m["C"], m["in_dim"]
    dec = make_decoder(in_dim, N, C, seed=m["fill_seed"]).to(DEV)
    rng = np.random.default_rng(m["z_seed"])
    z = torch.from_numpy(rng.standard_normal((B, in_dim)).astype(np.float32)).to(DEV).requires_grad_(True)
    r = torch.from_numpy(rng.standard_normal((B, C, T, N)).astype(np.float32)).to(DEV)
    y = dec(z)
    assert tuple(y.shape) == (B, C, T, N)
    (y * r).sum().backward()
    assert abs(y.double().norm().item() - float(g["out_l2"])) <= TOL * float(g["out_l2"])
    assert np.allclose(syn.checksum(y, 64)["samples"], g["out_samples"], rtol=1e-4, atol=1e-5)
    _close(z.grad, g["dz"], what="dz")
    for name, p in dec.named_parameters():
        check_against_record(g, "grad.", name, p.grad, 2e-4)


@pytest.mark.parametrize("tag", ["chamfer_B2_N32_C4", "chamfer_B2_N150_C5"])
def test_chamfer_vs_golden(tag):
    g, m = load_golden(tag)
    B, N, C = m["B"], m["N"], m["C"]
    gts = syn.synthetic_pcs(B, T, N, C, seed=m["gts_seed"]).to(DEV).permute(0, 3, 1, 2)      # strided view
    preds = (syn.synthetic_pcs(B, T, N, C, seed=m["preds_seed"]) * 0.7 + 0.1).permute(0, 3, 1, 2)
    preds = preds.contiguous().to(DEV).requires_grad_(True)
    loss_fn = SeqChamferLoss()
    l = loss_fn(preds, gts)
    l.backward()
    assert abs(l.item() - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    _close(loss_fn(preds.detach(), gts, avg_out=False), g["loss_per_seq"], 2e-5, what="per-seq loss")
    _close(preds.grad, g["dpreds"], 1e-4, what="dpreds")
    # avg_out=False backward with a non-trivial upstream gradient
    p2 = preds.detach().clone().requires_grad_(True)
    w = torch.tensor([0.5, -2.0], device=DEV)[:B]
    (loss_fn(p2, gts, avg_out=False) * w).sum().backward()
    pc = preds.detach().cpu().clone().requires_grad_(True)
    (O.seq_chamfer_loss(pc, gts.cpu(), avg_out=False) * w.cpu()).sum().backward()
    _close(p2.grad, pc.grad, 1e-4, what="dpreds (per-seq weights)")
    # identity property
    assert abs(loss_fn(gts, gts).item()) < 1e-4


@pytest.mark.parametrize("tag", ["disc_B6_K4", "disc_B16_K8"])
def test_discriminator_and_wgan_gp_vs_golden(tag):
    g, m = load_golden(tag)
    K = m["K"]
    disc = make_disc(K, seed=m["fill_seed"]).to(DEV)
    fv, z, alphas = (torch.from_numpy(g[k]).to(DEV) for k in ("fv", "z", "alphas"))
    gt = torch.from_numpy(g["gt"]).to(DEV)
    oh = torch.nn.functional.one_hot(gt, K).float()
    _close(disc(z, oh), g["real"], what="D(real)")
    _close(disc(fv, oh), g["fake"], what="D(fake)")
    params = ops._disc_params(disc)
    losses, grads = ops.disc_wgan_gp(z, fv, oh, alphas.reshape(-1).contiguous(), params, 15.0)
    assert abs(losses[1].item() - float(g["gp"])) <= TOL * abs(float(g["gp"]))
    assert abs(losses[0].item() - float(g["d_loss"])) <= TOL * abs(float(g["d_loss"]))
    for (name, _), gr in zip(disc.named_parameters(), grads):
        if name == "model.4.bias":
            # sum(+1/B) + sum(-1/B): exactly zero in the reference, so b3 never moves under Adam
            assert float(gr.abs().max()) == 0.0
            continue
        check_against_record(g, "grad.", name, gr, 2e-4, scale_floor=1e-3)
    # first-order autograd through the drop-in module vs the oracle
    sd = sd_clone(disc.cpu())
    disc.to(DEV)
    for v in sd.values():
        v.requires_grad_(True)
    xc = fv.cpu().clone().requires_grad_(True)
    w = torch.linspace(-1, 1, fv.shape[0]).view(-1, 1)
    (O.cg_discriminator_forward(xc, oh.cpu(), sd) * w).sum().backward()
    xg = fv.clone().requires_grad_(True)
    disc.zero_grad()
    (disc(xg, oh) * w.to(DEV)).sum().backward()
    _close(xg.grad, xc.grad, what="dD/dx")
    for name, p in disc.named_parameters():
        _close(p.grad, sd[name].grad, 2e-4, floor=1e-2, what=name)   # sum(w) = 0: db3 is ~0 here


def _trainer_from_golden(m, precision="fp32", fused=True):
    B, N, C, K = m["B"], m["N"], m["C"], m["K"]
    constants.NFEATURES = C
    cfg = dict(constants.CONFIG)
    cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B, LR=1e-4, B1=0.9, B2=0.99,
               GP_WEIGHT=15, ADV_WEIGHT=1, SUP_LATENT_DIM=32)
    tr = PCAATrainer(cfg, precision=precision, fused_decoder_update=fused)
    s = m["fill_seeds"]
    for mod, seed in zip((tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head,
                          tr.discriminator_projection_head), s):
        syn.deterministic_fill_(mod, seed)
    tr.finalize()
    tr.train()
    return tr


@pytest.mark.parametrize("fused", [False, True])
def test_v4_train_steps_vs_golden(fused):
    """``fused=True``: the trainer's default since round 5 -- the decoder's wide layers take their Adam update inside the
    fp32-product weight-gradient kernel, so their gradients never exist (``gradless_ranges``); the trajectory and the
    parameters after every step are held to the same gates against the reference's golden as the unfused run."""
    g, m = load_golden("v4_B6_N32_C4_K4")
    B, N, C, K, steps = m["B"], m["N"], m["C"], m["K"], m["steps"]
    tr = _trainer_from_golden(m, fused=fused)
    tr.set_prior_means(torch.from_numpy(g["means"]))
    for s in range(steps):
        pcs = syn.synthetic_pcs(B, T, N, C, seed=m["pcs_seed0"] + s).to(DEV).permute(0, 3, 1, 2)
        gt = syn.synthetic_labels(B, K, seed=m["gt_seed0"] + s).to(DEV)
        z0 = syn.synthetic_z0(B, 32, seed=m["z0_seed0"] + s).to(DEV)
        al = syn.synthetic_alphas(B, seed=m["alpha_seed0"] + s).to(DEV)
        out = tr.step(pcs, gt, z0, al)
        got = np.array([out[k].item() for k in ("d_loss", "gp", "rec_loss", "loss_g", "sup_loss", "tot_loss")])
        ref = g[f"s{s}.losses"]
        # step 0 starts from identical state: the 1e-4 gate.  Later steps start from states that
        # differ by Adam's +-lr steps on near-zero-gradient elements (their sign is rounding noise
        # on both sides), so the per-step gate widens with the trajectory.
        tol = TOL if s == 0 else 5e-4 * s
        assert np.allclose(got, ref, rtol=tol, atol=1e-5), (s, got, ref)
        assert np.array_equal(out["preds"].cpu().numpy(), g[f"s{s}.preds"]), "argmax labels must be bit-exact"
        _close(out["sup_fvs"], g[f"s{s}.sup_fvs"], tol, what=f"sup_fvs step {s}")
        _close(out["out_labels"], g[f"s{s}.out_labels"], tol, what=f"out_labels step {s}")
        if s == 0:
            for name, _ in tr.discriminator.named_parameters():
                if name == "model.4.bias":
                    assert float(tr.flat_d.grad_views["D." + name].abs().max()) == 0.0
                    continue
                check_against_record(g, "s0.dgrad.", name, tr.flat_d.grad_views["D." + name], 2e-4, scale_floor=1e-3)
            wscale = max(float(np.abs(g[k]).max()) for k in g.files
                         if k.startswith("s0.ggrad.E.") and k.endswith("weight::full"))
            gradless = 0
            for name, gv in tr.flat_g.grad_views.items():
                if is_pre_bn_bias(name):
                    assert float(gv.abs().max()) <= 1e-4 * wscale + 1e-4
                    continue
                off = tr.flat_g.offsets[tr.flat_g.names.index(name)]
                if any(lo <= off < hi for lo, hi in tr.gradless_ranges):
                    assert float(gv.abs().max()) == 0.0, "a fused layer must not write a weight gradient"
                    gradless += 1
                    continue
                check_against_record(g, "s0.ggrad.", name, gv, 5e-4)
            # the wide layers (stored zero-padded to multiples of 64 at this N) take the fused update, none with fused=False
            assert gradless == len(tr.gradless_ranges) and (gradless >= 2) == fused, (gradless, tr.gradless_ranges)
        if s in (0, steps - 1):
            for nm, mod in tr.modules().items():
                for name, v in mod.state_dict().items():
                    if is_pre_bn_bias(name):
                        continue
                    if name.endswith("running_mean"):
                        check_against_record(g, f"s{s}.param.{nm}.", name, v, 2e-5, scale_floor=5.0)
                        continue
                    key = f"s{s}.param.{nm}.{name}::full"
                    if key in g.files and v.dtype.is_floating_point:
                        # Adam's step is lr * m/(sqrt(v)+eps): where |grad| is within ~10x of eps the
                        # update depends on the gradient's last bits, so allow a fraction of one lr step
                        # per optimiser step on the worst element, and demand a tight mean error
                        err = (v.detach().cpu().double() - torch.from_numpy(g[key]).double()).abs()
                        scale = float(np.abs(g[key]).max())
                        assert err.max().item() <= 5e-5 * scale + 0.5e-4 * (s + 1), (name, err.max().item())
                        # (the mean gate widens with the trajectory like the loss gates above: the states
                        # the later steps start from differ by Adam's rounding-noise-signed steps)
                        assert err.mean().item() <= 2e-6 * (s + 1) * max(scale, 1.0), (name, err.mean().item())
                        continue
                    check_against_record(g, f"s{s}.param.{nm}.", name, v, 5e-5)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_v4_graphed_steps_match_eager_and_golden(precision):
    """step_graphed (hipGraph capture + replay, device-side Adam step count) performs exactly the
    steps step() does: same golden trajectory (fp32), and the same state as an eagerly stepped twin
    after 4 steps (1 eager warm-up, 1 capture+replay, 2 replays with fresh inputs)."""
    g, m = load_golden("v4_B6_N32_C4_K4")
    B, N, C, K, steps = m["B"], m["N"], m["C"], m["K"], m["steps"]
    twins = [_trainer_from_golden(m, precision), _trainer_from_golden(m, precision)]
    for tr in twins:
        tr.set_prior_means(torch.from_numpy(g["means"]))
    keys = ("d_loss", "gp", "rec_loss", "loss_g", "sup_loss", "tot_loss")
    for s in range(4):
        pcs = syn.synthetic_pcs(B, T, N, C, seed=m["pcs_seed0"] + s).to(DEV).permute(0, 3, 1, 2)
        gt = syn.synthetic_labels(B, K, seed=m["gt_seed0"] + s).to(DEV)
        z0 = syn.synthetic_z0(B, 32, seed=m["z0_seed0"] + s).to(DEV)
        al = syn.synthetic_alphas(B, seed=m["alpha_seed0"] + s).to(DEV)
        ref = twins[0].step(pcs, gt, z0, al)
        ref = {k: v.clone() for k, v in ref.items()}
        out = twins[1].step_graphed(pcs, gt, z0, al, warmup=1)
        got = np.array([out[k].item() for k in keys])
        want = np.array([ref[k].item() for k in keys])
        # both twins run the same kernels; only the order of the fp64 statistics atomics may differ.  In
        # bf16 mode that noise decides bf16 roundings, which Adam's sign-like first steps amplify: the two
        # trajectories agree to the bf16-mode tolerance (2e-2), not to round-off
        # (fp32: round-off at step 0, from identical state; later steps start from states that differ by
        # Adam's rounding-noise-signed first steps, like the golden trajectory gates above)
        rt = (1e-5 if s == 0 else 2e-4 * s) if precision == "fp32" else 2e-2
        assert np.allclose(got, want, rtol=rt, atol=1e-6 if precision == "fp32" else 2e-2), (s, got, want)
        if precision == "fp32":
            assert torch.equal(out["preds"], ref["preds"])
            if s < steps:
                tol = TOL if s == 0 else 5e-4 * s
                assert np.allclose(got, g[f"s{s}.losses"], rtol=tol, atol=1e-5), (s, got, g[f"s{s}.losses"])
                assert np.array_equal(out["preds"].cpu().numpy(), g[f"s{s}.preds"])
    assert twins[1].flat_g.step == twins[0].flat_g.step == 4
    assert int(twins[1].flat_g.step_dev.item()) == 4 and int(twins[1].flat_d.step_dev.item()) == 4
    if precision == "fp32":
        for a, b in ((twins[0].flat_g, twins[1].flat_g), (twins[0].flat_d, twins[1].flat_d)):
            # Adam moves a parameter by at most lr = 1e-4 per step, in the direction of the gradient's sign:
            # elements whose gradient is rounding noise can go opposite ways in the two twins (2 lr per
            # step at worst), everything else agrees to fp32 round-off -- hence a loose max, a tight mean
            assert float((a.p - b.p).abs().max()) <= 2e-4 * 4
            assert float((a.p - b.p).abs().mean()) <= 2e-6
        for (n0, v0), (n1, v1) in zip(twins[0].encoder.state_dict().items(), twins[1].encoder.state_dict().items()):
            if n0.endswith("num_batches_tracked"):
                assert int(v0) == int(v1) == 4, n0


def test_v4_step_bf16_mode_close_to_fp32():
    """bf16-MFMA throughput mode: fp32 accumulation, bf16 PointNet activations.
    Stated tolerance: losses within 2e-2 relative, embeddings within 5e-2 of
    their scale; argmax agreement reported, not required (bf16 cannot be
    bit-exact through four BatchNorm layers)."""
    g, m = load_golden("v4_B6_N32_C4_K4")
    B, N, C, K = m["B"], m["N"], m["C"], m["K"]
    tr = _trainer_from_golden(m, precision="bf16")
    tr.set_prior_means(torch.from_numpy(g["means"]))
    pcs = syn.synthetic_pcs(B, T, N, C, seed=m["pcs_seed0"]).to(DEV).permute(0, 3, 1, 2)
    gt = syn.synthetic_labels(B, K, seed=m["gt_seed0"]).to(DEV)
    out = tr.step(pcs, gt, syn.synthetic_z0(B, 32, seed=m["z0_seed0"]).to(DEV),
                  syn.synthetic_alphas(B, seed=m["alpha_seed0"]).to(DEV))
    got = np.array([out[k].item() for k in ("d_loss", "gp", "rec_loss", "loss_g", "sup_loss", "tot_loss")])
    ref = g["s0.losses"]
    assert np.allclose(got, ref, rtol=2e-2, atol=2e-2), (got, ref)
    _close(out["sup_fvs"], g["s0.sup_fvs"], 5e-2, what="bf16 sup_fvs")


def test_decoder_bf16_mode_wide_layers():
    """bf16 mode streams the wide decoder layers through the 256x256 bf16 MFMA kernel
    (fp32 sources rounded to bf16 in LDS, fp32 accumulation): compare with the fp32 path."""
    B, N, C = 8, 64, 4
    dec = make_decoder(64, N, C, seed=1).to(DEV)
    rng = np.random.default_rng(9)
    z = torch.from_numpy(rng.standard_normal((B, 64)).astype(np.float32)).to(DEV)
    r = torch.from_numpy(rng.standard_normal((B, C * T * N)).astype(np.float32)).to(DEV)
    out32, acts32 = F_hip.decoder_forward(dec, z, "fp32")
    g32, dz32 = F_hip.decoder_backward(dec, acts32, r, mode="fp32")
    out16, acts16 = F_hip.decoder_forward(dec, z, "bf16")
    g16, dz16 = F_hip.decoder_backward(dec, acts16, r, mode="bf16")
    _close(out16, out32, 2e-2, what="decoder output bf16 vs fp32")
    _close(dz16, dz32, 3e-2, what="decoder dz bf16 vs fp32")
    for k in g32:
        _close(g16[k], g32[k], 3e-2, what=k)


def test_v4_step_vs_oracle_config_like_shapes():
    """Seeded comparison with the CPU oracle at a mid-size shape the oracle
    finishes in seconds (B=8, N=64, C=5, K=8) -- covers C=5 and a wider decoder."""
    B, N, C, K = 8, 64, 5, 8
    m = dict(B=B, N=N, C=C, K=K, fill_seeds=[20, 21, 22, 23, 24])
    tr = _trainer_from_golden(m)
    means = O.sample_distant_points(32, K, 10, 10).float()
    tr.set_prior_means(means)
    def sd_cpu(mod):
        return {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()}

    st = O.V4State(*(sd_cpu(mod) for mod in (tr.encoder, tr.decoder, tr.discriminator,
                                             tr.decoder_projection_head,
                                             tr.discriminator_projection_head)), means, C, T, N, K)
    cfg = dict(LR=1e-4, B1=0.9, B2=0.99, GP_WEIGHT=15, ADV_WEIGHT=1)
    for s in range(2):
        pcs = syn.synthetic_pcs(B, T, N, C, seed=500 + s)
        gt = syn.synthetic_labels(B, K, seed=600 + s)
        z0 = syn.synthetic_z0(B, 32, seed=700 + s)
        al = syn.synthetic_alphas(B, seed=800 + s)
        ref = O.v4_train_step(st, pcs.permute(0, 3, 1, 2), gt, z0, al, cfg)
        out = tr.step(pcs.to(DEV).permute(0, 3, 1, 2), gt.to(DEV), z0.to(DEV), al.to(DEV))
        for k in ("d_loss", "gp", "rec_loss", "loss_g", "sup_loss", "tot_loss"):
            assert abs(out[k].item() - ref[k].item()) <= TOL * abs(ref[k].item()) + 1e-5, (s, k, out[k].item(), ref[k].item())
        assert torch.equal(out["preds"].cpu(), ref["preds"])
        _close(out["sup_fvs"], ref["sup_fvs"], what="sup_fvs")


def test_state_dict_roundtrip_and_manifest():
    g, _ = load_golden("misc")
    man = json.loads(str(g["manifest_N32_C4_K8"]))
    constants.NFEATURES = 4
    enc = models.CGEncoder(8, nmax_points=32, use_projection_head=True).to(DEV)
    assert list(enc.state_dict().keys()) == list(man["E"].keys())
    sd = {k: torch.randn_like(v) if v.dtype.is_floating_point else v for k, v in enc.state_dict().items()}
    enc.load_state_dict(sd)
    for k, v in enc.state_dict().items():
        assert torch.equal(v, sd[k])


def test_cpu_input_is_refused():
    constants.NFEATURES = 4
    enc = models.CGEncoder(4, nmax_points=32).float()
    with pytest.raises(RuntimeError, match="no CPU path"):
        enc(torch.zeros(2, 4, 30, 32))


def test_full_size_properties_config2():
    """Size-independent checks at BASELINE config 2 (B=64, T=30, N=128, C=4):
    BatchNorm'd features are finite, the pooled PointNet output equals the mean of
    per-sequence evaluation (eval mode => no cross-sample coupling), Chamfer of a
    cloud with itself is 0 and is permutation invariant."""
    F_hip.set_precision("fp32")
    B, N, C, K = 64, 128, 4, 8
    enc = make_encoder(K, N, C, True, seed=0).to(DEV).eval()
    x = syn.synthetic_pcs(B, T, N, C, seed=1234).to(DEV).permute(0, 3, 1, 2)
    with torch.no_grad():
        oc, fv = enc(x)
        oc1, fv1 = enc(x[5:6])
    assert torch.isfinite(oc).all() and torch.isfinite(fv).all()
    _close(fv[5:6], fv1, 1e-5, what="eval-mode per-sequence independence")
    perm = torch.randperm(N, device=DEV)
    with torch.no_grad():
        _, fvp = enc(x[:4][:, :, :, perm])
    _close(fvp, fv[:4], 1e-4, what="point-permutation invariance of the set encoder")
    loss_fn = SeqChamferLoss()
    assert abs(loss_fn(x, x).item()) < 1e-3
    a = loss_fn(x[:, :, :, perm], x).item()
    assert abs(a) < 1e-3


# ---------------------------------------------------------------- ablation variant 1 (GaussianMeanLearner centroids)
def _v1_trainer(m, learn=False):
    B, N, C, K = m["B"], m["N"], m["C"], m["K"]
    constants.NFEATURES = C
    cfg = dict(constants.CONFIG)
    cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B, LR=1e-4, B1=0.9, B2=0.99,
               GP_WEIGHT=15, ADV_WEIGHT=1, SUP_LATENT_DIM=32)
    tr = PCAATrainer(cfg, precision="fp32", variant="v1", learn_centroids=learn)
    for mod, seed in zip((tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head, tr.mean_learner),
                         m["fill_seeds"]):
        syn.deterministic_fill_(mod, seed)
    tr.finalize()
    tr.train()
    return tr


def _v1_inputs(m, s):
    B, N, C, K = m["B"], m["N"], m["C"], m["K"]
    return (syn.synthetic_pcs(B, T, N, C, seed=m["pcs_seed0"] + s), syn.synthetic_labels(B, K, seed=m["gt_seed0"] + s),
            syn.synthetic_z0(B, 32, seed=m["z0_seed0"] + s), syn.synthetic_alphas(B, seed=m["alpha_seed0"] + s))


def test_v1_train_steps_vs_golden():
    """train_variant1's step as the reference executes it (the mean learner gets no gradient: its
    Variable(z0 + mus) detaches) against the trajectory generated from the reference's own modules."""
    g, m = load_golden("v1_B6_N32_C4_K4")
    steps = m["steps"]
    tr = _v1_trainer(m)
    ml0 = {k: v.detach().clone() for k, v in tr.mean_learner.state_dict().items()}
    for s in range(steps):
        pcs, gt, z0, al = _v1_inputs(m, s)
        out = tr.step(pcs.to(DEV).permute(0, 3, 1, 2), gt.to(DEV), z0.to(DEV), al.to(DEV))
        got = np.array([out[k].item() for k in ("d_loss", "gp", "rec_loss", "loss_g", "sup_loss", "tot_loss")])
        tol = TOL if s == 0 else 5e-4 * s
        assert np.allclose(got, g[f"s{s}.losses"], rtol=tol, atol=1e-5), (s, got, g[f"s{s}.losses"])
        assert np.array_equal(out["preds"].cpu().numpy(), g[f"s{s}.preds"])
        _close(out["sup_fvs"], g[f"s{s}.sup_fvs"], tol, what=f"sup_fvs step {s}")
    for name, v in tr.discriminator.state_dict().items():
        check_against_record(g, f"s{steps - 1}.param.D.", name, v, 5e-5, scale_floor=1.0)
    for name, v in tr.mean_learner.state_dict().items():
        if name.endswith("weight") or name.endswith("bias"):
            assert torch.equal(v, ml0[name]), f"{name}: the reference never updates the mean learner"
        elif name.endswith("running_var"):
            check_against_record(g, f"s{steps - 1}.param.ML.", name, v, 1e-4, scale_floor=1.0)
        elif name.endswith("num_batches_tracked"):
            assert int(v) == steps
    cent = tr.learned_centroids()
    _close(cent, g["centroids_train_mode"], 1e-3, floor=1e-3, what="learned centroids (train-mode BN over the K one-hots)")


def test_v1_learn_centroids_option_vs_oracle():
    """learn_centroids=True (the variant's stated intent, not the reference's behaviour): z stays attached,
    pcaa_disc_wgan_gp returns d(d_loss)/dz and the mean learner is trained -- against the oracle's autograd."""
    g, m = load_golden("v1_B6_N32_C4_K4")
    K, C, N = m["K"], m["C"], m["N"]
    tr = _v1_trainer(m, learn=True)

    def sd_cpu(mod):
        return {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()}

    st = O.V1State(sd_cpu(tr.encoder), sd_cpu(tr.decoder), sd_cpu(tr.discriminator), sd_cpu(tr.decoder_projection_head),
                   sd_cpu(tr.mean_learner), C, T, N, K)
    cfg = dict(LR=1e-4, B1=0.9, B2=0.99, GP_WEIGHT=15, ADV_WEIGHT=1)
    pcs, gt, z0, al = _v1_inputs(m, 0)
    ref = O.v1_train_step(st, pcs.permute(0, 3, 1, 2), gt, z0, al, cfg, attach_centroids=True)
    out = tr.step(pcs.to(DEV).permute(0, 3, 1, 2), gt.to(DEV), z0.to(DEV), al.to(DEV))
    for k in ("d_loss", "gp", "rec_loss", "loss_g", "sup_loss", "tot_loss"):
        assert abs(out[k].item() - ref[k].item()) <= TOL * abs(ref[k].item()) + 1e-5, (k, out[k].item(), ref[k].item())
    # the gradient shrinks ~100x per BatchNorm layer on the way down (6 rows, one-hot inputs: 0.28 at the output
    # layer, 1e-6 at the first), each BatchNorm backward cancelling O(1) terms: the error is judged against the
    # scale of the chain's gradient, not each tensor's own
    gscale = max(float(v.abs().max()) for k, v in ref["d_grads"].items() if k.startswith("ML.") and v is not None)
    for name, _ in tr.mean_learner.named_parameters():
        gr = ref["d_grads"]["ML." + name]
        got = tr.flat_d.grad_views["ML." + name]
        if name in ("model.0.bias", "model.3.bias", "model.6.bias"):       # feed a BatchNorm: analytically zero
            assert float(got.abs().max()) == 0.0
            continue
        err = float((got.cpu().double() - gr.double()).abs().max())
        assert err <= 5e-5 * gscale, (name, err, gscale)
    for name, v in tr.mean_learner.state_dict().items():
        if name.endswith("weight") and v.dim() == 2:
            err = (v.cpu() - st.gml[name]).abs().max().item()
            # one Adam step moves an element by lr in the direction of its gradient's sign: a noise-level
            # gradient can send the two sides opposite ways (2 lr); the mean must agree far better
            assert err <= 2.1e-4, (name, err)
            if name in ("model.6.weight", "model.9.weight"):     # gradients well above Adam's eps: signs agree
                assert (v.cpu() - st.gml[name]).abs().mean().item() <= 1e-5, name


def test_checkpoints_are_compact_and_reload(tmp_path):
    """Checkpoint files of the reference's layout (<name>_{E,G,D,GPH,DPH}.pt, PCAA_ablation.py:1073-1117): after
    finalize() parameters are views into the trainer's flat buffers, the files must still hold only their own
    tensors and load into fresh modules of the reference's shapes."""
    g, m = load_golden("v4_B6_N32_C4_K4")
    tr = _trainer_from_golden(m)
    tr.set_prior_means(torch.from_numpy(g["means"]))
    tr.save_checkpoints(str(tmp_path), "mdl")
    sizes = {k: (tmp_path / f"mdl_{k}.pt").stat().st_size for k in ("E", "G", "D", "GPH", "DPH")}
    n_enc = sum(v.numel() * v.element_size() for v in tr.encoder.state_dict().values())
    assert sizes["E"] <= n_enc + (1 << 16), sizes           # not the 40 MB flat buffer
    assert sizes["D"] < (1 << 17) and sizes["GPH"] < (1 << 16)
    constants.NFEATURES = m["C"]
    enc2 = models.CGEncoder(m["K"], nmax_points=m["N"], use_projection_head=True)
    enc2.load_state_dict(torch.load(tmp_path / "mdl_E.pt", map_location="cpu"))
    for (k, a), (_, b) in zip(enc2.state_dict().items(), tr.encoder.state_dict().items()):
        assert torch.equal(a, b.cpu()), k



def test_empty_and_single_inputs():
    """Edge cases of the module boundary: an empty batch is refused with ValueError in both modes (the reference's
    train-mode BatchNorm raises ValueError too); a single sequence (B = 1: T*N rows feed the BatchNorm statistics)
    runs the train-mode forward and matches the oracle."""
    F_hip.set_precision("fp32")
    K, N, C = 4, 32, 4
    enc = make_encoder(K, N, C, True, seed=5)
    sd = sd_clone(enc)                                  # CPU copies for the oracle
    enc = enc.to(DEV)
    dec = make_decoder(32, N, C, seed=6).to(DEV)
    disc = make_disc(K, seed=7).to(DEV)
    for mode in (True, False):
        for mod in (enc, dec, disc):
            mod.train(mode)
        with pytest.raises(ValueError):
            enc(torch.empty(0, C, T, N, device=DEV))
        with pytest.raises(ValueError):
            dec(torch.empty(0, 32, device=DEV))
        with pytest.raises(ValueError):
            disc(torch.empty(0, 32, device=DEV), torch.empty(0, K, device=DEV))
    enc.train()
    xpm = syn.synthetic_pcs(1, T, N, C, seed=11)
    logits, fv = enc(xpm.to(DEV).permute(0, 3, 1, 2))
    ref = O.cg_encoder_forward(xpm.permute(0, 3, 1, 2), sd, True, True)
    _close(logits, ref[0], what="B=1 logits")
    _close(fv, ref[1], what="B=1 sup_fv")


def test_dil_temp_conv1d_refuses_output_widths_the_elementwise_kernels_cannot_take():
    """The HIP path's BatchNorm / ELU passes move 16-byte quads of channels: an output width that is not a multiple of
    four (never used by the reference: DTC_FILTERS, constants.py:37) is refused at construction, not deep inside forward."""
    for bad in (10, 12, 2048):
        with pytest.raises(NotImplementedError, match="multiple of 4"):
            models.DilTempConv1d(6, bad, 2)


@pytest.mark.parametrize("cin,cout,d", [(6, 32, 2), (5, 8, 1), (10, 16, 4)])
def test_dil_temp_conv1d_odd_channel_counts_forward_backward_vs_oracle(cin, cout, d):
    """the temporal block's weight gradients are deferred into one grouped launch that stages 16-byte
    quads (cout % 4, 3 cin % 4).  A DilTempConv1d whose INPUT width is not a multiple of four -- outside the product's own
    widths, inside the drop-in module's contract (models.py:46-55) -- must keep the per-layer product: forward, input
    gradient and the parameter gradients against the oracle under autograd."""
    F_hip.set_precision("fp32")
    B = 3
    layer = models.DilTempConv1d(cin, cout, d).float()
    syn.deterministic_fill_(layer, 7)
    sd = {k: v.detach().clone().double() for k, v in layer.state_dict().items()}
    layer = layer.to(DEV).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, cin, T, generator=g)
    r = torch.randn(B, cout, T, generator=g)
    xg = x.to(DEV).requires_grad_(True)
    out = layer(xg)
    (out * r.to(DEV)).sum().backward()
    names = ("conv1d.weight", "batch_norm.weight", "batch_norm.bias")
    xr = x.double().requires_grad_(True)
    for k in names:
        sd[k].requires_grad_(True)
    ref = O.dil_temp_conv1d(xr, sd, "", d, training=True, update_stats=False)
    grads = torch.autograd.grad((ref * r.double()).sum(), [xr] + [sd[k] for k in names])
    _close(out, ref, what="forward")
    _close(xg.grad, grads[0], 5e-4, what="dx")
    got = dict(layer.named_parameters())
    for k, gr in zip(names, grads[1:]):
        _close(got[k].grad, gr, 5e-4, what=k)
