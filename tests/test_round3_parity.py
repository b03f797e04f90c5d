"""Round-3 parity evidence (GPU): a reference for EVERY shape bench.py times.  Since round 5 the train-step references
come from goldens generated from the REFERENCE ITSELF at these shapes (tests/golden/full_B*_N*.npz,
make_golden_fullsize.py: one train_variant4 iteration with bench.py's fills and seeds) -- `_oracle_step` serves a shape
from its golden when one exists and from the CPU oracle otherwise; the eval-mode encoder tests use the oracle.

* BASELINE config[3] (point-subsampling sweep) at the benchmarked batch: one V4 train step at B=64 for N=64 and
  N=256 -- the launch paths only these shapes take (split-K choices, 16 statistics replicas, the skinny decoder
  kernels at M=64, the 627 M-parameter decoder) -- in fp32 parity mode (1e-4, labels bit-exact) and in the bf16
  throughput mode (stated bf16 tolerance) against ``oracle.v4_train_step``.  (N=128 is tests/test_round2_parity.py,
  N=32 at B=64 runs the same kernels as N=64 with a quarter of the rows and is covered against the fp32 mode in
  tests/test_configs.py.)
* BASELINE config[4] (open-set inference, B=1024): the bf16 eval-mode encoder with the fused GEMM epilogues against
  an ORACLE forward on 64 of the 1024 sequences (round 2 compared HIP-bf16 with HIP-fp32 only).
* the bf16 eval-mode encoder at N=150 (the reference's default NMAX) and N=256 against the oracle.
"""
import numpy as np
import pytest
import torch

from helpers import T, is_pre_bn_bias, make_encoder
from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip, synthetic as syn
from opensetgaitrecognition_pcaa_amd.train import PCAATrainer
from oracle import pcaa_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
LOSS_KEYS = ("d_loss", "gp", "rec_loss", "loss_g", "sup_loss", "tot_loss")
SEEDS = [0, 1, 2, 3, 4]                       # bench.py's fills
B_BENCH, C_BENCH, K_BENCH = 64, 4, 8


def _cfg(B, N, K):
    cfg = dict(constants.CONFIG)
    cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B, LR=1e-4, B1=0.9, B2=0.99, GP_WEIGHT=15,
               ADV_WEIGHT=1, SUP_LATENT_DIM=32)
    return cfg


def _trainer(N, precision, B=B_BENCH):
    constants.NFEATURES = C_BENCH
    tr = PCAATrainer(_cfg(B, N, K_BENCH), precision=precision, fused_decoder_update=False)
    mods = (tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head, tr.discriminator_projection_head)
    for mod, seed in zip(mods, SEEDS):
        syn.deterministic_fill_(mod, seed)
    return tr, mods


def _inputs(N, B=B_BENCH):
    return (syn.synthetic_pcs(B, T, N, C_BENCH, seed=1234), syn.synthetic_labels(B, K_BENCH, seed=1235),
            syn.synthetic_z0(B, 32, seed=1236), syn.synthetic_alphas(B, seed=1237))


_ORACLE_CACHE = {}


def _golden_step(N, B):
    """The same record from a full-size golden of the REFERENCE (tests/golden/make_golden_fullsize.py), where one exists:
    round 5 -- the N=256 step cost the GPU suite a minute of host time per run through the oracle, and a golden pins the
    shape by the reference itself.  Large gradient tensors come as (l2, 1 024 strided samples)."""
    import os
    from helpers import GOLDEN, full_golden
    if not os.path.exists(os.path.join(GOLDEN, f"full_B{B}_N{N}.npz")):
        return None
    g, m = full_golden(B, N)
    keep = {k: torch.tensor(float(v)) for k, v in zip(LOSS_KEYS, g["losses"])}
    keep.update(preds=torch.from_numpy(g["preds"]), sup_fvs=torch.from_numpy(g["sup_fvs"]),
                out_labels=torch.from_numpy(g["out_labels"]), nsample=int(m["nsample"]))
    keep["g_small"], keep["g_dec"], keep["g_big"], keep["d_grads"] = {}, {}, {}, {}
    for key in g.files:
        kind = key.rsplit("::", 1)[-1]
        if key.startswith("ggrad.") and kind in ("full", "l2"):
            name = key[len("ggrad."):].rsplit("::", 1)[0]
            if kind == "full":
                keep["g_small"][name] = torch.from_numpy(g[key])
            else:
                cs = {"l2": float(g[key]), "samples": g[f"ggrad.{name}::samples"]}
                keep["g_dec" if name.startswith("G.") else "g_big"][name] = cs
        elif key.startswith("dgrad.") and kind == "full":
            keep["d_grads"][key[len("dgrad."):].rsplit("::", 1)[0]] = torch.from_numpy(g[key])
    return keep, torch.from_numpy(g["means"])


def _oracle_step(N, B=B_BENCH):
    """One oracle V4 step at B=64 (N=64: ~10 s, N=256: ~1 min of host CPU and ~35 GB of host memory), shared by the
    fp32 and bf16 tests of that N; only what the tests compare is kept.  A shape with a full-size reference golden
    (N=256) is served from it instead."""
    if (N, B) in _ORACLE_CACHE:
        return _ORACLE_CACHE[(N, B)]
    gold = _golden_step(N, B)
    if gold is not None:
        _ORACLE_CACHE[(N, B)] = gold
        return gold
    # (what is kept per shape is small -- losses, embeddings, encoder gradients, decoder checksums -- so every shape stays
    # cached for the session; round 5: clearing per shape made fp32[64], fp32[256], bf16[64], bf16[256] compute each
    # oracle step twice, two minutes of the GPU suite)
    saved = constants.NFEATURES
    tr, mods = _trainer(N, "fp32", B)
    constants.NFEATURES = saved
    means = O.sample_distant_points(32, K_BENCH, 10, 10).float()
    st = O.V4State(*({k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for m in mods),
                   means, C_BENCH, T, N, K_BENCH)
    del tr
    torch.cuda.empty_cache()
    pcs, gt, z0, al = _inputs(N, B)
    ref = O.v4_train_step(st, pcs.permute(0, 3, 1, 2), gt, z0, al, _cfg(B, N, K_BENCH))
    keep = {k: ref[k] for k in LOSS_KEYS + ("preds", "sup_fvs", "out_labels")}
    # encoder / head gradients in full, the decoder's as (l2, 64 strided samples): dense5 alone is 1.9 GB at N=256
    keep["g_small"] = {k: v for k, v in ref["g_grads"].items() if v is not None and not k.startswith("G.")}
    keep["g_dec"] = {k: syn.checksum(v, 64) for k, v in ref["g_grads"].items() if v is not None and k.startswith("G.")}
    keep["d_grads"] = ref["d_grads"]
    del ref, st
    _ORACLE_CACHE[(N, B)] = (keep, means)
    return keep, means


def _hip_step(N, precision, means, B=B_BENCH, graphed=False):
    """One HIP step from the filled state.  ``graphed``: through PCAATrainer.step_graphed -- the captured hipGraph,
    replayed -- which is the path bench.py's sweep leg TIMES at N=32 (``prefers_graph``).  A throw-away trainer runs
    one eager step first (the library's one-time initialisation must not fall into a capture), then the trainer under
    test captures on its very first call (warmup=0): the replay IS its first step, from the oracle's state."""
    pcs, gt, z0, al = _inputs(N, B)
    args = (pcs.to(DEV).permute(0, 3, 1, 2), gt.to(DEV), z0.to(DEV), al.to(DEV))
    if graphed:
        warm, _ = _trainer(N, precision, B)
        warm.set_prior_means(means)
        warm.finalize()
        warm.train()
        warm.step(*args)
        torch.cuda.synchronize()
        del warm
    tr, _ = _trainer(N, precision, B)
    tr.set_prior_means(means)
    tr.finalize()
    tr.train()
    if graphed:
        out = tr.step_graphed(*args, warmup=0)
        assert len(tr._graphs) == 1 and next(iter(tr._graphs.values()))["graph"] is not None, "the step was not captured"
    else:
        out = tr.step(*args)
    torch.cuda.synchronize()
    return tr, out


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("N", [64, 256])
def test_sweep_shape_fp32_step_vs_oracle_at_bench_batch(N):
    ref, means = _oracle_step(N)
    tr, out = _hip_step(N, "fp32", means)
    _check_parity_grade(tr, out, ref, f"config[3] N={N} B=64 fp32")
    del tr, out
    torch.cuda.empty_cache()


def _check_parity_grade(tr, out, ref, what):
    """the fp32-grade gates: losses / embeddings / logits 1e-4, labels bit-exact, gradients 5e-4 relative l2"""
    for k in LOSS_KEYS:
        assert abs(out[k].item() - ref[k].item()) <= 1e-4 * abs(ref[k].item()) + 1e-5, (k, out[k].item(), ref[k].item())
    # argmax labels bit-exact; a sample whose fp32 top-2 margin is below the 1e-4 tolerance is flagged, not waved through
    lg = ref["out_labels"]
    top2 = lg.topk(2, dim=1).values
    tied = (top2[:, 0] - top2[:, 1]) <= 1e-4 * lg.abs().max()
    same = out["preds"].cpu() == ref["preds"]
    assert bool(same[~tied].all()), "argmax labels must be bit-exact"
    if bool(tied.any()):
        print(f"{what}: {int(tied.sum())} samples with a top-2 logit margin below 1e-4 of scale; "
              f"{int((~same & tied).sum())} of them differ")
    scale = ref["sup_fvs"].abs().max().item()
    assert (out["sup_fvs"].cpu() - ref["sup_fvs"]).abs().max().item() <= 1e-4 * scale
    assert (out["out_labels"].cpu() - lg).abs().max().item() <= 1e-4 * lg.abs().max().item()
    wscale = max(float(v.abs().max()) for k, v in ref["g_small"].items() if k.startswith("E.") and k.endswith("weight"))
    worst = ("", 0.0)
    for name, gref in ref["g_small"].items():
        mine = tr.flat_g.grad_views[name].detach().cpu()
        if is_pre_bn_bias(name):
            assert float(mine.abs().max()) <= 1e-4 * wscale + 1e-4, name
            continue
        rel = float((mine.double() - gref.double()).norm() / (gref.double().norm() + 1e-30))
        worst = max(worst, (name, rel), key=lambda t: t[1])
        assert rel <= 5e-4, (name, rel)
    for name, cs in list(ref["g_dec"].items()) + list(ref.get("g_big", {}).items()):
        mine = syn.checksum(tr.flat_g.grad_views[name], ref.get("nsample", 64))
        assert abs(mine["l2"] - cs["l2"]) <= 5e-4 * cs["l2"], (name, mine["l2"], cs["l2"])
        floor = cs["l2"] / np.sqrt(tr.flat_g.grad_views[name].numel())
        assert np.abs(mine["samples"] - cs["samples"]).max() <= 2e-3 * max(np.abs(cs["samples"]).max(), floor), name
    for name, gref in ref["d_grads"].items():
        if gref is None:
            continue
        mine = tr.flat_d.grad_views["D." + name].detach().cpu()
        if name == "model.4.bias":
            assert float(mine.abs().max()) == 0.0
            continue
        rel = float((mine.double() - gref.double()).norm() / (gref.double().norm() + 1e-30))
        assert rel <= 5e-4, (name, rel)
    print(f"{what} vs oracle: worst encoder gradient rel-l2 {worst[1]:.2e} ({worst[0]})")


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("N", [64, 256])
def test_sweep_shape_bf16_step_vs_oracle_at_bench_batch(N):
    """the mode bench.py's ``sweep`` leg times, against the ORACLE: losses 2e-2, embeddings 5e-2 of scale, weight
    gradients 5e-2 relative l2, argmax agreement reported and gated at 0.9."""
    ref, means = _oracle_step(N)
    tr, out = _hip_step(N, "bf16", means)
    _check_bf16_grade(tr, out, ref, f"config[3] N={N} B=64 bf16")


def _check_bf16_grade(tr, out, ref, what):
    for k in LOSS_KEYS:
        assert np.isfinite(out[k].item())
        assert abs(out[k].item() - ref[k].item()) <= 2e-2 * abs(ref[k].item()) + 2e-2, (k, out[k].item(), ref[k].item())
    scale = ref["sup_fvs"].abs().max().item()
    err = (out["sup_fvs"].cpu() - ref["sup_fvs"]).abs().max().item()
    assert err <= 5e-2 * scale, (err, scale)
    agree = (out["preds"].cpu() == ref["preds"]).float().mean().item()
    rels = {}
    for name, gref in ref["g_small"].items():
        if is_pre_bn_bias(name) or not name.endswith("weight") or gref.dim() < 2:
            continue
        mine = tr.flat_g.grad_views[name].detach().cpu()
        rels[name] = float((mine.double() - gref.double()).norm() / (gref.double().norm() + 1e-30))
    for name, cs in list(ref["g_dec"].items()) + list(ref.get("g_big", {}).items()):
        if name.endswith("weight") and not is_pre_bn_bias(name):
            l2 = syn.checksum(tr.flat_g.grad_views[name], 64)["l2"]
            assert abs(l2 - cs["l2"]) <= 5e-2 * cs["l2"], (name, l2, cs["l2"])
    print(f"{what} vs oracle: argmax agreement {agree:.4f}, sup_fv err {err / scale:.2e} of scale, "
          f"worst weight-gradient rel-l2 {max(rels.values()):.2e} ({max(rels, key=rels.get)})")
    assert agree >= 0.9
    assert max(rels.values()) <= 5e-2, rels


@pytest.mark.timeout(900)
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_sweep_shape_n32_graphed_step_vs_oracle_at_bench_batch(precision):
    """bench.py's sweep point N=32 is timed through hipGraph replay at B=64; this is that path
    (capture + replay of the 4-stream step) at that shape against the ORACLE, fp32 at the parity gates and bf16 -- the
    mode the sweep leg runs -- at the bf16 gates."""
    ref, means = _oracle_step(32)
    tr, out = _hip_step(32, precision, means, graphed=True)
    (_check_parity_grade if precision == "fp32" else _check_bf16_grade)(tr, out, ref, f"config[3] N=32 B=64 {precision} hipGraph replay")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("precision", ["fp32", "fp16x3", "bf16"])
def test_reference_default_shape_step_vs_oracle(precision):
    """the reference's own operating point -- constants.py:29,55: BATCH_SIZE = 16, NMAX = 150,
    4 features (what train_variant4(CONFIG) runs and bench.py's ``ref_default`` leg times) -- one step against the
    oracle: the parity-grade modes at the fp32 gates, bf16 at the bf16 gates.  N = 150 is the shape whose decoder
    widths (1125 ... 18000) are stored zero-padded to multiples of 64 inside the flat buffers."""
    ref, means = _oracle_step(150, 16)
    tr, out = _hip_step(150, precision, means, B=16)
    (_check_bf16_grade if precision == "bf16" else _check_parity_grade)(tr, out, ref, f"reference default B=16 N=150 {precision}")


def _oracle_eval(enc, x_cpu):
    sd = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
    with torch.no_grad():
        return O.cg_encoder_forward(x_cpu.contiguous(), sd, True, training=False)


@pytest.mark.timeout(900)
def test_config4_bf16_eval_encoder_vs_oracle_on_64_of_1024():
    """config[4]'s quoted path -- bf16, BatchNorm + ELU (+ the mean over the frame's points) in the GEMM epilogues, one
    batch of 1024 -- against the CPU oracle's eval-mode forward on every 16th sequence."""
    N, C, K = 128, 4, 8
    enc = make_encoder(K, N, C, True, seed=0).to(DEV).eval()
    pcs = syn.synthetic_pcs(1024, T, N, C, seed=5)
    x = pcs.to(DEV).permute(0, 3, 1, 2)
    with torch.no_grad():
        l16, f16, st = F_hip.encoder_forward(enc, x, False, "bf16")
    assert [s.y is None for s in st.pn] == [True, True, True, True], "the fused-epilogue path must be the one that ran"
    idx = torch.arange(0, 1024, 16)
    ref_oc, ref_fv = _oracle_eval(enc, pcs[idx].permute(0, 3, 1, 2))
    scale = ref_fv.abs().max().item()
    err = (f16.cpu()[idx] - ref_fv).abs().max().item()
    agree = (l16.argmax(1).cpu()[idx] == O.predicted_labels(ref_oc)).float().mean().item()
    print(f"config[4] bf16 fused eval encoder vs ORACLE on 64 of 1024 sequences: label agreement {agree:.4f}, "
          f"embedding err {err / scale:.2e} of scale")
    assert err <= 5e-2 * scale
    assert agree >= 0.95
    # the parity-grade fp32 path on the same 64 sequences: 1e-4, labels bit-exact
    with torch.no_grad():
        l32, f32, _ = F_hip.encoder_forward(enc, x[idx.to(DEV)].contiguous(), False, "fp32")
    assert (f32.cpu() - ref_fv).abs().max().item() <= 1e-4 * scale
    assert torch.equal(l32.argmax(1).cpu(), O.predicted_labels(ref_oc))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("N,B", [(150, 64), (256, 8)])
def test_bf16_eval_encoder_vs_oracle_other_widths(N, B):
    """bf16 eval-mode encoder at the reference's default NMAX=150 (B=64: the rows are whole 256-row tiles, so layers
    2-3 take the fused affine epilogue; the mean over 150 points is the separate pass) and at N=256."""
    C, K = 4, 8
    enc = make_encoder(K, N, C, True, seed=0).to(DEV).eval()
    pcs = syn.synthetic_pcs(B, T, N, C, seed=6)
    with torch.no_grad():
        l16, f16, st = F_hip.encoder_forward(enc, pcs.to(DEV).permute(0, 3, 1, 2), False, "bf16")
    assert st.pn[1].y is None and st.pn[2].y is None, "layers 2-3 must have run with the fused eval epilogue"
    ref_oc, ref_fv = _oracle_eval(enc, pcs.permute(0, 3, 1, 2))
    scale = ref_fv.abs().max().item()
    err = (f16.cpu() - ref_fv).abs().max().item()
    agree = (l16.argmax(1).cpu() == O.predicted_labels(ref_oc)).float().mean().item()
    print(f"bf16 eval encoder vs oracle, N={N} B={B}: label agreement {agree:.4f}, embedding err {err / scale:.2e} of scale")
    assert err <= 5e-2 * scale
    assert agree >= (0.95 if B >= 32 else 0.87)
