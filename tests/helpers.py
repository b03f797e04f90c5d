"""Shared helpers for the test-suite (CPU and GPU)."""
import json
import os

import numpy as np
import torch

from opensetgaitrecognition_pcaa_amd import constants, models, synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
T = constants.NSTEPS


def load_golden(tag):
    g = np.load(os.path.join(GOLDEN, tag + ".npz"), allow_pickle=False)
    meta = json.loads(str(g["meta"])) if "meta" in g.files else {}
    return g, meta


def make_encoder(K, N, C, head, seed=0):
    constants.NFEATURES = C
    m = models.CGEncoder(K, nmax_points=N, use_projection_head=head).float()
    syn.deterministic_fill_(m, seed)
    return m


def make_decoder(in_dim, N, C, seed=1):
    constants.NFEATURES = C
    m = models.CGDecoder(input_dim=in_dim, nmax_points=N).float()
    syn.deterministic_fill_(m, seed)
    return m


def make_disc(K, seed=2):
    m = models.CGDiscriminator(K).float()
    syn.deterministic_fill_(m, seed)
    return m


def make_head(i, o, seed):
    m = torch.nn.Sequential(torch.nn.Linear(i, o), torch.nn.ELU()).float()
    syn.deterministic_fill_(m, seed)
    return m


def sd_clone(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check_against_record(g, prefix, name, value, tol, scale_floor=0.0):
    """compare ``value`` with the golden entry (full tensor or checksum)."""
    if value is None:
        assert f"{prefix}{name}::none" in g.files, f"{name}: golden has a gradient, got None"
        return
    if f"{prefix}{name}::full" in g.files:
        ref = torch.from_numpy(g[f"{prefix}{name}::full"])
        err = float((value.detach().cpu().double() - ref.double()).abs().max())
        den = max(float(ref.double().abs().max()), scale_floor)
        assert err <= tol * den + 1e-12, f"{name}: abs err {err:.3e} vs scale {den:.3e}"
    elif f"{prefix}{name}::l2" in g.files:
        cs = syn.checksum(value, len(g[f"{prefix}{name}::samples"]))      # (16 strided samples in rounds 1-4's files, 1 024 in the full-size ones)
        l2 = float(g[f"{prefix}{name}::l2"])
        den = max(l2, scale_floor)
        assert abs(cs["l2"] - l2) <= tol * den, f"{name}: l2 {cs['l2']} vs {l2}"
        ref_s = g[f"{prefix}{name}::samples"]
        smax = max(np.abs(ref_s).max(), scale_floor)
        assert np.abs(cs["samples"] - ref_s).max() <= tol * max(smax, l2 / np.sqrt(value.numel())) * 4, \
            f"{name}: samples differ"
    elif f"{prefix}{name}::none" in g.files:
        raise AssertionError(f"{name}: golden has no gradient but got one")
    else:
        raise KeyError(f"{prefix}{name} not in golden")


# conv biases that feed a BatchNorm: their gradient is analytically zero
# (BN removes the mean), the reference's value is rounding noise.
def is_pre_bn_bias(name):
    return name.endswith("module.0.bias") or name.endswith("conv1d.bias")


def ring_allreduce_bf16(grads):
    """What a ring all-reduce of bf16 gradient buckets computes, element for element: every rank rounds its fp32
    bucket to bf16 once; the buffer is cut into W chunks and chunk c is accumulated hop by hop along the ring,
    starting at rank (c + 1) % W -- each hop adds the local bf16 value to the incoming partial sum in fp32 and rounds
    the result to bf16 for the next hop (RCCL's bf16 sum) -- W - 1 roundings of the partial sum in a fixed, per-chunk
    order; the all-gather phase then copies the finished chunk.  ``grads``: list of W equally sized fp32 tensors
    (one per rank).  Returns the reduced tensor as fp32 (every rank ends with the same bits)."""
    W = len(grads)
    flat = [g.reshape(-1) for g in grads]
    n = flat[0].numel()
    out = torch.empty(n, dtype=torch.float32, device=flat[0].device)
    bounds = [n * c // W for c in range(W + 1)]
    for c in range(W):
        lo, hi = bounds[c], bounds[c + 1]
        order = [(c + 1 + i) % W for i in range(W)]
        acc = flat[order[0]][lo:hi].bfloat16()
        for r in order[1:]:
            acc = (acc.float() + flat[r][lo:hi].bfloat16().float()).bfloat16()
        out[lo:hi] = acc.float()
    return out.view_as(grads[0])


OUTLIER_CAP = 50.0


def compare_record_l2(g, prefix, name, value, tol):
    """``value`` against a golden record in the relative-l2 sense of the full-size parity tests: a full tensor by
    ||a - b|| / ||b|| <= tol; a large one (l2, strided samples) by its l2 and by its samples, of which at most 1 % may lie
    outside 4 tol of the samples' scale and none outside OUTLIER_CAP tol.  (Elementwise max-abs is the wrong gate at full size: a near-tie in one of
    Chamfer's 245 760 argmins resolved the other way moves single gradient elements by 1e-3 of the tensor's largest --
    between the reference's own contraction order and any other.)"""
    if f"{prefix}{name}::full" in g.files:
        b = g[f"{prefix}{name}::full"].astype(np.float64)
        a = value.detach().cpu().double().numpy().reshape(b.shape)
        rel = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
        assert rel <= tol, f"{name}: rel-l2 {rel:.3e}"
        return rel
    ref_s = g[f"{prefix}{name}::samples"]
    cs = syn.checksum(value, len(ref_s))
    l2 = float(g[f"{prefix}{name}::l2"])
    assert abs(cs["l2"] - l2) <= tol * l2, f"{name}: l2 {cs['l2']} vs {l2}"
    scale = max(np.abs(ref_s).max(), l2 / np.sqrt(value.numel()))
    dev = np.abs(cs["samples"] - ref_s)
    bad = (dev > 4 * tol * scale).mean()
    assert bad <= 0.01, f"{name}: {bad:.3%} of the samples differ"
    # the allowed 1 % is for Chamfer's near-tie argmins (a few 1e-3 of the scale); it is not a licence for a corrupted
    # tile edge or row: no sample at all may be further out than OUTLIER_CAP x tol of the scale (round-5 advisor finding)
    worst = float(dev.max()) / scale
    assert worst <= OUTLIER_CAP * tol, f"{name}: a sample is {worst:.3e} of the scale away (cap {OUTLIER_CAP * tol:.1e})"
    return abs(cs["l2"] - l2) / l2


LABEL_REPORT = []      # one entry per check_step_against_full_golden call: the near-tie accounting of the label gate


def full_golden(B, N):
    """One train_variant4 iteration of the REFERENCE at a benchmarked shape (tests/golden/make_golden_fullsize.py:
    bench.py's fills and input seeds).  -> (npz, meta)"""
    return load_golden(f"full_B{B}_N{N}")


def check_step_against_full_golden(g, losses, preds, sup_fvs, out_labels, grads_g, grads_d, tol=1e-4, gtol=5e-4, what=""):
    """The fp32-grade gates against a full-size golden: losses / embeddings / logits ``tol`` relative, argmax labels
    bit-exact (a sample whose reference top-2 margin is below the tolerance is excluded from that gate AND reported: the
    count and how many of them differ are printed and appended to LABEL_REPORT), every
    recorded gradient ``gtol`` in the relative-l2 sense (compare_record_l2; large ones: l2 and 1 024 strided samples).
    ``grads_g`` {"E.<name>" / "GPH.<name>" / "G.<name>": tensor}, ``grads_d`` {"<name>": tensor}; a missing name is skipped."""
    ref = g["losses"]
    got = np.array([float(v) for v in losses])
    assert np.allclose(got, ref, rtol=tol, atol=1e-5), (what, got, ref)
    lg = torch.from_numpy(g["out_labels"])
    top2 = lg.topk(2, dim=1).values
    tied = (top2[:, 0] - top2[:, 1]) <= tol * lg.abs().max()
    same = preds.cpu() == torch.from_numpy(g["preds"])
    # "flag the sample rather than silently accept a flip" -- the count of near-tied samples and how
    # many of THOSE came out differently is printed with every call (pytest -s / -rP shows it; LABEL_REPORT keeps it for
    # the caller), and a flip of any untied sample fails
    n_tied, n_tied_diff = int(tied.sum()), int((~same[tied]).sum())
    LABEL_REPORT.append({"what": what, "samples": int(same.numel()), "tied": n_tied, "tied_differ": n_tied_diff,
                         "untied_differ": int((~same[~tied]).sum())})
    print(f"[labels] {what or 'step'}: {same.numel()} samples, {n_tied} with a reference top-2 margin <= {tol:g} of scale "
          f"({n_tied_diff} of them differ), untied differing: {int((~same[~tied]).sum())}")
    assert bool(same[~tied].all()), f"{what}: argmax labels must be bit-exact"
    fv = torch.from_numpy(g["sup_fvs"])
    assert (sup_fvs.cpu() - fv).abs().max().item() <= tol * fv.abs().max().item(), what
    assert (out_labels.cpu() - lg).abs().max().item() <= tol * lg.abs().max().item(), what
    checked = 0
    wscale = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("ggrad.E.") and k.endswith("weight::full"))
    for name, t in grads_g.items():
        if is_pre_bn_bias(name):
            assert float(t.abs().max()) <= 1e-4 * wscale + 1e-4, name
            continue
        if any(f"ggrad.{name}::{kind}" in g.files for kind in ("full", "l2")):
            compare_record_l2(g, "ggrad.", name, t, gtol)
            checked += 1
    for name, t in grads_d.items():
        if name == "model.4.bias":
            assert float(t.abs().max()) <= 1e-6
            continue
        compare_record_l2(g, "dgrad.", name, t, gtol)
        checked += 1
    return checked
