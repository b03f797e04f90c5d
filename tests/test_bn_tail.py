"""The BatchNorm finalize carried by the launch that produces the statistics (csrc/bn_tail.h, round 3): the last
workgroup of the producer writes the coefficients instead of a stand-alone bn_finalize / bn_bwd_finalize launch.
Same arithmetic on the same replica sums, so the two forms must agree to the noise of the fp64 atomics' order."""
import numpy as np
import pytest
import torch

from helpers import T
from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip, ops, synthetic as syn
from opensetgaitrecognition_pcaa_amd._lib import KC, PCAA_BF16
from opensetgaitrecognition_pcaa_amd.train import PCAATrainer

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bn(ch, seed):
    bn = torch.nn.BatchNorm1d(ch).to(DEV)
    syn.deterministic_fill_(bn, seed)
    return bn


@pytest.mark.parametrize("cin,cout", [(512, 512), (512, 1024)])
def test_gemm_carries_forward_finalize_repeatedly(cin, cout):
    """the LDS-DMA GEMM with the statistics epilogue, 30 launches back to back from pre-read (L2-warm) statistics
    buffers: the carried finalize equals the stand-alone kernel every time (a stale or early read of another
    workgroup's atomics would show as a wrong mean / rstd on some launch)"""
    P = 256 * 240                                   # 240 row tiles x cout/256 column tiles: every CU gets several
    g = torch.Generator(device=DEV).manual_seed(1)
    x = (torch.randn(P, cin, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(cout, cin, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(cout, device=DEV, generator=g) * 0.1
    taken0 = ops.TAILS["taken"]
    for it in range(30):
        bn_a, bn_b = _bn(cout, 7), _bn(cout, 7)
        stats = ops.new_stats(cout, DEV)
        _ = stats.sum().item()                      # the statistics lines are in this XCD's L2 / the host waited
        tail = ops.BnTailFwd(P, bias, bn_a, cout)
        y = ops.gemm(x, KC, w, KC, P, cout, cin, colstats=stats, out_dtype=torch.bfloat16, math=PCAA_BF16, tail=tail)
        ref = ops.bn_finalize(stats, P, bias, bn_b, cout)
        torch.cuda.synchronize()
        for a, b, nm in zip(tail.out, ref, ("scale", "shift", "mean", "rstd")):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), (it, nm, (a - b).abs().max().item())
        assert torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-6, atol=1e-7)
        assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-6, atol=1e-7)
        assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 1
        assert int(stats._pcaa_counter[0].item()) == 0, "the finalizer leaves the arrival counter at zero"
    assert ops.TAILS["taken"] - taken0 == 30, "the LDS-DMA launch must have carried every finalize"
    # sanity of the values themselves against torch
    yf = y.float()
    assert torch.allclose(tail.out[2], yf.mean(0), rtol=2e-2, atol=2e-3)


def test_small_producers_carry_their_finalize():
    """pointnet_in (forward statistics, backward statistics) and the mean-pool backward statistics"""
    P, C, cout = 64 * 30 * 32, 4, 512
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(P, C, device=DEV, generator=g)
    W = torch.randn(cout, C, device=DEV, generator=g) * 0.5
    bias = torch.randn(cout, device=DEV, generator=g) * 0.1
    bn_a, bn_b = _bn(cout, 3), _bn(cout, 3)
    t0 = ops.TAILS["taken"]
    stats = ops.new_stats(cout, DEV)
    tail = ops.BnTailFwd(P, bias, bn_a, cout)
    ops.pointnet_in_fwd(x, W, None, None, stats, tail=tail)
    ref = ops.bn_finalize(stats, P, bias, bn_b, cout)
    for a, b in zip(tail.out, ref):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    scale, shift, mean, rstd = ref
    da = (torch.randn(P, cout, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
    btail = ops.BnTailBwd(P, bn_a, mean, rstd, cout)
    st2 = ops.pointnet_in_bwd_stats(da, x, W, scale, shift, mean, rstd, tail=btail)
    ref2 = ops.bn_bwd_finalize(st2, P, bn_b, mean, rstd, cout)
    for a, b in zip(btail.out, ref2):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    # mean-pool backward statistics
    groups, ch = 1920, 1024
    dpool = torch.randn(groups, ch, device=DEV, generator=g)
    e = torch.randn(2, groups, ch, device=DEV, generator=g)
    bn_c = _bn(ch, 4)
    m2, r2 = torch.randn(ch, device=DEV, generator=g), torch.rand(ch, device=DEV, generator=g) + 0.5
    ptail = ops.BnTailBwd(groups * 128, bn_c, m2, r2, ch)
    st3 = ops.bn_pool_bwd_stats(dpool, e, 1.0 / 128, tail=ptail)
    ref3 = ops.bn_bwd_finalize(st3, groups * 128, bn_c, m2, r2, ch)
    for a, b in zip(ptail.out, ref3):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert ops.TAILS["taken"] - t0 == 3


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_train_step_with_and_without_carried_finalizes(precision):
    """one V4 train step from the same state with the finalizes carried by their producers and as stand-alone
    launches: same losses, same parameters (to the order noise of the fp64 atomics)"""
    B, N, C, K = 8, 32, 4, 4
    outs = {}
    for enabled in (True, False):
        ops.TAILS["enabled"] = enabled
        try:
            constants.NFEATURES = C
            cfg = dict(constants.CONFIG)
            cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B)
            tr = PCAATrainer(cfg, precision=precision)
            for i, m in enumerate((tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head,
                                   tr.discriminator_projection_head)):
                syn.deterministic_fill_(m, 50 + i)
            tr.sample_prior_means()
            tr.finalize()
            tr.train()
            t0, s0 = ops.TAILS["taken"], ops.TAILS["standalone"]
            for s in range(2):
                out = tr.step(syn.synthetic_pcs(B, T, N, C, seed=60 + s).to(DEV).permute(0, 3, 1, 2),
                              syn.synthetic_labels(B, K, seed=70 + s).to(DEV), syn.synthetic_z0(B, 32, seed=80 + s).to(DEV),
                              syn.synthetic_alphas(B, seed=90 + s).to(DEV))
            torch.cuda.synchronize()
            outs[enabled] = ({k: out[k].item() for k in ("d_loss", "rec_loss", "sup_loss", "tot_loss")},
                             tr.flat_g.p.detach().clone(), {k: v.detach().clone() for k, v in tr.encoder.state_dict().items()},
                             ops.TAILS["taken"] - t0, ops.TAILS["standalone"] - s0)
            del tr
        finally:
            ops.TAILS["enabled"] = True
    (la, pa, sa, taken, alone), (lb, pb, sb, taken_off, alone_off) = outs[True], outs[False]
    print(f"{precision}: finalizes per 2 steps carried {taken} / stand-alone {alone}  (switch off: {taken_off} / {alone_off})")
    assert taken_off == 0 and alone_off == 40
    # bf16: everything but the K-split first temporal layer (its statistics come from the slab reduction); fp32: the
    # exact-fp32 GEMM kernels and the two-pass BatchNorm backward of PointNet layers 2-3 do not carry tails
    assert taken >= (36 if precision == "bf16" else 26) and taken + alone == 40
    # with the switch off the first layer's statistics also come from a pass over [P, cout] instead of the points'
    # moments: equal to ~1e-6, which bf16 storage of the activations turns into bf16-level differences downstream
    ltol, mtol = (1e-3, 2e-5) if precision == "bf16" else (1e-5, 1e-7)
    for k in la:
        assert abs(la[k] - lb[k]) <= ltol * abs(lb[k]) + 1e-6, (k, la[k], lb[k])
    # an Adam step flips (2 lr) where a gradient is rounding noise; two steps here
    assert (pa - pb).abs().max().item() <= (2.5e-4 if precision != "bf16" else 4.5e-4)
    assert (pa - pb).abs().mean().item() <= mtol
    for k in sa:
        if sa[k].dtype.is_floating_point:
            assert torch.allclose(sa[k], sb[k], rtol=1e-4 if precision != "bf16" else 2e-3,
                                  atol=3e-4 if precision != "bf16" else 4.5e-4), k
        else:
            assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("C", [4, 5])
def test_pointnet_in_onepass_backward_equals_two_passes(C):
    """first PointNet layer, bf16 mode: statistics + G = dz^T.x from one read of the gradient, then
    dW = c0*G + c1*(W.x^T x) + c2*sum x -- against the two-pass form (statistics; then dy formed per element and
    contracted with the points) and against an fp64 evaluation of the same formulas."""
    P, cout = 64 * 30 * 64, 512
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(P, C, device=DEV, generator=g) * torch.tensor([1.0, 1.0, 0.5, 1.0, 10.0][:C], device=DEV)
    W = torch.randn(cout, C, device=DEV, generator=g) * 0.5
    bn = _bn(cout, 5)
    stats = ops.new_stats(cout, DEV)
    ops.pointnet_in_fwd(x, W, None, None, stats)
    scale, shift, mean, rstd = ops.bn_finalize(stats, P, None, bn, cout, update_running=False)
    da = (torch.randn(P, cout, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
    # two passes
    st = ops.pointnet_in_bwd_stats(da, x, W, scale, shift, mean, rstd)
    coef, dg, db = ops.bn_bwd_finalize(st, P, bn, mean, rstd, cout)
    dW2 = ops.pointnet_in_bwd_wgrad(da, x, W, scale, shift, coef)
    # one pass
    tail = ops.BnTailBwd(P, bn, mean, rstd, cout)
    dW1 = ops.pointnet_in_bwd_onepass(da, x, W, scale, shift, mean, rstd, tail)
    coef1, dg1, db1 = tail.out
    assert torch.allclose(coef1, coef, rtol=1e-5, atol=1e-7) and torch.allclose(dg1, dg, rtol=1e-5, atol=1e-5)
    assert torch.allclose(db1, db, rtol=1e-5, atol=1e-5)
    # fp64 reference of dy^T x
    xd, Wd = x.double(), W.double()
    y = xd @ Wd.t()
    z = y * scale.double() + shift.double()
    dz = da.double() * torch.where(z > 0, torch.ones_like(z), torch.exp(z))
    dy = coef[0].double() * dz + coef[1].double() * y + coef[2].double()
    ref = dy.t() @ xd
    den = ref.norm().item()
    e1, e2 = (dW1.double() - ref).norm().item() / den, (dW2.double() - ref).norm().item() / den
    print(f"C={C}: one-pass rel-l2 {e1:.2e}, two-pass rel-l2 {e2:.2e}")
    assert e1 <= 2e-3 and e2 <= 2e-3


@pytest.mark.parametrize("C", [4, 5])
def test_first_layer_coefficients_from_moments(C):
    """sum y = W.sum x and sum y^2 = W^T (x^T x) W: the first PointNet layer's train-mode BatchNorm coefficients and
    running statistics from the points' moments equal those of the statistics pass over [P, cout]"""
    P, cout = 64 * 30 * 128, 512
    g = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn(P, C, device=DEV, generator=g) * torch.tensor([1.0, 1.0, 0.5, 1.0, 10.0][:C], device=DEV) + 0.3
    W = torch.randn(cout, C, device=DEV, generator=g) * 0.5
    bias = torch.randn(cout, device=DEV, generator=g) * 0.1
    bn_a, bn_b = _bn(cout, 8), _bn(cout, 8)
    scale, shift, mean, rstd, mom = ops.pointnet_in_moment_coeffs(x, W, bias, bn_a)
    stats = ops.new_stats(cout, DEV)
    ops.pointnet_in_fwd(x, W, None, None, stats)
    ref = ops.bn_finalize(stats, P, bias, bn_b, cout)
    xd = x.double()
    assert torch.allclose(mom[:64].view(8, 8)[:C, :C], xd.t() @ xd, rtol=1e-6)
    assert torch.allclose(mom[64:64 + C], xd.sum(0), rtol=1e-6, atol=1e-3)
    for a, b, nm in zip((scale, shift, mean, rstd), ref, ("scale", "shift", "mean", "rstd")):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-6), (nm, (a - b).abs().max().item())
    assert torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn_a.num_batches_tracked) == 1
    # against fp64 torch
    y = xd @ W.double().t()
    assert torch.allclose(mean.double(), y.mean(0), rtol=1e-5, atol=1e-6)
    assert torch.allclose(rstd.double(), 1.0 / torch.sqrt(y.var(0, unbiased=False) + bn_a.eps), rtol=1e-5)


@pytest.mark.parametrize("C", [4, 5])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_first_layer_with_offcentre_features(C, dtype):
    """Round-3 advisor finding: features whose mean is far larger than their deviation (range in metres, power in dB).
    The moment form of the statistics (var = w^T (x^T x / P) w - (w.m)^2) and the one-pass weight gradient
    (dW = c0 G + c1 W.x^T x + c2 sum x) both cancel leading digits there.  Round 4: the moments are accumulated in fp64
    from exact products and G against the points centred on their mean; both must match an fp64 evaluation as well as
    the two-pass kernels do on the same inputs."""
    P, cout = 64 * 30 * 64, 512
    g = torch.Generator(device=DEV).manual_seed(11)
    offs = torch.tensor([40.0, -25.0, 3.0, 0.5, 900.0][:C], device=DEV)          # mean / deviation up to 3 000
    devs = torch.tensor([0.3, 0.2, 0.5, 1.0, 0.3][:C], device=DEV)
    x = torch.randn(P, C, device=DEV, generator=g) * devs + offs
    W = torch.randn(cout, C, device=DEV, generator=g) * 0.5
    bias = torch.randn(cout, device=DEV, generator=g) * 0.1
    bn = _bn(cout, 5)
    xd, Wd = x.double(), W.double()
    y = xd @ Wd.t()
    # forward coefficients from the moments against fp64
    scale, shift, mean, rstd, mom = ops.pointnet_in_moment_coeffs(x, W, bias, bn, update_running=False)
    var_ref = y.var(0, unbiased=False)
    rstd_ref = 1.0 / torch.sqrt(var_ref + bn.eps)
    e_mean = ((mean.double() - y.mean(0)).abs() / (y.std(0) + 1e-12)).max().item()     # (`mean` is that of the bias-free y)
    e_rstd = ((rstd.double() - rstd_ref).abs() / rstd_ref).max().item()
    print(f"C={C}: moment-form mean error {e_mean:.2e} of a deviation, rstd rel {e_rstd:.2e}")
    assert e_mean <= 2e-3 and e_rstd <= 1e-5        # (fp32 storage of `mean` itself: ~6e-8 |mean| / deviation)
    da = (torch.randn(P, cout, device=DEV, generator=g) * 0.1).to(dtype)
    tail = ops.BnTailBwd(P, bn, mean, rstd, cout)
    dW1 = ops.pointnet_in_bwd_onepass(da, x, W, scale, shift, mean, rstd, tail, mom=mom)
    coef, dg, db = tail.out
    # fp64 evaluation of the layer's backward from the same da, with the fp64 statistics
    z = (y - y.mean(0)) * rstd_ref * bn.weight.double() + bn.bias.double()
    dz = da.double() * torch.where(z > 0, torch.ones_like(z), torch.exp(z))
    yhat = (y - y.mean(0)) * rstd_ref
    dy = bn.weight.double() * rstd_ref * (dz - dz.mean(0) - yhat * (dz * yhat).mean(0))
    ref = dy.t() @ xd
    e1 = (dW1.double() - ref).norm().item() / ref.norm().item()
    # the two-pass form on the same inputs (dy rounded per element, contracted with the raw points)
    st = ops.pointnet_in_bwd_stats(da, x, W, scale, shift, mean, rstd)
    coef2, _, _ = ops.bn_bwd_finalize(st, P, bn, mean, rstd, cout)
    dW2 = ops.pointnet_in_bwd_wgrad(da, x, W, scale, shift, coef2)
    e2 = (dW2.double() - ref).norm().item() / ref.norm().item()
    print(f"C={C} {dtype}: off-centre one-pass rel-l2 {e1:.2e}, two-pass {e2:.2e}")
    assert e1 <= 2e-3, e1
