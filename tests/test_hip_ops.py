"""Op-level parity of the HIP kernels (through the C ABI) against fp64 CPU
arithmetic / the oracle.  GPU only."""
import numpy as np
import pytest
import torch

from opensetgaitrecognition_pcaa_amd import ops
from opensetgaitrecognition_pcaa_amd._lib import ACT_ELU, ACT_NONE, KC, PCAA_BF16, PCAA_F32, RC
from oracle import pcaa_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(shape, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def _mk(layout, rows, K, seed, dtype):
    """operand with logical shape (rows, K) stored per layout; returns (device tensor, logical fp64)."""
    logical = _rand((rows, K), seed)
    if dtype == torch.bfloat16:
        logical = logical.bfloat16().float()
    stored = logical if layout == KC else logical.t().contiguous()
    return stored.to(DEV).to(dtype).contiguous(), logical.double()


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 70, 45), (64, 15360 // 8, 960), (1000, 512, 512),
                                   (37, 5, 1125), (256, 4, 512), (300, 130, 33)])
@pytest.mark.parametrize("al,bl", [(KC, KC), (KC, RC), (RC, KC), (RC, RC)])
def test_gemm_f32_layouts(M, N, K, al, bl):
    A, Ad = _mk(al, M, K, 1, torch.float32)
    B, Bd = _mk(bl, N, K, 2, torch.float32)
    bias = _rand((N,), 3).to(DEV)
    C = ops.gemm(A, al, B, bl, M, N, K, bias=bias)
    ref = Ad @ Bd.t() + bias.cpu().double()
    err = (C.cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * K ** 0.5 * ref.abs().max().item() + 1e-6, err


@pytest.mark.parametrize("M,N,K,sk", [(64, 960, 7680, 8), (512, 512, 24576, 16), (130, 70, 4000, 5)])
def test_gemm_split_k_atomic(M, N, K, sk):
    A, Ad = _mk(RC, M, K, 4, torch.float32)
    B, Bd = _mk(RC, N, K, 5, torch.float32)
    C = ops.gemm(A, RC, B, RC, M, N, K, split_k=sk, accumulate=True)
    ref = Ad @ Bd.t()
    assert (C.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() * 4
    # accumulate on top of an existing tensor, with bias
    C0 = _rand((M, N), 6).to(DEV)
    bias = _rand((N,), 7).to(DEV)
    C1 = ops.gemm(A, RC, B, RC, M, N, K, out=C0.clone(), bias=bias, split_k=sk, accumulate=True)
    ref1 = ref + C0.cpu().double() + bias.cpu().double()
    assert (C1.cpu().double() - ref1).abs().max().item() <= 1e-5 * ref1.abs().max().item() * 4


@pytest.mark.parametrize("M,N,K", [(3840, 512, 512), (1000, 1024, 64), (300, 96, 40)])
def test_gemm_colstats(M, N, K):
    A, Ad = _mk(KC, M, K, 8, torch.float32)
    B, Bd = _mk(KC, N, K, 9, torch.float32)
    bias = _rand((N,), 10).to(DEV)
    stats = ops.new_stats(N, DEV)
    C = ops.gemm(A, KC, B, KC, M, N, K, bias=bias, colstats=stats)
    acc = Ad @ Bd.t()
    s = stats.sum(0).cpu()
    # fp32 accumulators carry ~1e-7 * sqrt(K) relative rounding each; sums over M rows random-walk
    scale = acc.abs().max().item()
    assert (s[0] - acc.sum(0)).abs().max().item() <= 2e-6 * scale * M ** 0.5 + 1e-4
    assert torch.allclose(s[1], (acc * acc).sum(0), rtol=2e-6, atol=1e-3)
    assert (C.cpu().double() - (acc + bias.cpu().double())).abs().max().item() < 1e-5 * scale


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (1000, 512, 512), (3840, 1024, 1024), (130, 72, 520)])
@pytest.mark.parametrize("bdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_gemm_bf16(M, N, K, bdt, cdt):
    A, Ad = _mk(KC, M, K, 11, torch.bfloat16)
    B, Bd = _mk(KC, N, K, 12, bdt)
    Bd = Bd.float().bfloat16().double()     # fp32 B operands are rounded to bf16 when staged
    bias = _rand((N,), 13).to(DEV)
    stats = ops.new_stats(N, DEV)
    C = ops.gemm(A, KC, B, KC, M, N, K, bias=bias, colstats=stats, out_dtype=cdt, math=PCAA_BF16)
    acc = Ad @ Bd.t()
    ref = acc + bias.cpu().double()
    tol = 1e-5 if cdt == torch.float32 else 1e-2
    assert (C.cpu().double() - ref).abs().max().item() <= tol * ref.abs().max().item()
    s = stats.sum(0).cpu()
    assert torch.allclose(s[0], acc.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(s[1], (acc * acc).sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("M,N,K,sk", [(512, 1024, 4000, 4), (1024, 1024, 30 * 128 * 3, 16), (256, 128, 520, 1),
                                      (520, 264, 1000, 3)])
def test_gemm_bf16_wgrad_rc_rc(M, N, K, sk):
    """dW = dy^T . a : both operands row-contiguous bf16 (transpose-read LDS image), split-K atomics.
    Exact-integer operands make a swapped fragment map impossible to miss."""
    rng = np.random.default_rng(33)
    Ai = torch.from_numpy(rng.integers(-3, 4, (M, K)).astype(np.float32))
    Bi = torch.from_numpy(rng.integers(-3, 4, (N, K)).astype(np.float32))
    A = Ai.t().contiguous().to(DEV).bfloat16()
    B = Bi.t().contiguous().to(DEV).bfloat16()
    C = ops.gemm(A, RC, B, RC, M, N, K, split_k=sk, accumulate=sk > 1, math=PCAA_BF16)
    ref = Ai.double() @ Bi.double().t()
    assert torch.equal(C.cpu().double(), ref)
    # random bf16 data
    A2, A2d = _mk(RC, M, K, 34, torch.bfloat16)
    B2, B2d = _mk(RC, N, K, 35, torch.bfloat16)
    C2 = ops.gemm(A2, RC, B2, RC, M, N, K, split_k=sk, accumulate=sk > 1, math=PCAA_BF16)
    ref2 = A2d @ B2d.t()
    assert (C2.cpu().double() - ref2).abs().max().item() <= 2e-5 * ref2.abs().max().item()


def test_gemm_bf16_big_tile_exact_integers():
    """asymmetric exact-integer check of the 256x256-tile KC x KC kernel (catches row/col swaps)."""
    M, N, K = 700, 384, 200
    rng = np.random.default_rng(36)
    Ai = torch.from_numpy(rng.integers(-4, 5, (M, K)).astype(np.float32))
    Bi = torch.from_numpy(rng.integers(-4, 5, (N, K)).astype(np.float32))
    C = ops.gemm(Ai.to(DEV).bfloat16(), KC, Bi.to(DEV), KC, M, N, K, math=PCAA_BF16)
    assert torch.equal(C.cpu().double(), Ai.double() @ Bi.double().t())


@pytest.mark.parametrize("lay,M,N,K,sk", [(KC, 512, 256, 128, 1), (KC, 768, 512, 1024, 1), (RC, 256, 512, 640, 2),
                                          (RC, 1024, 512, 64 * 40, 5), (KC, 256, 256, 64, 1)])
def test_gemm_bf16_lds_dma_exact_integers(lay, M, N, K, sk):
    """whole-tile bf16 x bf16 shapes take the LDS-DMA kernel (swizzled source addresses, swizzled
    fragment reads): exact small-integer operands must reproduce the integer product exactly."""
    rng = np.random.default_rng(37)
    Ai = torch.from_numpy(rng.integers(-3, 4, (M, K)).astype(np.float32))
    Bi = torch.from_numpy(rng.integers(-3, 4, (N, K)).astype(np.float32))
    A = (Ai if lay == KC else Ai.t().contiguous()).to(DEV).bfloat16()
    B = (Bi if lay == KC else Bi.t().contiguous()).to(DEV).bfloat16()
    bias = torch.from_numpy(rng.integers(-2, 3, (N,)).astype(np.float32)).to(DEV)
    stats = ops.new_stats(N, DEV) if sk == 1 else None
    C = ops.gemm(A, lay, B, lay, M, N, K, bias=bias, colstats=stats, split_k=sk, accumulate=sk > 1, math=PCAA_BF16)
    acc = Ai.double() @ Bi.double().t()
    assert torch.equal(C.cpu().double(), acc + bias.cpu().double())
    if stats is not None:
        s = stats.sum(0).cpu()
        assert torch.equal(s[0], acc.sum(0))
        assert torch.allclose(s[1], (acc * acc).sum(0), rtol=1e-6)     # squares exceed fp32's exact-integer range
        Cb = ops.gemm(A, lay, B, lay, M, N, K, out_dtype=torch.bfloat16, math=PCAA_BF16)
        assert torch.equal(Cb.float().cpu().double(), acc.float().bfloat16().double())


@pytest.mark.parametrize("M,N,K", [(64, 1920, 960), (64, 960, 1920), (37, 320, 704), (8, 128, 128), (1, 192, 256), (64, 960, 64),
                                   (64, 15360 // 2, 7680 // 2)])
def test_skinny_linear_layers_exact_integers(M, N, K):
    """batch-skinny (M <= 64) decoder kernels: weights streamed HBM -> MFMA fragments.  Small
    integers are exact in bf16, so forward / dgrad / wgrad must equal the integer products
    (asymmetric shapes, ragged M, N not a multiple of the 128-row workgroup tile)."""
    assert ops.skinny_supported(M, N, K)
    rng = np.random.default_rng(39)
    x = torch.from_numpy(rng.integers(-3, 4, (M, K)).astype(np.float32))
    W = torch.from_numpy(rng.integers(-3, 4, (N, K)).astype(np.float32))
    b = torch.from_numpy(rng.integers(-2, 3, (N,)).astype(np.float32))
    dz = torch.from_numpy(rng.integers(-3, 4, (M, N)).astype(np.float32))
    xd, Wd, bd, dzd = x.to(DEV), W.to(DEV), b.to(DEV), dz.to(DEV)
    ref = x.double() @ W.double().t() + b.double()
    y = ops.skinny_linear_fwd(xd, Wd, bd, ACT_NONE)
    assert torch.equal(y.cpu().double(), ref)
    ye = ops.skinny_linear_fwd(xd, Wd, bd, ACT_ELU)
    assert torch.allclose(ye.cpu().double(), torch.nn.functional.elu(ref), rtol=1e-6, atol=1e-7)
    # dgrad: plain, accumulate, and with the ELU'(a_prev) factor of the layer below
    dref = dz.double() @ W.double()
    dx = ops.skinny_linear_dgrad(dzd, Wd)
    assert torch.equal(dx.cpu().double(), dref)
    init = torch.from_numpy(rng.integers(-5, 6, (M, K)).astype(np.float32))
    dx2 = ops.skinny_linear_dgrad(dzd, Wd, out=init.to(DEV), accumulate=True)
    assert torch.equal(dx2.cpu().double(), dref + init.double())
    a_prev = torch.from_numpy(rng.uniform(-0.9, 2.0, (M, K)).astype(np.float32))
    dx3 = ops.skinny_linear_dgrad(dzd, Wd, a_prev=a_prev.to(DEV))
    fac = torch.where(a_prev > 0, torch.ones_like(a_prev), a_prev + 1).double()
    assert torch.allclose(dx3.cpu().double(), dref * fac, rtol=1e-6, atol=1e-6)
    # wgrad
    dW = ops.skinny_linear_wgrad(dzd, xd)
    assert torch.equal(dW.cpu().double(), dz.double().t() @ x.double())


def test_skinny_linear_layers_random_bf16_rounding():
    """random fp32 operands: result equals the fp64 product of the bf16-ROUNDED operands to fp32
    accumulation accuracy (the only approximation is the operand rounding)."""
    M, N, K = 64, 3840, 1920
    rng = np.random.default_rng(40)
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32))
    W = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    dz = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
    r = lambda t: t.bfloat16().double()
    y = ops.skinny_linear_fwd(x.to(DEV), W.to(DEV), None, ACT_NONE).cpu().double()
    ref = r(x) @ r(W).t()
    assert (y - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    dx = ops.skinny_linear_dgrad(dz.to(DEV), W.to(DEV)).cpu().double()
    ref = r(dz) @ r(W)
    assert (dx - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    dW = ops.skinny_linear_wgrad(dz.to(DEV), x.to(DEV)).cpu().double()
    ref = r(dz).t() @ r(x)
    assert (dW - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(64, 3840, 1920), (37, 1920, 960), (5, 256, 64)])
def test_skinny_linear_layers_exact_variants(M, N, K):
    """the fp32-product variants (the parity modes' decoder): fp32 accuracy against the fp64 product of the UNROUNDED
    operands, in all three passes, with the fused bias + ELU / ELU' reductions"""
    rng = np.random.default_rng(44)
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32))
    W = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
    dz = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
    a_prev = torch.from_numpy(rng.uniform(-0.9, 2.0, (M, K)).astype(np.float32))
    y = ops.skinny_linear_fwd(x.to(DEV), W.to(DEV), b.to(DEV), ACT_ELU, exact=True).cpu().double()
    z = x.double() @ W.double().t() + b.double()
    ref = torch.where(z > 0, z, torch.expm1(z))
    assert (y - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    dx = ops.skinny_linear_dgrad(dz.to(DEV), W.to(DEV), a_prev=a_prev.to(DEV), exact=True).cpu().double()
    ref = (dz.double() @ W.double()) * torch.where(a_prev > 0, torch.ones_like(a_prev), a_prev + 1).double()
    assert (dx - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    dW = ops.skinny_linear_wgrad(dz.to(DEV), x.to(DEV), exact=True).cpu().double()
    ref = dz.double().t() @ x.double()
    assert (dW - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(512, 256, 320), (768, 512, 1024), (1024, 1024, 512), (1320, 512, 512), (200, 256, 1024)])
def test_gemm_dgrad_bn_fused_epilogue(M, N, K):
    """dgrad fused with ELU' and the BatchNorm-backward statistics of the layer below: must agree with
    the separate chain  da = dy.Wt^T (bf16) ; stats = bn_act_bwd_stats(y, da) ; dz = da*ELU'(z)."""
    assert ops.gemm_dgrad_bn_supported(M, N, K)
    rng = np.random.default_rng(41)
    dy = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(DEV).bfloat16()
    Wt = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(DEV).bfloat16()
    # (y and dy sit in front of NaN-filled memory: a partial last row tile must neither read the rows past M into its
    # statistics -- round 4: a wave whose 128 rows all lay past M read y there and 0 * ELU'(NaN) poisoned the column sums,
    # a NaN step or a memory access fault depending on what the allocator had left behind -- nor write them)
    ybig = torch.full((M + 300, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ybig[:M] = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(DEV).bfloat16()
    y = ybig[:M]
    dybig = torch.full((M + 300, K), float("nan"), dtype=torch.bfloat16, device=DEV)
    dybig[:M] = dy
    dy = dybig[:M]
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, N).astype(np.float32)).to(DEV)
    shift = torch.from_numpy(rng.uniform(-0.5, 0.5, N).astype(np.float32)).to(DEV)
    mean = torch.from_numpy(rng.uniform(-0.2, 0.2, N).astype(np.float32)).to(DEV)
    rstd = torch.from_numpy(rng.uniform(0.5, 2.0, N).astype(np.float32)).to(DEV)
    dz, st = ops.gemm_dgrad_bn(dy, Wt, y, scale, shift, mean, rstd)
    assert bool(torch.isfinite(dz.float()).all()) and bool(torch.isfinite(st).all())
    da = ops.gemm(dy, KC, Wt, KC, M, N, K, out_dtype=torch.bfloat16, math=PCAA_BF16)
    dz_ref, st_ref = ops.bn_act_bwd_dz(y, scale, shift, mean, rstd, da=da)
    assert (dz.float() - dz_ref.float()).abs().max().item() <= 1e-2 * max(1.0, dz_ref.float().abs().max().item())
    sa, sb = st.sum(0).cpu(), st_ref.sum(0).cpu()
    # (the separate chain rounds da to bf16 before the statistics; the fused epilogue forms them from the fp32
    # accumulators -- since round 3 they no longer pass through a bf16 LDS image -- so the two differ by that rounding)
    assert (sa - sb).abs().max().item() <= 5e-3 * max(1.0, sb.abs().max().item())
    # exact fp64 check of the statistics against the dz the kernel itself wrote is not possible (they are
    # accumulated before the bf16 rounding); against fp64 of the unrounded product instead
    da64 = dy.float().cpu().double() @ Wt.float().cpu().double().t()
    z = y.float().cpu().double() * scale.cpu().double() + shift.cpu().double()
    dz64 = da64 * torch.where(z > 0, torch.ones_like(z), torch.exp(z))
    yh = (y.float().cpu().double() - mean.cpu().double()) * rstd.cpu().double()
    assert (sa[0] - dz64.sum(0)).abs().max().item() <= 5e-3 * max(1.0, dz64.sum(0).abs().max().item())
    assert (sa[1] - (dz64 * yh).sum(0)).abs().max().item() <= 5e-3 * max(1.0, (dz64 * yh).sum(0).abs().max().item())


def test_skinny_rejects_unsupported_shapes():
    assert not ops.skinny_supported(65, 1920, 960)
    assert not ops.skinny_supported(64, 1200, 960)
    assert not ops.skinny_supported(64, 960, 32)
    with pytest.raises(RuntimeError):
        ops.skinny_linear_fwd(torch.zeros(65, 128, device=DEV), torch.zeros(128, 128, device=DEV), None, ACT_NONE)


@pytest.mark.parametrize("M,N,K,sk,math", [(1024, 512, 64 * 40, 5, PCAA_BF16), (256, 256, 64 * 6, 3, PCAA_BF16),
                                           (520, 264, 1000, 3, PCAA_BF16), (130, 68, 900, 4, PCAA_F32)])
def test_gemm_slab_split_k(M, N, K, sk, math):
    """split-K through per-split slabs + reduce (no atomics): exact on integer data, for the
    LDS-DMA kernel (whole tiles), the register-staged bf16 kernel (ragged) and the fp32 kernel."""
    rng = np.random.default_rng(38)
    Ai = torch.from_numpy(rng.integers(-3, 4, (M, K)).astype(np.float32))
    Bi = torch.from_numpy(rng.integers(-3, 4, (N, K)).astype(np.float32))
    dt = torch.bfloat16 if math == PCAA_BF16 else torch.float32
    A = Ai.t().contiguous().to(DEV).to(dt)
    B = Bi.t().contiguous().to(DEV).to(dt)
    ref = Ai.double() @ Bi.double().t()
    C = ops.gemm_slabs(A, RC, B, RC, M, N, K, sk, math=math)
    assert torch.equal(C.cpu().double(), ref)
    C0 = torch.from_numpy(rng.integers(-5, 6, (M, N)).astype(np.float32)).to(DEV)
    C1 = ops.gemm_slabs(A, RC, B, RC, M, N, K, sk, out=C0.clone(), accumulate=True, math=math)
    assert torch.equal(C1.cpu().double(), ref + C0.cpu().double())


@pytest.mark.parametrize("al,bl", [(KC, KC), (RC, RC), (KC, RC)])
def test_gemm_f32_math_bf16_storage(al, bl):
    M, N, K = 520, 260, 1000
    A, Ad = _mk(al, M, K, 14, torch.bfloat16)
    B, Bd = _mk(bl, N, K, 15, torch.bfloat16)
    C = ops.gemm(A, al, B, bl, M, N, K)
    ref = Ad @ Bd.t()
    assert (C.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    B2, B2d = _mk(bl, N, K, 16, torch.float32)
    C2 = ops.gemm(A, al, B2, bl, M, N, K)
    ref2 = Ad @ B2d.t()
    assert (C2.cpu().double() - ref2).abs().max().item() <= 1e-5 * ref2.abs().max().item()


class _BN:
    def __init__(self, ch, seed):
        self.weight = (1 + 0.1 * _rand((ch,), seed)).to(DEV)
        self.bias = (0.1 * _rand((ch,), seed + 1)).to(DEV)
        self.running_mean = (0.1 * _rand((ch,), seed + 2)).to(DEV)
        self.running_var = (1 + 0.1 * _rand((ch,), seed + 3).abs()).to(DEV)
        self.num_batches_tracked = torch.zeros((), dtype=torch.int64, device=DEV)
        self.momentum, self.eps = 0.1, 1e-5


@pytest.mark.parametrize("rows,ch,dtype", [(3840, 1024, torch.float32), (1000, 512, torch.float32),
                                           (180, 16, torch.float32), (3840, 512, torch.bfloat16)])
def test_bn_forward_backward_chain(rows, ch, dtype):
    """gemm(+stats) -> finalize -> bn_act_fwd ; bwd_dz -> finalize -> dy  vs fp64 autograd."""
    K = 64
    A, Ad = _mk(KC, rows, K, 20, torch.float32)
    W, Wd = _mk(KC, ch, K, 21, torch.float32)
    lin_b = (0.3 * _rand((ch,), 22)).to(DEV)
    bn = _BN(ch, 23)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    stats = ops.new_stats(ch, DEV)
    # the stored pre-BN tensor is the bias-free accumulator (the bias cancels in BatchNorm)
    y = ops.gemm(A, KC, W, KC, rows, ch, K, colstats=stats, out_dtype=dtype)
    scale, shift, mean, rstd = ops.bn_finalize(stats, rows, lin_b, bn, ch)
    a = ops.bn_act_fwd(y, scale, shift)
    # fp64 reference with autograd
    yd = (Ad @ Wd.t() + lin_b.cpu().double()).requires_grad_(True)
    mu = yd.mean(0)
    var = ((yd - mu) ** 2).mean(0)
    zd = (yd - mu) / torch.sqrt(var + 1e-5) * bn.weight.cpu().double() + bn.bias.cpu().double()
    ad = torch.where(zd > 0, zd, torch.expm1(zd))
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert torch.allclose(mean.cpu().double() + lin_b.cpu().double(), mu.detach(), rtol=1e-5, atol=1e-5)
    assert (a.float().cpu().double() - ad.detach()).abs().max().item() <= tol * max(1.0, ad.abs().max().item())
    assert torch.allclose(bn.running_mean.cpu().double(), 0.9 * rm0.cpu().double() + 0.1 * mu.detach(), atol=1e-5)
    assert torch.allclose(bn.running_var.cpu().double(),
                          0.9 * rv0.cpu().double() + 0.1 * var.detach() * rows / (rows - 1), rtol=1e-4, atol=1e-5)
    assert int(bn.num_batches_tracked.item()) == 1
    # backward
    g = _rand((rows, ch), 24)
    (ad * g.double()).sum().backward()
    da = g.to(DEV).to(dtype)
    dz, st2 = ops.bn_act_bwd_dz(y, scale, shift, mean, rstd, da=da)
    coef, dgamma, dbeta = ops.bn_bwd_finalize(st2, rows, bn, mean, rstd, ch)
    dy = ops.bn_bwd_dy(dz, y, coef)
    ref = yd.grad
    gtol = 1e-4 if dtype == torch.float32 else 5e-2
    assert (dy.float().cpu().double() - ref).abs().max().item() <= gtol * ref.abs().max().item()
    # pooled variants
    G = rows // 30
    pooled = ops.bn_act_meanpool_fwd(y[: G * 30].contiguous(), scale, shift, G, 30)
    refp = ad.detach()[: G * 30].view(G, 30, ch).mean(1)
    assert (pooled.cpu().double() - refp).abs().max().item() <= tol * max(1.0, refp.abs().max().item())
    # training variant: same pooled values + the per-group sums from which the backward statistics of
    # a pooled layer follow without re-reading y: must equal the statistics pass over (dpool, y)
    yp = y[: G * 30].contiguous()
    pooled2, e = ops.bn_act_meanpool_fwd(yp, scale, shift, G, 30, mean, rstd)
    assert torch.equal(pooled2, pooled)
    dpool = _rand((G, ch), 25).to(DEV)
    st_a = ops.bn_act_bwd_stats(yp, scale, shift, mean, rstd, dpool=dpool, group_rows=30, pool_scale=1.0 / 30)
    st_b = ops.bn_pool_bwd_stats(dpool, e, 1.0 / 30)
    sa, sb = st_a.sum(0).cpu(), st_b.sum(0).cpu()
    stol = 2e-6 if dtype == torch.float32 else 2e-3
    assert (sa - sb).abs().max().item() <= stol * max(1.0, sa.abs().max().item())


def test_bias_act_elu_colsum_sum():
    x = _rand((70, 333), 30).to(DEV)
    b = _rand((333,), 31).to(DEV)
    y = ops.bias_act_(x.clone(), b, ACT_ELU)
    ref = O.elu(x.cpu() + b.cpu())
    assert torch.allclose(y.cpu(), ref, atol=1e-6)
    da = _rand((70, 333), 32).to(DEV)
    dz = ops.elu_bwd_from_out(da, y)
    z = x.cpu() + b.cpu()
    refd = da.cpu() * torch.where(z > 0, torch.ones_like(z), torch.exp(z))
    assert torch.allclose(dz.cpu(), refd, atol=1e-5)
    assert torch.allclose(ops.colsum(x).cpu(), x.cpu().sum(0), atol=1e-4)
    assert abs(ops.total(x, 0.5).item() - 0.5 * x.cpu().double().sum().item()) < 1e-3
    assert torch.allclose(ops.rowsum(x, 2.0).cpu(), 2 * x.cpu().sum(1), atol=1e-4)


@pytest.mark.parametrize("d", [1, 2, 4])
def test_dtc_im2col_col2im_adjoint(d):
    B, T, Cin = 3, 30, 16
    a = _rand((B * T, Cin), 40).to(DEV)
    col = ops.dtc_im2col(a, B, T, Cin, d)
    a3 = a.cpu().view(B, T, Cin)
    ref = torch.zeros(B, T, Cin, 3)
    for tap in range(3):
        sh = (2 - tap) * d
        if sh < T:
            ref[:, sh:, :, tap] = a3[:, : T - sh]
    assert torch.equal(col.cpu(), ref.view(B * T, Cin * 3))
    # adjoint: <im2col(a), g> == <a, col2im(g)>
    g = _rand((B * T, Cin * 3), 41).to(DEV)
    back = ops.dtc_col2im(g, B, T, Cin, d)
    lhs = (col.cpu().double() * g.cpu().double()).sum()
    rhs = (a.cpu().double() * back.cpu().double()).sum()
    assert abs(lhs - rhs) < 1e-6 * abs(lhs) + 1e-6


def test_pack_points_and_prior():
    x = _rand((3, 5, 30, 17), 50).to(DEV)
    assert torch.equal(ops.pack_points(x).cpu(), x.cpu().permute(0, 2, 3, 1).contiguous())
    z0 = _rand((6, 32), 51).to(DEV)
    means = _rand((4, 32), 52).to(DEV)
    gt = torch.tensor([0, 3, 1, 1, 2, 0], device=DEV)
    z, oh = ops.prior_sample(z0, means, gt, 4)
    assert torch.allclose(z.cpu(), z0.cpu() + means.cpu()[gt.cpu()])
    assert torch.equal(oh.cpu(), torch.nn.functional.one_hot(gt.cpu(), 4).float())


def test_adam_matches_reference_formula():
    n = 1003
    p = _rand((n + 1,), 60)[:n].clone()
    params = {"p": p.clone()}
    state = {}
    pd, m, v = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    # 16-B aligned storage
    for s in range(1, 4):
        g = _rand((n,), 60 + s)
        O.adam_step(params, {"p": g}, state, 1e-4, 0.9, 0.99)
        ops.adam_step_(pd, g.to(DEV), m, v, 1e-4, 0.9, 0.99, 1e-8, s)
        assert torch.allclose(pd.cpu(), params["p"], rtol=1e-6, atol=1e-7), s


@pytest.mark.parametrize("M,N,K", [(64, 256, 128), (37, 192, 160), (5, 128, 1024), (64, 960, 1920), (16, 4544, 2304)])
def test_skinny_wgrad_adam_equals_wgrad_then_adam_bitwise(M, N, K):
    """pcaa_skinny_linear_wgrad_adam == pcaa_skinny_linear_wgrad followed by pcaa_adam_step_dev, bit for bit
    (parameters and both moments), over three optimizer steps with changing gradients."""
    from opensetgaitrecognition_pcaa_amd.train import StepCount
    W0 = _rand((N, K), 300, 0.05).to(DEV)
    Wa, Wb = W0.clone(), W0.clone()
    ma, va, mb, vb = (torch.zeros_like(W0) for _ in range(4))
    ca, cb = StepCount(DEV), StepCount(DEV)
    for s in range(3):
        dz = _rand((M, N), 301 + 2 * s, 0.3).to(DEV)
        x = _rand((M, K), 302 + 2 * s, 1.0).to(DEV)
        ca.advance(1e-3, 0.9, 0.99)
        cb.advance(1e-3, 0.9, 0.99)
        dW = ops.skinny_linear_wgrad(dz, x)
        ops.adam_step_dev_(Wa, dW, ma, va, 0.9, 0.99, 1e-8, ca.coef_dev)
        ops.skinny_linear_wgrad_adam_(dz, x, Wb, mb, vb, 0.9, 0.99, 1e-8, cb.coef_dev)
        assert torch.equal(Wa, Wb) and torch.equal(ma, mb) and torch.equal(va, vb), s
    assert not torch.equal(Wa, W0)
    # the fp32-product variants (parity modes) hold the same identity
    Wa, Wb = W0.clone(), W0.clone()
    ma, va, mb, vb = (torch.zeros_like(W0) for _ in range(4))
    ca, cb = StepCount(DEV), StepCount(DEV)
    for s in range(2):
        dz = _rand((M, N), 321 + 2 * s, 0.3).to(DEV)
        x = _rand((M, K), 322 + 2 * s, 1.0).to(DEV)
        ca.advance(1e-3, 0.9, 0.99)
        cb.advance(1e-3, 0.9, 0.99)
        dW = ops.skinny_linear_wgrad(dz, x, exact=True)
        ops.adam_step_dev_(Wa, dW, ma, va, 0.9, 0.99, 1e-8, ca.coef_dev)
        ops.skinny_linear_wgrad_adam_(dz, x, Wb, mb, vb, 0.9, 0.99, 1e-8, cb.coef_dev, exact=True)
        assert torch.equal(Wa, Wb) and torch.equal(ma, mb) and torch.equal(va, vb), ("exact", s)


@pytest.mark.parametrize("M,N,K", [(64, 256, 128), (37, 192, 160), (5, 128, 1024), (128, 256, 128), (6, 512, 256), (192, 960, 1920),
                                   (256, 192, 160), (300, 128, 96), (512, 384, 512), (512, 1920, 960)])
def test_skinny_wgrad_adam_rows_gathered_update(M, N, K):
    """Round 5 (dp_gather): the fused weight-gradient + Adam update from the ranks' stacked rows, M = world * B up to
    512.  M <= 64: the bits of the single-process kernel.  Beyond: the gradient it forms is the bf16-operand product with
    fp32 accumulation over ALL rows -- checked through the Adam state it leaves: exp_avg = (1 - beta1) * grad after one step
    from zero moments, against an fp64 product of the rounded operands (1e-5 of the largest entry); the buffers' rows
    behind M (zeros, as the trainer allocates them) and a NaN-free W outside [N, K] are left alone."""
    from opensetgaitrecognition_pcaa_amd.train import StepCount
    R = ops.gathered_rows_alloc(M)
    W0 = _rand((N, K), 400, 0.05).to(DEV)
    dzb = torch.zeros((R, N), device=DEV)
    xb = torch.zeros((R, K), device=DEV)
    dzb[:M] = _rand((M, N), 401, 0.3).to(DEV)
    xb[:M] = _rand((M, K), 402, 1.0).to(DEV)
    if M < R:
        xb[M:] = 7.0                      # finite garbage behind the valid rows: must meet zero dz fragments
        dzb[M:] = 3.0                     # ... and dz rows behind M are masked by the kernel itself
    W, m, v = W0.clone(), torch.zeros_like(W0), torch.zeros_like(W0)
    c = StepCount(DEV)
    c.advance(1e-3, 0.9, 0.99)
    ops.skinny_linear_wgrad_adam_rows_(dzb, xb, M, W, m, v, 0.9, 0.99, 1e-8, c.coef_dev, grad_scale=0.5)
    torch.cuda.synchronize()
    g64 = 0.5 * (dzb[:M].bfloat16().double().t() @ xb[:M].bfloat16().double())
    got = m.double() / 0.1                # exp_avg after the first step = (1 - beta1) * grad
    assert (got - g64).abs().max().item() <= 1e-5 * g64.abs().max().item() + 1e-7
    assert torch.isfinite(W).all() and not torch.equal(W, W0)
    if M <= 64:
        Wb, mb, vb = W0.clone(), torch.zeros_like(W0), torch.zeros_like(W0)
        cb = StepCount(DEV)
        cb.advance(1e-3, 0.9, 0.99)
        ops.skinny_linear_wgrad_adam_(dzb[:M], xb[:M], Wb, mb, vb, 0.9, 0.99, 1e-8, cb.coef_dev, grad_scale=0.5)
        assert torch.equal(W, Wb) and torch.equal(m, mb) and torch.equal(v, vb), "M <= 64: the single-process kernel's bits"
    else:
        # the rows in 64-row chunks through the two-kernel path, summed: the same gradient up to fp32 summation order
        dW = sum(ops.skinny_linear_wgrad(dzb[r:min(r + 64, M)], xb[r:min(r + 64, M)]) for r in range(0, M, 64))
        Wc, mc, vc = W0.clone(), torch.zeros_like(W0), torch.zeros_like(W0)
        cc = StepCount(DEV)
        cc.advance(1e-3, 0.9, 0.99)
        ops.adam_step_dev_(Wc, dW, mc, vc, 0.9, 0.99, 1e-8, cc.coef_dev, 0.5)
        assert (m - mc).abs().max().item() <= 2e-6 * mc.abs().max().item()
        assert (W - Wc).abs().max().item() <= 2.1e-3          # one Adam step of lr 1e-3: sign flips where the gradient is rounding noise
        assert (W - Wc).abs().mean().item() <= 1e-6
    with pytest.raises(ValueError):
        ops.skinny_linear_wgrad_adam_rows_(dzb[:8], xb[:8], 600, W, m, v, 0.9, 0.99, 1e-8, c.coef_dev)


@pytest.mark.parametrize("rows,N,K", [(64, 256, 128), (37, 192, 160), (5, 130, 96), (64, 1920, 960)])
def test_pack_rows_t16_is_the_transposed_rounded_operand_pair(rows, N, K):
    """Round 6: pcaa_pack_rows_t16 -- column c of dz, then of x, as one 128-B row of 64 bf16 batch rows (nearest even),
    zeros behind ``rows``; strided sources (leading dimension > width) are honoured."""
    dzf = _rand((rows, N + 8), 810, 0.3).to(DEV)
    xf = _rand((rows, K + 4), 811, 1.0).to(DEV)
    dz, x = dzf[:, :N], xf[:, :K]
    assert dz.stride(0) == N + 8
    from opensetgaitrecognition_pcaa_amd import _lib
    out = torch.full((ops.packed_chunk_elems(N, K),), float("nan"), dtype=torch.bfloat16, device=DEV)
    _lib.check(_lib.load().pcaa_pack_rows_t16(dz.data_ptr(), dz.stride(0), N, x.data_ptr(), x.stride(0), K, rows,
                                              out.data_ptr(), torch.cuda.current_stream().cuda_stream), "pack")
    want = torch.zeros((N + K, 64), dtype=torch.bfloat16, device=DEV)
    want[:N, :rows] = dz.t().bfloat16()
    want[N:, :rows] = x.t().bfloat16()
    assert torch.equal(out.view(N + K, 64), want)
    assert torch.equal(ops.pack_rows_t16(dz.contiguous(), x.contiguous()), out)
    with pytest.raises(ValueError):
        ops.pack_rows_t16(torch.zeros((65, N), device=DEV), torch.zeros((65, K), device=DEV))


@pytest.mark.parametrize("chunks,rows,N,K", [(1, 64, 256, 128), (1, 37, 192, 160), (2, 64, 256, 128), (3, 16, 130, 96),
                                             (4, 64, 960, 1920), (5, 64, 384, 512), (8, 64, 1920, 960), (8, 40, 128, 1024),
                                             (8, 64, 3840, 1920)])
def test_skinny_wgrad_adam_t16_packed_gathered_update(chunks, rows, N, K):
    """Round 6 (the data-parallel decoder update from PACKED gathered operands): W <- Adam(W, s * sum_c dz_c^T x_c) over up
    to 8 chunks of <= 64 rows.  The gradient it forms is the bf16-operand product with fp32 accumulation over all chunks --
    checked through exp_avg = (1 - beta1) * grad after one step from zero moments against an fp64 product of the rounded
    operands (1e-5 of the largest entry), and against the rows kernel of round 5 on the same stacked rows (the same
    products in another summation order); chunks behind ``chunks`` in the buffer are not read."""
    from opensetgaitrecognition_pcaa_amd.train import StepCount
    W0 = _rand((N, K), 900, 0.05).to(DEV)
    ce = ops.packed_chunk_elems(N, K)
    packed = torch.full((chunks + 1, ce + 64), float("nan"), dtype=torch.bfloat16, device=DEV)[:, :ce]     # a strided buffer
    dzs = [_rand((rows, N), 901 + 2 * c, 0.3).to(DEV) for c in range(chunks)]
    xs = [_rand((rows, K), 902 + 2 * c, 1.0).to(DEV) for c in range(chunks)]
    for c in range(chunks):
        packed[c] = ops.pack_rows_t16(dzs[c], xs[c])
    W, m, v = W0.clone(), torch.zeros_like(W0), torch.zeros_like(W0)
    cnt = StepCount(DEV)
    cnt.advance(1e-3, 0.9, 0.99)
    ops.skinny_linear_wgrad_adam_t16_(packed, chunks, W, m, v, 0.9, 0.99, 1e-8, cnt.coef_dev, grad_scale=0.5)
    torch.cuda.synchronize()
    g64 = 0.5 * sum(dzs[c].bfloat16().double().t() @ xs[c].bfloat16().double() for c in range(chunks))
    got = m.double() / 0.1
    assert (got - g64).abs().max().item() <= 1e-5 * g64.abs().max().item() + 1e-7
    assert torch.isfinite(W).all() and not torch.equal(W, W0)
    # the rows kernel on the same rows stacked (rows < 64: zero rows in between change nothing)
    M = chunks * 64
    R = ops.gathered_rows_alloc(M)
    dzb, xb = torch.zeros((R, N), device=DEV), torch.zeros((R, K), device=DEV)
    for c in range(chunks):
        dzb[64 * c:64 * c + rows] = dzs[c]
        xb[64 * c:64 * c + rows] = xs[c]
    Wr, mr, vr = W0.clone(), torch.zeros_like(W0), torch.zeros_like(W0)
    cr = StepCount(DEV)
    cr.advance(1e-3, 0.9, 0.99)
    ops.skinny_linear_wgrad_adam_rows_(dzb, xb, M, Wr, mr, vr, 0.9, 0.99, 1e-8, cr.coef_dev, grad_scale=0.5)
    assert (m - mr).abs().max().item() <= 2e-6 * mr.abs().max().item()
    assert (W - Wr).abs().max().item() <= 2.1e-3 and (W - Wr).abs().mean().item() <= 1e-6
    with pytest.raises(ValueError):
        ops.skinny_linear_wgrad_adam_t16_(packed, chunks + 2, W, m, v, 0.9, 0.99, 1e-8, cnt.coef_dev)


@pytest.mark.parametrize("M,N,K", [(64, 256, 128), (37, 192, 160), (64, 960, 1920), (16, 4544, 2304)])
def test_skinny_wgrad_bf16_output_is_the_rounded_fp32_gradient(M, N, K):
    """pcaa_skinny_linear_wgrad_bf16 (the data-parallel step's bf16 gradient buckets are produced directly): the same
    accumulators as the fp32 form, rounded to nearest-even once -- bit-identical to casting the fp32 gradient."""
    dz = _rand((M, N), 700, 0.3).to(DEV)
    x = _rand((M, K), 701, 1.0).to(DEV)
    ref = ops.skinny_linear_wgrad(dz, x)
    out = torch.full((N, K), float("nan"), dtype=torch.bfloat16, device=DEV)
    got = ops.skinny_linear_wgrad(dz, x, out=out)
    assert got.dtype == torch.bfloat16 and torch.equal(got, ref.bfloat16())


def test_gemm_tile_loop_with_two_launches_competing_for_the_cus():
    """The LDS-DMA GEMM starts one workgroup per CU and lets them draw tiles from per-XCD ticket counters (one set per
    stream).  Two such launches on two streams at once: each gets its CUs late / piecemeal, every tile must still be
    computed exactly once by each, and the counters must be back at zero for the next launch."""
    P, cin, cout = 61440, 512, 1024            # 960 tiles: more than the 256 workgroups
    a = [(_rand((P, cin), 900 + i, 0.5)).to(DEV).bfloat16() for i in range(2)]
    w = [(_rand((cout, cin), 910 + i, 0.05)).to(DEV).bfloat16() for i in range(2)]
    ref = [(a[i].float() @ w[i].float().t()) for i in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    out = [torch.empty((P, cout), dtype=torch.bfloat16, device=DEV) for _ in range(2)]
    torch.cuda.synchronize()
    for rep in range(4):
        for o in out:
            o.zero_()
        torch.cuda.synchronize()
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                for _ in range(2):                     # back-to-back launches on one stream reuse its counters
                    ops.gemm(a[i], KC, w[i], KC, P, cout, cin, out=out[i], out_dtype=torch.bfloat16, math=PCAA_BF16)
        torch.cuda.synchronize()
        for i in range(2):
            err = (out[i].float() - ref[i]).abs().max().item() / ref[i].abs().max().item()
            assert err < 1e-2, (rep, i, err)


def test_cross_entropy_and_preds():
    B, K = 37, 6
    x = _rand((B, K), 70, 3.0)
    x[3] = -20.0            # saturated ELU logits: exact ties -> first index
    x[5, 2] = x[5, 4] = 5.0
    t = torch.from_numpy(np.random.default_rng(71).integers(0, K, B))
    xg = x.clone().requires_grad_(True)
    ref = O.cross_entropy(xg, t)
    ref.backward()
    loss, dl, preds = ops.cross_entropy(x.to(DEV), t.to(DEV), want_loss=True, want_grad=True, want_preds=True)
    assert abs(loss.item() - ref.item()) < 1e-5
    assert torch.allclose(dl.cpu(), xg.grad, atol=1e-6)
    assert torch.equal(preds.cpu(), O.predicted_labels(x))


@pytest.mark.parametrize("P,C,cout,dtype", [(5000, 4, 512, torch.float32), (3333, 5, 512, torch.float32),
                                            (4096, 4, 512, torch.bfloat16), (700, 3, 64, torch.float32)])
def test_pointnet_input_layer_kernels(P, C, cout, dtype):
    x = _rand((P, C), 80).to(DEV)
    W = _rand((cout, C), 81).to(DEV)
    b = _rand((cout,), 82).to(DEV)
    stats = ops.new_stats(cout, DEV)
    y = ops.pointnet_in_fwd(x, W, b, dtype, stats)
    acc = x.cpu().double() @ W.cpu().double().t()
    ref = acc + b.cpu().double()
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert (y.float().cpu().double() - ref).abs().max().item() <= tol * ref.abs().max().item()
    s = stats.sum(0).cpu()
    assert torch.allclose(s[0], acc.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[1], (acc * acc).sum(0), rtol=1e-5, atol=1e-3)
    dy = _rand((P, cout), 83).to(DEV).to(dtype)
    dW = ops.pointnet_in_wgrad(dy, x)
    refw = dy.float().cpu().double().t() @ x.cpu().double()
    assert (dW.cpu().double() - refw).abs().max().item() <= 2e-6 * refw.abs().max().item() * P ** 0.5 / 10 + 1e-5
    # recompute path (y never stored): statistics-only forward, apply, and the two backward passes over
    # da must agree with the materialised chain y -> bn_act_fwd ; bn_act_bwd_stats -> dy -> wgrad
    bn = _BN(cout, 84)
    st_r = ops.new_stats(cout, DEV)
    assert ops.pointnet_in_fwd(x, W, None, None, st_r) is None
    assert torch.allclose(st_r.sum(0).cpu(), s, rtol=1e-12, atol=1e-9)
    y0 = ops.pointnet_in_fwd(x, W, None, torch.float32)                  # bias-free pre-activation
    scale, shift, mean, rstd = ops.bn_finalize(st_r, P, b, bn, cout)
    a_ref = ops.bn_act_fwd(y0, scale, shift)
    a_rec = ops.pointnet_in_apply(x, W, scale, shift, dtype)
    atol_ = 1e-6 if dtype == torch.float32 else 1e-2
    assert (a_rec.float() - a_ref).abs().max().item() <= atol_ * max(1.0, a_ref.abs().max().item())
    da = _rand((P, cout), 85).to(DEV).to(dtype)
    st_a = ops.bn_act_bwd_stats(y0, scale, shift, mean, rstd, da=da.float())
    st_b = ops.pointnet_in_bwd_stats(da, x, W, scale, shift, mean, rstd)
    sa, sb = st_a.sum(0).cpu(), st_b.sum(0).cpu()
    assert (sa - sb).abs().max().item() <= (2e-6 if dtype == torch.float32 else 2e-3) * max(1.0, sa.abs().max().item())
    coef, _, _ = ops.bn_bwd_finalize(st_a, P, bn, mean, rstd, cout)
    dy_ref = ops.bn_bwd_dy_fused(y0, scale, shift, coef, da=da.float())
    dW_ref = ops.pointnet_in_wgrad(dy_ref, x).cpu().double()
    dW_rec = ops.pointnet_in_bwd_wgrad(da, x, W, scale, shift, coef).cpu().double()
    wtol = 2e-5 if dtype == torch.float32 else 2e-3
    assert (dW_rec - dW_ref).abs().max().item() <= wtol * max(1e-3, dW_ref.abs().max().item())


# ---------------------------------------------------------------- fused MLP heads (heads.hip)
@pytest.mark.parametrize("B,K,head,proj,sup", [(64, 8, True, True, True), (6, 4, True, True, True),
                                               (16, 6, False, False, True), (37, 2, True, False, False),
                                               (5, 8, False, True, True)])
def test_heads_fwd_bwd_vs_fp64_autograd(B, K, head, proj, sup):
    """One-launch forward / backward of MLP_sup1 -> (MLP_head) -> MLP_sup2 (+ decoder projection head)
    against fp64 torch autograd of the same layers (reference models.py:285-292, PCAA_ablation.py:778-781)."""
    d2 = 16 if head else 32
    x4 = _rand((B, 512), 1)
    P = {"W1": _rand((32, 512), 2, 512 ** -0.5), "b1": _rand((32,), 3, 0.1),
         "W2": _rand((K, d2), 4, d2 ** -0.5), "b2": _rand((K,), 5, 0.1)}
    if head:
        P.update(Wh=_rand((16, 32), 6, 32 ** -0.5), bh=_rand((16,), 7, 0.1))
    if proj:
        P.update(Wg=_rand((64, 32), 8, 32 ** -0.5), bg=_rand((64,), 9, 0.1))
    d_logits, d_sup, d_hproj = _rand((B, K), 10), _rand((B, 32), 11) if sup else None, _rand((B, 64), 12) if proj else None
    # fp64 reference
    R = {k: v.double().requires_grad_(True) for k, v in P.items()}
    xr = x4.double().requires_grad_(True)
    elu = torch.nn.functional.elu
    sup_r = elu(xr @ R["W1"].t() + R["b1"])
    h_r = elu(sup_r @ R["Wh"].t() + R["bh"]) if head else sup_r
    log_r = elu(h_r @ R["W2"].t() + R["b2"])
    obj = (log_r * d_logits.double()).sum()
    if sup:
        obj = obj + (sup_r * d_sup.double()).sum()
    if proj:
        hp_r = elu(sup_r @ R["Wg"].t() + R["bg"])
        obj = obj + (hp_r * d_hproj.double()).sum()
    obj.backward()
    # HIP
    G = {k: v.to(DEV) for k, v in P.items()}
    xg = x4.to(DEV)
    s, h, lg, hp = ops.heads_fwd(xg, G["W1"], G["b1"], G.get("Wh"), G.get("bh"), G["W2"], G["b2"], G.get("Wg"), G.get("bg"))

    def close(a, ref, what, tol=2e-5):
        err = (a.cpu().double() - ref.detach()).abs().max().item()
        assert err <= tol * max(ref.detach().abs().max().item(), 1e-3), (what, err)

    close(s, sup_r, "sup_fv"); close(lg, log_r, "logits")
    if head:
        close(h, h_r, "h")
    if proj:
        close(hp, hp_r, "hproj")
    if B > 64:
        return
    outs, dx4 = ops.heads_bwd(xg, s, h, lg, hp, G["W1"], G.get("Wh"), G["W2"], G.get("Wg"), d_logits.to(DEV),
                              d_sup.to(DEV) if sup else None, d_hproj.to(DEV) if proj else None)
    close(dx4, xr.grad, "dx4")
    for k, nm in (("dW1", "W1"), ("db1", "b1"), ("dW2", "W2"), ("db2", "b2"), ("dWh", "Wh"), ("dbh", "bh"),
                  ("dWg", "Wg"), ("dbg", "bg")):
        if nm in R:
            close(outs[k], R[nm].grad, k)


def test_heads_fwd_large_batch_and_unsupported_backward():
    B, K = 1024, 8
    x4 = _rand((B, 512), 21).to(DEV)
    W1, b1 = _rand((32, 512), 22, 0.05).to(DEV), _rand((32,), 23, 0.1).to(DEV)
    W2, b2 = _rand((K, 32), 24, 0.2).to(DEV), _rand((K,), 25, 0.1).to(DEV)
    s, h, lg, hp = ops.heads_fwd(x4, W1, b1, None, None, W2, b2)
    ref = torch.nn.functional.elu(x4.double() @ W1.double().t() + b1.double())
    assert (s.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert h is None and hp is None and tuple(lg.shape) == (B, K)
    assert not ops.heads_supported(B, K, 512, 32, 0, 0, True)      # backward keeps all rows in LDS: B <= 64
    assert not ops.heads_supported(8, K, 256, 32, 0, 0, False)


# ---------------------------------------------------------------- fused temporal-block layer (dtc_fused.hip)
@pytest.mark.parametrize("B,T,cin,cout,d,act", [(3, 30, 16, 32, 2, True), (2, 30, 1024, 16, 1, False),
                                                (5, 30, 256, 512, 4, True), (4, 7, 64, 128, 1, True),
                                                (64, 30, 32, 64, 4, True),
                                                # the two-sequence kernels (round 4): 64-column and 32-column workgroups,
                                                # an odd batch, a short sequence, a ragged last chunk of columns
                                                (64, 30, 256, 512, 4, True), (64, 30, 128, 256, 2, True),
                                                (5, 17, 128, 80, 2, True), (7, 30, 160, 64, 1, False)])
def test_dtc_conv_fwd_vs_fp64_conv1d(B, T, cin, cout, d, act):
    """Implicit-im2col causal dilated Conv1d (models.py:59-68, 75-76: padding 2d both sides, last 2d
    outputs dropped) with the previous layer's BatchNorm+ELU applied on load, against fp64 torch."""
    src = _rand((B * T, cin), 31)
    W = _rand((cout, cin, 3), 32, (3 * cin) ** -0.5)
    scale = (_rand((cin,), 33, 0.3) + 1.0) if act else None
    shift = _rand((cin,), 34, 0.3) if act else None
    a = src.double()
    if act:
        a = torch.nn.functional.elu(a * scale.double() + shift.double())
    x = a.view(B, T, cin).permute(0, 2, 1)                                # [B, cin, T]
    ref = torch.nn.functional.conv1d(x, W.double(), padding=2 * d, dilation=d)[:, :, :-2 * d]
    ref = ref.permute(0, 2, 1).reshape(B * T, cout)
    stats = torch.zeros((ops.NREP, 2, cout), dtype=torch.float64, device=DEV)
    y, col = ops.dtc_conv_fwd(src.to(DEV), scale.to(DEV) if act else None, shift.to(DEV) if act else None,
                              W.view(cout, cin * 3).to(DEV), B, T, d, stats=stats, want_col=True)
    scl = ref.abs().max().item()
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-6 * (3 * cin) ** 0.5 * scl + 1e-6
    st = stats.sum(0).cpu()
    assert torch.allclose(st[0], ref.sum(0), rtol=1e-4, atol=1e-4 * scl * (B * T) ** 0.5)
    assert torch.allclose(st[1], (ref * ref).sum(0), rtol=1e-4, atol=1e-6)
    col_ref = ops.dtc_im2col(a.float().to(DEV).contiguous(), B, T, cin, d)
    assert (col - col_ref).abs().max().item() <= 1e-6 * max(a.abs().max().item(), 1.0)


def test_fused_gemm_entry_points_state_the_tile_loops_domain():
    """Round 5: the 4-wave tile loops are the only LDS-DMA GEMM kernels left (the 8-wave kernel is gone); they need a
    contraction of at least five 64-deep steps.  The predicates say so, and an entry point called outside the domain
    reports an ARGUMENT error (callers -- functional.py -- then take the unfused chain), not a launch failure."""
    assert ops.gemm_dgrad_bn_supported(512, 256, 320) and not ops.gemm_dgrad_bn_supported(512, 256, 256)
    assert ops.gemm_split3_supported(512, 256, 128) and not ops.gemm_split3_supported(512, 256, 64)
    a = torch.zeros((256, 64), dtype=torch.bfloat16, device=DEV)
    w = torch.zeros((256, 64), dtype=torch.bfloat16, device=DEV)
    with pytest.raises((ValueError, RuntimeError)):
        ops.gemm_affine_elu(a, w, torch.ones(256, device=DEV), torch.zeros(256, device=DEV))
    # a plain product outside the domain is served by the register-staged kernel
    c = ops.gemm(torch.ones((256, 64), dtype=torch.bfloat16, device=DEV), KC, torch.ones((256, 64), dtype=torch.bfloat16, device=DEV),
                 KC, 256, 256, 64, math=PCAA_BF16)
    assert bool((c == 64.0).all())


@pytest.mark.parametrize("M,N,K", [(256, 256, 320), (512, 1024, 512), (3840, 512, 512), (1320, 512, 512)])
def test_gemm_affine_elu_epilogue(M, N, K):
    """Eval-mode PointNet layer in one launch: ELU(scale * (a @ W^T) + shift), bf16 in/out, fp32 accumulate."""
    a = _rand((M, K), 61).to(DEV).to(torch.bfloat16)
    W = (_rand((N, K), 62, K ** -0.5)).to(DEV).to(torch.bfloat16)
    scale = (_rand((N,), 63, 0.3) + 1.0).to(DEV)
    shift = _rand((N,), 64, 0.5).to(DEV)
    out = ops.gemm_affine_elu(a, W, scale, shift)
    ref = torch.nn.functional.elu((a.double() @ W.double().t()) * scale.double() + shift.double())
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == (M, N)
    err = (out.double() - ref).abs().max().item()
    assert err <= 2 ** -7 * ref.abs().max().item(), err           # one bf16 rounding of the output
    # and it equals the two-pass path (GEMM -> bf16 y -> BN+ELU pass) up to y's extra bf16 rounding
    y = ops.gemm(a, KC, W, KC, M, N, K, out_dtype=torch.bfloat16, math=PCAA_BF16)
    two = ops.bn_act_fwd(y, scale, shift)
    assert (out.float() - two.float()).abs().max().item() <= 2 ** -6 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K,pool", [(512, 256, 384, 32), (1024, 1024, 512, 64), (3840, 1024, 1024, 128), (960, 512, 512, 64),
                                         (1440, 256, 1024, 32)])
def test_gemm_affine_elu_meanpool_epilogue(M, N, K, pool):
    """Last eval-mode PointNet layer: BN (affine) + ELU + mean over the frame's points in the GEMM epilogue."""
    a = _rand((M, K), 71).to(DEV).to(torch.bfloat16)
    W = (_rand((N, K), 72, K ** -0.5)).to(DEV).to(torch.bfloat16)
    scale = (_rand((N,), 73, 0.3) + 1.0).to(DEV)
    shift = _rand((N,), 74, 0.5).to(DEV)
    out = ops.gemm_affine_elu(a, W, scale, shift, pool_rows=pool)
    act = torch.nn.functional.elu((a.double() @ W.double().t()) * scale.double() + shift.double())
    ref = act.view(M // pool, pool, N).mean(1)
    assert out.dtype == torch.float32 and tuple(out.shape) == (M // pool, N)
    assert (out.double() - ref).abs().max().item() <= 2e-5 * max(act.abs().max().item(), 1.0)


def test_wgan_gp_dz_vs_autograd():
    """d(d_loss)/dz of the critic step (the gradient variant 1 would send into the mean learner) against
    autograd through the oracle's loss (double backward through the gradient penalty)."""
    B, K = 16, 8
    disc = {k: v for k, v in zip(("model.0.weight", "model.0.bias", "model.2.weight", "model.2.bias", "model.4.weight",
                                  "model.4.bias"),
                                 (_rand((64, 32 + K), 81, 0.2), _rand((64,), 82, 0.1), _rand((32, 64), 83, 0.2),
                                  _rand((32,), 84, 0.1), _rand((1, 32), 85, 0.3), _rand((1,), 86, 0.1)))}
    z = _rand((B, 32), 87); fv = _rand((B, 32), 88)
    oh = torch.nn.functional.one_hot(torch.arange(B) % K, K).float()
    al = torch.from_numpy(np.random.default_rng(89).random((B, 1), dtype=np.float32))
    zr = z.clone().requires_grad_(True)
    d_loss, gp = O.wgan_gp_d_loss(disc, fv, oh, zr, al, 15.0)
    ref = torch.autograd.grad(d_loss, zr)[0]
    losses, grads, dz = ops.disc_wgan_gp(z.to(DEV), fv.to(DEV), oh.to(DEV), al.reshape(-1).to(DEV),
                                         [v.to(DEV) for v in disc.values()], 15.0, want_dz=True)
    assert abs(losses[0].item() - d_loss.item()) <= 1e-4 * abs(d_loss.item()) + 1e-6
    assert (dz.cpu() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-7


@pytest.mark.parametrize("B,T,cin,cout,d", [(3, 30, 16, 32, 2), (2, 30, 1024, 16, 1), (5, 30, 256, 512, 4),
                                            (4, 7, 64, 128, 1), (64, 30, 32, 64, 4), (2, 30, 20, 36, 2),
                                            # the two-sequence kernels: two staging passes (512 contraction channels),
                                            # 64-column workgroups, an odd batch with a short sequence
                                            (64, 30, 256, 512, 4), (64, 30, 384, 128, 1), (5, 17, 80, 160, 2),
                                            (64, 30, 128, 256, 2)])
def test_dtc_conv_dgrad_vs_fp64_autograd(B, T, cin, cout, d):
    """Adjoint of the causal dilated convolution w.r.t. its input (implicit col2im) against fp64 autograd of
    conv1d(padding=2d)[..., :-2d] (models.py:59-68, 75-76), and against the unfused dcol = dy.W + col2im."""
    W = _rand((cout, cin, 3), 91, (3 * cin) ** -0.5)
    dy = _rand((B * T, cout), 92)
    x = torch.zeros((B, cin, T), dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv1d(x, W.double(), padding=2 * d, dilation=d)[:, :, :-2 * d]
    y.backward(dy.double().view(B, T, cout).permute(0, 2, 1))
    ref = x.grad.permute(0, 2, 1).reshape(B * T, cin)
    W2d = W.view(cout, cin * 3).to(DEV)
    da, _, _ = ops.dtc_conv_dgrad(dy.to(DEV), W2d, B, T, cin, d)
    scl = ref.abs().max().item()
    assert (da.cpu().double() - ref).abs().max().item() <= 2e-6 * (3 * cout) ** 0.5 * scl + 1e-6
    dcol = ops.gemm(dy.to(DEV), KC, W2d, RC, B * T, cin * 3, cout, out_dtype=torch.float32)
    two = ops.dtc_col2im(dcol, B, T, cin, d)
    assert (da - two).abs().max().item() <= 1e-5 * scl


@pytest.mark.parametrize("B,T,cin,cout,d", [(3, 30, 16, 32, 2), (5, 30, 256, 512, 4), (64, 30, 32, 64, 4),
                                            (64, 30, 256, 512, 4), (64, 30, 384, 128, 1), (7, 19, 128, 256, 2)])
def test_dtc_conv_dgrad_fused_bn_halves(B, T, cin, cout, d):
    """dy formed on load (dy = c0*dz + c1*y + c2) and the epilogue for the layer below (dz_below = da*ELU'(z),
    statistics {sum dz, sum dz*yhat}) against the separate passes."""
    W2d = _rand((cout, cin * 3), 101, (3 * cin) ** -0.5).to(DEV)
    dz, y = _rand((B * T, cout), 102).to(DEV), _rand((B * T, cout), 103).to(DEV)
    coef = torch.stack([_rand((cout,), 104, 0.3) + 1.0, _rand((cout,), 105, 0.1), _rand((cout,), 106, 0.1)]).to(DEV).contiguous()
    yb = _rand((B * T, cin), 107).to(DEV)
    scale, shift = (_rand((cin,), 108, 0.3) + 1.0).to(DEV), _rand((cin,), 109, 0.3).to(DEV)
    mean, rstd = _rand((cin,), 110, 0.2).to(DEV), (_rand((cin,), 111, 0.1).abs() + 0.8).to(DEV)
    out, stats, dy = ops.dtc_conv_dgrad(None, W2d, B, T, cin, d, dz=dz, y=y, coef=coef, want_dy=True,
                                        below=(yb, scale, shift, mean, rstd))
    dy_ref = ops.bn_bwd_dy(dz, y, coef)
    assert torch.equal(dy, dy_ref) or (dy - dy_ref).abs().max().item() <= 1e-6 * dy_ref.abs().max().item()
    da_ref, _, _ = ops.dtc_conv_dgrad(dy_ref, W2d, B, T, cin, d)
    z = yb.double() * scale.double() + shift.double()
    dz_ref = da_ref.double() * torch.where(z > 0, torch.ones_like(z), torch.exp(z))
    scl = dz_ref.abs().max().item()
    assert (out.double() - dz_ref).abs().max().item() <= 2e-6 * scl + 1e-7
    st = stats.sum(0)
    yhat = (yb.double() - mean.double()) * rstd.double()
    assert torch.allclose(st[0], dz_ref.sum(0), rtol=1e-4, atol=1e-5 * scl * (B * T) ** 0.5)
    assert torch.allclose(st[1], (dz_ref * yhat).sum(0), rtol=1e-4, atol=1e-5 * scl * (B * T) ** 0.5 * yhat.abs().max().item())



# ---------------------------------------------------------------------------------------------------------
# split-fp16 products (round 3, precision "fp16x3")
# ---------------------------------------------------------------------------------------------------------
def test_gemm_split3_vs_fp64_both_layouts():
    """pcaa_gemm_split3 / pcaa_gemm_slabs_split3 on [hi | lo] fp16 operand images: hi.hi + lo.hi + hi.lo must sit at
    the fp32 level from the fp64 product (plain bf16 operands: ~2e-3; a bf16 pair: 4.5e-6, measured in round 3)."""
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(11)
    M, N, K = 2048, 512, 1024
    A = torch.randn(M, K, device=dev, generator=g)
    B = torch.randn(N, K, device=dev, generator=g) * 0.05
    ai, bi = ops.split_f16(A, scale=ops.SPLIT_SCALE_ACT), ops.split_f16(B)
    assert tuple(ai.img.shape) == (M, 2 * K) and ai.img.dtype == torch.float16
    # the image is exactly hi | lo
    hi = A.to(torch.float16)
    assert torch.equal(ai.img[:, :K], hi) and torch.equal(ai.img[:, K:], (A - hi.float()).to(torch.float16))
    assert (bi.float() - B).abs().max().item() <= 1e-6 * B.abs().max().item()
    stats = ops.new_stats(N, dev)
    C = ops.gemm_split3(ai, bi, KC, M, N, K, colstats=stats)
    ref = A.double() @ B.double().t()
    err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
    plain = ((A.to(torch.bfloat16).float() @ B.to(torch.bfloat16).float().t()).double() - ref).abs().max() / ref.abs().max()
    f32 = ((A @ B.t()).double() - ref).abs().max() / ref.abs().max()
    print(f"split3 KC: rel err {err:.2e} (plain bf16 operands {plain.item():.2e}, torch fp32 matmul {f32.item():.2e})")
    assert err <= 2e-6
    assert torch.allclose(stats.sum(0)[0], C.double().sum(0), rtol=1e-6, atol=1e-3)
    assert torch.allclose(stats.sum(0)[1], (C.double() ** 2).sum(0), rtol=1e-6, atol=1e-3)
    # transposed image of a weight matrix (the dgrad's operand)
    bt = ops.split_f16(B, transpose=True)
    assert tuple(bt.shape) == (K, N) and (bt.float() - B.t()).abs().max().item() <= 1e-6 * B.abs().max().item()
    # RC x RC (weight gradient: contraction over the rows), slab split-K; gradients are small numbers
    P, co, ci = 64 * 256, 512, 256
    dy = torch.randn(P, co, device=dev, generator=g) * 1e-5
    a = torch.randn(P, ci, device=dev, generator=g)
    dW = ops.gemm_slabs_split3(ops.split_f16(dy, scale=ops.SPLIT_SCALE_GRAD), ops.split_f16(a, scale=ops.SPLIT_SCALE_ACT),
                               co, ci, P, 8)
    refw = dy.double().t() @ a.double()
    errw = ((dW.double() - refw).abs().max() / refw.abs().max()).item()
    print(f"split3 RC (slab split-K): rel err {errw:.2e}")
    assert errw <= 2e-6


def test_split_image_producers():
    """bn_act_fwd_split / bn_bwd_dy_fused_split / pointnet_in_apply(split) write the [hi | lo] image of what their fp32
    forms compute"""
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(12)
    rows, ch = 3840, 512
    y = torch.randn(rows, ch, device=dev, generator=g)
    scale, shift = torch.rand(ch, device=dev, generator=g) + 0.5, torch.randn(ch, device=dev, generator=g) * 0.1
    a32 = ops.bn_act_fwd(y, scale, shift)
    ai = ops.bn_act_fwd_split(y, scale, shift)
    hi = a32.to(torch.float16)
    assert torch.equal(ai.img[:, :ch], hi) and torch.equal(ai.img[:, ch:], (a32 - hi.float()).to(torch.float16))
    coef = torch.randn(3, ch, device=dev, generator=g)
    da = torch.randn(rows, ch, device=dev, generator=g) * 1e-4
    dy32 = ops.bn_bwd_dy_fused(y, scale, shift, coef * 1e-4, da=da.clone())
    dyi = ops.bn_bwd_dy_fused_split(y, scale, shift, coef * 1e-4, da=da)
    assert (dyi.float() - dy32).abs().max().item() <= 2e-6 * dy32.abs().max().item()
    dpool = torch.randn(rows // 128, ch, device=dev, generator=g) * 1e-3
    dy32p = ops.bn_bwd_dy_fused(y, scale, shift, coef * 1e-4, dpool=dpool, group_rows=128, pool_scale=1 / 128)
    dyip = ops.bn_bwd_dy_fused_split(y, scale, shift, coef * 1e-4, dpool=dpool, group_rows=128, pool_scale=1 / 128)
    assert (dyip.float() - dy32p).abs().max().item() <= 2e-6 * dy32p.abs().max().item()
    x = torch.randn(rows, 4, device=dev, generator=g)
    W = torch.randn(ch, 4, device=dev, generator=g) * 0.5
    a1 = ops.pointnet_in_apply(x, W, scale, shift, torch.float32)
    a1i = ops.pointnet_in_apply(x, W, scale, shift, ops.SplitImage.dtype)
    assert (a1i.float() - a1).abs().max().item() <= 1e-6 * a1.abs().max().item()


def test_gemm_dgrad_bn_split3_vs_separate_chain():
    """the fused dgrad of the fp16x3 mode (split operands, fp32 y / dz, statistics in the epilogue) against the separate
    chain gemm_split3 -> bn_act_bwd_dz on the same operands, and bn_bwd_dy_split against bn_bwd_dy_fused_split"""
    M, N, K = 768, 512, 256
    rng = np.random.default_rng(45)
    dy = torch.from_numpy((rng.standard_normal((M, K)) * 1e-3).astype(np.float32)).to(DEV)
    Wt = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(DEV)
    y = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(DEV)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, N).astype(np.float32)).to(DEV)
    shift = torch.from_numpy(rng.uniform(-0.5, 0.5, N).astype(np.float32)).to(DEV)
    mean = torch.from_numpy(rng.uniform(-0.2, 0.2, N).astype(np.float32)).to(DEV)
    rstd = torch.from_numpy(rng.uniform(0.5, 2.0, N).astype(np.float32)).to(DEV)
    dyi, wti = ops.split_f16(dy, scale=ops.SPLIT_SCALE_GRAD), ops.split_f16(Wt, scale=ops.SPLIT_SCALE_WEIGHT)
    dz, st = ops.gemm_dgrad_bn_split3(dyi, wti, y, scale, shift, mean, rstd)
    da = ops.gemm_split3(dyi, wti, KC, M, N, K)
    dz_ref, st_ref = ops.bn_act_bwd_dz(y, scale, shift, mean, rstd, da=da)
    assert (dz - dz_ref).abs().max().item() <= 1e-6 * max(1e-30, dz_ref.abs().max().item())
    sa, sb = st.sum(0).cpu(), st_ref.sum(0).cpu()
    assert (sa - sb).abs().max().item() <= 1e-5 * sb.abs().max().item()
    # fp64 of the unrounded product
    da64 = dy.cpu().double() @ Wt.cpu().double().t()
    z = y.cpu().double() * scale.cpu().double() + shift.cpu().double()
    dz64 = da64 * torch.where(z > 0, torch.ones_like(z), torch.exp(z))
    assert (dz.cpu().double() - dz64).abs().max().item() <= 2e-5 * dz64.abs().max().item()
    coef = torch.from_numpy((rng.standard_normal((3, N)) * 1e-4).astype(np.float32)).to(DEV)   # gradient-sized: the image's 2^16 scale
    a = ops.bn_bwd_dy_split(dz, y, coef).float()
    b = (coef[0] * dz + coef[1] * y + coef[2])
    assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()


def test_split_image_range_guard_saturates_and_flags():
    """Round-3 advisor finding: the fp16x3 mode's [hi | lo] images hold |scale * v| <= 65504 only.  A value beyond that
    is saturated (finite image: no inf - inf = NaN in the product) and the device flag is raised; ops.range_check()
    -- called by PCAATrainer.check() -- then raises.  In-range tensors leave the flag alone."""
    ops.range_check()                                        # clear anything an earlier test left
    y = torch.randn(256, 64, device=DEV) * 0.1
    scale, shift = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    a = ops.bn_act_fwd_split(y, scale, shift)
    torch.cuda.synchronize()
    ops.range_check()                                        # in range: no complaint
    # a gradient spike: |dy| * 2^16 > 65504  <=>  |dy| >= ~1.0
    dz = torch.randn(256, 64, device=DEV) * 1e-3
    dz[17, 5] = 3.0
    coef = torch.stack([torch.ones(64), torch.zeros(64), torch.zeros(64)]).to(DEV).contiguous()
    dy = ops.bn_bwd_dy_split(dz, y, coef)
    torch.cuda.synchronize()
    assert torch.isfinite(dy.img.float()).all()
    assert abs(dy.float()[17, 5].item() - 65504.0 / ops.SPLIT_SCALE_GRAD) < 1e-6      # saturated, not inf
    ok = torch.ones_like(dz, dtype=torch.bool); ok[17, 5] = False
    assert torch.allclose(dy.float()[ok], dz[ok], rtol=2e-6, atol=1e-12)
    with pytest.raises(FloatingPointError):
        ops.range_check()
    ops.range_check()                                        # the check reset the flag
    # an activation beyond fp16's range (ELU is unbounded above), a weight beyond 65504 / 2^8, and a NaN
    big = torch.full((256, 64), 7.0e4, device=DEV)
    a = ops.bn_act_fwd_split(big, scale, shift)
    assert torch.isfinite(a.img.float()).all()
    with pytest.raises(FloatingPointError):
        ops.range_check()
    w = torch.randn(64, 64, device=DEV); w[3, 3] = 300.0
    ops.split_f16(w)
    with pytest.raises(FloatingPointError):
        ops.range_check()
    y2 = y.clone(); y2[0, 0] = float("nan")
    a2 = ops.bn_act_fwd_split(y2, scale, shift)
    # a NaN raises the flag AND stays NaN (it used to come out as -65504, so a diverged step showed
    # finite losses until the once-per-epoch check); its neighbours are untouched
    assert torch.isnan(a2.float()[0, 0]) and torch.isfinite(a2.float()[0, 1:]).all() and torch.isfinite(a2.float()[1:]).all()
    with pytest.raises(FloatingPointError):
        ops.range_check()
    # the device argument resolves like the flag's own key: an index-less "cuda" is the current device
    ops.bn_act_fwd_split(big, scale, shift)
    with pytest.raises(FloatingPointError):
        ops.range_check("cuda")
    ops.bn_act_fwd_split(big, scale, shift)
    with pytest.raises(FloatingPointError):
        ops.range_check(torch.device("cuda", torch.cuda.current_device()))
    # the first layer's producer
    x = torch.randn(512, 4, device=DEV) * 1e3
    Wp = torch.randn(512, 4, device=DEV) * 100.0
    ops.pointnet_in_apply(x, Wp, torch.ones(512, device=DEV), torch.zeros(512, device=DEV), ops.SplitImage.dtype)
    with pytest.raises(FloatingPointError):
        ops.range_check()


@pytest.mark.parametrize("M,N,K", [(2048, 512, 512), (1320, 512, 1024), (72000 // 8, 1024, 512), (130, 256, 320)])
@pytest.mark.parametrize("cdt", [torch.bfloat16, torch.float32])
def test_gemm_v2_tile_loop_vs_register_staged_kernel_and_fp64(M, N, K, cdt):
    """The 4-wave tile loop (csrc/gemm_v2.h) against an fp64 product of the same bf16 operands, with the BatchNorm
    column statistics, for whole and PARTIAL last row tiles (M % 256 != 0: rows past M are requested out of the buffer
    range, read as zeros and are not stored) -- and against the register-staged 256 x 256 kernel, the one the library
    falls back to when the loop declines a launch (pcaa_gemm_v2_enable(0); both accumulate in fp32 from bf16 operands:
    equal up to the order of the K steps).  (Rounds 1-4 compared bitwise with the 8-wave loop, removed in round 5.)"""
    a = _rand((M, K), 71).to(DEV).to(torch.bfloat16)
    w = (_rand((N, K), 72, K ** -0.5)).to(DEV).to(torch.bfloat16)
    ref = a.double().cpu() @ w.double().cpu().t()
    guard = torch.full((M + 64, N), 7.0, dtype=cdt, device=DEV)            # the rows behind M must stay untouched
    out = guard[:M]
    stats = ops.new_stats(N, DEV)
    ops.gemm_v2_enable(True)
    try:
        ops.gemm(a, KC, w, KC, M, N, K, colstats=stats, out=out, out_dtype=cdt, math=PCAA_BF16)
        torch.cuda.synchronize()
        tol = (2 ** -7 if cdt == torch.bfloat16 else 1e-5) * ref.abs().max().item()
        assert (out.double().cpu() - ref).abs().max().item() <= tol
        assert bool((guard[M:] == 7.0).all()), "a partial tile wrote past the last row"
        ssum = stats.sum(0).cpu()
        assert (ssum[0] - ref.sum(0)).abs().max().item() <= 1e-4 * ref.abs().sum(0).max().item()
        assert (ssum[1] - (ref * ref).sum(0)).abs().max().item() <= 1e-4 * (ref * ref).sum(0).max().item()
        ops.gemm_v2_enable(False)
        old = torch.empty_like(out)
        stats_old = ops.new_stats(N, DEV)
        ops.gemm(a, KC, w, KC, M, N, K, colstats=stats_old, out=old, out_dtype=cdt, math=PCAA_BF16)
        torch.cuda.synchronize()
        assert (old.double().cpu() - ref).abs().max().item() <= tol
        assert (old.float() - out.float()).abs().max().item() <= (2 ** -7 if cdt == torch.bfloat16 else 2e-6) * ref.abs().max().item()
        so = stats_old.sum(0).cpu()
        assert (so[0] - ssum[0]).abs().max().item() <= 1e-5 * ref.abs().sum(0).max().item()
    finally:
        ops.gemm_v2_enable(True)


def test_gemm_split3_partial_row_tile():
    """the fp16x3 mode's product and fused dgrad on a row count that is not a multiple of 256 (the reference's default
    shape: B = 16, N = 150 -> 72 000 rows): against the fp64 product / the separate chain"""
    M, N, K = 1320, 512, 512
    a = _rand((M, K), 81).to(DEV)
    w = _rand((N, K), 82, K ** -0.5).to(DEV)
    assert ops.gemm_split3_supported(M, N, K)
    ai, wi = ops.split_f16(a, scale=ops.SPLIT_SCALE_ACT), ops.split_f16(w)
    stats = ops.new_stats(N, DEV)
    y = ops.gemm_split3(ai, wi, KC, M, N, K, colstats=stats)
    ref = a.double().cpu() @ w.double().cpu().t()
    assert (y.double().cpu() - ref).abs().max().item() <= 4e-6 * ref.abs().max().item()
    assert (stats.sum(0).cpu()[0] - ref.sum(0)).abs().max().item() <= 1e-5 * ref.abs().sum(0).max().item()


# ---------------------------------------------------------------------------------------------------------
# temporal block on the bf16 MFMA pipe (round 4: the bf16 throughput mode's variants of the two kernels)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T,cin,cout,d,act", [(3, 30, 16, 32, 2, True), (2, 30, 1024, 16, 1, False), (5, 30, 256, 512, 4, True),
                                                (64, 30, 128, 256, 2, True), (4, 30, 32, 64, 4, True), (2, 17, 64, 128, 1, True),
                                                (64, 30, 256, 512, 4, True), (5, 17, 128, 80, 2, True)])
def test_dtc_conv_fwd_bf16_variant(B, T, cin, cout, d, act):
    """pcaa_dtc_conv_fwd_bf16 against fp64 conv1d of the bf16-ROUNDED operands (the kernel rounds the activated input
    and the weights as it builds the fragments; products exact, fp32 accumulate) and against the exact-fp32 kernel at
    the bf16 tolerance; statistics of its own output; the im2col matrix is the fp32 one, unchanged."""
    src = _rand((B * T, cin), 31)
    W = _rand((cout, cin, 3), 32, (3 * cin) ** -0.5)
    scale = (_rand((cin,), 33, 0.3) + 1.0) if act else None
    shift = _rand((cin,), 34, 0.3) if act else None
    args = (src.to(DEV), scale.to(DEV) if act else None, shift.to(DEV) if act else None, W.view(cout, cin * 3).to(DEV), B, T, d)
    stats = torch.zeros((ops.NREP, 2, cout), dtype=torch.float64, device=DEV)
    y16, col16 = ops.dtc_conv_fwd(*args, stats=stats, want_col=True, bf16=True)
    y32, col32 = ops.dtc_conv_fwd(*args, want_col=True)
    assert torch.equal(col16, col32)
    # reference on the rounded operands: the activation as the kernel stages it (fp32 ELU of the fp32 affine), rounded
    a = src.to(DEV)
    if act:
        a = ops.bn_act_fwd(a, scale.to(DEV), shift.to(DEV))
    a16 = a.bfloat16().double().cpu().view(B, T, cin).permute(0, 2, 1)
    ref = torch.nn.functional.conv1d(a16, W.bfloat16().double(), padding=2 * d, dilation=d)[:, :, :-2 * d]
    ref = ref.permute(0, 2, 1).reshape(B * T, cout)
    scl = ref.abs().max().item()
    assert (y16.cpu().double() - ref).abs().max().item() <= 3e-5 * scl + 1e-6, "fp32 accumulation of exact bf16 products"
    assert (y16 - y32).abs().max().item() <= 2e-2 * y32.abs().max().item()
    st = stats.sum(0).cpu()
    assert torch.allclose(st[0], y16.double().cpu().sum(0), rtol=1e-5, atol=1e-5 * scl * (B * T) ** 0.5)
    assert torch.allclose(st[1], (y16.double().cpu() ** 2).sum(0), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,T,cin,cout,d", [(3, 30, 16, 32, 2), (2, 30, 1024, 16, 1), (5, 30, 256, 512, 4), (64, 30, 32, 64, 4),
                                            (4, 30, 128, 256, 2), (64, 30, 256, 512, 4), (64, 30, 384, 128, 1), (5, 17, 80, 160, 2)])
def test_dtc_conv_dgrad_bf16_variant(B, T, cin, cout, d):
    """pcaa_dtc_conv_dgrad_bf16 with the fused BatchNorm halves (dy formed on load, ELU' + statistics of the layer below)
    against the exact-fp32 kernel: dy identical (formed in fp32 before the rounding), dz_below and the statistics at the
    bf16 tolerance, and exact on bf16-representable operands."""
    W2d = _rand((cout, cin * 3), 101, (3 * cin) ** -0.5).to(DEV)
    dz, y = _rand((B * T, cout), 102).to(DEV), _rand((B * T, cout), 103).to(DEV)
    coef = torch.stack([_rand((cout,), 104, 0.3) + 1.0, _rand((cout,), 105, 0.1), _rand((cout,), 106, 0.1)]).to(DEV).contiguous()
    yb = _rand((B * T, cin), 107).to(DEV)
    scale, shift = (_rand((cin,), 108, 0.3) + 1.0).to(DEV), _rand((cin,), 109, 0.3).to(DEV)
    mean, rstd = _rand((cin,), 110, 0.2).to(DEV), (_rand((cin,), 111, 0.1).abs() + 0.8).to(DEV)
    fused = cout <= 512
    kw = dict(dz=dz, y=y, coef=coef, want_dy=True, below=(yb, scale, shift, mean, rstd)) if fused else {}
    dy_in = None if fused else ops.bn_bwd_dy(dz, y, coef)
    o32 = ops.dtc_conv_dgrad(dy_in, W2d, B, T, cin, d, **kw)
    o16 = ops.dtc_conv_dgrad(dy_in, W2d, B, T, cin, d, bf16=True, **kw)
    if fused:
        assert torch.equal(o16[2], o32[2])
        scl = o32[0].abs().max().item()
        assert (o16[0] - o32[0]).abs().max().item() <= 2e-2 * scl
        s16, s32 = o16[1].sum(0), o32[1].sum(0)
        assert (s16 - s32).abs().max().item() <= 2e-2 * s32.abs().max().item() + 1e-3 * scl * (B * T) ** 0.5
    else:
        assert (o16[0] - o32[0]).abs().max().item() <= 2e-2 * o32[0].abs().max().item()
    # exact on bf16-representable operands (small integers): the two kernels then agree to fp32 summation order
    Wi = torch.randint(-3, 4, (cout, cin * 3), device=DEV).float()
    dyi = torch.randint(-4, 5, (B * T, cout), device=DEV).float()
    a32, _, _ = ops.dtc_conv_dgrad(dyi, Wi, B, T, cin, d)
    a16, _, _ = ops.dtc_conv_dgrad(dyi, Wi, B, T, cin, d, bf16=True)
    assert torch.equal(a16, a32)


# ---------------------------------------------------------------------------------------------------------
# decoder weights as bf16 images (round 4, ABI 14): the weight-streaming kernels read half the bytes
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(64, 1920, 960), (64, 3840, 1920), (37, 256, 128), (64, 15360, 7680)])
def test_skinny_w16_image_is_bitwise_the_fp32_stream(M, N, K):
    """pcaa_skinny_linear_fwd_w16 / _dgrad_w16 on the bf16 image of W against the kernels that round W in registers: the
    image holds exactly those roundings, so every output bit agrees."""
    x, W, b = _rand((M, K), 201).to(DEV), _rand((N, K), 202, K ** -0.5).to(DEV), _rand((N,), 203, 0.1).to(DEV)
    dz, a_prev = _rand((M, N), 204).to(DEV), _rand((M, K), 205).to(DEV)
    W16 = W.bfloat16()
    assert torch.equal(ops.cast_bf16(W, want_transposed=False)[0], W16)
    y0 = ops.skinny_linear_fwd(x, W, b, ACT_ELU)
    y1 = ops.skinny_linear_fwd(x, W, b, ACT_ELU, W16=W16)
    assert torch.equal(y0, y1)
    d0 = ops.skinny_linear_dgrad(dz, W, a_prev=a_prev)
    d1 = ops.skinny_linear_dgrad(dz, W, a_prev=a_prev, W16=W16)
    assert torch.equal(d0, d1)


def test_gemm_group_rc_f32_vs_fp64():
    """pcaa_gemm_group_rc_f32: six products of the temporal block's weight-gradient shapes (B*T = 1920 rows) in one launch,
    accumulated into zeroed results, against fp64."""
    R = 1920
    shapes = [(16, 3072), (32, 48), (64, 96), (128, 192), (256, 384), (512, 768)]
    prods, refs = [], []
    for i, (m, n) in enumerate(shapes):
        A, B = _rand((R, m), 300 + i), _rand((R, n), 320 + i)
        C = torch.zeros((m, n), dtype=torch.float32, device=DEV)
        prods.append((A.to(DEV), B.to(DEV), C, ops.pick_split_k(m, n, R)))
        refs.append(A.double().t() @ B.double())
    ops.gemm_group_rc_f32(prods)
    for (A, B, C, sk), ref in zip(prods, refs):
        assert (C.cpu().double() - ref).abs().max().item() <= 2e-6 * R ** 0.5 * ref.abs().max().item() + 1e-6
    # a second launch accumulates
    ops.gemm_group_rc_f32(prods[:2])
    assert (prods[0][2].cpu().double() - 2 * refs[0]).abs().max().item() <= 4e-6 * R ** 0.5 * refs[0].abs().max().item() + 1e-6
    assert (prods[2][2].cpu().double() - refs[2]).abs().max().item() <= 2e-6 * R ** 0.5 * refs[2].abs().max().item() + 1e-6
