"""Forward/backward engines of the PCAA modules on the HIP path, and the
``torch.autograd.Function`` wrappers that make the drop-in modules of
:mod:`.models` behave like the reference's (``models.py`` of the reference;
see each function for file:line).

Layout: activations are point-major ``[rows, channels]`` (rows = B*T*N points
or B*T time steps).  The module-facing tensors keep the reference's logical
shapes (``[B,C,T,N]``, ``[B,C,T]``) as zero-copy permuted views.

Numerics modes (``set_precision``):
  * ``"fp32"``  -- fp32 storage, exact-fp32 MFMA: the parity mode (1e-4 gate).
  * ``"fp16x3"`` -- (round 3) parity-grade without the 1/16-rate fp32 MFMA: everything as in ``"fp32"`` except the
    PointNet layers' 2-4 products, whose fp32 operands are kept as [hi | lo] fp16 images (times a power of two
    that keeps them in fp16's normal range) and multiplied as hi.hi + lo.hi + hi.lo on the f16 MFMA pipe
    (pcaa_gemm_split3; fp32 accumulation, 22 mantissa bits per operand): passes the fp32 mode's 1e-4 /
    bit-exact-label / 5e-4-gradient gates unchanged (embedding error 2.0e-6 of scale against exact fp32's 2.5e-6 at
    the benchmarked size) at about half its step time.
  * ``"bf16"``  -- PointNet activations stored in bf16, PointNet GEMMs on the
    bf16 MFMA pipe with fp32 accumulation; everything else fp32.
"""
import contextlib
import os
import itertools

import torch
from torch.autograd.function import once_differentiable

from . import ops
from ._lib import ACT_ELU, ACT_NONE, KC, PCAA_BF16, PCAA_F32, RC

_W16_CACHE = {}           # weight data_ptr -> transposed bf16 shadow made in the forward pass of this step
_WSPLIT_T_CACHE = {}      # fp16x3 mode: weight data_ptr -> SplitImage of W^T (forward -> backward of the same step)
_PRECISION = {"mode": "fp32"}
_SYNC_BN = {"group": None}


# Optional section marks: with a list installed (set_marks), mark(name) records a timing event on the
# current stream at a few section boundaries of the step (tools/step_sections.py prints the elapsed
# times between them).  Off by default: no events, no cost.
_MARKS = None


def set_marks(lst):
    global _MARKS
    _MARKS = lst


def mark(name):
    if _MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        _MARKS.append((name, ev))


def set_precision(mode: str):
    if mode not in ("fp32", "bf16", "fp16x3"):
        raise ValueError("precision must be 'fp32', 'bf16' or 'fp16x3'")
    _PRECISION["mode"] = mode


def get_precision() -> str:
    return _PRECISION["mode"]


def set_sync_bn_group(group):
    """``None`` (per-rank BatchNorm statistics, stock DDP semantics) or a
    torch.distributed process group over which the fp64 batch statistics are
    all-reduced (SyncBN: the 8-GPU step equals the single-process global-batch
    step of the reference)."""
    _SYNC_BN["group"] = group


@contextlib.contextmanager
def sync_bn_group(group):
    """``with sync_bn_group(g):`` -- SyncBN over ``g`` inside the block, the previous setting restored after it
    (how PCAATrainer scopes its group to its own train-mode work)."""
    prev = _SYNC_BN["group"]
    _SYNC_BN["group"] = group if group is not None else prev
    try:
        yield
    finally:
        _SYNC_BN["group"] = prev


def _sync_stats(stats, count):
    g = _SYNC_BN["group"]
    if g is None:
        return count
    import torch.distributed as dist
    ev = _SYNC_BN.get("events")
    if ev is not None:
        # PCAATrainer.time_comm: these all-reduces are synchronous on the main stream -- all of their time is exposed
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dist.all_reduce(stats, group=g)
        e1.record()
        ev.append((e0, e1))
    else:
        dist.all_reduce(stats, group=g)
    _SYNC_BN["collectives"] = _SYNC_BN.get("collectives", 0) + 1          # PCAATrainer folds these into its comm record
    _SYNC_BN["payload_bytes"] = _SYNC_BN.get("payload_bytes", 0) + stats.numel() * stats.element_size()
    return count * dist.get_world_size(g)


def _sync_fn():
    """the ``sync`` argument of ops.BnTailFwd / BnTailBwd: None unless SyncBN is on (then the statistics are all-reduced
    before a stand-alone finalize; the producer cannot carry it)"""
    return _sync_stats if _SYNC_BN["group"] is not None else None


def _require_gpu(x, what):
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise RuntimeError(f"{what}: input must live on the HIP device; this package has no CPU path "
                           "(the CPU restatement lives in oracle/ and is test infrastructure only)")
    if x.numel() == 0:
        # the reference's train-mode BatchNorm raises ValueError on an empty batch ("Expected more than 1 value per
        # channel"); its eval mode would hand back empty tensors -- here both refuse (error behaviour)
        raise ValueError(f"{what}: empty input {tuple(x.shape)}")


def _point_major(x):
    """logical [B,C,T,N] -> contiguous [B,T,N,C] storage (no copy if x already
    is a permuted view of point-major storage)."""
    xp = x.permute(0, 2, 3, 1)
    if xp.is_contiguous() and x.dtype == torch.float32:
        return xp
    return ops.pack_points(x.float() if x.dtype != torch.float32 else x)


# ======================================================================
# Linear + BatchNorm + ELU stacks (PointNetBlock models.py:82-105 and
# TemporalConvolutionBlock models.py:108-160)
# ======================================================================
class _LayerSave:
    __slots__ = ("a_in", "col", "y", "scale", "shift", "mean", "rstd", "rows", "cin", "cout", "dil", "pool_e", "mom")

    def __init__(self):
        self.pool_e = None
        self.mom = None


def _linear_bn(a_in, W2d, lin_bias, bn, training, mode, first_layer):
    """y = a_in @ W2d^T (the linear bias is NOT added: it cancels in train-mode BatchNorm and
    is folded into ``shift`` in eval mode; see bn_finalize_kernel) with BatchNorm statistics;
    returns the bias-free y and the BN coefficients that apply to it."""
    rows, cin = a_in.shape
    cout = W2d.shape[0]
    stats = ops.new_stats(cout, a_in.device) if training else None
    # the finalize of these statistics rides on the launch that produces them where that launch can carry it
    # (csrc/bn_tail.h: its last workgroup writes the coefficients); else tail.resolve() ran the stand-alone kernel
    tail = ops.BnTailFwd(rows, lin_bias, bn, cout, sync=_sync_fn()) if training else None
    use_bf16 = (mode == "bf16") and a_in.dtype == torch.bfloat16 and cin % 8 == 0
    out_dtype = torch.bfloat16 if (mode == "bf16" and first_layer is not None) else torch.float32
    if isinstance(a_in, ops.SplitImage):
        # fp16x3 mode: both operands as [hi | lo] images, three f16 MFMA passes, fp32 result
        w_img = ops.split_f16(W2d)
        if training:
            _WSPLIT_T_CACHE[W2d.data_ptr()] = ops.split_f16(W2d, transpose=True)      # the dgrad's operand
        y = ops.gemm_split3(a_in, w_img, KC, rows, cout, cin, colstats=stats, tail=tail)
    elif first_layer is not None and a_in.dtype == torch.float32 and cin <= 8 and ops.pointnet_in_ok(cin, cout):
        # raw points -> first PointNet layer: C-wide contraction, HBM-bound streaming kernel
        y = ops.pointnet_in_fwd(a_in, W2d, None, out_dtype, stats, tail=tail)
    elif use_bf16:
        # bf16 shadow of the weights so both operands stream by LDS-DMA (the transposed copy
        # serves the dgrad GEMM of the backward pass)
        w16, wt16 = ops.cast_bf16(W2d, True, training)
        _W16_CACHE[W2d.data_ptr()] = wt16
        y = ops.gemm(a_in, KC, w16, KC, rows, cout, cin, colstats=stats, out_dtype=out_dtype, math=PCAA_BF16, tail=tail)
    else:
        tiles = ((rows + 127) // 128) * ((cout + 127) // 128)
        sk = ops.pick_split_k(rows, cout, cin, target_blocks=512)
        if (training and out_dtype == torch.float32 and tiles < 128 and sk > 1 and cout % 4 == 0
                and 256 % (cout // 4) == 0 and cout <= 1024):
            # few output tiles, long K (the temporal block): split K through slabs; the reduction
            # pass also produces the BatchNorm statistics the GEMM epilogue would have
            y = ops.gemm_slabs(a_in, KC, W2d, KC, rows, cout, cin, sk, math=PCAA_F32, colstats=stats, tail=tail)
        else:
            y = ops.gemm(a_in, KC, W2d, KC, rows, cout, cin, colstats=stats, out_dtype=out_dtype, math=PCAA_F32, tail=tail)
    if training:
        scale, shift, mean, rstd = tail.out
        count = tail.count_out
    else:
        scale, shift = ops.bn_eval_coeffs(bn, cout, lin_bias)
        mean = rstd = None
        count = rows
    return y, scale, shift, mean, rstd, count


def pointnet_forward(xp2d, layers, training, mode, pool_rows=0):
    """xp2d: [P, C] fp32.  Returns (output, saves).  If pool_rows>0 the last
    layer's ELU output is mean-pooled over groups of pool_rows rows (fp32
    [P/pool_rows, ch]); else the last activation [P, ch] is returned."""
    saves = []
    a = xp2d
    nl = len(layers)
    # fp16x3 mode: the activations between the layers are [hi | lo] images when every later product is one the
    # LDS-DMA kernel serves (whole 256-tiles); else the layer stack runs in exact fp32
    split = mode == "fp16x3" and nl > 1 and all(
        ops.gemm_split3_supported(xp2d.shape[0], l.module[0].weight.shape[0], l.module[0].weight.shape[1])
        and l.module[0].weight.shape[1] % 256 == 0 for l in layers[1:])
    for li, layer in enumerate(layers):
        conv, bn = layer.module[0], layer.module[1]
        cout, cin = conv.weight.shape[0], conv.weight.shape[1]
        W2d = conv.weight.view(cout, cin)
        if (a.dtype == torch.float32 and cin <= 8 and ops.pointnet_in_ok(cin, cout)
                and not (li == nl - 1 and pool_rows)):
            # raw points -> first layer: y = x.W^T costs cin FMAs per element, less than reading it back, so
            # it is never stored: statistics pass, then a = ELU(BN(y)) straight from the points
            rows = a.shape[0]
            mom = None
            if training and _MOMENT_STATS and _sync_fn() is None and ops.TAILS["enabled"]:
                # y = x.W^T is linear in the points: the layer's batch statistics follow from the points' C x C second
                # moments (a 4 MB read) -- no statistics pass over [P, cout]; the backward reuses the moments
                scale, shift, mean, rstd, mom = ops.pointnet_in_moment_coeffs(a, W2d, conv.bias, bn)
                count = rows
            elif training:
                stats = ops.new_stats(cout, a.device)
                tail = ops.BnTailFwd(rows, conv.bias, bn, cout, sync=_sync_fn())
                ops.pointnet_in_fwd(a, W2d, None, None, stats, tail=tail)
                scale, shift, mean, rstd = tail.out
                count = tail.count_out
            else:
                scale, shift = ops.bn_eval_coeffs(bn, cout, conv.bias)
                mean = rstd = None
                count = rows
            s = _LayerSave()
            s.a_in, s.col, s.y, s.scale, s.shift, s.mean, s.rstd = a, None, None, scale, shift, mean, rstd
            s.rows, s.cin, s.cout, s.dil = count, cin, cout, 0
            s.mom = mom
            saves.append(s)
            a = ops.pointnet_in_apply(a, W2d, scale, shift, torch.bfloat16 if mode == "bf16" else
                                      (ops.SplitImage.dtype if split else torch.float32))
            continue
        last_pool = li == nl - 1 and pool_rows
        if (_FUSE_EVAL_EPILOGUE and not training and mode == "bf16" and a.dtype == torch.bfloat16
                and (not last_pool or pool_rows in (32, 64, 128))
                and ops.gemm_dgrad_bn_supported(a.shape[0], cout, cin)):
            # eval mode: BatchNorm is a fixed per-channel affine map -> BN + ELU (and the mean over the frame's
            # points for the last layer) in the GEMM epilogue: no stored y, no separate pass
            scale, shift = ops.bn_eval_coeffs(bn, cout, conv.bias)
            w16, _ = ops.cast_bf16(W2d, True, False)
            a_next = ops.gemm_affine_elu(a, w16, scale, shift, pool_rows if last_pool else 0)
            s = _LayerSave()
            s.a_in, s.col, s.y, s.scale, s.shift, s.mean, s.rstd = a, None, None, scale, shift, None, None
            s.rows, s.cin, s.cout, s.dil = a.shape[0], cin, cout, 0
            saves.append(s)
            if last_pool:
                return a_next, saves
            a = a_next
            continue
        y, scale, shift, mean, rstd, count = _linear_bn(a, W2d, conv.bias, bn, training, mode, True)
        s = _LayerSave()
        s.a_in, s.col, s.y, s.scale, s.shift, s.mean, s.rstd = a, None, y, scale, shift, mean, rstd
        s.rows, s.cin, s.cout, s.dil = count, cin, cout, 0
        saves.append(s)
        if li == nl - 1 and pool_rows:
            if training:
                out, s.pool_e = ops.bn_act_meanpool_fwd(y, scale, shift, y.shape[0] // pool_rows, pool_rows, mean, rstd)
            else:
                out = ops.bn_act_meanpool_fwd(y, scale, shift, y.shape[0] // pool_rows, pool_rows)
            return out, saves
        a = ops.bn_act_fwd_split(y, scale, shift) if (split and li < nl - 1) else ops.bn_act_fwd(y, scale, shift)
    return a, saves


class _FusedGrad:
    """What the fused dgrad (ops.gemm_dgrad_bn) hands to the layer below instead of da:
    dz = da * ELU'(z) and that layer's BatchNorm-backward statistics."""
    __slots__ = ("dz", "stats", "fin")

    def __init__(self, dz, stats, fin=None):
        # fin: (coef, dgamma, dbeta) of that layer when the producing launch carried its finalize (ops.BnTailBwd)
        self.dz, self.stats, self.fin = dz, stats, fin


# Side stream for the SMALL weight-gradient products (temporal block, MLP heads): they only feed the
# optimizer at the end of the step, are 10-50 us launches on a handful of CUs, and are independent of the
# dgrad that continues the chain -- run beside it they leave the backward's critical path.  Set per step by
# the trainer (set_wgrad_stream), which joins the stream before the encoder-gradient all-reduce / Adam.
_WGRAD_STREAM = None


def set_wgrad_stream(stream):
    global _WGRAD_STREAM
    _WGRAD_STREAM = stream


class _on_wgrad_stream:
    """``with _on_wgrad_stream(t1, t2, ...)``: run the body on the wgrad side stream after everything
    enqueued so far; the listed tensors (inputs allocated on the main stream) are kept alive for it."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.ctx = None

    def __enter__(self):
        st = _WGRAD_STREAM
        if st is None:
            return self
        ev = torch.cuda.Event()
        ev.record()                      # (on the current stream)
        self.ctx = ops.on_stream(st)
        self.ctx.__enter__()
        st.wait_event(ev)
        for t in self.tensors:
            if t is not None:
                t.record_stream(st)
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


# Fusions that have an unfused fallback for the shapes their kernels do not take.  They were A/B-ed through
# environment switches in round 1 (docs/LAB_LOG.md section 4) and are plain constants now: the product has one path per shape.
# PointNet weight gradients on the wgrad side stream: measured no change (they fill the chip either way) -> off.
_BIG_WGRAD_ASIDE = False
# ... but the split-K slab reductions that follow them (12 us each, three per step, needed only by Adam) do leave it
_REDUCE_ASIDE = os.environ.get("PCAA_REDUCE_ASIDE", "1") != "0"
_FUSE_DGRAD_BN = True
# (the same fusion into the dgrad above the FIRST layer -- y rebuilt from the points in the epilogue -- measured slower in
# round 1: the layer-2 dgrad went 0.195 -> 0.361 ms to save a 0.10 ms statistics pass; it lived in the 8-wave kernel only
# and was removed with it in round 5)
# first PointNet layer, bf16 mode: one pass over the incoming gradient instead of two (ops.pointnet_in_bwd_onepass)
_ONEPASS_IN_BWD = os.environ.get("PCAA_ONEPASS_IN_BWD", "1") != "0"
# first PointNet layer, forward: BatchNorm statistics from the points' second moments instead of a pass over [P, cout]
_MOMENT_STATS = os.environ.get("PCAA_MOMENT_STATS", "1") != "0"


def _bn_layer_backward(s, bn, W2d, mode, da=None, dpool=None, group_rows=0, pool_scale=1.0,
                       need_dinput=True, lhs=None, outs=None, below=None, below_W=None, dgrad_fn=None,
                       below_bn=None, below_outs=None, wgrad_math=PCAA_F32, defer_wgrad=None):
    """Backward of one (linear, BN, ELU) layer.  ``lhs`` is the GEMM's left
    operand ([rows, K]: the input activation or the im2col matrix).  ``outs`` =
    (dW, dgamma, dbeta) destination views (dW PRE-ZEROED: the trainer's flat
    gradient buffer) or None to allocate.  ``da`` may be a _FusedGrad (the dgrad of the layer
    above already applied ELU' and reduced the statistics).  ``below``: the saved state of the
    layer that produced ``lhs`` -- if it qualifies, this layer's dgrad is fused with the first
    half of ITS backward and a _FusedGrad is returned as d_lhs.
    Returns (dW2d, dgamma, dbeta, d_lhs or None)."""
    y = s.y
    rows_local, cout = y.shape
    d_lhs = None
    dgrad_done = False
    if isinstance(da, _FusedGrad):
        stats = da.stats
        if da.fin is not None:
            coef, dgamma, dbeta = da.fin          # finalized by the launch that produced the statistics
        else:
            _sync_stats(stats, 0)
            coef, dgamma, dbeta = ops.bn_bwd_finalize(stats, s.rows, bn, s.mean, s.rstd, cout,
                                                      dgamma=outs[1] if outs else None, dbeta=outs[2] if outs else None)
        if need_dinput and dgrad_fn is not None and getattr(dgrad_fn, "forms_dy", False):
            # the adjoint kernel forms dy = c0*dz + c1*y + c2 while it stages its operand and hands it back
            # for the weight gradient: no separate elementwise pass
            d_lhs, dy = dgrad_fn(None, dz=da.dz, y=y, coef=coef)
            dgrad_done = True
        elif isinstance(lhs, ops.SplitImage):
            # fp16x3 mode: dz (fp32) came out of the split-operand dgrad's epilogue; dy only feeds the two products below
            dy = ops.bn_bwd_dy_split(da.dz, y, coef)
        else:
            dy = ops.bn_bwd_dy(da.dz, y, coef, out=da.dz)
    else:
        # pass 1: statistics only (reads da/dpool and y, writes nothing); pass 2: dy directly, with
        # dz = da*ELU'(z) recomputed in registers -- dz is never materialised
        if da is not None and da.dtype != y.dtype:
            da = da.to(y.dtype)
        pool_e = getattr(s, "pool_e", None)
        tail = ops.BnTailBwd(s.rows, bn, s.mean, s.rstd, cout, dgamma=outs[1] if outs else None,
                             dbeta=outs[2] if outs else None, sync=_sync_fn())
        if dpool is not None and pool_e is not None:
            # mean-pooled layer: the statistics follow from the forward's per-group sums, y is not re-read
            ops.bn_pool_bwd_stats(dpool, pool_e, pool_scale, tail=tail)
        else:
            ops.bn_act_bwd_stats(y, s.scale, s.shift, s.mean, s.rstd, da=da, dpool=dpool,
                                 group_rows=group_rows, pool_scale=pool_scale, tail=tail)
        coef, dgamma, dbeta = tail.out
        if isinstance(lhs, ops.SplitImage):
            # fp16x3 mode: dy only feeds the two products below -> written as its [hi | lo] image
            dy = ops.bn_bwd_dy_fused_split(y, s.scale, s.shift, coef, da=da, dpool=dpool, group_rows=group_rows,
                                           pool_scale=pool_scale)
        else:
            dy = ops.bn_bwd_dy_fused(y, s.scale, s.shift, coef, da=da, dpool=dpool, group_rows=group_rows,
                                     pool_scale=pool_scale, out=da)
    K = lhs.shape[1]
    dW_out = outs[0].view(cout, K) if outs else None
    # dW[cout, K] = dy^T . lhs   (contraction over the rows: both operands row-contiguous)
    wgrad_bf16 = (mode == "bf16" and dy.dtype == torch.bfloat16 and lhs.dtype == torch.bfloat16
                  and cout >= 256 and K >= 128 and cout % 8 == 0 and K % 8 == 0)
    if isinstance(dy, ops.SplitImage):
        sk = ops.pick_split_k(cout, K, 3 * rows_local, target_blocks=256, bk=64, tile=256)
        dW = ops.gemm_slabs_split3(dy, lhs, cout, K, rows_local, sk, out=dW_out)
    elif lhs.dtype == torch.float32 and K <= 8 and s.col is None and ops.pointnet_in_ok(K, cout):
        dW = ops.pointnet_in_wgrad(dy, lhs, out=dW_out, out_is_zero=True)
    elif wgrad_bf16:
        sk = ops.pick_split_k(cout, K, rows_local, target_blocks=256, bk=64, tile=256)
        with _on_wgrad_stream(dy, lhs) if (_BIG_WGRAD_ASIDE and dW_out is not None) else contextlib.nullcontext():
            if sk > 1:
                # the slab sum only feeds the optimizer: off the main stream (joined before the encoder's Adam)
                dW = ops.gemm_slabs(dy, RC, lhs, RC, cout, K, rows_local, sk, out=dW_out, math=PCAA_BF16,
                                    reduce_ctx=_on_wgrad_stream if (_REDUCE_ASIDE and dW_out is not None) else None)
            else:
                dW = ops.gemm(dy, RC, lhs, RC, cout, K, rows_local, out=dW_out, math=PCAA_BF16)
    else:
        # (wgrad_math = PCAA_BF16: the temporal block's weight gradients in the bf16 throughput mode -- the fp32 operands
        # are rounded to bf16 on their way into the fragments of the 256x256-tile kernel; fp32 accumulate)
        big = wgrad_math == PCAA_BF16 and cout >= 128 and K >= 128 and cout % 8 == 0
        sk = ops.pick_split_k(cout, K, rows_local, target_blocks=256, bk=64, tile=256) if big else ops.pick_split_k(cout, K, rows_local)
        wm = PCAA_BF16 if big else PCAA_F32
        # the grouped launch stages 16-B quads of both operands: cout % 4, K % 4 and 16-B aligned bases (ADVICE round 4:
        # channel counts outside the product's own, e.g. DilTempConv1d(6, 10), keep the per-layer product below)
        groupable = (cout % 4 == 0 and K % 4 == 0 and dy.data_ptr() % 16 == 0 and lhs.data_ptr() % 16 == 0
                     and (dW_out is None or dW_out.data_ptr() % 16 == 0))
        if (defer_wgrad is not None and not big and groupable and dy.dtype == torch.float32
                and lhs.dtype == torch.float32):
            # the caller launches this product together with its siblings (dtc_backward: one grouped launch)
            dW = dW_out if dW_out is not None else torch.zeros((cout, K), dtype=torch.float32, device=dy.device)
            defer_wgrad.append((dy, lhs, dW, sk))
        elif dW_out is not None:
            with _on_wgrad_stream(dy, lhs):
                dW = ops.gemm(dy, RC, lhs, RC, cout, K, rows_local, out=dW_out, split_k=sk, accumulate=True, math=wm)
        else:
            dW = ops.gemm(dy, RC, lhs, RC, cout, K, rows_local, out=dW_out, split_k=sk, accumulate=sk > 1, math=wm)
    if dgrad_done:
        pass
    elif need_dinput and dgrad_fn is not None:
        d_lhs = dgrad_fn(dy)              # the caller's own adjoint (temporal block: implicit col2im)
        if isinstance(d_lhs, tuple):
            d_lhs = d_lhs[0]
    elif need_dinput and isinstance(dy, ops.SplitImage):
        wt_img = _WSPLIT_T_CACHE.pop(W2d.data_ptr(), None)            # [K, 2 cout] image of W^T made by the forward
        if wt_img is None or tuple(wt_img.shape) != (K, cout):
            wt_img = ops.split_f16(W2d, transpose=True)
        if (_FUSE_DGRAD_BN and below is not None and below.y is not None and below.y.dtype == torch.float32
                and below.mean is not None and tuple(below.y.shape) == (rows_local, K)
                and ops.gemm_dgrad_bn_split3_supported(rows_local, K, cout)):
            # as in the bf16 mode: ELU' and the BatchNorm-backward statistics of the layer below in the dgrad's epilogue
            # (round 3; before, this mode re-read da and y in a statistics pass of its own: 0.6 ms per step)
            btail = None
            if below_bn is not None:
                btail = ops.BnTailBwd(below.rows, below_bn, below.mean, below.rstd, K,
                                      dgamma=below_outs[1] if below_outs else None,
                                      dbeta=below_outs[2] if below_outs else None, sync=_sync_fn())
            d_lhs = _FusedGrad(*ops.gemm_dgrad_bn_split3(dy, wt_img, below.y, below.scale, below.shift, below.mean,
                                                         below.rstd, tail=btail), fin=btail.out if btail else None)
        else:
            d_lhs = ops.gemm_split3(dy, wt_img, KC, rows_local, K, cout)
    elif need_dinput:
        if mode == "bf16" and dy.dtype == torch.bfloat16 and cout % 8 == 0:
            Wt = _W16_CACHE.pop(W2d.data_ptr(), None)      # bf16 [K, cout] made by the forward pass
            if Wt is None or tuple(Wt.shape) != (K, cout):
                _, Wt = ops.cast_bf16(W2d, False, True)
            if (_FUSE_DGRAD_BN and below is not None and below.y is not None and below.y.dtype == torch.bfloat16
                    and below.mean is not None and tuple(below.y.shape) == (rows_local, K)
                    and ops.gemm_dgrad_bn_supported(rows_local, K, cout)):
                # dgrad fused with ELU' and the BatchNorm-backward statistics of the layer below (and, when the caller
                # handed that layer's BatchNorm over, their finalize)
                btail = None
                if below_bn is not None:
                    btail = ops.BnTailBwd(below.rows, below_bn, below.mean, below.rstd, K,
                                          dgamma=below_outs[1] if below_outs else None,
                                          dbeta=below_outs[2] if below_outs else None, sync=_sync_fn())
                d_lhs = _FusedGrad(*ops.gemm_dgrad_bn(dy, Wt, below.y, below.scale, below.shift, below.mean,
                                                      below.rstd, tail=btail), fin=btail.out if btail else None)
            else:
                d_lhs = ops.gemm(dy, KC, Wt, KC, rows_local, K, cout, out_dtype=torch.bfloat16, math=PCAA_BF16)
        else:
            d_lhs = ops.gemm(dy, KC, W2d, RC, rows_local, K, cout,
                             out_dtype=torch.float32)
    return dW, dgamma, dbeta, d_lhs


def _layer_outs(gout, prefix, wname, gname, bname):
    if gout is None:
        return None
    return gout[prefix + wname], gout[prefix + gname], gout[prefix + bname]


def pointnet_backward(saves, layers, mode, d_last=None, dpool=None, pool_rows=0, need_dx=False, gout=None,
                      prefix="pc_block.pointnet"):
    """Returns ({param_name_suffix: grad} per layer list, dx2d or None).  ``gout``:
    optional {state_dict name: pre-zeroed gradient view} to write into."""
    grads = []
    da = d_last
    for li in range(len(layers) - 1, -1, -1):
        layer, s = layers[li], saves[li]
        conv, bn = layer.module[0], layer.module[1]
        W2d = conv.weight.view(s.cout, s.cin)
        need_in = li > 0 or need_dx
        outs = _layer_outs(gout, f"{prefix}{li + 1}.", "module.0.weight", "module.1.weight", "module.1.bias")
        # the layer below: its BatchNorm and gradient destinations, for the finalize the fused dgrad can carry
        below_bn = layers[li - 1].module[1] if li > 0 else None
        below_outs = _layer_outs(gout, f"{prefix}{li}.", "module.0.weight", "module.1.weight", "module.1.bias") if li > 0 else None
        below_W = None
        if s.y is None and da is not None and not need_in:
            # recompute path of the first layer: two passes over da (one, when the dgrad above already applied
            # ELU' and reduced the statistics), nothing else is read or written
            fused = isinstance(da, _FusedGrad)
            if fused and da.fin is not None:
                coef, dg, db = da.fin
            elif fused:
                _sync_stats(da.stats, 0)
                coef, dg, db = ops.bn_bwd_finalize(da.stats, s.rows, bn, s.mean, s.rstd, s.cout,
                                                   dgamma=outs[1] if outs else None, dbeta=outs[2] if outs else None)
            else:
                tail = ops.BnTailBwd(s.rows, bn, s.mean, s.rstd, s.cout, dgamma=outs[1] if outs else None,
                                     dbeta=outs[2] if outs else None, sync=_sync_fn())
                if (_ONEPASS_IN_BWD and (da.dtype == torch.bfloat16 or (mode == "fp16x3" and da.dtype == torch.float32))
                        and outs is not None and outs[0].is_contiguous()):
                    # bf16 and fp16x3 modes: statistics and G = dz^T.x from ONE read of da; the weight gradient is then a
                    # combination of G with the points' second moments (exact-fp32 mode keeps the two passes: there
                    # dy is formed per element before the contraction, as the oracle's autograd does)
                    dW = ops.pointnet_in_bwd_onepass(da, s.a_in, W2d, s.scale, s.shift, s.mean, s.rstd, tail,
                                                     mom=s.mom, out=outs[0].view(s.cout, s.cin))
                    coef, dg, db = tail.out
                    zb = gout[f"{prefix}{li + 1}.module.0.bias"] if gout is not None else torch.zeros_like(conv.bias)
                    grads.append({"module.0.weight": dW.view_as(conv.weight), "module.0.bias": zb,
                                  "module.1.weight": dg, "module.1.bias": db})
                    da = None
                    continue
                ops.pointnet_in_bwd_stats(da, s.a_in, W2d, s.scale, s.shift, s.mean, s.rstd, tail=tail)
                coef, dg, db = tail.out
            dW = ops.pointnet_in_bwd_wgrad(da.dz if fused else da, s.a_in, W2d, s.scale, s.shift, coef,
                                           out=outs[0].view(s.cout, s.cin) if outs else None, out_is_zero=True,
                                           dz_is_pre=fused)
            dprev = None
        elif li == len(layers) - 1 and dpool is not None:
            dW, dg, db, dprev = _bn_layer_backward(s, bn, W2d, mode, dpool=dpool, group_rows=pool_rows,
                                                   pool_scale=1.0 / pool_rows, need_dinput=need_in, lhs=s.a_in,
                                                   outs=outs, below=saves[li - 1] if li > 0 else None,
                                                   below_W=below_W, below_bn=below_bn, below_outs=below_outs)
        else:
            if s.y is None:      # recompute layer, but the caller wants the gradient w.r.t. the points: rebuild y
                s.y = ops.pointnet_in_fwd(s.a_in, W2d, None, da.dtype)
            dW, dg, db, dprev = _bn_layer_backward(s, bn, W2d, mode, da=da, need_dinput=need_in, lhs=s.a_in,
                                                   outs=outs, below=saves[li - 1] if li > 0 else None,
                                                   below_W=below_W, below_bn=below_bn, below_outs=below_outs)
        # the conv bias gradient is analytically zero (BatchNorm removes the mean)
        zb = gout[f"{prefix}{li + 1}.module.0.bias"] if gout is not None else torch.zeros_like(conv.bias)
        grads.append({"module.0.weight": dW.view_as(conv.weight), "module.0.bias": zb,
                      "module.1.weight": dg, "module.1.bias": db})
        da = dprev
    grads.reverse()
    return grads, da


_FUSE_DTC = True
_FUSE_DTC_BWD = True


# The temporal block's products in the bf16 throughput mode (PCAA_DTC_BF16): "wide" (default) = the bf16 matrix pipe where
# the two-sequence kernels take the layer (contraction over >= 128 channels, >= 64 output channels: layers 5 and 6 and
# their adjoints -- csrc/dtc_fused.hip, dtc_pair_kernel), exact fp32 products elsewhere; "all" = the one-sequence bf16
# variants as well (measured no faster than their fp32 forms: docs/LAB_LOG.md section 9); "0" = fp32 products everywhere.
_DTC_BF16 = os.environ.get("PCAA_DTC_BF16", "wide")


def _dtc_bf16(mode, kc, nc, adj=False):
    """kc: channels the product contracts over, nc: channels it produces (forward: cin, cout; adjoint: cout, cin) --
    the same rule as pair_takes() in csrc/dtc_fused.hip"""
    if mode != "bf16" or _DTC_BF16 == "0":
        return False
    if _DTC_BF16 in ("all", "1"):
        return True
    which = os.environ.get("PCAA_DTC_PAIR", "1")[:1]
    on = which not in ("0", "a" if not adj else "f")
    return on and kc >= 128 and kc % 32 == 0 and nc >= 64 and nc % 4 == 0


def dtc_forward(a2d, B, T, layers, training, pool_time, mode="fp32"):
    """a2d: [B*T, Cin] fp32 rows (b,t).  Causal dilated conv = (implicit) im2col + contraction."""
    saves = []
    a = a2d
    nl = len(layers)
    fused = _FUSE_DTC and a2d.dtype == torch.float32 and all(
        ops.dtc_conv_supported(T, l.conv1d.weight.shape[1], l.conv1d.weight.shape[0]) for l in layers)
    prev = None          # (scale, shift) of the layer whose bias-free output `a` is (fused path)
    for li, layer in enumerate(layers):
        conv, bn = layer.conv1d, layer.batch_norm
        cout, cin = conv.weight.shape[0], conv.weight.shape[1]
        W2d = conv.weight.view(cout, cin * 3)
        if fused:
            # one launch: implicit im2col + BN/ELU of the previous layer on load + contraction + statistics
            stats = ops.new_stats(cout, a.device) if training else None
            tail = ops.BnTailFwd(a.shape[0], conv.bias, bn, cout, sync=_sync_fn()) if training else None
            y, col = ops.dtc_conv_fwd(a, prev[0] if prev else None, prev[1] if prev else None, W2d, B, T,
                                      layer.dilation, stats=stats, want_col=training, tail=tail,
                                      bf16=_dtc_bf16(mode, cin, cout))
            if training:
                scale, shift, mean, rstd = tail.out
                count = tail.count_out
            else:
                scale, shift = ops.bn_eval_coeffs(bn, cout, conv.bias)
                mean = rstd = None
                count = y.shape[0]
            a_in = None
        else:
            col = ops.dtc_im2col(a, B, T, cin, layer.dilation)
            y, scale, shift, mean, rstd, count = _linear_bn(col, W2d, conv.bias, bn, training, "fp32", None)
            a_in = a
        s = _LayerSave()
        s.a_in, s.col, s.y, s.scale, s.shift, s.mean, s.rstd = a_in, col, y, scale, shift, mean, rstd
        s.rows, s.cin, s.cout, s.dil = count, cin, cout, layer.dilation
        saves.append(s)
        if li == nl - 1 and pool_time:
            if training:
                out, s.pool_e = ops.bn_act_meanpool_fwd(y, scale, shift, B, T, mean, rstd)
                return out, saves
            return ops.bn_act_meanpool_fwd(y, scale, shift, B, T), saves
        if fused and li < nl - 1:
            a, prev = y, (scale, shift)
        else:
            a = ops.bn_act_fwd(y, scale, shift)
    return a, saves


# measured (round 4, same box, 2 x 3 windows of 20 steps): N=128 5.64 / 5.56 on, 5.57 / 5.58 off; N=32 2.16 / 2.16 on, 2.13 / 2.09 off -- the
# 256x256-tile kernel with an atomic epilogue is no better than the exact-fp32 128x128 one on these small products: off
_DTC_WGRAD_BF16 = os.environ.get("PCAA_DTC_WGRAD_BF16", "0") != "0"


# the temporal block's weight gradients: "wg" = one grouped launch on the weight-gradient stream, "main" = on the calling
# stream, "0" = one launch per layer (rounds 1-3)
_DTC_WGRAD_GROUP = os.environ.get("PCAA_DTC_WGRAD_GROUP", "wg")


def dtc_backward(saves, layers, B, T, d_last=None, dpool=None, need_dx=True, gout=None, prefix="tc_block.dtc", mode="fp32"):
    grads = []
    wmath = PCAA_BF16 if (mode == "bf16" and _DTC_WGRAD_BF16) else PCAA_F32
    defer = [] if _DTC_WGRAD_GROUP != "0" else None
    da = d_last
    for li in range(len(layers) - 1, -1, -1):
        layer, s = layers[li], saves[li]
        conv, bn = layer.conv1d, layer.batch_norm
        W2d = conv.weight.view(s.cout, s.cin * 3)
        need_in = li > 0 or need_dx
        outs = _layer_outs(gout, f"{prefix}{li + 1}.", "conv1d.weight", "batch_norm.weight", "batch_norm.bias")
        fused = _FUSE_DTC and _FUSE_DTC_BWD and T <= 32 and s.cin % 4 == 0 and s.cout % 4 == 0
        # the adjoint w.r.t. the layer input in one launch (implicit col2im), with the first half of the
        # BatchNorm+ELU backward of the layer below in its epilogue (dz and the two column sums: that layer then
        # starts at bn_bwd_finalize); else dcol = dy . W, then col2im
        dgrad_fn = None
        if fused:
            sb = saves[li - 1] if (li > 0 and s.cout <= 512 and saves[li - 1].mean is not None) else None
            sb_bn = layers[li - 1].batch_norm if sb is not None else None
            sb_outs = _layer_outs(gout, f"{prefix}{li}.", "conv1d.weight", "batch_norm.weight", "batch_norm.bias") \
                if sb is not None else None

            def dgrad_fn(dy, dz=None, y=None, coef=None, W2d=W2d, s=s, sb=sb, sb_bn=sb_bn, sb_outs=sb_outs):
                btail = None
                if sb is not None:
                    btail = ops.BnTailBwd(sb.rows, sb_bn, sb.mean, sb.rstd, s.cin,
                                          dgamma=sb_outs[1] if sb_outs else None, dbeta=sb_outs[2] if sb_outs else None,
                                          sync=_sync_fn())
                out, stats, dy_used = ops.dtc_conv_dgrad(
                    dy, W2d, B, T, s.cin, s.dil, dz=dz, y=y, coef=coef, want_dy=dy is None,
                    below=(sb.y, sb.scale, sb.shift, sb.mean, sb.rstd) if sb else None, tail=btail,
                    bf16=_dtc_bf16(mode, s.cout, s.cin, adj=True))
                return (_FusedGrad(out, stats, fin=btail.out) if sb else out), dy_used

            dgrad_fn.forms_dy = s.cout <= 512
        if li == len(layers) - 1 and dpool is not None:
            dW, dg, db, dcol = _bn_layer_backward(s, bn, W2d, "fp32", dpool=dpool, group_rows=T,
                                                  pool_scale=1.0 / T, need_dinput=need_in, lhs=s.col, outs=outs,
                                                  dgrad_fn=dgrad_fn, wgrad_math=wmath, defer_wgrad=defer)
        else:
            dW, dg, db, dcol = _bn_layer_backward(s, bn, W2d, "fp32", da=da, need_dinput=need_in, lhs=s.col,
                                                  outs=outs, dgrad_fn=dgrad_fn, wgrad_math=wmath, defer_wgrad=defer)
        zb = gout[f"{prefix}{li + 1}.conv1d.bias"] if gout is not None else torch.zeros_like(conv.bias)
        grads.append({"conv1d.weight": dW.view_as(conv.weight), "conv1d.bias": zb,
                      "batch_norm.weight": dg, "batch_norm.bias": db})
        if not need_in:
            da = None
        elif fused:
            da = dcol                       # already the gradient w.r.t. the layer's input
        else:
            da = ops.dtc_col2im(dcol, B, T, s.cin, s.dil)
    if defer:
        # the block's weight gradients dW_l = dy_l^T . col_l, all in one launch behind the dgrad chain (round 4: they were
        # six launches of 26-47 us in a chain beside it; measured without them the N = 32 step was 0.17 ms shorter)
        tensors = [t for d in defer for t in d[:3]]
        for k in range(0, len(defer), 8):
            if _DTC_WGRAD_GROUP == "main":
                ops.gemm_group_rc_f32(defer[k:k + 8])
            else:
                with _on_wgrad_stream(*tensors):
                    ops.gemm_group_rc_f32(defer[k:k + 8])
    grads.reverse()
    return grads, da


# ======================================================================
# small dense layers (Linear + ELU): MLP heads, projection heads, decoder
# ======================================================================
def _wide_bf16(mode, M, N, K):
    """Route a dense layer through the 256x256-tile bf16 MFMA kernel: only the
    weight-streaming decoder layers qualify (bf16 throughput mode, wide, 8-aligned)."""
    return mode == "bf16" and N >= 512 and K >= 512 and N % 8 == 0 and K % 8 == 0 and M % 8 == 0


def _skinny(mode, M, N, K):
    """Batch-skinny streaming kernels (gemm_skinny.hip): bf16 throughput mode, M <= 64 rows."""
    return mode == "bf16" and ops.skinny_supported(M, N, K)


def _skinny_exact(mode, M, N, K):
    """The same kernels with fp32 products (the parity modes): round 3 -- before, those modes ran the decoder on the
    128x128-tile fp32 GEMM (half of every tile padding; the weight gradient, a 64-deep contraction, 2-3 ms per step)."""
    return mode in ("fp32", "fp16x3") and ops.skinny_supported(M, N, K)


def linear_act_forward(x, lin, act, mode="fp32", W16=None):
    """act(x @ W^T + b); x [M,K] fp32 -> [M,N] fp32.  fp32 MFMA, or (bf16 mode,
    wide layers) bf16 MFMA with fp32 accumulation: the layer is bound by
    streaming W from HBM either way, the bf16 pipe just keeps the MFMA time out
    of the way of the stream."""
    M, K = x.shape
    N = lin.weight.shape[0]
    if _skinny(mode, M, N, K):
        # W16: the weight's bf16 image, passed by a caller that vouches for it (PCAATrainer._refresh_w16): same result
        return ops.skinny_linear_fwd(x, lin.weight, lin.bias, act, W16=W16)
    if _skinny_exact(mode, M, N, K):
        return ops.skinny_linear_fwd(x, lin.weight, lin.bias, act, exact=True)
    if _wide_bf16(mode, M, N, K):
        sk = ops.pick_split_k(M, N, K, target_blocks=256, bk=64, tile=256)
        y = ops.gemm(x, KC, lin.weight, KC, M, N, K, split_k=sk, accumulate=True, math=PCAA_BF16)
        return ops.bias_act_(y, lin.bias, act)
    sk = ops.pick_split_k(M, N, K)
    if sk > 1:
        y = ops.gemm(x, KC, lin.weight, KC, M, N, K, split_k=sk, accumulate=True)
        return ops.bias_act_(y, lin.bias, act)
    y = ops.gemm(x, KC, lin.weight, KC, M, N, K, bias=lin.bias if act == ACT_NONE else None)
    if act != ACT_NONE:
        ops.bias_act_(y, lin.bias, act)
    return y


def linear_act_backward(x, a_out, lin, act, d_out, need_dx=True, dW_out=None, db_out=None, dx_init=None,
                        mode="fp32", d_is_pre=False, fuse_elu_in=False, update=None, W16=None, defer_db=None):
    """d_out is the gradient w.r.t. the layer output (d_is_pre: already w.r.t. its
    pre-activation).  Returns (dW, db, dx).  fuse_elu_in (skinny path only, x = ELU output of
    the layer below): dx is multiplied by ELU'(x), i.e. it is the gradient w.r.t. that layer's
    pre-activation -- the caller passes it on with d_is_pre=True.
    ``update`` (skinny path only): callable ``update(dz, x)`` that forms the weight gradient AND applies the
    optimizer to the weight in one kernel (ops.skinny_linear_wgrad_adam_) LATER -- the callback itself must not touch the
    weight: it is called before the layer's dgrad, which reads it -- and dW is returned as None.
    ``defer_db`` (a list; with ``update`` and a ``db_out`` destination): the bias gradient's column sum is not launched here
    but noted as (dz, db_out) for the caller to launch later (decoder_backward: all of them behind ONE hand-over to the
    weight-gradient stream -- every hand-over is an event record on this stream, ~7 us of idle queue between two kernels)."""
    M, K = x.shape
    N = lin.weight.shape[0]
    dz = ops.elu_bwd_from_out(d_out, a_out) if (act == ACT_ELU and not d_is_pre) else d_out
    dz2 = dz.view(M, N)
    if update is not None:
        if not (_skinny(mode, M, N, K) or _skinny_exact(mode, M, N, K)):
            raise RuntimeError("linear_act_backward: a fused weight update is only served by the skinny path")
        if defer_db is not None and db_out is not None and _WGRAD_STREAM is not None:
            defer_db.append(((dz2,), lambda dz2=dz2, db_out=db_out: ops.colsum(dz2, out=db_out)))
            db = db_out
        else:
            with _on_wgrad_stream(dz2):
                db = ops.colsum(dz2, out=db_out)
        # the callback only NOTES the operands (single process: the kernels run later, on the Adam side stream, once the whole
        # decoder backward -- whose dgrads read the weights they overwrite -- is enqueued) or packs and sends them (data
        # parallel); nothing in it writes the weight, so it runs ahead of this layer's dgrad and the pack + all-gather of
        # the data-parallel step leave while the dgrad streams the weight
        update(dz2, x)
        dx = None
        if need_dx:
            dx = ops.skinny_linear_dgrad(dz2, lin.weight, a_prev=x if fuse_elu_in else None, out=dx_init,
                                         accumulate=dx_init is not None, exact=_skinny_exact(mode, M, N, K),
                                         W16=W16)
        return None, db, dx
    exact = _skinny_exact(mode, M, N, K)
    if _skinny(mode, M, N, K) or exact:
        if dW_out is not None and db_out is not None and defer_db is not None and _WGRAD_STREAM is not None:
            # (decoder_backward's one hand-over to the weight-gradient stream takes this layer's two products as well)
            def both(dz2=dz2, x=x, db_out=db_out, dW_out=dW_out, exact=exact):
                ops.colsum(dz2, out=db_out)
                ops.skinny_linear_wgrad(dz2, x, out=dW_out, exact=exact)
            defer_db.append(((dz2, x), both))
            db, dW = db_out, dW_out
        elif dW_out is not None and db_out is not None:
            # the weight-gradient write stream (and the bias gradient) beside the dgrad read stream of the same layer
            with _on_wgrad_stream(dz2, x):
                db = ops.colsum(dz2, out=db_out)
                dW = ops.skinny_linear_wgrad(dz2, x, out=dW_out, exact=exact)
        else:
            db = ops.colsum(dz2, out=db_out)
            dW = ops.skinny_linear_wgrad(dz2, x, out=dW_out, exact=exact)
        dx = None
        if need_dx:
            dx = ops.skinny_linear_dgrad(dz2, lin.weight, a_prev=x if fuse_elu_in else None, out=dx_init,
                                         accumulate=dx_init is not None, exact=exact, W16=W16)
        return dW, db, dx
    if fuse_elu_in:
        raise RuntimeError("linear_act_backward: fuse_elu_in is only served by the skinny path")
    db = ops.colsum(dz2, out=db_out)
    wide = _wide_bf16(mode, M, N, K)
    if dW_out is not None and not wide:
        with _on_wgrad_stream(dz2, x):
            dW = ops.gemm(dz2, RC, x, RC, N, K, M, out=dW_out, math=PCAA_F32)
    else:
        dW = ops.gemm(dz2, RC, x, RC, N, K, M, out=dW_out, math=PCAA_BF16 if wide else PCAA_F32)
    dx = None
    if need_dx:
        if wide:
            sk = ops.pick_split_k(M, K, N, target_blocks=256, bk=64, tile=256)
            math = PCAA_BF16
        else:
            sk = ops.pick_split_k(M, K, N)
            math = PCAA_F32
        if dx_init is not None:
            dx = ops.gemm(dz2, KC, lin.weight, RC, M, K, N, out=dx_init, split_k=sk, accumulate=True, math=math)
        else:
            dx = ops.gemm(dz2, KC, lin.weight, RC, M, K, N, split_k=sk, accumulate=sk > 1 or wide, math=math)
    return dW, db, dx


# ======================================================================
# CGEncoder (models.py:232-292)
# ======================================================================
class EncoderState:
    pass


_FUSE_HEADS = True
_FUSE_EVAL_EPILOGUE = True


def _heads_mods(enc, gph):
    l1 = enc.MLP_sup1[0]
    lh = enc.MLP_head[0] if enc.use_projection_head else None
    l2 = enc.MLP_sup2[0]
    lg = gph[0] if gph is not None else None
    return l1, lh, l2, lg


def _heads_fusable(enc, gph, B, backward):
    if not _FUSE_HEADS:
        return False
    l1, lh, l2, lg = _heads_mods(enc, gph)
    return ops.heads_supported(B, l2.weight.shape[0], l1.weight.shape[1], l1.weight.shape[0],
                               lh.weight.shape[0] if lh is not None else 0,
                               lg.weight.shape[0] if lg is not None else 0, backward) and \
        l2.weight.shape[1] == (lh.weight.shape[0] if lh is not None else l1.weight.shape[0]) and \
        (lg is None or lg.weight.shape[1] == l1.weight.shape[0]) and \
        (lh is None or lh.weight.shape[1] == l1.weight.shape[0])


def encoder_forward(enc, x, training, mode=None, gph=None):
    """``gph``: optional decoder projection head ``Sequential(Linear(32,64), ELU)`` evaluated in the
    same launch as the MLP heads (``st.hproj``)."""
    mode = get_precision() if mode is None else mode
    _require_gpu(x, "CGEncoder")
    if x.dim() != 4:
        raise ValueError(f"CGEncoder expects [B,C,T,N], got {tuple(x.shape)}")
    B, C, T, N = x.shape
    l1 = enc.pc_block.pointnet1.module[0]
    if C != l1.weight.shape[1]:
        raise RuntimeError(f"CGEncoder: input has {C} features, first layer expects {l1.weight.shape[1]}")
    if N != enc.nmax_points:
        raise RuntimeError(f"CGEncoder: N={N} points but nmax_points={enc.nmax_points} "
                           "(the reference's AvgPool2d((1,nmax_points)) would emit >1 column)")
    st = EncoderState()
    st.B, st.C, st.T, st.N, st.mode, st.training = B, C, T, N, mode, training
    xp = _point_major(x).view(B * T * N, C)
    st.xp = xp
    mark("enc_fwd.begin")
    x2, st.pn = pointnet_forward(xp, enc.pc_block.layers(), training, mode, pool_rows=N)   # [B*T, 1024]
    st.x2 = x2
    mark("enc_fwd.pointnet")
    x4, st.dtc = dtc_forward(x2, B, T, enc.tc_block.layers(), training, pool_time=True, mode=mode)     # [B, 512]
    st.x4 = x4
    mark("enc_fwd.dtc")
    st.h = st.hproj = None
    if _heads_fusable(enc, gph, B, False):
        m1, mh, m2, mg = _heads_mods(enc, gph)
        st.sup_fv, st.h, st.logits, st.hproj = ops.heads_fwd(
            x4, m1.weight, m1.bias, mh.weight if mh is not None else None, mh.bias if mh is not None else None,
            m2.weight, m2.bias, mg.weight if mg is not None else None, mg.bias if mg is not None else None)
        return st.logits, st.sup_fv, st
    st.sup_fv = linear_act_forward(x4, enc.MLP_sup1[0], ACT_ELU)
    h = st.sup_fv
    if enc.use_projection_head:
        st.h = linear_act_forward(st.sup_fv, enc.MLP_head[0], ACT_ELU)
        h = st.h
    st.logits = linear_act_forward(h, enc.MLP_sup2[0], ACT_ELU)
    if gph is not None:
        st.hproj = linear_act_forward(st.sup_fv, gph[0], ACT_ELU)
    return st.logits, st.sup_fv, st


def encoder_backward(enc, st, d_logits, d_supfv, need_dx=False, gout=None, before_pointnet=None,
                     gph=None, d_hproj=None, gph_gout=None, after_heads=None):
    """Returns ({state_dict-style name: grad}, dx [B,C,T,N] view or None).
    ``gout``: optional {name: gradient view}; split-K products accumulate into
    them, so they must arrive ZEROED (the trainer zeroes its flat buffer once).
    ``gph`` / ``d_hproj`` / ``gph_gout``: the decoder projection head evaluated by encoder_forward(gph=...),
    the gradient w.r.t. its output and optional (dW, db) destinations: its backward runs in the heads'
    launch; its gradients are returned under "GPH.0.weight" / "GPH.0.bias"."""
    if not st.training:
        raise RuntimeError("CGEncoder backward in eval mode is not implemented on the HIP path "
                           "(the reference only differentiates the train-mode encoder)")
    g = {}
    B, T, N = st.B, st.T, st.N

    def dst(name):
        return (gout[name + ".weight"], gout[name + ".bias"]) if gout is not None else (None, None)

    if d_hproj is not None and (gph is None or st.hproj is None):
        raise RuntimeError("encoder_backward: d_hproj needs the head passed to encoder_forward(gph=...)")
    if _heads_fusable(enc, gph if d_hproj is not None else None, B, True):
        m1, mh, m2, mg = _heads_mods(enc, gph if d_hproj is not None else None)
        outs = {}
        outs["dW1"], outs["db1"] = dst("MLP_sup1.0")
        outs["dW2"], outs["db2"] = dst("MLP_sup2.0")
        if mh is not None:
            outs["dWh"], outs["dbh"] = dst("MLP_head.0")
        if mg is not None and gph_gout is not None:
            outs["dWg"], outs["dbg"] = gph_gout
        outs, dx4 = ops.heads_bwd(
            st.x4, st.sup_fv, st.h, st.logits, st.hproj if mg is not None else None, m1.weight,
            mh.weight if mh is not None else None, m2.weight, mg.weight if mg is not None else None,
            d_logits.contiguous() if d_logits is not None else None,
            d_supfv.contiguous() if d_supfv is not None else None,
            d_hproj.contiguous() if mg is not None else None, outs)
        g["MLP_sup1.0.weight"], g["MLP_sup1.0.bias"] = outs["dW1"], outs["db1"]
        g["MLP_sup2.0.weight"], g["MLP_sup2.0.bias"] = outs["dW2"], outs["db2"]
        if mh is not None:
            g["MLP_head.0.weight"], g["MLP_head.0.bias"] = outs["dWh"], outs["dbh"]
        if mg is not None:
            g["GPH.0.weight"], g["GPH.0.bias"] = outs["dWg"], outs["dbg"]
    else:
        dsup = d_supfv.contiguous().clone() if d_supfv is not None else torch.zeros_like(st.sup_fv)
        if d_hproj is not None:
            gw, gb = gph_gout if gph_gout is not None else (None, None)
            dW, db, dsup = linear_act_backward(st.sup_fv, st.hproj, gph[0], ACT_ELU, d_hproj.contiguous(),
                                               dx_init=dsup, dW_out=gw, db_out=gb)
            g["GPH.0.weight"], g["GPH.0.bias"] = dW, db
        if d_logits is not None:
            h = st.h if enc.use_projection_head else st.sup_fv
            w_o, b_o = dst("MLP_sup2.0")
            if enc.use_projection_head:
                dW, db, dh = linear_act_backward(h, st.logits, enc.MLP_sup2[0], ACT_ELU, d_logits.contiguous(),
                                                 dW_out=w_o, db_out=b_o)
                g["MLP_sup2.0.weight"], g["MLP_sup2.0.bias"] = dW, db
                w_o, b_o = dst("MLP_head.0")
                dW, db, dsup = linear_act_backward(st.sup_fv, st.h, enc.MLP_head[0], ACT_ELU, dh, dx_init=dsup,
                                                   dW_out=w_o, db_out=b_o)
                g["MLP_head.0.weight"], g["MLP_head.0.bias"] = dW, db
            else:
                dW, db, dsup = linear_act_backward(h, st.logits, enc.MLP_sup2[0], ACT_ELU, d_logits.contiguous(),
                                                   dx_init=dsup, dW_out=w_o, db_out=b_o)
                g["MLP_sup2.0.weight"], g["MLP_sup2.0.bias"] = dW, db
        else:
            for nm in ("MLP_sup2.0", "MLP_head.0"):
                mod = getattr(enc, nm.split(".")[0], None)
                if mod is not None:
                    g[nm + ".weight"] = gout[nm + ".weight"] if gout is not None else torch.zeros_like(mod[0].weight)
                    g[nm + ".bias"] = gout[nm + ".bias"] if gout is not None else torch.zeros_like(mod[0].bias)
        w_o, b_o = dst("MLP_sup1.0")
        dW, db, dx4 = linear_act_backward(st.x4, st.sup_fv, enc.MLP_sup1[0], ACT_ELU, dsup, dW_out=w_o, db_out=b_o)
        g["MLP_sup1.0.weight"], g["MLP_sup1.0.bias"] = dW, db
    mark("enc_bwd.heads")
    if after_heads is not None:
        after_heads()            # trainer hook: the heads' backward is enqueued, the temporal block follows
    dg, dx2 = dtc_backward(st.dtc, enc.tc_block.layers(), B, T, dpool=dx4, need_dx=True, gout=gout, mode=st.mode)
    mark("enc_bwd.dtc")
    for i, d in enumerate(dg, start=1):
        for k, v in d.items():
            g[f"tc_block.dtc{i}.{k}"] = v
    if before_pointnet is not None:
        before_pointnet()        # trainer hook: everything enqueued so far is the latency-bound part of the backward
    pg, dxp = pointnet_backward(st.pn, enc.pc_block.layers(), st.mode, dpool=dx2, pool_rows=N, need_dx=need_dx,
                                gout=gout)
    for i, d in enumerate(pg, start=1):
        for k, v in d.items():
            g[f"pc_block.pointnet{i}.{k}"] = v
    mark("enc_bwd.pointnet")
    dx = None
    if need_dx:
        dx = dxp.float().view(B, T, N, st.C).permute(0, 3, 1, 2)
    return g, dx


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, x, *params):
        logits, sup_fv, st = encoder_forward(enc, x, enc.training)
        ctx.enc, ctx.st = enc, st
        ctx.need_dx = x.requires_grad
        ctx.names = [n for n, _ in enc.named_parameters()]
        return logits, sup_fv

    @staticmethod
    @once_differentiable
    def backward(ctx, d_logits, d_supfv):
        g, dx = encoder_backward(ctx.enc, ctx.st, d_logits, d_supfv, need_dx=ctx.need_dx)
        return (None, dx) + tuple(g.get(n) for n in ctx.names)


def cg_encoder(enc, x):
    _require_gpu(x, "CGEncoder")
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in enc.parameters())):
        return _EncoderFn.apply(enc, x, *enc.parameters())
    logits, sup_fv, _ = encoder_forward(enc, x, enc.training)
    return logits, sup_fv


# ---------------------------------------------------------------- encoder trunk (OR-CED baseline, models.py:446-505)
class _TrunkFn(torch.autograd.Function):
    """PointNetBlock -> mean over the points -> TemporalConvolutionBlock -> mean over time: x [B,C,T,N] -> x4 [B,512],
    the part ORCEDEncoder shares with CGEncoder, on the same kernels (its three Linear heads follow in torch)."""

    @staticmethod
    def forward(ctx, enc, x, *params):
        B, C, T, N = x.shape
        mode = get_precision()
        xp = _point_major(x).view(B * T * N, C)
        x2, pn = pointnet_forward(xp, enc.pc_block.layers(), enc.training, mode, pool_rows=N)
        x4, dtc = dtc_forward(x2, B, T, enc.tc_block.layers(), enc.training, pool_time=True)
        ctx.enc, ctx.pn, ctx.dtc, ctx.shape, ctx.mode, ctx.training = enc, pn, dtc, (B, C, T, N), mode, enc.training
        ctx.names = [n for n, _ in itertools.chain(enc.pc_block.named_parameters(prefix="pc_block"),
                                                   enc.tc_block.named_parameters(prefix="tc_block"))]
        return x4

    @staticmethod
    @once_differentiable
    def backward(ctx, dx4):
        if not ctx.training:
            raise RuntimeError("ORCEDEncoder backward in eval mode is not implemented on the HIP path")
        B, C, T, N = ctx.shape
        enc = ctx.enc
        g = {}
        dg, dx2 = dtc_backward(ctx.dtc, enc.tc_block.layers(), B, T, dpool=dx4.contiguous().float(), need_dx=True)
        for i, d in enumerate(dg, start=1):
            for k, v in d.items():
                g[f"tc_block.dtc{i}.{k}"] = v
        pg, _ = pointnet_backward(ctx.pn, enc.pc_block.layers(), ctx.mode, dpool=dx2, pool_rows=N, need_dx=False)
        for i, d in enumerate(pg, start=1):
            for k, v in d.items():
                g[f"pc_block.pointnet{i}.{k}"] = v
        return (None, None) + tuple(g.get(n) for n in ctx.names)


class _OrcedHeadsFn(torch.autograd.Function):
    """ORCEDEncoder's MLP_mu / MLP_logvar, the reparametrisation and MLP_classification (models.py:489-505) in one
    launch (csrc/orced.hip); backward in two."""

    @staticmethod
    def forward(ctx, x4, eps, Wmu, bmu, Wlv, blv, Wc, bc):
        x4, eps = x4.contiguous(), eps.contiguous().float()
        logits, sup, mu, logvar = ops.orced_heads_fwd(x4, Wmu, bmu, Wlv, blv, eps, Wc, bc)
        ctx.save_for_backward(x4, eps, logvar, sup, Wmu, Wlv, Wc)
        return logits, sup, mu, logvar

    @staticmethod
    @once_differentiable
    def backward(ctx, d_logits, d_sup, d_mu, d_logvar):
        x4, eps, logvar, sup, Wmu, Wlv, Wc = ctx.saved_tensors
        c = lambda t: None if t is None else t.contiguous().float()
        dx4, dWmu, dbmu, dWlv, dblv, dWc, dbc = ops.orced_heads_bwd(x4, eps, logvar, sup, Wmu, Wlv, Wc, c(d_logits), c(d_sup),
                                                                    c(d_mu), c(d_logvar), need_dx=ctx.needs_input_grad[0])
        return dx4, None, dWmu, dbmu, dWlv, dblv, dWc, dbc


def orced_heads(enc, x4, eps):
    """(logits, sup_fv, vae_mu, vae_logvar) of an ORCEDEncoder from its trunk output and the caller's eps draw"""
    _require_gpu(x4, "ORCEDEncoder heads")
    lm, ll, lc = enc.MLP_mu[0], enc.MLP_logvar[0], enc.MLP_classification[0]
    return _OrcedHeadsFn.apply(x4, eps, lm.weight, lm.bias, ll.weight, ll.bias, lc.weight, lc.bias)


class _KlFn(torch.autograd.Function):
    """CG_kl_divergence (utils.py:72-85): forward and the three gradients in one launch each."""

    @staticmethod
    def forward(ctx, mu, logvar, mu_k):
        mu, logvar, mu_k = mu.contiguous().float(), logvar.contiguous().float(), mu_k.contiguous().float()
        # loss and the three UNSCALED gradients from the one launch (as _CeFn): the backward multiplies by the upstream
        # gradient on the device -- no host read of a device scalar (round-3 advisor finding: float(g) synchronised
        # every step and cannot be captured into a hipGraph)
        loss, grads = ops.orced_kl(mu, logvar, mu_k, want_loss=True, gscale=1.0)
        ctx.save_for_backward(*grads)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        dm, dl, dk = ctx.saved_tensors
        return dm * g, dl * g, dk * g


def cg_kl_divergence(mu, logvar, mu_k):
    _require_gpu(mu, "CG_kl_divergence")
    return _KlFn.apply(mu, logvar, mu_k)


class _CeFn(torch.autograd.Function):
    """torch.nn.functional.cross_entropy (mean) on the device kernel of the PCAA step (pcaa_cross_entropy: loss and
    softmax - onehot gradient in the same launch)."""

    @staticmethod
    def forward(ctx, logits, target):
        logits = logits.contiguous().float()
        loss, dl, _ = ops.cross_entropy(logits, target, want_loss=True, want_grad=True, grad_scale=1.0)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


def cross_entropy_loss(logits, target):
    _require_gpu(logits, "cross_entropy_loss")
    return _CeFn.apply(logits, target)


def encoder_trunk(enc, x):
    _require_gpu(x, "ORCEDEncoder")
    if x.dim() != 4 or x.shape[3] != enc.nmax_points:
        raise RuntimeError(f"ORCEDEncoder expects [B,C,T,{enc.nmax_points}], got {tuple(x.shape)}")
    params = list(enc.pc_block.parameters()) + list(enc.tc_block.parameters())
    return _TrunkFn.apply(enc, x, *params)


# ---------------------------------------------------------------- standalone blocks
class _PointNetStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layers, training, x, *params):
        B, C, T, N = x.shape
        xp = _point_major(x).view(B * T * N, C)
        a, saves = pointnet_forward(xp, layers, training, "fp32")
        ctx.layers, ctx.saves, ctx.shape = layers, saves, (B, C, T, N)
        ctx.need_dx = x.requires_grad
        ctx.training = training
        return a.view(B, T, N, -1).permute(0, 3, 1, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        if not ctx.training:
            raise RuntimeError("PointNet backward in eval mode is not implemented on the HIP path")
        B, C, T, N = ctx.shape
        d = gout.permute(0, 2, 3, 1).contiguous().view(B * T * N, -1)
        grads, dxp = pointnet_backward(ctx.saves, ctx.layers, "fp32", d_last=d, need_dx=ctx.need_dx)
        flat = []
        for gd in grads:
            flat += [gd["module.0.weight"], gd["module.0.bias"], gd["module.1.weight"], gd["module.1.bias"]]
        dx = dxp.view(B, T, N, C).permute(0, 3, 1, 2) if ctx.need_dx else None
        return (None, None, dx) + tuple(flat)


def pointnet_stack(x, layers, training):
    _require_gpu(x, "PointNet")
    params = []
    for l in layers:
        params += [l.module[0].weight, l.module[0].bias, l.module[1].weight, l.module[1].bias]
    return _PointNetStackFn.apply(layers, training, x, *params)


class _DtcStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layers, training, x, *params):
        B, C, T = x.shape
        a2d = x.permute(0, 2, 1).contiguous().view(B * T, C)
        a, saves = dtc_forward(a2d, B, T, layers, training, pool_time=False)
        ctx.layers, ctx.saves, ctx.shape, ctx.training = layers, saves, (B, C, T), training
        return a.view(B, T, -1).permute(0, 2, 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        if not ctx.training:
            raise RuntimeError("DilTempConv1d backward in eval mode is not implemented on the HIP path")
        B, C, T = ctx.shape
        d = gout.permute(0, 2, 1).contiguous().view(B * T, -1)
        grads, dx2d = dtc_backward(ctx.saves, ctx.layers, B, T, d_last=d, need_dx=True)
        flat = []
        for gd in grads:
            flat += [gd["conv1d.weight"], gd["conv1d.bias"], gd["batch_norm.weight"], gd["batch_norm.bias"]]
        return (None, None, dx2d.view(B, T, C).permute(0, 2, 1)) + tuple(flat)


def dtc_stack(x, layers, training):
    _require_gpu(x, "DilTempConv1d")
    params = []
    for l in layers:
        params += [l.conv1d.weight, l.conv1d.bias, l.batch_norm.weight, l.batch_norm.bias]
    return _DtcStackFn.apply(layers, training, x.float(), *params)


# ======================================================================
# CGDecoder (models.py:340-385)
# ======================================================================
def decoder_forward(dec, z, mode=None, images=None):
    """``images`` {layer number: bf16 image of that layer's weight}: streamed instead of the fp32 matrix by the bf16
    mode's weight-streaming kernels (same rounding, same result); only a caller that keeps them current passes them."""
    mode = get_precision() if mode is None else mode
    _require_gpu(z, "CGDecoder")
    if z.dim() != 2 or z.shape[1] != dec.dense1.weight.shape[1]:
        raise RuntimeError(f"CGDecoder: expected [B,{dec.dense1.weight.shape[1]}], got {tuple(z.shape)}")
    acts = [z.contiguous().float()]
    # _pcaa_pad (set by PCAATrainer.finalize for ragged widths): the layers as zero-padded [Np,Kp] weights;
    # the chain then runs in the padded widths (padded activations are exactly 0) and the output is cut back
    layers = getattr(dec, "_pcaa_pad", None) or dec.dense_layers()
    for i, lin in enumerate(layers):
        acts.append(linear_act_forward(acts[-1], lin, ACT_ELU if i < 4 else ACT_NONE, mode,
                                       W16=(images or {}).get(i + 1)))
    out = acts[-1]
    S = dec.dense5.weight.shape[0]
    if out.shape[1] != S:
        out = out[:, :S].contiguous()
    return out, acts


def decoder_backward(dec, acts, d_out, need_dz=True, grads_out=None, dz_init=None, mode=None, after_layer=None,
                     updates=None, images=None):
    """``grads_out`` {"denseI.weight"/"denseI.bias": destination}: for a padded decoder (see decoder_forward)
    these are the PADDED gradient tensors; without grads_out the returned gradients are cut to the parameter
    shapes.  ``updates`` {layer number: update(dz, x)}: those layers' weight gradients are formed and consumed by
    the caller's fused optimizer kernel (linear_act_backward) and come back as None."""
    mode = get_precision() if mode is None else mode
    padded = getattr(dec, "_pcaa_pad", None)
    layers = padded or dec.dense_layers()
    g = {}
    d = d_out.contiguous().view(d_out.shape[0], -1)
    if d.shape[1] != acts[-1].shape[1]:
        dp = torch.zeros_like(acts[-1])                  # padded output columns carry no gradient
        dp[:, :d.shape[1]].copy_(d)
        d = dp
    pre = False           # d is w.r.t. the layer's pre-activation (ELU' already applied by the dgrad above)
    defer_db = []         # (dz, db destination) of the layers whose update is fused: their column sums leave together below
    for i in range(4, -1, -1):
        lin = layers[i]
        nm = f"dense{i + 1}"
        dW_out = grads_out[nm + ".weight"] if grads_out else None
        db_out = grads_out[nm + ".bias"] if grads_out else None
        M, K = acts[i].shape
        fuse = i > 0 and (_skinny(mode, M, lin.weight.shape[0], K) or _skinny_exact(mode, M, lin.weight.shape[0], K))    # acts[i] is an ELU output for i >= 1
        dW, db, d = linear_act_backward(acts[i], acts[i + 1], lin, ACT_ELU if i < 4 else ACT_NONE, d,
                                        need_dx=(i > 0 or need_dz), dW_out=dW_out, db_out=db_out,
                                        dx_init=dz_init if i == 0 else None, mode=mode, d_is_pre=pre,
                                        fuse_elu_in=fuse, update=(updates or {}).get(i + 1), W16=(images or {}).get(i + 1),
                                        defer_db=defer_db if ((updates or {}).get(i + 1) is not None or after_layer is None) else None)
        pre = fuse
        if after_layer is not None:
            after_layer(i + 1)       # trainer hook: layer i+1's weight / bias gradients are enqueued
        dW = dW.view_as(lin.weight) if dW is not None else None
        if padded and grads_out is None:
            ref = dec.dense_layers()[i]                  # cut the padding off: gradients in the parameter shapes
            db = db[:ref.bias.shape[0]]
            dW = dW[:ref.weight.shape[0], :ref.weight.shape[1]] if dW is not None else None
        g[nm + ".weight"], g[nm + ".bias"] = dW, db
    if defer_db:
        with _on_wgrad_stream(*[t for ts, _ in defer_db for t in ts]):
            for _, launch in defer_db:
                launch()
    return g, d


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dec, z, *params):
        out, acts = decoder_forward(dec, z)
        ctx.dec, ctx.acts, ctx.need_dz = dec, acts, z.requires_grad
        ctx.names = [n for n, _ in dec.named_parameters()]
        return out.view(-1, dec.n_features, dec.n_steps, dec.nmax_points)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        g, dz = decoder_backward(ctx.dec, ctx.acts, gout.contiguous(), need_dz=ctx.need_dz)
        return (None, dz) + tuple(g.get(n) for n in ctx.names)   # bn1..4: None (unused, like the reference)


def cg_decoder(dec, z):
    _require_gpu(z, "CGDecoder")
    return _DecoderFn.apply(dec, z, *dec.parameters())


# ======================================================================
# CGDiscriminator (models.py:405-421).  Twice differentiable: the reference's D-step differentiates THROUGH
# torch.autograd.grad(D(interp), interp, create_graph=True) (PCAA_ablation.py:955-976), so the backward of this
# Function is itself a Function (_DiscBwdFn) whose backward is the closed-form second-order kernel
# (pcaa_disc_backward_backward).  The fused trainer does not go through autograd at all
# (ops.disc_wgan_gp); this is what lets the reference's own loop body drive the drop-in module unchanged.
# ======================================================================
class _DiscBwdFn(torch.autograd.Function):
    """(x, label, gout, params) -> (dx, dlabel, dparams...) = the first-order backward, as a differentiable op.
    Its own backward supports an incoming gradient w.r.t. ``dx`` (the gradient-penalty use)."""

    @staticmethod
    def forward(ctx, disc, want, x, label, gout, *params):
        ctx.set_materialize_grads(False)
        dx, dl, grads = ops.disc_backward(x, label, list(params), gout, want_dx=want[0], want_dlabel=want[1],
                                          want_params=want[2])
        ctx.save_for_backward(x, label, gout, *params)
        outs = (dx, dl) + (tuple(grads) if grads is not None else (None,) * 6)
        ctx.mark_non_differentiable(*[o for o in outs[1:] if o is not None])
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, g_dx, *g_rest):
        if any(g is not None for g in g_rest):
            raise NotImplementedError("CGDiscriminator: only the input gradient dD/dx can be differentiated again "
                                      "(the WGAN-GP penalty); gradients of parameter gradients are not implemented")
        if g_dx is None:
            return (None,) * (5 + 6)
        x, label, gout, *params = ctx.saved_tensors
        need = ctx.needs_input_grad        # (disc, want, x, label, gout, *params)
        dx2, dl2, dgo, grads = ops.disc_backward_backward(
            x, label, params, gout, g_dx.contiguous().float(), want_dx=need[2], want_dlabel=need[3],
            want_dgout=need[4], want_params=any(need[5:]))
        return (None, None, dx2, dl2, dgo) + (tuple(grads) if grads is not None else (None,) * 6)


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disc, x, label, *params):
        ctx.disc = disc
        ctx.save_for_backward(x, label, *params)       # the inputs themselves: they keep their place in the graph
        ctx.need = (ctx.needs_input_grad[1], ctx.needs_input_grad[2], any(ctx.needs_input_grad[3:]))
        return ops.disc_forward(x, label, list(params))

    @staticmethod
    def backward(ctx, gout):
        x, label, *params = ctx.saved_tensors
        gout = gout.contiguous().view(-1)
        if torch.is_grad_enabled():
            # create_graph=True: the input gradient stays attached to (x, gout, parameters)
            outs = _DiscBwdFn.apply(ctx.disc, ctx.need, x, label, gout, *params)
            return (None,) + tuple(outs)
        dx, dl, grads = ops.disc_backward(x, label, params, gout, want_dx=ctx.need[0], want_dlabel=ctx.need[1],
                                          want_params=ctx.need[2])
        return (None, dx, dl) + (tuple(grads) if grads is not None else (None,) * 6)


def cg_discriminator(disc, x, label):
    _require_gpu(x, "CGDiscriminator")
    # (conversions happen out here, under autograd, so that what the Function saves are its own inputs)
    return _DiscFn.apply(disc, x.contiguous().float(), label.contiguous().float(), *ops._disc_params(disc))


# ======================================================================
# GaussianMeanLearner (models.py:424-443): Linear+BN1d+ELU x3, Linear
# ======================================================================
class _GmlFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gml, x, *params):
        m = gml.model
        x = x.contiguous().float()
        saves = []
        a = x
        for lin_i, bn_i in ((0, 1), (3, 4), (6, 7)):
            lin, bn = m[lin_i], m[bn_i]
            y, scale, shift, mean, rstd, count = _linear_bn(a, lin.weight, lin.bias, bn, gml.training, "fp32", None)
            s = _LayerSave()
            s.a_in, s.col, s.y, s.scale, s.shift, s.mean, s.rstd = a, None, y, scale, shift, mean, rstd
            s.rows, s.cin, s.cout, s.dil = count, lin.weight.shape[1], lin.weight.shape[0], 0
            saves.append(s)
            a = ops.bn_act_fwd(y, scale, shift)
        out = linear_act_forward(a, m[9], ACT_NONE)
        ctx.gml, ctx.saves, ctx.a_last, ctx.training = gml, saves, a, gml.training
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        if not ctx.training:
            raise RuntimeError("GaussianMeanLearner backward in eval mode is not implemented on the HIP path")
        m = ctx.gml.model
        dW9, db9, da = linear_act_backward(ctx.a_last, None, m[9], ACT_NONE, gout.contiguous())
        out = []
        for (lin_i, bn_i), s in reversed(list(zip(((0, 1), (3, 4), (6, 7)), ctx.saves))):
            lin, bn = m[lin_i], m[bn_i]
            dW, dg, db, da = _bn_layer_backward(s, bn, lin.weight, "fp32", da=da, need_dinput=True, lhs=s.a_in)
            out = [dW, torch.zeros_like(lin.bias), dg, db] + out
        return (None, da) + tuple(out) + (dW9, db9)


def gaussian_mean_learner(gml, x):
    _require_gpu(x, "GaussianMeanLearner")
    return _GmlFn.apply(gml, x, *gml.parameters())


# ======================================================================
# SeqChamferLoss (utils.py:88-132)
# ======================================================================
class _ChamferFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, gts, avg_out):
        B, C, T, N = preds.shape
        want = preds.requires_grad
        scale = 1.0 / (B * T) if avg_out else 1.0 / T
        fl, dp = ops.chamfer(preds.float(), gts.float(), want_grad=want, grad_scale=scale)
        ctx.dp, ctx.avg_out, ctx.B = dp, avg_out, B
        if avg_out:
            return ops.total(fl, scale)
        return ops.rowsum(fl, scale)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        if ctx.dp is None:
            return None, None, None
        gout = gout.contiguous().float()
        if ctx.avg_out:
            return ops.scale_by_device_scalar(ctx.dp, gout), None, None
        return ops.scale_rows(ctx.dp, gout), None, None


def seq_chamfer_loss(preds, gts, avg_out=True):
    _require_gpu(preds, "SeqChamferLoss")
    _require_gpu(gts, "SeqChamferLoss")
    return _ChamferFn.apply(preds, gts, bool(avg_out))
