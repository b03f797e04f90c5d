"""Thin tensor-level wrappers over the C ABI (include/pcaa_hip.h).

Every function takes torch CUDA(=HIP) tensors, validates what the kernels
assume (device, dtype, contiguity, shapes) on the host BEFORE launching, and
launches on torch's current stream.  torch is used for allocation only.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_ELU, ACT_NONE, KC, PCAA_BF16, PCAA_F32, RC, check

PCAA_SPLIT_F16 = 2      # include/pcaa_hip.h

NREP = 16          # replicas of a BatchNorm statistics row (spreads fp64 atomics)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _p(t):
    return None if t is None else t.data_ptr()      # (ctypes takes the int for a void*: no c_void_p object per argument)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _s():
    """the current stream's handle.  Through torch's C entry points when it has them: torch.cuda.current_stream() walks
    ~10 Python frames (device-index resolution, an availability check that reads os.environ) -- 8 us a call, 86 calls a
    step = a sixth of the step's host time (cProfile, round 3)"""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


_get_cur_stream = getattr(torch._C, "_cuda_getCurrentStream", None)
_set_cur_stream = getattr(torch._C, "_cuda_setStream", None)


def current_stream():
    """torch.cuda.current_stream() of the current device without its ~8 us of Python (see _s)."""
    if _get_cur_stream is None or _raw_device is None:
        return torch.cuda.current_stream()
    sd = _get_cur_stream(_raw_device())
    return torch.cuda.Stream(stream_id=sd[0], device_index=sd[1], device_type=sd[2])


class on_stream:
    """``with on_stream(st):`` = ``with torch.cuda.stream(st):`` for a stream of the CURRENT device, through the same C
    entry points (torch._C._cuda_setStream) minus the Python around them: the step switches streams ~20 times."""
    __slots__ = ("st", "prev")

    def __init__(self, st):
        self.st = st
        self.prev = None

    def __enter__(self):
        # torch._C._cuda_getCurrentStream / _cuda_setStream / _cuda_getDevice are PRIVATE torch entry points (present in
        # the 2.x series this image ships): when any is missing, or the stream belongs to another device than the
        # current one (the raw path saves and restores the current device's stream only -- round-3 advisor finding),
        # the public context manager runs instead
        if (_set_cur_stream is None or _get_cur_stream is None or _raw_device is None
                or self.st.device_index != _raw_device()):
            self.prev = torch.cuda.stream(self.st)
            self.prev.__enter__()
            return self
        self.prev = _get_cur_stream(_raw_device())
        st = self.st
        _set_cur_stream(stream_id=st.stream_id, device_index=st.device_index, device_type=st.device_type)
        return self

    def __exit__(self, *exc):
        if isinstance(self.prev, tuple):
            _set_cur_stream(stream_id=self.prev[0], device_index=self.prev[1], device_type=self.prev[2])
        else:
            self.prev.__exit__(*exc)
        return False


def _dt(t):
    if t.dtype == torch.float32:
        return PCAA_F32
    if t.dtype == torch.bfloat16:
        return PCAA_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _chk(t, name, dtype=None, dim=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the HIP device, got "
                           f"{'cpu tensor' if isinstance(t, torch.Tensor) else type(t)} "
                           "(this package has no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if dim is not None and t.dim() != dim:
        raise ValueError(f"{name}: expected {dim}-D, got shape {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous (shape {tuple(t.shape)}, strides {t.stride()})")
    return t


class LaunchTimer:
    """Optional per-launch HIP-event timing of pcaa_gemm calls (bench.py uses it
    to measure the dominant kernel's average duration inside the timed region;
    events are recorded on the stream the kernels are launched on)."""

    def __init__(self, only_prefix=None):
        self.records = []      # (key, flops, bytes, events)
        # every timed launch costs two event packets on the stream (~0.3 ms per step when all ~45
        # GEMM launches are timed); bench.py restricts the timing to the kernel it reports
        self.only_prefix = only_prefix

    def wants(self, key):
        return self.only_prefix is None or key.startswith(self.only_prefix)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, flops, nbytes, ev in self.records:
            if not ev.valid():
                continue
            a = agg.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            a["launches"] += 1
            a["ms"] += ev.elapsed_ms()
            a["flops"] += flops
            a["bytes"] += nbytes
        return agg


class _TorchEvents:
    """two marker events around a call (each drains the queue: ~25 us on top of the kernel)"""

    def __init__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e1 = torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def end(self):
        self.e1.record()
        return self

    def valid(self):
        return True

    def elapsed_ms(self):
        return self.e0.elapsed_time(self.e1)


class _KernelEvents:
    """start/stop events attached to the next LDS-DMA GEMM launch itself (pcaa_time_next_gemm): the
    kernel's own begin/end timestamps, as rocprofv3 reports them."""

    def __init__(self):
        lib = _lib.load()
        self.a, self.b = ctypes.c_void_p(), ctypes.c_void_p()
        check(lib.pcaa_timing_events_create(ctypes.byref(self.a), ctypes.byref(self.b)), "pcaa_timing_events_create")
        check(lib.pcaa_time_next_gemm(self.a, self.b), "pcaa_time_next_gemm")
        self.ok = False

    def end(self):
        lib = _lib.load()
        self.ok = not lib.pcaa_timing_pending()       # consumed by the launch?
        if not self.ok:
            lib.pcaa_time_next_gemm(None, None)
        return self

    def valid(self):
        return self.ok

    def elapsed_ms(self):
        ms = ctypes.c_float()
        check(_lib.load().pcaa_timing_elapsed_ms(self.a, self.b, ctypes.byref(ms)), "pcaa_timing_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            _lib.load().pcaa_timing_events_destroy(self.a, self.b)
        except Exception:
            pass


def _begin_timing(key):
    return _KernelEvents() if key.startswith(("gemm_bf16_v2_kernel", "gemm_bf16_v2rc_kernel")) else _TorchEvents()


TIMER = None


def set_timer(timer):
    global TIMER
    TIMER = timer


class StatsPool:
    """One fp64 buffer for all BatchNorm statistics of a step, cleared by ONE
    fill at the start of the step instead of one per layer and direction."""

    def __init__(self, device, capacity_doubles=4 << 20):
        self.buf = torch.zeros(capacity_doubles, dtype=torch.float64, device=device)
        self.off = 0

    def begin(self):
        if self.off:
            self.buf[:self.off].zero_()
        self.off = 0

    def take_raw(self, n):
        """n zeroed doubles (16-B granular) from the pool, or None when it is full"""
        n = (n + 1) // 2 * 2
        if self.off + n > self.buf.numel():
            return None
        v = self.buf[self.off:self.off + n]
        self.off += n
        return v

    def take(self, ch):
        # + 2 doubles: the arrival counter of a finalize carried by the producer (BnTailFwd / BnTailBwd below), zeroed
        # with the statistics by begin(); 16-B granularity keeps every buffer aligned
        n = NREP * 2 * ch
        if self.off + n + 2 > self.buf.numel():
            return None
        v = self.buf[self.off:self.off + n].view(NREP, 2, ch)
        v._pcaa_counter = self.buf[self.off + n:self.off + n + 1].view(torch.int32)
        self.off += n + 2
        return v


STATS_POOL = None


def set_stats_pool(pool):
    global STATS_POOL
    STATS_POOL = pool


def new_stats(ch, device):
    if STATS_POOL is not None and STATS_POOL.buf.device == torch.device(device):
        v = STATS_POOL.take(ch)
        if v is not None:
            return v
    buf = torch.zeros(NREP * 2 * ch + 2, dtype=torch.float64, device=device)
    v = buf[:NREP * 2 * ch].view(NREP, 2, ch)
    v._pcaa_counter = buf[NREP * 2 * ch:NREP * 2 * ch + 1].view(torch.int32)
    return v


# ------------------------------------------------------------------ finalize carried by the producer (csrc/bn_tail.h)
# A BnTailFwd / BnTailBwd describes the BatchNorm finalize of the statistics a launch is about to accumulate.  The
# producer's wrapper calls arm(stats) right before its launch and resolve(stats) right after: if the launch could carry
# the tail (its last workgroup then writes the coefficients) resolve() has nothing to do, otherwise -- SyncBN (``sync``:
# the statistics are all-reduced first), a producer or shape that does not carry tails -- it runs the stand-alone kernel.
TAILS = {"enabled": os.environ.get("PCAA_BN_TAIL", "1") != "0", "taken": 0, "standalone": 0}


class BnTailFwd:
    def __init__(self, count, lin_bias, bn, ch, sync=None, update_running=True):
        self.count, self.lin_bias, self.bn, self.ch, self.sync, self.update_running = count, lin_bias, bn, ch, sync, update_running
        self.out = None
        self.armed = False
        self.count_out = count            # the count the statistics were averaged over (global under SyncBN)

    def arm(self, stats):
        counter = getattr(stats, "_pcaa_counter", None)
        if self.sync is not None or counter is None or not TAILS["enabled"]:
            return
        dev, bn, ch = stats.device, self.bn, self.ch
        scale = torch.empty(ch, dtype=torch.float32, device=dev)
        self.out = (scale, torch.empty_like(scale), torch.empty_like(scale), torch.empty_like(scale))
        rm = bn.running_mean if self.update_running else None
        rv = bn.running_var if self.update_running else None
        nbt = bn.num_batches_tracked if self.update_running else None
        check(_lib.load().pcaa_bn_tail_arm_fwd(_p(stats), NREP, int(self.count), _p(self.lin_bias), _p(bn.weight), _p(bn.bias),
                                               _p(rm), _p(rv), _p(nbt), BN_MOMENTUM if bn.momentum is None else bn.momentum,
                                               bn.eps, *(_p(t) for t in self.out), ch, _p(counter)), "pcaa_bn_tail_arm_fwd")
        self.armed = True

    def resolve(self, stats):
        """-> (scale, shift, mean, rstd)"""
        lib = _lib.load()
        if self.armed and not lib.pcaa_bn_tail_pending():
            TAILS["taken"] += 1
            return self.out
        if self.armed:
            lib.pcaa_bn_tail_disarm()
        count = self.sync(stats, self.count) if self.sync is not None else self.count
        self.count_out = count
        TAILS["standalone"] += 1
        self.out = bn_finalize(stats, count, self.lin_bias, self.bn, self.ch, self.update_running)
        return self.out


class BnTailBwd:
    def __init__(self, count, bn, mean, rstd, ch, dgamma=None, dbeta=None, sync=None):
        self.count, self.bn, self.mean, self.rstd, self.ch, self.sync = count, bn, mean, rstd, ch, sync
        self.dgamma, self.dbeta = dgamma, dbeta
        self.out = None
        self.armed = False

    def arm(self, stats):
        counter = getattr(stats, "_pcaa_counter", None)
        if self.sync is not None or counter is None or not TAILS["enabled"]:
            return
        dev, ch = stats.device, self.ch
        coef = torch.empty((3, ch), dtype=torch.float32, device=dev)
        dgamma = torch.empty(ch, dtype=torch.float32, device=dev) if self.dgamma is None else self.dgamma
        dbeta = torch.empty(ch, dtype=torch.float32, device=dev) if self.dbeta is None else self.dbeta
        self.out = (coef, dgamma, dbeta)
        check(_lib.load().pcaa_bn_tail_arm_bwd(_p(stats), NREP, int(self.count), _p(self.bn.weight), _p(self.mean),
                                               _p(self.rstd), _p(coef), _p(dgamma), _p(dbeta), ch, _p(counter)),
              "pcaa_bn_tail_arm_bwd")
        self.armed = True

    def resolve(self, stats):
        """-> (coef, dgamma, dbeta)"""
        lib = _lib.load()
        if self.armed and not lib.pcaa_bn_tail_pending():
            TAILS["taken"] += 1
            return self.out
        if self.armed:
            lib.pcaa_bn_tail_disarm()
        if self.sync is not None:
            self.sync(stats, 0)
        TAILS["standalone"] += 1
        self.out = bn_bwd_finalize(stats, self.count, self.bn, self.mean, self.rstd, self.ch, dgamma=self.dgamma,
                                   dbeta=self.dbeta)
        return self.out


# ------------------------------------------------------------------ GEMM
_GEMM_V2 = {"on": os.environ.get("PCAA_GEMM_V2", "1") != "0"}


def gemm_v2_enable(on=True):
    """Lab switch (pcaa_gemm_v2_enable): with ``on=False`` the 4-wave tile loops (csrc/gemm_v2.h) decline every launch --
    plain products then take the register-staged 256 x 256 kernel, the fused entry points report their shapes unsupported.
    (Rounds 1-4 routed to the 8-wave loop here; round 5 removed it.)"""
    _GEMM_V2["on"] = bool(on)
    check(_lib.load().pcaa_gemm_v2_enable(int(bool(on))), "pcaa_gemm_v2_enable")


_GEMM_V2_RC = os.environ.get("PCAA_GEMM_V2_RC", "1")[:1] != "0"     # read the same way by csrc/gemm_bf16.hip


def _v2_takes(K, split_k=1, accumulate=False):
    """the dispatch rule of csrc/gemm_bf16.hip (launch_dma): the 4-wave loop takes a KC x KC launch without K splits whose
    contraction is at least five 64-deep steps long"""
    return _GEMM_V2["on"] and K // 64 >= 5 and int(split_k) <= 1 and not accumulate


def _dma_key(out_dtype, layout, v2=False):
    """LaunchTimer / PMC key of one LDS-DMA GEMM instantiation (rocprofv3 lists them as separate kernels): the 4-wave
    loops' v2::gemm_bf16_v2_kernel<__bf16, 0, false> = PointNet forward / plain dgrad, v2::gemm_bf16_v2rc_kernel = the
    weight gradients; anything they decline runs on the register-staged gemm_bf16_big_kernel (timed with stream events)."""
    dt = 'bf16' if out_dtype == torch.bfloat16 else 'f32'
    if v2 and layout == KC:
        return f"gemm_bf16_v2_kernel<{dt},plain>"
    if v2 and layout == RC and dt == 'f32' and _GEMM_V2["on"] and _GEMM_V2_RC:
        return "gemm_bf16_v2rc_kernel<f32>"      # the weight gradients on the 4-wave loop (whole 256 x 256 tiles)
    lay = "KC" if layout == KC else "RC"
    return f"gemm_bf16_big_kernel<{dt},{lay},{lay}>"


def gemm(A, a_layout, B, b_layout, M, N, K, *, lda=None, ldb=None, out=None, out_dtype=torch.float32,
         bias=None, colstats=None, split_k=1, accumulate=False, math=PCAA_F32, tail=None):
    """out[M,N] (=|+=) A(M,K) . B(K,N) (+bias).  A/B are 2-D contiguous tensors
    whose storage order is given by the layout flag (see pcaa_hip.h).  ``tail``: the BatchNorm finalize of
    ``colstats`` (BnTailFwd), carried by this launch when it can be."""
    _chk(A, "gemm.A", dim=2)
    _chk(B, "gemm.B", dim=2)
    ea = (M, K) if a_layout == KC else (K, M)
    eb = (N, K) if b_layout == KC else (K, N)
    if tuple(A.shape) != ea or tuple(B.shape) != eb:
        raise ValueError(f"gemm: operand shapes {tuple(A.shape)} {tuple(B.shape)} do not match "
                         f"M={M} N={N} K={K} layouts {a_layout},{b_layout}")
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    atomic = split_k > 1 or accumulate
    if out is None:
        if atomic:
            out = torch.zeros((M, N), dtype=torch.float32, device=A.device)
        else:
            out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    else:
        _chk(out, "gemm.out")
        if out.numel() != M * N:
            raise ValueError(f"gemm: out has {out.numel()} elements, need {M * N}")
    if bias is not None:
        _chk(bias, "gemm.bias", torch.float32)
        if bias.numel() != N:
            raise ValueError("gemm: bias length")
    if colstats is not None:
        _chk(colstats, "gemm.colstats", torch.float64)
        if tuple(colstats.shape) != (NREP, 2, N):
            raise ValueError("gemm: colstats shape")
    lib = _lib.load()
    timer = TIMER
    if timer is not None:
        if math != PCAA_BF16:
            key = "gemm_f32_kernel"
        elif (A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and a_layout == b_layout
              and (M % 256 == 0 or (a_layout == KC and _v2_takes(K, split_k, accumulate))) and N % 256 == 0 and K % 64 == 0):
            key = _dma_key(out.dtype, a_layout, _v2_takes(K, split_k, accumulate))   # same dispatch rule as pcaa_launch_gemm_bf16_big
        else:
            key = "gemm_bf16_big_kernel"
        timer = timer if timer.wants(key) else None
    if timer is not None:
        ev = _begin_timing(key)
    if tail is not None:
        tail.arm(colstats)
    check(lib.pcaa_gemm(math, _p(A), _dt(A), a_layout, lda, _p(B), _dt(B), b_layout, ldb,
                        _p(out), _dt(out), N, M, N, K, _p(bias), _p(colstats), NREP,
                        int(split_k), int(bool(accumulate)), _s()), "pcaa_gemm")
    if timer is not None:
        ev.end()
        nbytes = A.numel() * A.element_size() + B.numel() * B.element_size() + out.numel() * out.element_size()
        timer.records.append((key, 2.0 * M * N * K, float(nbytes), ev))
    if tail is not None:
        tail.resolve(colstats)
    return out


def gemm_group_rc_f32(products):
    """``products``: up to 8 tuples (A [K,M] fp32, B [K,N] fp32, C [M,N] fp32 -- accumulated into, the caller zeroes it --,
    split_k): all of them C += A^T . B on the exact-fp32 matrix pipe in ONE launch (pcaa_gemm_group_rc_f32)."""
    n = len(products)
    if not 1 <= n <= 8:
        raise ValueError("gemm_group_rc_f32: 1..8 products")
    ptr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    ints = lambda vs: (ctypes.c_int * n)(*[int(v) for v in vs])
    for A, B, C, sk in products:
        _chk(A, "gemm_group.A", torch.float32, 2)
        _chk(B, "gemm_group.B", torch.float32, 2)
        _chk(C, "gemm_group.C", torch.float32)
        if A.shape[0] != B.shape[0] or C.numel() != A.shape[1] * B.shape[1] or A.stride(0) != A.shape[1] or B.stride(0) != B.shape[1]:
            raise ValueError("gemm_group_rc_f32: A [K,M], B [K,N] with contiguous rows, C [M,N]")
    check(_lib.load().pcaa_gemm_group_rc_f32(
        n, ptr([p[0] for p in products]), ptr([p[1] for p in products]), ptr([p[2] for p in products]),
        ints([p[0].shape[1] for p in products]), ints([p[1].shape[1] for p in products]), ints([p[0].shape[0] for p in products]),
        ints([p[3] for p in products]), _s()), "pcaa_gemm_group_rc_f32")


def gemm_slabs(A, a_layout, B, b_layout, M, N, K, split_k, out=None, accumulate=False, math=PCAA_BF16,
               colstats=None, tail=None, reduce_ctx=None):
    """Split-K product without atomics: every split writes its partial [M,N] slab, a second
    launch sums the slabs into ``out`` (=|+=).  Used for the long-K weight gradients, where the
    atomic epilogue (256 KB of fp32 atomics per workgroup) cost as much as the MFMA loop."""
    _chk(A, "gemm_slabs.A", dim=2)
    _chk(B, "gemm_slabs.B", dim=2)
    if (M * N) % 4:
        raise ValueError("gemm_slabs: M*N must be a multiple of 4")
    lib = _lib.load()
    ns = lib.pcaa_gemm_num_splits(math, K, int(split_k))
    stride = M * N
    slabs = torch.empty(ns * stride, dtype=torch.float32, device=A.device)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    else:
        _chk(out, "gemm_slabs.out", torch.float32)
        if out.numel() != M * N:
            raise ValueError("gemm_slabs: out size")
    timer = TIMER
    if timer is not None:
        dma = (math == PCAA_BF16 and A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and a_layout == b_layout
               and M % 256 == 0 and N % 256 == 0 and K % 64 == 0)
        key = _dma_key(torch.float32, a_layout, M % 256 == 0 and N % 256 == 0 and (K // ns) % 64 == 0) if dma else ("gemm_bf16_big_kernel" if math == PCAA_BF16 else "gemm_f32_kernel")
        timer = timer if timer.wants(key) else None
    if timer is not None:
        ev = _begin_timing(key)
    check(lib.pcaa_gemm_slabs(math, _p(A), _dt(A), a_layout, A.stride(0), _p(B), _dt(B), b_layout, B.stride(0),
                              _p(slabs), stride, M, N, K, int(split_k), _s()), "pcaa_gemm_slabs")
    if timer is not None:
        ev.end()
        nbytes = A.numel() * A.element_size() + B.numel() * B.element_size() + ns * stride * 4
        timer.records.append((key, 2.0 * M * N * K, float(nbytes), ev))
    if colstats is not None:
        # reduction fused with the BatchNorm column statistics of the result
        if accumulate:
            raise ValueError("gemm_slabs: colstats with accumulate is not supported")
        _chk(colstats, "gemm_slabs.colstats", torch.float64)
        check(lib.pcaa_splitk_reduce_stats(_p(slabs), ns, stride, _p(out), _p(colstats), NREP, M, N, _s()),
              "pcaa_splitk_reduce_stats")
        if tail is not None:
            tail.resolve(colstats)
        return out
    if reduce_ctx is not None:
        # ``reduce_ctx(slabs)``: a context manager under which the slab reduction is enqueued (functional's wgrad side
        # stream: the sum only feeds the optimizer, so it leaves the stream that carries the backward's chain)
        with reduce_ctx(slabs):
            check(lib.pcaa_splitk_reduce(_p(slabs), ns, stride, stride, _p(out), int(bool(accumulate)), _s()),
                  "pcaa_splitk_reduce")
        return out
    check(lib.pcaa_splitk_reduce(_p(slabs), ns, stride, stride, _p(out), int(bool(accumulate)), _s()),
          "pcaa_splitk_reduce")
    return out


def gemm_slabs_part(A, B, M, N, K, split_k, slabs, math=PCAA_BF16):
    """The weight-gradient form (RC x RC, contraction over the rows) over the rows the two views cover: writes the
    partial products of its splits into ``slabs`` (a float32 view with room for them, ``M*N`` apart) and returns their
    count.  Launch only -- the caller sums the slabs (``splitk_reduce``).  With this a long-K product can be cut in
    two launches on different streams (functional's overlapped weight gradients)."""
    _chk(A, "gemm_slabs_part.A", dim=2)
    _chk(B, "gemm_slabs_part.B", dim=2)
    if A.shape[0] != K or B.shape[0] != K or A.shape[1] != M or B.shape[1] != N:
        raise ValueError("gemm_slabs_part: operand views must be [K, M] and [K, N]")
    lib = _lib.load()
    ns = lib.pcaa_gemm_num_splits(math, K, int(split_k))
    stride = M * N
    _chk(slabs, "gemm_slabs_part.slabs", torch.float32)
    if slabs.numel() < ns * stride or not slabs.is_contiguous():
        raise ValueError("gemm_slabs_part: slab view too small")
    check(lib.pcaa_gemm_slabs(math, _p(A), _dt(A), RC, A.stride(0), _p(B), _dt(B), RC, B.stride(0),
                              _p(slabs), stride, M, N, K, int(split_k), _s()), "pcaa_gemm_slabs")
    return ns


def splitk_reduce(slabs, ns, M, N, out, accumulate=False):
    """out[M, N] (=|+=) the sum of ``ns`` slabs laid ``M*N`` apart"""
    _chk(out, "splitk_reduce.out", torch.float32)
    if out.numel() != M * N or slabs.numel() < ns * M * N:
        raise ValueError("splitk_reduce: sizes")
    check(_lib.load().pcaa_splitk_reduce(_p(slabs), int(ns), M * N, M * N, _p(out), int(bool(accumulate)), _s()),
          "pcaa_splitk_reduce")
    return out


def gemm_affine_elu(a, W16, scale, shift, pool_rows=0):
    """ELU(scale * (a[M,K] @ W16[N,K]^T) + shift) in one launch (eval-mode BatchNorm+ELU in the GEMM
    epilogue): bf16 [M,N], or with pool_rows in {32,64,128} its mean over groups of pool_rows consecutive
    rows, fp32 [M/pool_rows, N].  a, W16 bf16; shapes as gemm_dgrad_bn_supported."""
    _chk(a, "gemm_affine_elu.a", torch.bfloat16, 2)
    _chk(W16, "gemm_affine_elu.W16", torch.bfloat16, 2)
    _chk(scale, "gemm_affine_elu.scale", torch.float32)
    _chk(shift, "gemm_affine_elu.shift", torch.float32)
    M, K = a.shape
    N = W16.shape[0]
    if (W16.shape[1] != K or scale.numel() != N or shift.numel() != N or not gemm_dgrad_bn_supported(M, N, K)
            or pool_rows not in (0, 32, 64, 128)):
        raise ValueError(f"gemm_affine_elu: unsupported shapes a {tuple(a.shape)} W {tuple(W16.shape)} pool {pool_rows}")
    if pool_rows:
        out = torch.empty((M // pool_rows, N), dtype=torch.float32, device=a.device)
    else:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    timer = TIMER
    # eval-mode epilogues (BatchNorm affine + ELU [+ mean-pool])
    key = "gemm_bf16_v2_kernel<bf16,affine_elu>"
    timer = timer if (timer is not None and timer.wants(key)) else None
    if timer is not None:
        ev = _begin_timing(key)
    check(_lib.load().pcaa_gemm_affine_elu(_p(a), a.stride(0), _p(W16), W16.stride(0), _p(out), N, _p(scale),
                                           _p(shift), M, N, K, int(pool_rows), _s()), "pcaa_gemm_affine_elu")
    if timer is not None:
        ev.end()
        nbytes = 2 * (M * K + N * K) + out.numel() * out.element_size()
        timer.records.append((key, 2.0 * M * N * K, float(nbytes), ev))
    return out


def gemm_dgrad_bn_supported(M, N, K):
    return bool(_lib.load().pcaa_gemm_dgrad_bn_supported(int(M), int(N), int(K)))


def gemm_dgrad_bn(dy, Wt, y, scale, shift, mean, rstd, tail=None):
    """dz[M,N] = (dy[M,K] @ Wt[N,K]^T) * ELU'(y*scale+shift) plus the BatchNorm-backward statistics of
    the layer that owns y -- the dgrad of the layer above fused with the first half of this layer's
    backward.  Returns (dz bf16, stats).  (The variant that rebuilt y of the FIRST PointNet layer from the points in the
    epilogue -- measured slower than the separate statistics pass in round 1, never on -- left with the 8-wave kernel.)"""
    _chk(dy, "gemm_dgrad_bn.dy", torch.bfloat16, 2)
    _chk(Wt, "gemm_dgrad_bn.Wt", torch.bfloat16, 2)
    _chk(y, "gemm_dgrad_bn.y", torch.bfloat16, 2)
    M, K = dy.shape
    N = Wt.shape[0]
    if Wt.shape[1] != K or tuple(y.shape) != (M, N):
        raise ValueError("gemm_dgrad_bn: shape mismatch")
    dz = torch.empty_like(y)
    stats = new_stats(N, dy.device)
    timer = TIMER
    # its own instantiation (epilogue carries ELU' + statistics)
    key = "gemm_bf16_v2_kernel<bf16,dgrad_bn>"
    timer = timer if (timer is not None and timer.wants(key)) else None
    if timer is not None:
        ev = _begin_timing(key)
    if tail is not None:
        tail.arm(stats)
    check(_lib.load().pcaa_gemm_dgrad_bn(_p(dy), dy.stride(0), _p(Wt), Wt.stride(0), _p(y), _p(dz), dz.stride(0),
                                         _p(scale), _p(shift), _p(mean), _p(rstd), _p(stats), NREP, M, N, K,
                                         None, 0, None, _s()),
          "pcaa_gemm_dgrad_bn")
    if timer is not None:
        ev.end()
        nbytes = 2 * (M * K + N * K + 2 * M * N)
        timer.records.append((key, 2.0 * M * N * K, float(nbytes), ev))
    if tail is not None:
        tail.resolve(stats)
    return dz, stats


# ------------------------------------------------------------------ split-operand parity mode (pcaa_gemm_split3)
# power-of-two image scales that keep each kind of tensor in fp16's normal range (|x| in 6e-5 .. 65504)
SPLIT_SCALE_ACT = 1.0            # activations after BatchNorm + ELU: O(1)
SPLIT_SCALE_WEIGHT = 256.0       # weights ~ 1/sqrt(fan_in): their lo halves would be fp16 subnormals unscaled
SPLIT_SCALE_GRAD = 65536.0       # gradients of a batch-mean loss: 1e-8 .. 1e-2


class SplitImage:
    """The [hi | lo] fp16 image of an fp32 [rows, ch] tensor times ``scale`` (hi = fp16(s v), lo = fp16(s v - hi)):
    ``img`` is fp16 [rows, 2 ch]; ``shape`` / ``device`` are those of the fp32 tensor it stands for."""
    __slots__ = ("img", "rows", "ch", "scale")
    dtype = "split_fp16"

    def __init__(self, img, rows, ch, scale):
        self.img, self.rows, self.ch, self.scale = img, rows, ch, float(scale)

    @property
    def shape(self):
        return (self.rows, self.ch)

    @property
    def device(self):
        return self.img.device

    def float(self):
        """the fp32 tensor the image stands for (tests)"""
        return (self.img[:, :self.ch].float() + self.img[:, self.ch:].float()) / self.scale


# Range guard (round 4, advisor finding): fp16 images hold |scale * v| <= 65504 while the reference's fp32 tensors have no
# limit.  The producers saturate what leaves the range and raise a per-device flag (csrc/common.h, split_guard); the
# flag is registered right before every producer launch (split_image_empty is on each producer's path) and read by
# range_check() -- PCAATrainer.check() calls it -- which raises: the caller reruns in "fp32" (exact, no range limit).
_RANGE_FLAGS = {}


def _range_flag(device):
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _RANGE_FLAGS.get(idx)
    if t is None:
        t = _RANGE_FLAGS[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
    return t


def range_check(device=None, reset=True):
    """Raise FloatingPointError if an fp16x3-mode operand image saturated on ``device`` since the last check (one
    host sync).  The products of such a step are wrong; rerun it with precision "fp32"."""
    if device is None:
        idxs = list(_RANGE_FLAGS)
    else:
        dev = torch.device(device)       # an index-less "cuda" is the CURRENT device, as in _range_flag
        idxs = [dev.index if dev.index is not None else torch.cuda.current_device()]
    for idx in idxs:
        t = _RANGE_FLAGS.get(idx)
        if t is not None and int(t.item()):
            if reset:
                t.zero_()
            raise FloatingPointError(
                "fp16x3 mode: a value left the fp16 range of its [hi | lo] operand image (|scale * v| > 65504 or NaN; "
                f"scales: activations {SPLIT_SCALE_ACT:g}, weights {SPLIT_SCALE_WEIGHT:g}, gradients {SPLIT_SCALE_GRAD:g}); "
                "the image was saturated and this step's products are not fp32-grade -- use precision='fp32'")


def split_image_empty(rows, ch, device, scale):
    check(_lib.load().pcaa_set_range_flag(_p(_range_flag(device))), "pcaa_set_range_flag")
    return SplitImage(torch.empty((rows, 2 * ch), dtype=torch.float16, device=device), rows, ch, scale)


def split_f16(x2d, transpose=False, scale=SPLIT_SCALE_WEIGHT):
    """fp32 [rows, ch] -> SplitImage of x2d (``transpose``: of x2d^T, without materialising the transpose)"""
    _chk(x2d, "split_f16.x", torch.float32, 2)
    rows, ch = x2d.shape
    out = split_image_empty(ch, rows, x2d.device, scale) if transpose else split_image_empty(rows, ch, x2d.device, scale)
    check(_lib.load().pcaa_split_f16(_p(x2d), _p(out.img), rows, ch, int(bool(transpose)), float(scale), _s()),
          "pcaa_split_f16")
    return out


def gemm_split3_supported(M, N, K):
    return bool(_lib.load().pcaa_gemm_split3_supported(int(M), int(N), int(K)))


def gemm_split3(A, B, layout, M, N, K, colstats=None, tail=None, out=None):
    """out[M,N] fp32 = A . B^T (layout KC: A [M,K], B [N,K]) or A^T . B (RC: A [K,M], B [K,N]) from SplitImages:
    hi.hi + lo.hi + hi.lo on the f16 MFMA pipe, fp32 accumulate, rescaled by 1 / (A.scale B.scale)."""
    ea = (M, K) if layout == KC else (K, M)
    eb = (N, K) if layout == KC else (K, N)
    if tuple(A.shape) != ea or tuple(B.shape) != eb:
        raise ValueError(f"gemm_split3: operand shapes {A.shape} {B.shape} do not match M={M} N={N} K={K}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    timer = TIMER
    key = "gemm_bf16_v2_kernel<f32,split3>" if layout == KC else "gemm_bf16_v2rc_kernel<f32,split3>"
    timer = timer if (timer is not None and timer.wants(key)) else None
    if timer is not None:
        ev = _begin_timing(key)
    if tail is not None:
        tail.arm(colstats)
    check(_lib.load().pcaa_gemm_split3(_p(A.img), _p(B.img), layout, A.img.stride(0), B.img.stride(0), _p(out), N, M, N, K,
                                       _p(colstats), NREP, 1.0 / (A.scale * B.scale), _s()), "pcaa_gemm_split3")
    if timer is not None:
        ev.end()
        timer.records.append((key, 3 * 2.0 * M * N * K, float(4 * (A.img.numel() // 2 + B.img.numel() // 2 + M * N)), ev))
    if tail is not None:
        tail.resolve(colstats)
    return out


def gemm_slabs_split3(A, B, M, N, K, split_k, out=None):
    """The weight-gradient form (RC x RC: A [K,M], B [K,N] SplitImages, contraction over the K rows) with slab split-K."""
    if tuple(A.shape) != (K, M) or tuple(B.shape) != (K, N):
        raise ValueError("gemm_slabs_split3: operand shapes")
    lib = _lib.load()
    ns = lib.pcaa_gemm_split3_num_splits(K, int(split_k))
    stride = M * N
    slabs = torch.empty(ns * stride, dtype=torch.float32, device=A.device)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    timer = TIMER
    key = "gemm_bf16_v2rc_kernel<f32,split3>"
    timer = timer if (timer is not None and timer.wants(key)) else None
    if timer is not None:
        ev = _begin_timing(key)
    check(lib.pcaa_gemm_slabs_split3(_p(A.img), _p(B.img), RC, A.img.stride(0), B.img.stride(0), _p(slabs), stride, M, N, K,
                                     int(split_k), 1.0 / (A.scale * B.scale), _s()), "pcaa_gemm_slabs_split3")
    if timer is not None:
        ev.end()
        timer.records.append((key, 3 * 2.0 * M * N * K, float(4 * (A.img.numel() // 2 + B.img.numel() // 2) + ns * stride * 4), ev))
    check(lib.pcaa_splitk_reduce(_p(slabs), ns, stride, stride, _p(out), 0, _s()), "pcaa_splitk_reduce")
    return out


def bn_act_fwd_split(y, scale, shift):
    """a = ELU(y*scale+shift) as a SplitImage (fp32 y)"""
    _chk(y, "bn_act_fwd_split.y", torch.float32, 2)
    a = split_image_empty(y.shape[0], y.shape[1], y.device, SPLIT_SCALE_ACT)
    check(_lib.load().pcaa_bn_act_fwd_split(_p(y), _p(a.img), _p(scale), _p(shift), y.shape[0], y.shape[1], a.scale, _s()),
          "pcaa_bn_act_fwd_split")
    return a


def bn_bwd_dy_fused_split(y, scale, shift, coef, *, da=None, dpool=None, group_rows=0, pool_scale=1.0):
    """bn_bwd_dy_fused with dy written as a SplitImage (fp32 y / da)"""
    _chk(y, "bn_bwd_dy_fused_split.y", torch.float32, 2)
    rows, ch = y.shape
    dy = split_image_empty(rows, ch, y.device, SPLIT_SCALE_GRAD)
    check(_lib.load().pcaa_bn_bwd_dy_fused_split(_p(da), _p(dpool), int(group_rows), float(pool_scale), _p(y), _p(dy.img),
                                                 _p(scale), _p(shift), _p(coef), rows, ch, dy.scale, _s()),
          "pcaa_bn_bwd_dy_fused_split")
    return dy


def bn_bwd_dy_split(dz, y, coef):
    """dy = coef0*dz + coef1*y + coef2 as a SplitImage (fp32 dz, y): behind gemm_dgrad_bn_split3"""
    _chk(dz, "bn_bwd_dy_split.dz", torch.float32, 2)
    _chk(y, "bn_bwd_dy_split.y", torch.float32, 2)
    rows, ch = y.shape
    dy = split_image_empty(rows, ch, y.device, SPLIT_SCALE_GRAD)
    check(_lib.load().pcaa_bn_bwd_dy_split(_p(dz), _p(y), _p(dy.img), _p(coef), rows, ch, dy.scale, _s()),
          "pcaa_bn_bwd_dy_split")
    return dy


def gemm_dgrad_bn_split3_supported(M, N, K):
    return bool(_lib.load().pcaa_gemm_split3_supported(int(M), int(N), int(K)))


def gemm_dgrad_bn_split3(dy, wt, y, scale, shift, mean, rstd, tail=None):
    """dgrad of a PointNet layer on split-fp16 operands (dy [M, K], wt [N, K] SplitImages) fused with the first half of
    the BatchNorm + ELU backward of the layer below (fp32 y [M, N]): returns (dz fp32 [M, N], stats)."""
    M, K = dy.shape
    N = wt.shape[0]
    if wt.shape[1] != K or tuple(y.shape) != (M, N):
        raise ValueError("gemm_dgrad_bn_split3: operand shapes")
    _chk(y, "gemm_dgrad_bn_split3.y", torch.float32, 2)
    dz = torch.empty((M, N), dtype=torch.float32, device=y.device)
    stats = new_stats(N, y.device)
    if tail is not None:
        tail.arm(stats)
    check(_lib.load().pcaa_gemm_dgrad_bn_split3(_p(dy.img), dy.img.stride(0), _p(wt.img), wt.img.stride(0), _p(y), _p(dz), N,
                                                _p(scale), _p(shift), _p(mean), _p(rstd), _p(stats), NREP, M, N, K,
                                                1.0 / (dy.scale * wt.scale), _s()), "pcaa_gemm_dgrad_bn_split3")
    if tail is not None:
        tail.resolve(stats)
    return dz, stats


def pick_split_k(M, N, K, target_blocks=1024, bk=32, tile=128):
    tiles = ((M + tile - 1) // tile) * ((N + tile - 1) // tile)
    if tiles >= target_blocks:
        return 1
    s = max(1, target_blocks // tiles)
    return max(1, min(s, K // (4 * bk) if K >= 4 * bk else 1))


def cast_bf16(W2d, want_plain=True, want_transposed=True, out=None):
    """bf16 shadows of an fp32 weight matrix [R,C]: (W16 [R,C], W16t [C,R]).  ``out``: the plain image's destination."""
    _chk(W2d, "cast_bf16.W", torch.float32, 2)
    R, C = W2d.shape
    if out is not None:
        _chk(out, "cast_bf16.out", torch.bfloat16, 2)
        if tuple(out.shape) != (R, C):
            raise ValueError("cast_bf16: out shape")
    w = out if out is not None else (torch.empty((R, C), dtype=torch.bfloat16, device=W2d.device) if want_plain else None)
    wt = torch.empty((C, R), dtype=torch.bfloat16, device=W2d.device) if want_transposed else None
    check(_lib.load().pcaa_cast_bf16(_p(W2d), _p(w), _p(wt), R, C, _s()), "pcaa_cast_bf16")
    return w, wt


def pointnet_in_ok(C, cout):
    return 1 <= C <= 8 and cout % 4 == 0 and cout <= 1024 and 256 % (cout // 4) == 0


def pointnet_in_fwd(x2d, W2d, bias, out_dtype, stats=None, tail=None):
    """y[P,cout] = x2d[P,C] . W2d[cout,C]^T + bias (+ BatchNorm statistics; ``tail``: their finalize, BnTailFwd)."""
    _chk(x2d, "pointnet_in.x", torch.float32, 2)
    _chk(W2d, "pointnet_in.W", torch.float32, 2)
    P, C = x2d.shape
    cout = W2d.shape[0]
    if W2d.shape[1] != C or not pointnet_in_ok(C, cout):
        raise ValueError(f"pointnet_in_fwd: unsupported shape C={C} cout={cout}")
    if out_dtype is None:                    # statistics only (recompute path): y is never stored
        if stats is None:
            raise ValueError("pointnet_in_fwd: statistics-only call needs stats")
        if tail is not None:
            tail.arm(stats)
        check(_lib.load().pcaa_pointnet_in_fwd(_p(x2d), C, _p(W2d), None, None, PCAA_F32, P, cout, _p(stats), NREP,
                                               _s()), "pcaa_pointnet_in_fwd(stats)")
        if tail is not None:
            tail.resolve(stats)
        return None
    y = torch.empty((P, cout), dtype=out_dtype, device=x2d.device)
    if tail is not None and stats is not None:
        tail.arm(stats)
    check(_lib.load().pcaa_pointnet_in_fwd(_p(x2d), C, _p(W2d), _p(bias), _p(y), _dt(y), P, cout, _p(stats), NREP, _s()),
          "pcaa_pointnet_in_fwd")
    if tail is not None and stats is not None:
        tail.resolve(stats)
    return y


def pointnet_in_apply(x2d, W2d, scale, shift, out_dtype):
    """a[P,cout] = ELU((x2d . W2d^T) * scale + shift): first PointNet layer, y recomputed, not read."""
    _chk(x2d, "pointnet_in_apply.x", torch.float32, 2)
    _chk(W2d, "pointnet_in_apply.W", torch.float32, 2)
    P, C = x2d.shape
    cout = W2d.shape[0]
    if W2d.shape[1] != C or not pointnet_in_ok(C, cout):
        raise ValueError(f"pointnet_in_apply: unsupported shape C={C} cout={cout}")
    if out_dtype == SplitImage.dtype:
        a = split_image_empty(P, cout, x2d.device, SPLIT_SCALE_ACT)
        check(_lib.load().pcaa_pointnet_in_apply(_p(x2d), C, _p(W2d), _p(scale), _p(shift), _p(a.img), PCAA_SPLIT_F16, P,
                                                 cout, _s()), "pcaa_pointnet_in_apply(split)")
        return a
    a = torch.empty((P, cout), dtype=out_dtype, device=x2d.device)
    check(_lib.load().pcaa_pointnet_in_apply(_p(x2d), C, _p(W2d), _p(scale), _p(shift), _p(a), _dt(a), P, cout, _s()),
          "pcaa_pointnet_in_apply")
    return a


def pointnet_in_bwd_stats(da, x2d, W2d, scale, shift, mean, rstd, tail=None):
    _chk(da, "pointnet_in_bwd_stats.da", dim=2)
    _chk(x2d, "pointnet_in_bwd_stats.x", torch.float32, 2)
    P, cout = da.shape
    C = x2d.shape[1]
    if x2d.shape[0] != P or tuple(W2d.shape) != (cout, C) or not pointnet_in_ok(C, cout):
        raise ValueError("pointnet_in_bwd_stats: unsupported shape")
    stats = new_stats(cout, da.device)
    if tail is not None:
        tail.arm(stats)
    check(_lib.load().pcaa_pointnet_in_bwd_stats(_p(da), _dt(da), _p(x2d), C, _p(W2d), _p(scale), _p(shift), _p(mean),
                                                 _p(rstd), _p(stats), NREP, P, cout, _s()),
          "pcaa_pointnet_in_bwd_stats")
    if tail is not None:
        tail.resolve(stats)
    return stats


def points_moments(x2d):
    """fp64 second moments and sums of the points (pcaa_points_moments) for pointnet_in_bwd_onepass."""
    _chk(x2d, "points_moments.x", torch.float32, 2)
    lib = _lib.load()
    n = lib.pcaa_points_moments_size()
    mom = STATS_POOL.take_raw(n) if (STATS_POOL is not None and STATS_POOL.buf.device == x2d.device) else None
    if mom is None:
        mom = torch.zeros(n, dtype=torch.float64, device=x2d.device)
    else:
        mom = mom[:n]
    check(lib.pcaa_points_moments(_p(x2d), x2d.shape[1], x2d.shape[0], _p(mom), _s()), "pcaa_points_moments")
    return mom


def pointnet_in_moment_coeffs(x2d, W2d, lin_bias, bn, mom=None, update_running=True):
    """Train-mode BatchNorm coefficients of the first PointNet layer from the points' moments (no pass over [P, cout]):
    -> (scale, shift, mean, rstd, mom)."""
    _chk(x2d, "pointnet_in_moment_coeffs.x", torch.float32, 2)
    P, C = x2d.shape
    cout = W2d.shape[0]
    if mom is None:
        mom = points_moments(x2d)
    tail = BnTailFwd(P, lin_bias, bn, cout, update_running=update_running)
    stats = mom                       # the armed finalize is matched by this pointer
    stats._pcaa_counter = mom         # (no arrival counter is used by this producer; any non-null word)
    tail.arm(stats)
    if not tail.armed:
        raise RuntimeError("pointnet_in_moment_coeffs: the carried finalize is switched off")
    check(_lib.load().pcaa_pointnet_in_moment_stats(_p(mom), _p(W2d), C, cout, _s()), "pcaa_pointnet_in_moment_stats")
    TAILS["taken"] += 1
    return tail.out + (mom,)


def pointnet_in_bwd_onepass(da, x2d, W2d, scale, shift, mean, rstd, tail, mom=None, out=None):
    """Backward of the recompute layer from ONE read of the incoming gradient: BatchNorm-backward statistics (their
    finalize: ``tail``, an ops.BnTailBwd) and G = dz^T.x in one launch, then dW = c0*G + c1*(W.x^T x) + c2*sum x.
    Returns dW [cout, C] (``out`` is overwritten)."""
    _chk(da, "pointnet_in_bwd_onepass.da", dim=2)
    _chk(x2d, "pointnet_in_bwd_onepass.x", torch.float32, 2)
    P, cout = da.shape
    C = x2d.shape[1]
    if x2d.shape[0] != P or tuple(W2d.shape) != (cout, C) or not pointnet_in_ok(C, cout):
        raise ValueError("pointnet_in_bwd_onepass: unsupported shape")
    lib = _lib.load()
    if mom is None:
        mom = points_moments(x2d)
    # G is accumulated against the points centred on their mean (csrc/pointnet_in.hip, MODE 3); under SyncBN the pivot is
    # the GLOBAL mean -- the same on every rank, so that the dropped term pivot (x) sum dy cancels in the all-reduce
    pivot_mom, pivot_count = mom, P
    if tail.sync is not None:
        pivot_mom = mom.clone()
        pivot_count = tail.sync(pivot_mom, P)
    stats = new_stats(cout, da.device)
    # replica b % NREP of workgroup b; zeroed with the step's statistics (StatsPool.begin: one fill per step) -- a fill of
    # its own sat on the critical path between the last product of the backward and this launch
    nG = NREP * cout * C
    raw = STATS_POOL.take_raw((nG + 1) // 2) if (STATS_POOL is not None and STATS_POOL.buf.device == da.device) else None
    if raw is None:
        G = torch.zeros((NREP, cout, C), dtype=torch.float32, device=da.device)
    else:
        G = raw.view(torch.float32)[:nG].view(NREP, cout, C)
    tail.arm(stats)
    check(lib.pcaa_pointnet_in_bwd_onepass(_p(da), _dt(da), _p(x2d), C, _p(W2d), _p(scale), _p(shift), _p(mean), _p(rstd),
                                           _p(stats), NREP, _p(G), P, cout, _p(pivot_mom), 1.0 / pivot_count, _s()),
          "pcaa_pointnet_in_bwd_onepass")
    coef, _, _ = tail.resolve(stats)
    if out is None:
        out = torch.empty((cout, C), dtype=torch.float32, device=da.device)
    check(lib.pcaa_pointnet_in_bwd_combine(_p(G), NREP, _p(W2d), _p(mom), _p(coef), _p(out), cout, C, P, _p(pivot_mom),
                                           1.0 / pivot_count, _s()), "pcaa_pointnet_in_bwd_combine")
    return out


def pointnet_in_bwd_wgrad(da, x2d, W2d, scale, shift, coef, out=None, out_is_zero=False, dz_is_pre=False):
    _chk(da, "pointnet_in_bwd_wgrad.da", dim=2)
    _chk(x2d, "pointnet_in_bwd_wgrad.x", torch.float32, 2)
    P, cout = da.shape
    C = x2d.shape[1]
    if x2d.shape[0] != P or tuple(W2d.shape) != (cout, C) or not pointnet_in_ok(C, cout):
        raise ValueError("pointnet_in_bwd_wgrad: unsupported shape")
    if out is None:
        out = torch.zeros((cout, C), dtype=torch.float32, device=da.device)
    elif not out_is_zero:
        out.zero_()
    check(_lib.load().pcaa_pointnet_in_bwd_wgrad(_p(da), _dt(da), _p(x2d), C, _p(W2d), _p(scale), _p(shift), _p(coef),
                                                 _p(out), P, cout, int(bool(dz_is_pre)), _s()),
          "pcaa_pointnet_in_bwd_wgrad")
    return out


def pointnet_in_wgrad(dy, x2d, out=None, out_is_zero=False):
    _chk(dy, "pointnet_in_wgrad.dy", dim=2)
    _chk(x2d, "pointnet_in_wgrad.x", torch.float32, 2)
    P, cout = dy.shape
    C = x2d.shape[1]
    if x2d.shape[0] != P or not pointnet_in_ok(C, cout):
        raise ValueError("pointnet_in_wgrad: unsupported shape")
    if out is None:
        out = torch.zeros((cout, C), dtype=torch.float32, device=dy.device)
    elif not out_is_zero:
        out.zero_()
    check(_lib.load().pcaa_pointnet_in_wgrad(_p(dy), _dt(dy), _p(x2d), C, _p(out), P, cout, _s()),
          "pcaa_pointnet_in_wgrad")
    return out


# ------------------------------------------------------------------ BatchNorm pieces
def bn_finalize(stats, count, lin_bias, bn, ch, update_running=True):
    dev = stats.device
    scale = torch.empty(ch, dtype=torch.float32, device=dev)
    shift = torch.empty_like(scale)
    mean = torch.empty_like(scale)
    rstd = torch.empty_like(scale)
    lib = _lib.load()
    rm = bn.running_mean if update_running else None
    rv = bn.running_var if update_running else None
    nbt = bn.num_batches_tracked if update_running else None
    check(lib.pcaa_bn_finalize(_p(stats), NREP, int(count), _p(lin_bias), _p(bn.weight), _p(bn.bias),
                               _p(rm), _p(rv), _p(nbt), BN_MOMENTUM if bn.momentum is None else bn.momentum,
                               bn.eps, _p(scale), _p(shift), _p(mean), _p(rstd), ch, _s()), "pcaa_bn_finalize")
    return scale, shift, mean, rstd


def bn_eval_coeffs(bn, ch, lin_bias=None):
    dev = bn.weight.device
    scale = torch.empty(ch, dtype=torch.float32, device=dev)
    shift = torch.empty_like(scale)
    check(_lib.load().pcaa_bn_eval_coeffs(_p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var),
                                          _p(lin_bias), bn.eps, _p(scale), _p(shift), ch, _s()),
          "pcaa_bn_eval_coeffs")
    return scale, shift


def bn_act_fwd(y, scale, shift):
    _chk(y, "bn_act_fwd.y", dim=2)
    a = torch.empty_like(y)
    check(_lib.load().pcaa_bn_act_fwd(_p(y), _p(a), _dt(y), _p(scale), _p(shift), y.shape[0], y.shape[1], _s()),
          "pcaa_bn_act_fwd")
    return a


def bn_act_meanpool_fwd(y, scale, shift, groups, group_rows, mean=None, rstd=None):
    """Mean over each group of rows of ELU(BN(y)).  With mean/rstd (training) also returns the
    per-(group, channel) sums (e1, e2) that bn_pool_bwd_stats turns into the backward statistics."""
    _chk(y, "bn_act_meanpool_fwd.y", dim=2)
    if y.shape[0] != groups * group_rows:
        raise ValueError("bn_act_meanpool_fwd: rows != groups*group_rows")
    out = torch.empty((groups, y.shape[1]), dtype=torch.float32, device=y.device)
    e = None
    if mean is not None:
        e = torch.empty((2, groups, y.shape[1]), dtype=torch.float32, device=y.device)
    check(_lib.load().pcaa_bn_act_meanpool_fwd(_p(y), _dt(y), _p(scale), _p(shift), _p(mean), _p(rstd), _p(out),
                                               _p(e[0]) if e is not None else None,
                                               _p(e[1]) if e is not None else None,
                                               groups, group_rows, y.shape[1], _s()), "pcaa_bn_act_meanpool_fwd")
    return out if e is None else (out, e)


def bn_pool_bwd_stats(dpool, e, pool_scale, tail=None):
    """BatchNorm-backward statistics of a mean-pooled layer from the forward's (e1, e2) sums."""
    _chk(dpool, "bn_pool_bwd_stats.dpool", torch.float32, 2)
    _chk(e, "bn_pool_bwd_stats.e", torch.float32, 3)
    groups, ch = dpool.shape
    if tuple(e.shape) != (2, groups, ch):
        raise ValueError("bn_pool_bwd_stats: e shape")
    stats = new_stats(ch, dpool.device)
    if tail is not None:
        tail.arm(stats)
    check(_lib.load().pcaa_bn_pool_bwd_stats(_p(dpool), _p(e[0]), _p(e[1]), float(pool_scale), _p(stats), NREP,
                                             groups, ch, _s()), "pcaa_bn_pool_bwd_stats")
    if tail is not None:
        tail.resolve(stats)
    return stats


def bn_act_bwd_dz(y, scale, shift, mean, rstd, *, da=None, dpool=None, group_rows=0, pool_scale=1.0, out=None):
    _chk(y, "bn_act_bwd_dz.y", dim=2)
    rows, ch = y.shape
    if da is not None:
        _chk(da, "bn_act_bwd_dz.da", y.dtype, 2)
        if da.shape != y.shape:
            raise ValueError("bn_act_bwd_dz: da shape")
    else:
        _chk(dpool, "bn_act_bwd_dz.dpool", torch.float32, 2)
        if dpool.shape[0] * group_rows != rows or dpool.shape[1] != ch:
            raise ValueError("bn_act_bwd_dz: dpool shape")
    dz = out if out is not None else torch.empty_like(y)
    stats = new_stats(ch, y.device)
    check(_lib.load().pcaa_bn_act_bwd_dz(_p(da), _p(dpool), int(group_rows), float(pool_scale), _p(y), _p(dz),
                                         _dt(y), _p(scale), _p(shift), _p(mean), _p(rstd), _p(stats), NREP,
                                         rows, ch, _s()), "pcaa_bn_act_bwd_dz")
    return dz, stats


def bn_act_bwd_stats(y, scale, shift, mean, rstd, *, da=None, dpool=None, group_rows=0, pool_scale=1.0, tail=None):
    """Statistics-only pass of the BatchNorm backward (sum dz, sum dz*yhat): reads, writes nothing."""
    _chk(y, "bn_act_bwd_stats.y", dim=2)
    rows, ch = y.shape
    stats = new_stats(ch, y.device)
    check(_lib.load().pcaa_bn_act_bwd_dz(_p(da), _p(dpool), int(group_rows), float(pool_scale), _p(y), None,
                                         _dt(y), _p(scale), _p(shift), _p(mean), _p(rstd), _p(stats), NREP,
                                         rows, ch, _s()), "pcaa_bn_act_bwd_dz(stats)")
    if tail is not None:
        tail.resolve(stats)
    return stats


def bn_bwd_dy_fused(y, scale, shift, coef, *, da=None, dpool=None, group_rows=0, pool_scale=1.0, out=None):
    """dy = coef0*(da*ELU'(BN(y))) + coef1*y + coef2 in one pass (dz never stored)."""
    _chk(y, "bn_bwd_dy_fused.y", dim=2)
    rows, ch = y.shape
    dy = out if out is not None else torch.empty_like(y)
    check(_lib.load().pcaa_bn_bwd_dy_fused(_p(da), _p(dpool), int(group_rows), float(pool_scale), _p(y), _p(dy),
                                           _dt(y), _p(scale), _p(shift), _p(coef), rows, ch, _s()),
          "pcaa_bn_bwd_dy_fused")
    return dy


def bn_bwd_finalize(stats, count, bn, mean, rstd, ch, dgamma=None, dbeta=None):
    dev = stats.device
    coef = torch.empty((3, ch), dtype=torch.float32, device=dev)
    dgamma = torch.empty(ch, dtype=torch.float32, device=dev) if dgamma is None else dgamma
    dbeta = torch.empty(ch, dtype=torch.float32, device=dev) if dbeta is None else dbeta
    check(_lib.load().pcaa_bn_bwd_finalize(_p(stats), NREP, int(count), _p(bn.weight), _p(mean), _p(rstd),
                                           _p(coef), _p(dgamma), _p(dbeta), ch, _s()), "pcaa_bn_bwd_finalize")
    return coef, dgamma, dbeta


def bn_bwd_dy(dz, y, coef, out=None):
    dy = out if out is not None else torch.empty_like(dz)
    check(_lib.load().pcaa_bn_bwd_dy(_p(dz), _p(y), _p(dy), _dt(y), _p(coef), y.shape[0], y.shape[1], _s()),
          "pcaa_bn_bwd_dy")
    return dy


# ------------------------------------------------------------------ small helpers
def bias_act_(y, bias, act):
    _chk(y, "bias_act.y", torch.float32, 2)
    check(_lib.load().pcaa_bias_act(_p(y), _p(bias), act, y.shape[0], y.shape[1], _s()), "pcaa_bias_act")
    return y


def elu_bwd_from_out(da, a, out=None):
    _chk(da, "elu_bwd.da", torch.float32)
    _chk(a, "elu_bwd.a", torch.float32)
    dz = out if out is not None else torch.empty_like(da)
    check(_lib.load().pcaa_elu_bwd_from_out(_p(da), _p(a), _p(dz), da.numel(), _s()), "pcaa_elu_bwd_from_out")
    return dz


def colsum(x, out=None):
    _chk(x, "colsum.x", torch.float32, 2)
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device) if out is None else out
    check(_lib.load().pcaa_colsum(_p(x), _p(out), x.shape[0], x.shape[1], _s()), "pcaa_colsum")
    return out


# ------------------------------------------------------------------ batch-skinny Linear layers (decoder)
def skinny_supported(M, N, K):
    return bool(_lib.load().pcaa_skinny_supported(int(M), int(N), int(K)))


def _skinny_timed(fn, flops, nbytes):
    timer = TIMER
    if timer is None or not timer.wants("gemm_skinny_kernel"):
        return fn()
    ev = _TorchEvents()
    fn()
    ev.end()
    timer.records.append(("gemm_skinny_kernel", flops, float(nbytes), ev))


def _w16_image(W16, W, what):
    """the bf16 image of weight W (pcaa_skinny_linear_*_w16): same shape, contiguous rows, bf16"""
    _chk(W16, what, torch.bfloat16, 2)
    if tuple(W16.shape) != tuple(W.shape) or W16.stride(1) != 1:
        raise ValueError(f"{what}: the bf16 image must have the weight's shape {tuple(W.shape)}, got {tuple(W16.shape)}")
    return W16


def skinny_linear_fwd(x, W, bias, act, exact=False, W16=None):
    """act(x[M,K] @ W[N,K]^T + bias) with M <= 64: one streaming pass over W.  ``exact``: fp32 products on the fp32
    matrix pipe (the parity modes) instead of bf16-rounded operands.  ``W16``: the bf16 image of W (every element the
    weight rounded to nearest even -- what the kernel's own conversion produces): streamed instead of W, same result."""
    _chk(x, "skinny_fwd.x", torch.float32, 2)
    _chk(W, "skinny_fwd.W", torch.float32, 2)
    M, K = x.shape
    N = W.shape[0]
    if W.shape[1] != K:
        raise ValueError("skinny_linear_fwd: shape mismatch")
    lib = _lib.load()
    ns = lib.pcaa_skinny_splits(2 if exact else 0, M, N, K)
    ws = torch.empty(ns * M * N, dtype=torch.float32, device=x.device)
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    if W16 is not None and not exact:
        Wsrc, fn, wb = _w16_image(W16, W, "skinny_fwd.W16"), lib.pcaa_skinny_linear_fwd_w16, 2
    else:
        Wsrc, fn, wb = W, (lib.pcaa_skinny_linear_fwd_exact if exact else lib.pcaa_skinny_linear_fwd), 4
    _skinny_timed(lambda: check(fn(_p(x), x.stride(0), _p(Wsrc), Wsrc.stride(0), _p(bias), act,
                                   _p(y), _p(ws), ws.numel(), M, N, K, ns, _s()),
                                "pcaa_skinny_linear_fwd"), 2.0 * M * N * K, wb * N * K + 4 * (M * K + M * N))
    return y


def skinny_linear_dgrad(dz, W, a_prev=None, out=None, accumulate=False, exact=False, W16=None):
    """dx[M,K] (=|+=) (dz[M,N] @ W[N,K]) * ELU'(a_prev) (a_prev: ELU OUTPUT of the layer below or None).
    ``W16``: the bf16 image of W, streamed instead of it (see skinny_linear_fwd)."""
    _chk(dz, "skinny_dgrad.dz", torch.float32, 2)
    _chk(W, "skinny_dgrad.W", torch.float32, 2)
    M, N = dz.shape
    K = W.shape[1]
    if W.shape[0] != N:
        raise ValueError("skinny_linear_dgrad: shape mismatch")
    if a_prev is not None:
        _chk(a_prev, "skinny_dgrad.a_prev", torch.float32)
        if a_prev.numel() != M * K:
            raise ValueError("skinny_linear_dgrad: a_prev size")
    if out is None:
        if accumulate:
            raise ValueError("skinny_linear_dgrad: accumulate needs out")
        out = torch.empty((M, K), dtype=torch.float32, device=dz.device)
    else:
        _chk(out, "skinny_dgrad.out", torch.float32)
        if out.numel() != M * K:
            raise ValueError("skinny_linear_dgrad: out size")
    lib = _lib.load()
    ns = lib.pcaa_skinny_splits(3 if exact else 1, M, N, K)
    ws = torch.empty(ns * M * K, dtype=torch.float32, device=dz.device)
    if W16 is not None and not exact:
        Wsrc, fn, wb = _w16_image(W16, W, "skinny_dgrad.W16"), lib.pcaa_skinny_linear_dgrad_w16, 2
    else:
        Wsrc, fn, wb = W, (lib.pcaa_skinny_linear_dgrad_exact if exact else lib.pcaa_skinny_linear_dgrad), 4
    _skinny_timed(lambda: check(fn(_p(dz), dz.stride(0), _p(Wsrc), Wsrc.stride(0), _p(out),
                                   _p(a_prev), int(bool(accumulate)), _p(ws), ws.numel(),
                                   M, N, K, ns, _s()),
                                "pcaa_skinny_linear_dgrad"), 2.0 * M * N * K, wb * N * K + 4 * (M * K + M * N))
    return out


def skinny_linear_wgrad(dz, x, out=None, exact=False):
    """dW[N,K] = dz[M,N]^T @ x[M,K] with M <= 64: one streaming store pass over dW."""
    _chk(dz, "skinny_wgrad.dz", torch.float32, 2)
    _chk(x, "skinny_wgrad.x", torch.float32, 2)
    M, N = dz.shape
    K = x.shape[1]
    if x.shape[0] != M:
        raise ValueError("skinny_linear_wgrad: shape mismatch")
    if out is None:
        out = torch.empty((N, K), dtype=torch.float32, device=dz.device)
    else:
        _chk(out, "skinny_wgrad.out")
        if out.numel() != N * K or out.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("skinny_linear_wgrad: out must hold N*K fp32 or bf16 elements")
    lib = _lib.load()
    if out.dtype == torch.bfloat16:       # the gradient as it crosses the wire (bf16 gradient buckets)
        if exact:
            raise ValueError("skinny_linear_wgrad: the exact variant writes fp32")
        _skinny_timed(lambda: check(lib.pcaa_skinny_linear_wgrad_bf16(_p(dz), dz.stride(0), _p(x), x.stride(0), _p(out), K,
                                                                      M, N, K, _s()),
                                    "pcaa_skinny_linear_wgrad_bf16"), 2.0 * M * N * K, 2 * N * K + 4 * (M * K + M * N))
        return out
    fn = lib.pcaa_skinny_linear_wgrad_exact if exact else lib.pcaa_skinny_linear_wgrad
    _skinny_timed(lambda: check(fn(_p(dz), dz.stride(0), _p(x), x.stride(0), _p(out), K, M, N, K, _s()),
                                "pcaa_skinny_linear_wgrad"), 2.0 * M * N * K, 4 * (N * K + M * K + M * N))
    return out


def skinny_linear_wgrad_adam_(dz, x, W, exp_avg, exp_avg_sq, beta1, beta2, eps, coef_dev, grad_scale=1.0, exact=False):
    """W[N,K] <- Adam(W, dz[M,N]^T @ x[M,K]) in place, moments too (M <= 64): the weight gradient never reaches HBM.
    ``coef_dev``: the two step-dependent scalars of the optimizer step in progress (StepCount.coef_dev)."""
    _chk(dz, "skinny_wgrad_adam.dz", torch.float32, 2)
    _chk(x, "skinny_wgrad_adam.x", torch.float32, 2)
    M, N = dz.shape
    K = x.shape[1]
    if x.shape[0] != M:
        raise ValueError("skinny_linear_wgrad_adam_: shape mismatch")
    for t, nm in ((W, "W"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, "skinny_wgrad_adam." + nm, torch.float32)
        if tuple(t.shape) != (N, K):
            raise ValueError(f"skinny_linear_wgrad_adam_: {nm} must be [{N},{K}], got {tuple(t.shape)}")
    _chk(coef_dev, "skinny_wgrad_adam.coef", torch.float32)
    lib = _lib.load()
    fn = lib.pcaa_skinny_linear_wgrad_adam_exact if exact else lib.pcaa_skinny_linear_wgrad_adam
    _skinny_timed(lambda: check(fn(
        _p(dz), dz.stride(0), _p(x), x.stride(0), _p(W), _p(exp_avg), _p(exp_avg_sq), K, M, N, K, float(beta1),
        float(beta2), float(eps), float(grad_scale), _p(coef_dev), _s()), "pcaa_skinny_linear_wgrad_adam"),
        2.0 * M * N * K, 4 * (6 * N * K + M * K + M * N))
    return W


def gathered_rows_alloc(m):
    """rows a gathered operand of ``m`` valid rows must be allocated for (skinny_linear_wgrad_adam_rows_ reads whole
    64-row chunks: 64, 128, 256 or 512); None if m is beyond the kernel's 512"""
    for r in (64, 128, 256, 512):
        if m <= r:
            return r
    return None


def skinny_linear_wgrad_adam_rows_(dz_all, x_all, m, W, exp_avg, exp_avg_sq, beta1, beta2, eps, coef_dev, grad_scale=1.0):
    """The fused weight-gradient + Adam update from GATHERED rows (data parallel; pcaa_skinny_linear_wgrad_adam_rows):
    ``dz_all`` [R, N], ``x_all`` [R, K] hold the rows of all ranks stacked in their first ``m`` rows (R =
    gathered_rows_alloc(m); the rows behind m must be finite -- the buffers are zero-initialised once).
    W[N,K] <- Adam(W, grad_scale * dz_all[:m]^T @ x_all[:m]) in place, bf16 products."""
    _chk(dz_all, "skinny_wgrad_adam_rows.dz", torch.float32, 2)
    _chk(x_all, "skinny_wgrad_adam_rows.x", torch.float32, 2)
    R, N = dz_all.shape
    K = x_all.shape[1]
    need = gathered_rows_alloc(int(m))
    if need is None or x_all.shape[0] != R or R < need:
        raise ValueError(f"skinny_linear_wgrad_adam_rows_: {m} valid rows need buffers of {need} rows, got {R} / {x_all.shape[0]}")
    for t, nm in ((W, "W"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, "skinny_wgrad_adam_rows." + nm, torch.float32)
        if tuple(t.shape) != (N, K):
            raise ValueError(f"skinny_linear_wgrad_adam_rows_: {nm} must be [{N},{K}], got {tuple(t.shape)}")
    _chk(coef_dev, "skinny_wgrad_adam_rows.coef", torch.float32)
    _skinny_timed(lambda: check(_lib.load().pcaa_skinny_linear_wgrad_adam_rows(
        _p(dz_all), dz_all.stride(0), _p(x_all), x_all.stride(0), _p(W), _p(exp_avg), _p(exp_avg_sq), K, int(m), N, K,
        float(beta1), float(beta2), float(eps), float(grad_scale), _p(coef_dev), R, _s()),
        "pcaa_skinny_linear_wgrad_adam_rows"), 2.0 * m * N * K, 4 * (6 * N * K + m * K + m * N))
    return W


PACK_ROWS = 64        # batch rows of one packed chunk (pcaa_pack_rows_t16)
PACK_MAX_CHUNKS = 8   # chunks one pcaa_skinny_linear_wgrad_adam_t16 launch contracts over


def packed_chunk_elems(N, K):
    """bf16 elements of one rank's packed weight-gradient operands of an [N, K] layer: (N + K) columns x 64 rows"""
    return int(_lib.load().pcaa_packed_chunk_elems(int(N), int(K)))


def pack_rows_t16(dz, x, out=None):
    """One rank's two weight-gradient operands of a batch-skinny layer -- dz [rows, N], x [rows, K], rows <= 64 -- as ONE
    packed chunk [(N + K) * 64] bf16 (transposed: a 128-B row per column of dz, then of x; rounded to nearest even; zero
    behind ``rows``): the form skinny_linear_wgrad_adam_t16_ contracts over and the data-parallel step all-gathers."""
    _chk(dz, "pack_rows_t16.dz", torch.float32, 2)
    _chk(x, "pack_rows_t16.x", torch.float32, 2)
    rows, N = dz.shape
    K = x.shape[1]
    if x.shape[0] != rows or rows > PACK_ROWS:
        raise ValueError(f"pack_rows_t16: dz {tuple(dz.shape)} / x {tuple(x.shape)}: equal row counts <= {PACK_ROWS}")
    n = packed_chunk_elems(N, K)
    if out is None:
        out = torch.empty(n, dtype=torch.bfloat16, device=dz.device)
    else:
        _chk(out, "pack_rows_t16.out", torch.bfloat16)
        if out.numel() != n:
            raise ValueError(f"pack_rows_t16: out must hold {n} bf16 elements, got {out.numel()}")
    check(_lib.load().pcaa_pack_rows_t16(_p(dz), dz.stride(0), N, _p(x), x.stride(0), K, rows, _p(out), _s()),
          "pcaa_pack_rows_t16")
    return out


def skinny_linear_wgrad_adam_t16_(packed, chunks, W, exp_avg, exp_avg_sq, beta1, beta2, eps, coef_dev, grad_scale=1.0):
    """The fused weight-gradient + Adam update from the ranks' PACKED operands (data parallel, round 6;
    pcaa_skinny_linear_wgrad_adam_t16): ``packed`` [>= chunks, (N + K) * 64] bf16, one pack_rows_t16 chunk per rank
    (a gathered buffer).  W[N,K] <- Adam(W, grad_scale * sum over the chunks of dz_c^T @ x_c) in place, bf16 products."""
    if (not isinstance(packed, torch.Tensor) or not packed.is_cuda or packed.dtype != torch.bfloat16 or packed.dim() != 2
            or packed.stride(1) != 1):
        raise TypeError("skinny_linear_wgrad_adam_t16_: packed must be a [chunks, >= (N + K) * 64] bf16 tensor on the HIP "
                        "device with contiguous rows")
    for t, nm in ((W, "W"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, "skinny_wgrad_adam_t16." + nm, torch.float32, 2)
        if tuple(t.shape) != tuple(W.shape):
            raise ValueError(f"skinny_linear_wgrad_adam_t16_: {nm} must be {tuple(W.shape)}, got {tuple(t.shape)}")
    N, K = W.shape
    chunks = int(chunks)
    if not 1 <= chunks <= min(PACK_MAX_CHUNKS, packed.shape[0]) or packed.shape[1] < packed_chunk_elems(N, K):
        raise ValueError(f"skinny_linear_wgrad_adam_t16_: {chunks} chunks of an [{N},{K}] layer do not fit a packed buffer "
                         f"{tuple(packed.shape)} (at most {PACK_MAX_CHUNKS} chunks of {packed_chunk_elems(N, K)} elements)")
    _chk(coef_dev, "skinny_wgrad_adam_t16.coef", torch.float32)
    m = chunks * PACK_ROWS
    _skinny_timed(lambda: check(_lib.load().pcaa_skinny_linear_wgrad_adam_t16(
        _p(packed), packed.stride(0), chunks, _p(W), _p(exp_avg), _p(exp_avg_sq), K, N, K, float(beta1), float(beta2),
        float(eps), float(grad_scale), _p(coef_dev), _s()), "pcaa_skinny_linear_wgrad_adam_t16"),
        2.0 * m * N * K, 24 * N * K + 2 * m * (N + K))
    return W


def total(x, scale=1.0, out=None):
    _chk(x, "sum.x", torch.float32)
    out = torch.empty((), dtype=torch.float32, device=x.device) if out is None else _chk(out, "sum.out", torch.float32)
    if out.numel() != 1:
        raise ValueError("total: out must hold one float")
    check(_lib.load().pcaa_sum(_p(x), x.numel(), float(scale), _p(out), _s()), "pcaa_sum")
    return out


def rowsum(x, scale=1.0):
    _chk(x, "rowsum.x", torch.float32, 2)
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(_lib.load().pcaa_rowsum(_p(x), _p(out), x.shape[0], x.shape[1], float(scale), _s()), "pcaa_rowsum")
    return out


def scale_by_device_scalar(x, s):
    _chk(x, "scale.x", torch.float32)
    _chk(s, "scale.s", torch.float32)
    out = torch.empty_like(x)
    check(_lib.load().pcaa_scale_by_device_scalar(_p(x), _p(s), _p(out), x.numel(), _s()), "pcaa_scale_by_device_scalar")
    return out


def scale_rows(x, s):
    _chk(x, "scale_rows.x", torch.float32)
    _chk(s, "scale_rows.s", torch.float32)
    rows = s.numel()
    out = torch.empty_like(x)
    check(_lib.load().pcaa_scale_rows(_p(x), _p(s), _p(out), rows, x.numel() // rows, _s()), "pcaa_scale_rows")
    return out


def prior_sample(z0, means, gt, K):
    _chk(z0, "prior.z0", torch.float32, 2)
    _chk(means, "prior.means", torch.float32, 2)
    _chk(gt, "prior.gt", torch.int64, 1)
    B, D = z0.shape
    z = torch.empty_like(z0)
    oh = torch.empty((B, K), dtype=torch.float32, device=z0.device)
    check(_lib.load().pcaa_prior_sample(_p(z0), _p(means), _p(gt), B, K, D, _p(z), _p(oh), _s()), "pcaa_prior_sample")
    return z, oh


def pack_points(x):
    """logical [B,C,T,N] (any strides) -> contiguous point-major [B,T,N,C]."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        raise RuntimeError("pack_points: expected a 4-D float32 tensor on the HIP device")
    B, C, T, N = x.shape
    out = torch.empty((B, T, N, C), dtype=torch.float32, device=x.device)
    sb, sc, st, sn = x.stride()
    check(_lib.load().pcaa_pack_points(_p(x), sb, sc, st, sn, _p(out), B, C, T, N, _s()), "pcaa_pack_points")
    return out


def gather_rows(src, idx, out=None, err_flag=None):
    """out[r] = src[idx[r]] along dim 0 (device-side batch assembly; rows must be multiples of 16 bytes)."""
    if not (isinstance(src, torch.Tensor) and src.is_cuda and src.is_contiguous() and src.dim() >= 1):
        raise RuntimeError("gather_rows: src must be a contiguous tensor on the HIP device (this package has no CPU path)")
    _chk(idx, "gather_rows.idx", torch.int64, 1)
    row_bytes = src[0].numel() * src.element_size() if src.dim() > 1 else src.element_size()
    if row_bytes % 16:
        raise ValueError(f"gather_rows: rows of {row_bytes} bytes are not a multiple of 16")
    n = idx.numel()
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    elif out.shape != (n,) + tuple(src.shape[1:]) or out.dtype != src.dtype or not out.is_contiguous():
        raise ValueError("gather_rows: out does not match")
    if err_flag is not None:
        _chk(err_flag, "gather_rows.err_flag", torch.int32)
    if n:
        check(_lib.load().pcaa_gather_rows(_p(src), src.shape[0], row_bytes, _p(idx), _p(out), n, _p(err_flag), _s()),
              "pcaa_gather_rows")
    return out


def dtc_im2col(a, B, T, Cin, d):
    _chk(a, "im2col.a", torch.float32, 2)
    col = torch.empty((B * T, 3 * Cin), dtype=torch.float32, device=a.device)
    check(_lib.load().pcaa_dtc_im2col(_p(a), _p(col), B, T, Cin, d, _s()), "pcaa_dtc_im2col")
    return col


def dtc_col2im(dcol, B, T, Cin, d):
    _chk(dcol, "col2im.dcol", torch.float32, 2)
    da = torch.empty((B * T, Cin), dtype=torch.float32, device=dcol.device)
    check(_lib.load().pcaa_dtc_col2im(_p(dcol), _p(da), B, T, Cin, d, _s()), "pcaa_dtc_col2im")
    return da


# ------------------------------------------------------------------ losses
def chamfer(preds, gts, want_grad, grad_scale=1.0, grad_per_b=None):
    """preds/gts logical [B,C,T,N] fp32 with arbitrary strides.  Returns
    (frame_loss [B,T], dpreds contiguous [B,C,T,N] or None)."""
    for t, nm in ((preds, "preds"), (gts, "gts")):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4:
            raise RuntimeError(f"chamfer: {nm} must be a 4-D float32 tensor on the HIP device")
    if preds.shape != gts.shape:
        raise ValueError("chamfer: shape mismatch")
    B, C, T, N = preds.shape
    fl = torch.empty((B, T), dtype=torch.float32, device=preds.device)
    dp = torch.empty((B, C, T, N), dtype=torch.float32, device=preds.device) if want_grad else None
    ds = dp.stride() if want_grad else (0, 0, 0, 0)
    ps, gs = preds.stride(), gts.stride()
    check(_lib.load().pcaa_chamfer_fwd_bwd(_p(preds), ps[0], ps[1], ps[2], ps[3], _p(gts), gs[0], gs[1], gs[2], gs[3],
                                           B, T, N, C, _p(fl), _p(dp), ds[0], ds[1], ds[2], ds[3],
                                           float(grad_scale), _p(grad_per_b), _s()), "pcaa_chamfer_fwd_bwd")
    return fl, dp


def cross_entropy(logits, target=None, want_loss=True, want_grad=False, grad_scale=1.0, want_preds=False,
                  err_flag=None):
    """``err_flag`` (device int32[1]): the kernel raises it when a target is outside [0,K) and the caller checks
    it when convenient (the trainer: once per epoch).  Without it the targets are range-checked here, on the
    host (one synchronisation) -- torch's CrossEntropyLoss raises on such a label, so must this."""
    _chk(logits, "ce.logits", torch.float32, 2)
    B, K = logits.shape
    if target is not None:
        _chk(target, "ce.target", torch.int64, 1)
        if target.shape[0] != B:
            raise ValueError(f"cross_entropy: {target.shape[0]} targets for {B} rows of logits")
        if err_flag is None:
            lo, hi = int(target.min().item()), int(target.max().item())
            if lo < 0 or hi >= K:
                raise IndexError(f"cross_entropy: target {lo if lo < 0 else hi} is out of bounds for {K} classes")
        else:
            _chk(err_flag, "ce.err_flag", torch.int32)
    dev = logits.device
    loss = torch.empty((), dtype=torch.float32, device=dev) if (want_loss and target is not None) else None
    dl = torch.empty_like(logits) if want_grad else None
    pr = torch.empty(B, dtype=torch.int64, device=dev) if want_preds else None
    check(_lib.load().pcaa_cross_entropy(_p(logits), _p(target), B, K, _p(loss), _p(dl), float(grad_scale), _p(pr),
                                         _p(err_flag) if target is not None else None, _s()),
          "pcaa_cross_entropy")
    return loss, dl, pr


# ------------------------------------------------------------------ OR-CED heads (csrc/orced.hip)
def orced_heads_fwd(x4, Wmu, bmu, Wlv, blv, eps, Wc, bc):
    """-> (logits [B,K], sup_fv, mu, logvar [B,L])"""
    for t, nm in ((x4, "x4"), (Wmu, "Wmu"), (Wlv, "Wlv"), (eps, "eps"), (Wc, "Wc")):
        _chk(t, f"orced_heads_fwd.{nm}", torch.float32, 2)
    B, d_in = x4.shape
    L, K = Wmu.shape[0], Wc.shape[0]
    lib = _lib.load()
    if (tuple(Wmu.shape) != (L, d_in) or tuple(Wlv.shape) != (L, d_in) or tuple(eps.shape) != (B, L)
            or tuple(Wc.shape) != (K, L) or not lib.pcaa_orced_heads_supported(B, K, d_in, L)):
        raise ValueError("orced_heads_fwd: unsupported shapes")
    mu = torch.empty((B, L), dtype=torch.float32, device=x4.device)
    logvar, sup = torch.empty_like(mu), torch.empty_like(mu)
    logits = torch.empty((B, K), dtype=torch.float32, device=x4.device)
    check(lib.pcaa_orced_heads_fwd(_p(x4), _p(Wmu), _p(bmu), _p(Wlv), _p(blv), _p(eps), _p(Wc), _p(bc), _p(mu), _p(logvar),
                                   _p(sup), _p(logits), B, K, d_in, L, _s()), "pcaa_orced_heads_fwd")
    return logits, sup, mu, logvar


def orced_heads_bwd(x4, eps, logvar, sup, Wmu, Wlv, Wc, d_logits, d_sup, d_mu, d_logvar, need_dx=True):
    """-> (dx4 or None, dWmu, dbmu, dWlv, dblv, dWc, dbc)"""
    B, d_in = x4.shape
    L, K = Wmu.shape[0], Wc.shape[0]
    dev = x4.device
    for t in (d_logits, d_sup, d_mu, d_logvar):
        if t is not None:
            _chk(t, "orced_heads_bwd.grad", torch.float32, 2)
    ws = torch.empty(2 * B * L, dtype=torch.float32, device=dev)
    dWmu, dWlv = torch.empty_like(Wmu), torch.empty_like(Wlv)
    dbmu = torch.empty(L, dtype=torch.float32, device=dev)
    dblv = torch.empty_like(dbmu)
    dWc = torch.empty_like(Wc)
    dbc = torch.empty(K, dtype=torch.float32, device=dev)
    dx4 = torch.empty_like(x4) if need_dx else None
    check(_lib.load().pcaa_orced_heads_bwd(_p(x4), _p(eps), _p(logvar), _p(sup), _p(Wmu), _p(Wlv), _p(Wc), _p(d_logits),
                                           _p(d_sup), _p(d_mu), _p(d_logvar), _p(ws), _p(dWmu), _p(dbmu), _p(dWlv), _p(dblv),
                                           _p(dWc), _p(dbc), _p(dx4), B, K, d_in, L, _s()), "pcaa_orced_heads_bwd")
    return dx4, dWmu, dbmu, dWlv, dblv, dWc, dbc


def orced_kl(mu, logvar, mu_k, want_loss=True, gscale=None):
    """CG_kl_divergence and (``gscale`` given: the upstream gradient) its gradients w.r.t. mu, logvar, mu_k."""
    for t in (mu, logvar, mu_k):
        _chk(t, "orced_kl", torch.float32, 2)
    B, L = mu.shape
    if tuple(logvar.shape) != (B, L) or tuple(mu_k.shape) != (B, L):
        raise ValueError("orced_kl: shape mismatch")
    loss = torch.empty((), dtype=torch.float32, device=mu.device) if want_loss else None
    grads = (torch.empty_like(mu), torch.empty_like(mu), torch.empty_like(mu)) if gscale is not None else (None, None, None)
    check(_lib.load().pcaa_orced_kl(_p(mu), _p(logvar), _p(mu_k), _p(loss), _p(grads[0]), _p(grads[1]), _p(grads[2]),
                                    float(gscale if gscale is not None else 0.0), B, L, _s()), "pcaa_orced_kl")
    return loss, grads


# ------------------------------------------------------------------ discriminator
def _disc_params(m):
    lin = (m.model[0], m.model[2], m.model[4])
    return [lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight, lin[2].bias]


def disc_workspace(B, K, device):
    n = _lib.load().pcaa_disc_workspace_bytes(B, K)
    return torch.empty(n // 4, dtype=torch.float32, device=device)


def disc_forward(x, label, params):
    _chk(x, "disc.x", torch.float32, 2)
    _chk(label, "disc.label", torch.float32, 2)
    B, K = label.shape
    if x.shape != (B, 32) or params[0].shape != (64, 32 + K):
        raise ValueError("disc_forward: shapes")
    out = torch.empty((B, 1), dtype=torch.float32, device=x.device)
    check(_lib.load().pcaa_disc_forward(_p(x), _p(label), B, K, *[_p(p) for p in params], _p(out), _s()),
          "pcaa_disc_forward")
    return out


def disc_backward(x, label, params, gout, want_dx=True, want_dlabel=False, want_params=True, grads_out=None, dx_out=None):
    B, K = label.shape
    _chk(gout, "disc.gout", torch.float32)
    if gout.numel() != B:
        raise ValueError("disc_backward: gout size")
    dev = x.device
    if dx_out is not None:
        _chk(dx_out, "disc.dx_out", torch.float32)
        if tuple(dx_out.shape) != tuple(x.shape):
            raise ValueError("disc_backward: dx_out must have x's shape")
    dx = (dx_out if dx_out is not None else torch.empty_like(x)) if want_dx else None
    dl = torch.empty_like(label) if want_dlabel else None
    if want_params:
        grads = grads_out if grads_out is not None else [torch.empty_like(p) for p in params]
        ws = disc_workspace(B, K, dev)
    else:
        grads = [None] * 6
        ws = None
    check(_lib.load().pcaa_disc_backward(_p(x), _p(label), B, K, *[_p(p) for p in params], _p(gout), _p(dx), _p(dl),
                                         *[_p(g) for g in grads], _p(ws), 0 if ws is None else ws.numel() * 4, _s()),
          "pcaa_disc_backward")
    return dx, dl, (grads if want_params else None)


def disc_backward_backward(x, label, params, gout, gbar, want_dx=True, want_dlabel=False, want_dgout=True,
                           want_params=True):
    """Backward of disc_backward's ``dx`` output: gradients of sum <gbar, dx> w.r.t. (x, label, gout, params)."""
    for t, nm in ((x, "x"), (label, "label"), (gout, "gout"), (gbar, "gbar")):
        _chk(t, f"disc_bb.{nm}", torch.float32)
    B, K = label.shape
    if x.shape != (B, 32) or gbar.shape != (B, 32) or gout.numel() != B:
        raise ValueError("disc_backward_backward: shapes")
    dev = x.device
    dx2 = torch.empty_like(x) if want_dx else None
    dl2 = torch.empty_like(label) if want_dlabel else None
    dgo = torch.empty(B, dtype=torch.float32, device=dev) if want_dgout else None
    grads = [torch.empty_like(p) for p in params] if want_params else [None] * 6
    ws = disc_workspace(B, K, dev)
    check(_lib.load().pcaa_disc_backward_backward(_p(x), _p(label), B, K, *[_p(p) for p in params], _p(gout), _p(gbar),
                                                  _p(dx2), _p(dl2), _p(dgo), *[_p(g) for g in grads], _p(ws),
                                                  ws.numel() * 4, _s()), "pcaa_disc_backward_backward")
    return dx2, dl2, dgo, (grads if want_params else None)


def disc_wgan_gp(z, fv, label, alphas, params, gp_weight, grads_out=None, want_dz=False, losses_out=None):
    for t, nm in ((z, "z"), (fv, "fv"), (label, "label"), (alphas, "alphas")):
        _chk(t, f"wgan.{nm}", torch.float32)
    B, K = label.shape
    if z.shape != (B, 32) or fv.shape != (B, 32) or alphas.numel() != B:
        raise ValueError("disc_wgan_gp: shapes")
    dev = z.device
    losses = torch.empty(2, dtype=torch.float32, device=dev) if losses_out is None else _chk(losses_out, "wgan.losses_out", torch.float32)
    if losses.numel() != 2:
        raise ValueError("disc_wgan_gp: losses_out must hold 2 floats")
    grads = grads_out if grads_out is not None else [torch.empty_like(p) for p in params]
    ws = disc_workspace(B, K, dev)
    dz = torch.empty_like(z) if want_dz else None
    check(_lib.load().pcaa_disc_wgan_gp(_p(z), _p(fv), _p(label), _p(alphas), B, K, *[_p(p) for p in params],
                                        float(gp_weight), _p(losses), *[_p(g) for g in grads], _p(dz), _p(ws),
                                        ws.numel() * 4, _s()), "pcaa_disc_wgan_gp")
    if want_dz:
        return losses, grads, dz          # dz = d(d_loss)/dz (variant 1: learned centroids)
    return losses, grads


# ------------------------------------------------------------------ optimizer
def adam_step_(p, g, m, v, lr, b1, b2, eps, step, grad_scale=1.0, max_blocks=0):
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, f"adam.{nm}", torch.float32)
    n = p.numel()
    if not (g.numel() == n and m.numel() == n and v.numel() == n):
        raise ValueError("adam_step_: size mismatch")
    check(_lib.load().pcaa_adam_step(_p(p), _p(g), _p(m), _p(v), n, float(lr), float(b1), float(b2), float(eps),
                                     int(step), float(grad_scale), int(max_blocks), _s()), "pcaa_adam_step")


def adam_advance_(step_dev, coef_dev, lr, b1, b2):
    """Device-side optimizer step count: ``step_dev`` int32[1] += 1 and ``coef_dev`` float32[2] =
    (lr / (1 - b1^step), 1 / sqrt(1 - b2^step)) -- see pcaa_adam_advance."""
    _chk(step_dev, "adam.step_dev", torch.int32)
    _chk(coef_dev, "adam.coef_dev", torch.float32)
    if step_dev.numel() != 1 or coef_dev.numel() != 2:
        raise ValueError("adam_advance_: step_dev is int32[1], coef_dev float32[2]")
    check(_lib.load().pcaa_adam_advance(_p(step_dev), _p(coef_dev), float(lr), float(b1), float(b2), _s()),
          "pcaa_adam_advance")


def adam_step_dev_(p, g, m, v, b1, b2, eps, coef_dev, grad_scale=1.0, max_blocks=0):
    """adam_step_ with the step-dependent scalars read from ``coef_dev`` (adam_advance_)."""
    for nm, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _chk(t, f"adam.{nm}", torch.float32)
    _chk(coef_dev, "adam.coef_dev", torch.float32)
    n = p.numel()
    if not (g.numel() == m.numel() == v.numel() == n) or coef_dev.numel() != 2:
        raise ValueError("adam_step_dev_: size mismatch")
    check(_lib.load().pcaa_adam_step_dev(_p(p), _p(g), _p(m), _p(v), n, float(b1), float(b2), float(eps),
                                         _p(coef_dev), float(grad_scale), int(max_blocks), _s()),
          "pcaa_adam_step_dev")


def adam_step_dev_g16_(p, g16, m, v, b1, b2, eps, coef_dev, grad_scale=1.0, max_blocks=0):
    """adam_step_dev_ with a bf16 gradient (a reduced bf16 gradient bucket, consumed without widening it first)."""
    for nm, t in (("p", p), ("m", m), ("v", v)):
        _chk(t, f"adam_g16.{nm}", torch.float32)
    _chk(g16, "adam_g16.g", torch.bfloat16)
    _chk(coef_dev, "adam_g16.coef_dev", torch.float32)
    n = p.numel()
    if not (g16.numel() == m.numel() == v.numel() == n) or coef_dev.numel() != 2:
        raise ValueError("adam_step_dev_g16_: size mismatch")
    check(_lib.load().pcaa_adam_step_dev_g16(_p(p), _p(g16), _p(m), _p(v), n, float(b1), float(b2), float(eps),
                                             _p(coef_dev), float(grad_scale), int(max_blocks), _s()),
          "pcaa_adam_step_dev_g16")


# ------------------------------------------------------------------ fused MLP heads (heads.hip)
def heads_supported(B, K, d_in, d_sup, d_head, d_proj, backward):
    return bool(_lib.load().pcaa_heads_supported(int(B), int(K), int(d_in), int(d_sup), int(d_head), int(d_proj),
                                                 1 if backward else 0))


def heads_fwd(x4, W1, b1, Wh, bh, W2, b2, Wg=None, bg=None):
    """One launch for MLP_sup1 -> (MLP_head) -> MLP_sup2 and, if given, the decoder projection head.
    Returns (sup_fv [B,32], h [B,16] or None, logits [B,K], hproj [B,64] or None)."""
    _chk(x4, "heads_fwd.x4", torch.float32, 2)
    for nm, t in (("W1", W1), ("b1", b1), ("W2", W2), ("b2", b2)):
        _chk(t, f"heads_fwd.{nm}", torch.float32)
    B, K = x4.shape[0], W2.shape[0]
    d_head = Wh.shape[0] if Wh is not None else 0
    d_proj = Wg.shape[0] if Wg is not None else 0
    if (tuple(W1.shape) != (32, x4.shape[1]) or W2.shape[1] != (d_head or 32)
            or not heads_supported(B, K, x4.shape[1], 32, d_head, d_proj, False)):
        raise ValueError(f"heads_fwd: unsupported shapes x4 {tuple(x4.shape)} W1 {tuple(W1.shape)} W2 {tuple(W2.shape)}")
    dev = x4.device
    sup = torch.empty((B, 32), dtype=torch.float32, device=dev)
    h = torch.empty((B, d_head), dtype=torch.float32, device=dev) if Wh is not None else None
    logits = torch.empty((B, K), dtype=torch.float32, device=dev)
    hproj = torch.empty((B, d_proj), dtype=torch.float32, device=dev) if Wg is not None else None
    check(_lib.load().pcaa_heads_fwd(_p(x4), _p(W1), _p(b1), _p(Wh), _p(bh), _p(W2), _p(b2), _p(Wg), _p(bg),
                                     _p(sup), _p(h), _p(logits), _p(hproj), B, K, _s()), "pcaa_heads_fwd")
    return sup, h, logits, hproj


def heads_bwd(x4, sup_fv, h, logits, hproj, W1, Wh, W2, Wg, d_logits, d_sup, d_hproj, outs=None):
    """One launch for the backward of heads_fwd.  ``outs``: optional dict of destination views
    (dW1, db1, dWh, dbh, dW2, db2, dWg, dbg) -- missing ones are allocated.  Returns (grads dict, dx4)."""
    B, K = x4.shape[0], W2.shape[0]
    for nm, t in (("x4", x4), ("sup_fv", sup_fv), ("logits", logits), ("W1", W1), ("W2", W2)):
        _chk(t, f"heads_bwd.{nm}", torch.float32)
    for nm, t, shape in (("d_logits", d_logits, (B, K)), ("d_sup", d_sup, (B, 32)), ("d_hproj", d_hproj, (B, 64))):
        if t is not None:
            _chk(t, f"heads_bwd.{nm}", torch.float32)
            if tuple(t.shape) != shape:
                raise ValueError(f"heads_bwd.{nm}: shape {tuple(t.shape)}, expected {shape}")
    outs = dict(outs or {})
    dev = x4.device
    want = {"dW1": W1, "db1": W1.shape[0], "dW2": W2, "db2": K}
    if Wh is not None:
        want.update(dWh=Wh, dbh=Wh.shape[0])
    if d_hproj is not None:
        want.update(dWg=Wg, dbg=Wg.shape[0])
    for k, like in want.items():
        if outs.get(k) is None:
            outs[k] = (torch.empty_like(like) if isinstance(like, torch.Tensor)
                       else torch.empty(like, dtype=torch.float32, device=dev))
        _chk(outs[k], f"heads_bwd.{k}", torch.float32)
    dx4 = torch.empty_like(x4)
    check(_lib.load().pcaa_heads_bwd(_p(x4), _p(sup_fv), _p(h), _p(logits), _p(hproj), _p(W1), _p(Wh), _p(W2),
                                     _p(Wg), _p(d_logits), _p(d_sup), _p(d_hproj), _p(outs["dW1"]), _p(outs["db1"]),
                                     _p(outs.get("dWh")), _p(outs.get("dbh")), _p(outs["dW2"]), _p(outs["db2"]),
                                     _p(outs.get("dWg")), _p(outs.get("dbg")), _p(dx4), B, K, _s()), "pcaa_heads_bwd")
    return outs, dx4


# ------------------------------------------------------------------ fused temporal-block forward (dtc_fused.hip)
def dtc_conv_supported(T, cin, cout):
    return bool(_lib.load().pcaa_dtc_conv_supported(int(T), int(cin), int(cout)))


def dtc_conv_fwd(src, scale, shift, W2d, B, T, dilation, stats=None, want_col=False, tail=None, bf16=False):
    """One DilTempConv1d layer forward in one launch (see pcaa_dtc_conv_fwd): returns (y, col or None)."""
    _chk(src, "dtc_conv_fwd.src", torch.float32, 2)
    _chk(W2d, "dtc_conv_fwd.W", torch.float32, 2)
    rows, cin = src.shape
    cout = W2d.shape[0]
    if rows != B * T or W2d.shape[1] != cin * 3 or not dtc_conv_supported(T, cin, cout):
        raise ValueError(f"dtc_conv_fwd: unsupported shapes src {tuple(src.shape)} W {tuple(W2d.shape)} B={B} T={T}")
    if scale is not None:
        _chk(scale, "dtc_conv_fwd.scale", torch.float32)
        _chk(shift, "dtc_conv_fwd.shift", torch.float32)
        if scale.numel() != cin or shift.numel() != cin:
            raise ValueError("dtc_conv_fwd: scale/shift length")
    if stats is not None:
        _chk(stats, "dtc_conv_fwd.stats", torch.float64)
        if tuple(stats.shape) != (NREP, 2, cout):
            raise ValueError("dtc_conv_fwd: stats shape")
    y = torch.empty((rows, cout), dtype=torch.float32, device=src.device)
    col = torch.empty((rows, cin * 3), dtype=torch.float32, device=src.device) if want_col else None
    lib = _lib.load()
    fwd = lib.pcaa_dtc_conv_fwd_bf16 if bf16 else lib.pcaa_dtc_conv_fwd       # bf16: the throughput mode's MFMA variant
    ksplit = lib.pcaa_dtc_conv_ksplit(B, cin, cout)
    if ksplit > 1:
        stride = rows * cout
        slabs = torch.empty(ksplit * stride, dtype=torch.float32, device=src.device)
        check(fwd(_p(src), _p(scale), _p(shift), _p(W2d), _p(slabs), _p(col), None, NREP,
                                    B, T, cin, cout, int(dilation), ksplit, stride, _s()), "pcaa_dtc_conv_fwd")
        if stats is not None:
            check(lib.pcaa_splitk_reduce_stats(_p(slabs), ksplit, stride, _p(y), _p(stats), NREP, rows, cout, _s()),
                  "pcaa_splitk_reduce_stats")
            if tail is not None:
                tail.resolve(stats)
        else:
            check(lib.pcaa_splitk_reduce(_p(slabs), ksplit, stride, stride, _p(y), 0, _s()), "pcaa_splitk_reduce")
        return y, col
    if tail is not None and stats is not None:
        tail.arm(stats)
    check(fwd(_p(src), _p(scale), _p(shift), _p(W2d), _p(y), _p(col), _p(stats), NREP,
              B, T, cin, cout, int(dilation), 1, 0, _s()), "pcaa_dtc_conv_fwd")
    if tail is not None and stats is not None:
        tail.resolve(stats)
    return y, col


def dtc_conv_dgrad(dy, W2d, B, T, cin, dilation, dz=None, y=None, coef=None, want_dy=False, below=None, tail=None,
                   bf16=False):
    """Adjoint of the causal dilated convolution w.r.t. its input in one launch (pcaa_dtc_conv_dgrad).
    ``dy`` [B*T,cout], or None with ``dz``, ``y``, ``coef``: dy is formed on load (``want_dy``: also returned).
    ``below`` = (y, scale, shift, mean, rstd) of the layer below: the result is that layer's dz and its
    BatchNorm-backward statistics come back too.  Returns (out [B*T,cin], stats or None, dy or None)."""
    _chk(W2d, "dtc_conv_dgrad.W", torch.float32, 2)
    cout = W2d.shape[0]
    rows = B * T
    src = dy if dy is not None else dz
    _chk(src, "dtc_conv_dgrad.dy/dz", torch.float32, 2)
    if tuple(src.shape) != (rows, cout) or tuple(W2d.shape) != (cout, cin * 3) or T > 32 or cin % 4 or cout % 4:
        raise ValueError(f"dtc_conv_dgrad: unsupported shapes {tuple(src.shape)} W {tuple(W2d.shape)} B={B} T={T}")
    if dy is None:
        _chk(y, "dtc_conv_dgrad.y", torch.float32, 2)
        _chk(coef, "dtc_conv_dgrad.coef", torch.float32, 2)
        if tuple(y.shape) != (rows, cout) or tuple(coef.shape) != (3, cout):
            raise ValueError("dtc_conv_dgrad: y / coef shapes")
    lib = _lib.load()
    dev = src.device
    out = torch.empty((rows, cin), dtype=torch.float32, device=dev)
    dy_out = torch.empty((rows, cout), dtype=torch.float32, device=dev) if (want_dy and dy is None) else None
    ks = lib.pcaa_dtc_conv_dgrad_ksplit(B, cin, cout)
    dgrad = lib.pcaa_dtc_conv_dgrad_bf16 if bf16 else lib.pcaa_dtc_conv_dgrad
    stats = None
    ep = [None] * 5
    if below is not None:
        if ks > 1:
            raise ValueError("dtc_conv_dgrad: the fused epilogue needs cout <= 512")
        for t in below:
            _chk(t, "dtc_conv_dgrad.below", torch.float32)
        if tuple(below[0].shape) != (rows, cin) or any(t.numel() != cin for t in below[1:]):
            raise ValueError("dtc_conv_dgrad: below shapes")
        ep = list(below)
        stats = new_stats(cin, dev)
    if ks > 1:
        stride = rows * cin
        slabs = torch.empty(ks * stride, dtype=torch.float32, device=dev)
        check(dgrad(_p(dy), _p(dz), _p(y), _p(coef), _p(dy_out), _p(W2d), _p(slabs), None, None, None,
                                      None, None, None, NREP, B, T, cin, cout, int(dilation), ks, stride, _s()),
              "pcaa_dtc_conv_dgrad")
        check(lib.pcaa_splitk_reduce(_p(slabs), ks, stride, stride, _p(out), 0, _s()), "pcaa_splitk_reduce")
    else:
        if tail is not None and stats is not None:
            tail.arm(stats)
        check(dgrad(_p(dy), _p(dz), _p(y), _p(coef), _p(dy_out), _p(W2d), _p(out), *[_p(t) for t in ep],
                                      _p(stats), NREP, B, T, cin, cout, int(dilation), 1, 0, _s()),
              "pcaa_dtc_conv_dgrad")
        if tail is not None and stats is not None:
            tail.resolve(stats)
    return out, stats, (dy if dy is not None else dy_out)
