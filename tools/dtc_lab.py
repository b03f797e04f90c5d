#!/usr/bin/env python3
"""The temporal block's layers alone on the GPU (B=64, T=30): forward and adjoint per layer, with and without the
im2col / statistics side outputs.   python tools/dtc_lab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opensetgaitrecognition_pcaa_amd import ops

B, T, dev = 64, 30, "cuda"
chans = [1024, 16, 32, 64, 128, 256, 512]
dils = [1, 2, 4, 1, 2, 4]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


TRACE = "--trace" in sys.argv      # needs a library built with PCAA_HIPCC_EXTRA=-DPCAA_DTC_TRACE (dtc_fused.hip)
if TRACE:
    import ctypes
    from opensetgaitrecognition_pcaa_amd import _lib
    _tr = _lib.load().pcaa_lab_dtc_trace
    _tr.argtypes = [ctypes.c_void_p, ctypes.c_int]

    def phases(fn, nwg, names, base, reps=10):
        """average time of wave 0 in each phase (s_memtime ticks, 100 MHz)"""
        fn(); torch.cuda.synchronize()
        _tr(None, 1)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        _tr(buf, 1)
        return "  ".join(f"{n} {buf[base + i] / (nwg * reps) / 100.0:6.2f}" for i, n in enumerate(names)) + "   (s_memtime ticks / 100)"

for li in range(6):
    cin, cout, d = chans[li], chans[li + 1], dils[li]
    src = torch.randn(B * T, cin, device=dev)
    sc = torch.rand(cin, device=dev) + 0.5
    sh = torch.randn(cin, device=dev) * 0.1
    W = torch.randn(cout, cin * 3, device=dev) * 0.05
    stats = torch.zeros(ops.NREP, 2, cout, dtype=torch.float64, device=dev)
    for bf16 in (False, True):
        t_full = timed(lambda: ops.dtc_conv_fwd(src, sc, sh, W, B, T, d, stats=stats, want_col=True, bf16=bf16))
        t_nocol = timed(lambda: ops.dtc_conv_fwd(src, sc, sh, W, B, T, d, stats=stats, want_col=False, bf16=bf16))
        t_bare = timed(lambda: ops.dtc_conv_fwd(src, sc, sh, W, B, T, d, stats=None, want_col=False, bf16=bf16))
        t_noact = timed(lambda: ops.dtc_conv_fwd(src, None, None, W, B, T, d, stats=None, want_col=False, bf16=bf16))
        dy = torch.randn(B * T, cout, device=dev)
        t_dg = timed(lambda: ops.dtc_conv_dgrad(dy, W, B, T, cin, d, bf16=bf16))
        if TRACE and not bf16:
            ks = 8 if cin >= 1024 else 1
            print(f"   fwd   us/workgroup: " + phases(lambda: ops.dtc_conv_fwd(src, sc, sh, W, B, T, d, stats=stats, want_col=True),
                                                     ((B // 2) * (cout // (64 if cout >= 512 else 32)) if (cin >= 128 and ks == 1 and os.environ.get("PCAA_DTC_PAIR", "1") != "0")
                                                      else B * ((cout + 31) // 32) * ks), ["stage", "im2col", "contract", "combine+store", "stats", "tail"], 0))
            pair = cout >= 128 and os.environ.get("PCAA_DTC_PAIR", "1") != "0"     # the two-sequence kernel marks 0..5 in both directions
            print(f"   dgrad us/workgroup: " + phases(lambda: ops.dtc_conv_dgrad(dy, W, B, T, cin, d),
                                                     (B // 2) * ((cin + 31) // 32) if pair else B * ((cin + 31) // 32),
                                                     ["stage", "im2col", "contract", "combine+store", "stats", "tail"] if pair else
                                                     ["stage", "contract", "combine+store", "stats", "tail"], 0 if pair else 8))
        fl = 2.0 * B * T * cin * 3 * cout
        print(f"layer {li + 1} {cin:4d}->{cout:3d} d={d} {'bf16' if bf16 else 'fp32'}: fwd {t_full:6.1f} us (no col {t_nocol:6.1f}, no col/stats {t_bare:6.1f}, "
              f"no activation on load {t_noact:6.1f})  dgrad {t_dg:6.1f} us   {fl / 1e9:.2f} GFLOP")
