#!/usr/bin/env python3
"""A weight-gradient product (matrix pipes) beside a BatchNorm-backward elementwise pass (HBM) at the bench shape:
serial on one stream against the product cut in two -- a part on a CU-masked stream beside the elementwise pass, the
rest behind it on the main stream.   python tools/overlap_lab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opensetgaitrecognition_pcaa_amd import ops
from opensetgaitrecognition_pcaa_amd.ops import RC, PCAA_BF16

P, dev = 245760, "cuda"

# CU-masked streams are a lab hook (tools/microbench/streams_lab.hip), not part of the product ABI
import ctypes, subprocess
_HERE = os.path.dirname(os.path.abspath(__file__))
_LAB = os.path.join(_HERE, "microbench", "libstreams_lab.so")
if not os.path.exists(_LAB):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-o", _LAB,
                           os.path.join(_HERE, "microbench", "streams_lab.hip")])
_lab = ctypes.CDLL(_LAB)
_MASKED = {}


def masked_stream(n_cus, first_cu=0):
    key = (int(first_cu), int(n_cus))
    if key not in _MASKED:
        h = ctypes.c_void_p()
        if _lab.lab_stream_create_masked(int(first_cu), int(n_cus), ctypes.byref(h)) != 0:
            raise RuntimeError("lab_stream_create_masked failed")
        _MASKED[key] = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", torch.cuda.current_device()))
    return _MASKED[key]


def timed(fn, reps=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def case(cout, cin, ew_ch):
    """wgrad dW[cout, cin] = dy[P, cout]^T a[P, cin] beside bn_bwd_dy on [P, ew_ch]"""
    dy = (torch.randn(P, cout, device=dev) * 0.1).bfloat16()
    a = (torch.randn(P, cin, device=dev) * 0.5).bfloat16()
    dz = (torch.randn(P, ew_ch, device=dev) * 0.1).bfloat16()
    y = (torch.randn(P, ew_ch, device=dev) * 0.7).bfloat16()
    coef = torch.randn(3, ew_ch, device=dev) * 0.3
    out = torch.empty(cout, cin, device=dev)
    ntiles = (cout // 256) * (cin // 256)
    sk_full = 256 // ntiles
    slabs = torch.empty(4 * 256 * 256 * 256, device=dev)
    main = torch.cuda.current_stream()

    def wgrad_full():
        ns = ops.gemm_slabs_part(dy, a, cout, cin, P, sk_full, slabs)
        ops.splitk_reduce(slabs, ns, cout, cin, out)

    def ew():
        ops.bn_bwd_dy(dz, y, coef, out=dz)

    ref = None
    t_w, t_e = timed(wgrad_full), timed(ew)
    wgrad_full(); ref = out.clone()
    t_serial = timed(lambda: (ew(), wgrad_full()))
    print(f"dW[{cout},{cin}] + bn_bwd_dy[{ew_ch}]: wgrad {t_w:.3f}  elementwise {t_e:.3f}  serial {t_serial:.3f} ms")
    # the cost of the cut and of the stream hand-offs alone
    side = torch.cuda.Stream()
    for name, W, E in (("same stream", None, None), ("plain side stream", side, None)):
        rows_a = P // 4
        def cut(W=W):
            ev = torch.cuda.Event(); ev.record(main)
            if W is not None:
                with torch.cuda.stream(W):
                    W.wait_event(ev)
                    na = ops.gemm_slabs_part(dy[:rows_a], a[:rows_a], cout, cin, rows_a, sk_full, slabs)
            else:
                na = ops.gemm_slabs_part(dy[:rows_a], a[:rows_a], cout, cin, rows_a, sk_full, slabs)
            ew()
            nb = ops.gemm_slabs_part(dy[rows_a:], a[rows_a:], cout, cin, P - rows_a, sk_full, slabs[na * cout * cin:])
            if W is not None:
                main.wait_stream(W)
            ops.splitk_reduce(slabs, na + nb, cout, cin, out)
        print(f"    product cut 1/4 + 3/4, first part on the {name}: {timed(cut):.3f} ms")
    for ncu in (96, 128, 160, 192):
        W = masked_stream(ncu)
        E = masked_stream(256 - ncu, first_cu=ncu)
        sk_a = max(8, ncu // ntiles // 8 * 8)
        if sk_a * ntiles > ncu:
            continue
        # the elementwise pass alone on its share of the chip, the product part alone on its share
        def ew_on_E():
            ev = torch.cuda.Event(); ev.record(main)
            with torch.cuda.stream(E):
                E.wait_event(ev)
                ew()
            main.wait_stream(E)
        t_eE = timed(ew_on_E)
        for frac in (0.15, 0.25, 0.35, 0.5):
            import math
            unit = math.lcm(sk_a * 64, sk_full * 64)
            rows_a = int(P * frac) // unit * unit
            if rows_a == 0 or rows_a >= P:
                continue
            rows_b = P - rows_a

            def part_on_W():
                ev = torch.cuda.Event(); ev.record(main)
                with torch.cuda.stream(W):
                    W.wait_event(ev)
                    ops.gemm_slabs_part(dy[:rows_a], a[:rows_a], cout, cin, rows_a, sk_a, slabs)
                main.wait_stream(W)

            def overlapped():
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(W):
                    W.wait_event(ev)
                    na = ops.gemm_slabs_part(dy[:rows_a], a[:rows_a], cout, cin, rows_a, sk_a, slabs)
                with torch.cuda.stream(E):
                    E.wait_event(ev)
                    ew()
                main.wait_stream(E)
                main.wait_stream(W)
                nb = ops.gemm_slabs_part(dy[rows_a:], a[rows_a:], cout, cin, rows_b, sk_full, slabs[na * cout * cin:])
                ops.splitk_reduce(slabs, na + nb, cout, cin, out)

            t_a = timed(part_on_W)
            t = timed(overlapped)
            overlapped(); torch.cuda.synchronize()
            err = ((out - ref).norm() / ref.norm()).item()
            print(f"    {ncu:3d} CUs x {sk_a:2d} splits, {frac:.2f} of K beside the pass ({256 - ncu} CUs): alone {t_a:.3f} / {t_eE:.3f}, "
                  f"together + rest {t:.3f} ms  ({t_serial - t:+.3f})  rel diff {err:.1e}")


case(1024, 1024, 1024)     # wgrad of layer 4 beside dy of layer 3
case(1024, 512, 512)       # wgrad of layer 3 beside dy of layer 2
case(512, 512, 512)        # wgrad of layer 2 (beside the first layer's backward: a 252 MB read)
