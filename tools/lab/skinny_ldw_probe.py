#!/usr/bin/env python3
"""Does the decoder forward's 4.5 TB/s come from WHERE the weight rows fall in HBM?  The 7680 -> 15360 layer with its rows
at other leading dimensions (the C ABI takes ldw; the product keeps rows contiguous): same kernel, same bytes, padded rows.
Also the dgrad (contraction along the rows: 1 KB runs per row)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from opensetgaitrecognition_pcaa_amd import _lib
lib = _lib.load()
M, K, N = 64, 7680, 15360
dev = "cuda"
x = torch.randn(M, K, device=dev)
dz = torch.randn(M, N, device=dev)
b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
dx = torch.empty(M, K, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for pad in (0, 16, 32, 64, 128, 256, 512, 1024):
    ldw = K + pad
    Wb = torch.randn(N, ldw, device=dev) * 0.02
    ns = lib.pcaa_skinny_splits(0, M, N, K)
    ws = torch.empty(ns * M * N, device=dev)
    f = lambda: _lib.check(lib.pcaa_skinny_linear_fwd(x.data_ptr(), K, Wb.data_ptr(), ldw, b.data_ptr(), 1, y.data_ptr(), ws.data_ptr(),
                                                      ws.numel(), M, N, K, ns, st), "fwd")
    t_f = timed(f)
    nd = lib.pcaa_skinny_splits(1, M, N, K)
    wd = torch.empty(nd * M * K, device=dev)
    g = lambda: _lib.check(lib.pcaa_skinny_linear_dgrad(dz.data_ptr(), N, Wb.data_ptr(), ldw, dx.data_ptr(), None, 0, wd.data_ptr(),
                                                        wd.numel(), M, N, K, nd, st), "dgrad")
    t_g = timed(g)
    wb = 4.0 * N * K
    print(f"ldw = K + {pad:4d}: fwd {t_f:6.1f} us {wb / t_f / 1e6:5.2f} TB/s | dgrad {t_g:6.1f} us {wb / t_g / 1e6:5.2f} TB/s", flush=True)
    del Wb, ws, wd
