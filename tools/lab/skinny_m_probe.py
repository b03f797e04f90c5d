#!/usr/bin/env python3
"""What the 64-row operand costs the decoder's forward / dgrad: the 7680 -> 15360 layer at M = 64 / 32 / 8 / 1 rows (rows
past M are not loaded: the x / dz chunk traffic from L2, the slab stores and the reduction shrink with M; the weight stream
and the MFMA work per chunk do not)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from opensetgaitrecognition_pcaa_amd import ops
from opensetgaitrecognition_pcaa_amd._lib import ACT_ELU
K, N = 7680, 15360
dev = "cuda"
W = torch.randn(N, K, device=dev) * 0.02
b = torch.randn(N, device=dev)


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


for M in (64, 32, 8, 1):
    x = torch.randn(M, K, device=dev)
    dz = torch.randn(M, N, device=dev)
    t_f = timed(lambda: ops.skinny_linear_fwd(x, W, b, ACT_ELU))
    t_g = timed(lambda: ops.skinny_linear_dgrad(dz, W, a_prev=x))
    wb = 4.0 * N * K
    print(f"M = {M:2d}: fwd {t_f:6.1f} us {wb / t_f / 1e6:5.2f} TB/s | dgrad {t_g:6.1f} us {wb / t_g / 1e6:5.2f} TB/s", flush=True)
