#!/usr/bin/env python3
"""The LDS-DMA GEMM on square problems (uniform random operands), beside torch.matmul (hipBLASLt): where the K loop
dominates.  python tools/gemm_square.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opensetgaitrecognition_pcaa_amd import ops
from opensetgaitrecognition_pcaa_amd._lib import KC, PCAA_BF16
for n in (2048, 4096, 8192):
    x = (torch.rand(n, n, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(n, n, device="cuda") * 2 - 1).bfloat16()
    y = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * n ** 3
    for name, fn in (("hipBLASLt", lambda: torch.matmul(x, w.t(), out=y)),
                     ("ours", lambda: ops.gemm(x, KC, w, KC, n, n, n, out=y, out_dtype=torch.bfloat16, math=PCAA_BF16))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f"{n}^3 {name:10s} median {ts[5]:.3f} ms  {fl / ts[5] / 1e9:7.1f} TF   (min {ts[0]:.3f})")
    ref = (x[:256].float() @ w.float().t())
    err = (y[:256].float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"{n}^3 ours vs fp32 reference (256 rows): rel err {err:.2e}")
