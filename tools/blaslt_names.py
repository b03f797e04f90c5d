#!/usr/bin/env python3
"""Which hipBLASLt / rocBLAS kernels torch.matmul picks for the PointNet shapes (run under
rocprofv3 --kernel-trace --stats: the kernel names encode macro tile, MFMA shape, prefetch depths)."""
import torch
P = 245760
for cin, cout in ((512, 512), (512, 1024), (1024, 1024)):
    x = torch.randn(P, cin, device="cuda").bfloat16()
    w = torch.randn(cout, cin, device="cuda").bfloat16()
    for _ in range(5):
        y = x @ w.t()
    torch.cuda.synchronize()
