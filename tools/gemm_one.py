#!/usr/bin/env python3
"""A few launches of the three LDS-DMA GEMM roles on [245760,1024]x[1024,1024] for counter collection
(tools/pmc_gemm.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from opensetgaitrecognition_pcaa_amd import ops  # noqa: E402
from opensetgaitrecognition_pcaa_amd._lib import KC, PCAA_BF16, RC  # noqa: E402

P, cin, cout, dev = 245760, 1024, 1024, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=0.5: (torch.randn(*s, device=dev, generator=g) * sc).bfloat16()
x, W16, dy, yb = rnd(P, cin), rnd(cout, cin, sc=0.05), rnd(P, cout), rnd(P, cin)
Wt16 = W16.t().contiguous()
scale, shift, mean, rstd = (torch.rand(cin, device=dev) + 0.5 for _ in range(4))
y = torch.empty(P, cout, dtype=torch.bfloat16, device=dev)
dW = torch.empty(cout, cin, device=dev)
sk = ops.pick_split_k(cout, cin, P, target_blocks=256, bk=64, tile=256)
for _ in range(4):
    ops.gemm(x, KC, W16, KC, P, cout, cin, colstats=ops.new_stats(cout, dev), out=y, out_dtype=torch.bfloat16, math=PCAA_BF16)
    ops.gemm_dgrad_bn(dy, Wt16, yb, scale, shift, mean, rstd)
    ops.gemm_slabs(dy, RC, x, RC, cout, cin, P, sk, out=dW, math=PCAA_BF16)
torch.cuda.synchronize()
