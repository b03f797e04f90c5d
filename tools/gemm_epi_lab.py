#!/usr/bin/env python3
"""What the forward GEMM's epilogue pieces cost: with / without the BatchNorm statistics, bf16 / fp32 output, on the
three PointNet shapes (interleaved rounds in one process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opensetgaitrecognition_pcaa_amd import ops
from opensetgaitrecognition_pcaa_amd._lib import KC, PCAA_BF16

P, dev = 245760, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=0.5: (torch.randn(*s, device=dev, generator=g) * sc).bfloat16()


def timeit(fn, iters=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for cin, cout in ((512, 512), (512, 1024), (1024, 1024)):
    x, W = rnd(P, cin), rnd(cout, cin, sc=0.05)
    y16 = torch.empty(P, cout, dtype=torch.bfloat16, device=dev)
    stats = ops.new_stats(cout, dev)
    cases = {
        "bf16 out + stats": lambda: ops.gemm(x, KC, W, KC, P, cout, cin, colstats=stats, out=y16, out_dtype=torch.bfloat16, math=PCAA_BF16),
        "bf16 out": lambda: ops.gemm(x, KC, W, KC, P, cout, cin, out=y16, out_dtype=torch.bfloat16, math=PCAA_BF16),
    }
    t = {k: [] for k in cases}
    for _ in range(5):
        for k, fn in cases.items():
            fn()
            t[k].append(timeit(fn))
    fl = 2.0 * P * cin * cout
    for k, v in t.items():
        m = sorted(v)[len(v) // 2]
        print(f"[{cin}->{cout}] {k:18s} median {m:.3f} ms {fl / m / 1e9:7.1f} TF", flush=True)
