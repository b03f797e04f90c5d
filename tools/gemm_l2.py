#!/usr/bin/env python3
"""One PointNet GEMM shape in a loop, for PMC runs (L2 hit rate, fabric fetch bytes):
python tools/gemm_l2.py --cin 512 --cout 1024 [--mode fwd|wgrad] [--iters 10]."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from opensetgaitrecognition_pcaa_amd import ops  # noqa: E402
from opensetgaitrecognition_pcaa_amd._lib import KC, PCAA_BF16, RC  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=512)
ap.add_argument("--cout", type=int, default=1024)
ap.add_argument("--points", type=int, default=245760)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--mode", default="fwd")
ap.add_argument("--split", type=int, default=64)
a = ap.parse_args()
dev = "cuda"
P = a.points
x = (torch.randn(P, a.cin, device=dev) * 0.5).bfloat16()
W16 = (torch.randn(a.cout, a.cin, device=dev) * 0.05).bfloat16()
dy = (torch.randn(P, a.cout, device=dev) * 0.5).bfloat16()
y = torch.empty(P, a.cout, dtype=torch.bfloat16, device=dev)
dW = torch.empty(a.cout, a.cin, device=dev)


def run():
    if a.mode == "fwd":
        ops.gemm(x, KC, W16, KC, P, a.cout, a.cin, out=y, math=PCAA_BF16)
    else:
        ops.gemm_slabs(dy, RC, x, RC, a.cout, a.cin, P, a.split, out=dW, math=PCAA_BF16)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
print(f"{a.mode} cin={a.cin} cout={a.cout} stagger={os.environ.get('PCAA_GEMM_STAGGER', '0')}: {ms:.3f} ms "
      f"{2.0 * P * a.cin * a.cout / ms / 1e9:.0f} TFLOP/s", flush=True)
