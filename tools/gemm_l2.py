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

if os.environ.get("PCAA_GEMM_DIAG") == "20" and a.mode == "fwd":
    # phase stamps of the LAST launch (cycles of the shader clock; slot 7 = 100 MHz wall clock)
    import ctypes
    import numpy as np
    from opensetgaitrecognition_pcaa_amd import _lib
    nwg = min(8192, (P // 256) * (a.cout // 256))
    buf = np.zeros(nwg * 12, dtype=np.uint64)
    rc = _lib.load().pcaa_debug_gemm_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    assert rc == 0
    s = buf.reshape(nwg, 12).astype(np.int64)
    ph = {"launch->first DMA issued": s[:, 1] - s[:, 0], "first stage landed": s[:, 4] - s[:, 1],
          "K loop": s[:, 2] - s[:, 4], "epilogue (LDS transpose + C stores issued)": s[:, 3] - s[:, 2],
          "  per step: DMA issue": s[:, 8] // (a.cin // 64), "  per step: ds_read + MFMA issue": s[:, 9] // (a.cin // 64),
          "  per step: wait vmcnt(0)": s[:, 10] // (a.cin // 64), "  per step: barrier": s[:, 11] // (a.cin // 64),
          "statistics": s[:, 5] - s[:, 3], "stores retired": s[:, 6] - s[:, 5], "whole workgroup": s[:, 6] - s[:, 0]}
    for k, v in ph.items():
        print(f"  {k:46s} median {np.median(v):9.0f} cyc   p10 {np.percentile(v, 10):9.0f}   p90 {np.percentile(v, 90):9.0f}")
    wall = (s[:, 7].max() - s[:, 7].min()) / 100.0      # us between first and last workgroup start
    print(f"  workgroup starts span {wall:.1f} us; K steps per tile {a.cin // 64}")
