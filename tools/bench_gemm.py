#!/usr/bin/env python3
"""Micro-benchmark of pcaa_gemm on the PointNet shapes of BASELINE config[1]
(P = 64*30*128 = 245760 points).  Prints TFLOP/s per shape; used to iterate on
the kernels (python tools/bench_gemm.py [--iters 20])."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from opensetgaitrecognition_pcaa_amd import ops  # noqa: E402
from opensetgaitrecognition_pcaa_amd._lib import KC, PCAA_BF16, PCAA_F32, RC  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--points", type=int, default=245760)
    a = ap.parse_args()
    P = a.points
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)

    def rnd(*shape, dtype=torch.bfloat16):
        return (torch.randn(*shape, device=dev, generator=g) * 0.5).to(dtype)

    print(f"{'case':44s} {'ms':>9s} {'TFLOP/s':>9s}")
    for (cin, cout) in ((512, 512), (512, 1024), (1024, 1024)):
        x = rnd(P, cin)
        W = rnd(cout, cin, dtype=torch.float32)
        Wt = W.t().contiguous()
        W16, Wt16 = ops.cast_bf16(W)
        dy = rnd(P, cout)
        bias = rnd(cout, dtype=torch.float32)
        stats = ops.new_stats(cout, dev)
        y = torch.empty(P, cout, dtype=torch.bfloat16, device=dev)
        da = torch.empty(P, cin, dtype=torch.bfloat16, device=dev)
        fl = 2.0 * P * cin * cout
        ms = timeit(lambda: ops.gemm(x, KC, W, KC, P, cout, cin, bias=bias, colstats=stats, out=y, math=PCAA_BF16), a.iters)
        print(f"fwd  bf16 [{P},{cin}]x[{cout},{cin}]^T +stats        {ms:9.3f} {fl / ms / 1e9:9.1f}")
        ms = timeit(lambda: ops.gemm(dy, KC, Wt, KC, P, cin, cout, out=da, math=PCAA_BF16), a.iters)
        print(f"dgrd bf16 [{P},{cout}]x[{cin},{cout}]^T               {ms:9.3f} {fl / ms / 1e9:9.1f}")
        ms = timeit(lambda: ops.gemm(x, KC, W16, KC, P, cout, cin, bias=bias, colstats=stats, out=y, math=PCAA_BF16), a.iters)
        print(f"fwd  bf16 LDS-DMA (bf16 weights) +stats              {ms:9.3f} {fl / ms / 1e9:9.1f}")
        ms = timeit(lambda: ops.gemm(dy, KC, Wt16, KC, P, cin, cout, out=da, math=PCAA_BF16), a.iters)
        print(f"dgrd bf16 LDS-DMA (bf16 weights)                     {ms:9.3f} {fl / ms / 1e9:9.1f}")
        for sk in (16, 64):
            dW = torch.zeros(cout, cin, device=dev)
            try:
                ms = timeit(lambda: ops.gemm(dy, RC, x, RC, cout, cin, P, out=dW, split_k=sk, accumulate=True, math=PCAA_BF16), a.iters)
                print(f"wgrd bf16 [{cout},{cin}] K={P} split{sk:3d}               {ms:9.3f} {fl / ms / 1e9:9.1f}")
            except Exception as e:  # noqa: BLE001
                print("wgrd bf16 failed:", str(e)[:100])
            ms = timeit(lambda: ops.gemm_slabs(dy, RC, x, RC, cout, cin, P, sk, out=dW, math=PCAA_BF16), a.iters)
            print(f"wgrd bf16 slabs+reduce split{sk:3d}                        {ms:9.3f} {fl / ms / 1e9:9.1f}")
        if (cin, cout) == (1024, 1024):
            xf, dyf = x.float(), dy.float()
            yf = torch.empty(P, cout, device=dev)
            ms = timeit(lambda: ops.gemm(xf, KC, W, KC, P, cout, cin, bias=bias, colstats=stats, out=yf), max(3, a.iters // 4))
            print(f"fwd  fp32 [{P},{cin}]x[{cout},{cin}]^T +stats        {ms:9.3f} {fl / ms / 1e9:9.1f}")
            dWf = torch.zeros(cout, cin, device=dev)
            ms = timeit(lambda: ops.gemm(dyf, RC, xf, RC, cout, cin, P, out=dWf, split_k=16, accumulate=True), max(3, a.iters // 4))
            print(f"wgrd fp32 [{cout},{cin}] K={P} split 16               {ms:9.3f} {fl / ms / 1e9:9.1f}")
    # decoder-shaped skinny GEMMs (fp32)
    for (kin, nout) in ((960, 1920), (3840, 7680), (7680, 15360)):
        x = rnd(64, kin, dtype=torch.float32)
        W = rnd(nout, kin, dtype=torch.float32)
        dyv = rnd(64, nout, dtype=torch.float32)
        dW = torch.empty(nout, kin, device=dev)
        fl = 2.0 * 64 * kin * nout
        by = W.numel() * 4
        sk = ops.pick_split_k(64, nout, kin)
        ms = timeit(lambda: ops.gemm(x, KC, W, KC, 64, nout, kin, split_k=sk, accumulate=True), a.iters)
        print(f"dec fwd  [64,{kin}]x[{nout},{kin}]^T sk={sk:<3d}          {ms:9.3f} {fl / ms / 1e9:9.1f}  {by / ms / 1e6:7.0f} GB/s")
        sk = ops.pick_split_k(64, kin, nout)
        ms = timeit(lambda: ops.gemm(dyv, KC, W, RC, 64, kin, nout, split_k=sk, accumulate=True), a.iters)
        print(f"dec dX   [64,{nout}]x[{nout},{kin}]   sk={sk:<3d}          {ms:9.3f} {fl / ms / 1e9:9.1f}  {by / ms / 1e6:7.0f} GB/s")
        ms = timeit(lambda: ops.gemm(dyv, RC, x, RC, nout, kin, 64, out=dW), a.iters)
        print(f"dec dW   [{nout},{kin}] K=64                       {ms:9.3f} {fl / ms / 1e9:9.1f}  {by / ms / 1e6:7.0f} GB/s")
        for sk in (4, 8, 16):
            ms = timeit(lambda: ops.gemm(x, KC, W, KC, 64, nout, kin, split_k=sk, accumulate=True, math=PCAA_BF16), a.iters)
            print(f"dec fwd  bf16-math sk={sk:<3d}                          {ms:9.3f} {fl / ms / 1e9:9.1f}  {by / ms / 1e6:7.0f} GB/s")
            ms = timeit(lambda: ops.gemm(dyv, KC, W, RC, 64, kin, nout, split_k=sk, accumulate=True, math=PCAA_BF16), a.iters)
            print(f"dec dX   bf16-math sk={sk:<3d}                          {ms:9.3f} {fl / ms / 1e9:9.1f}  {by / ms / 1e6:7.0f} GB/s")
        ms = timeit(lambda: ops.gemm(dyv, RC, x, RC, nout, kin, 64, out=dW, math=PCAA_BF16), a.iters)
        print(f"dec dW   bf16-math                                 {ms:9.3f} {fl / ms / 1e9:9.1f}  {by / ms / 1e6:7.0f} GB/s")


if __name__ == "__main__":
    main()
