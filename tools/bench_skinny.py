#!/usr/bin/env python3
"""Micro-benchmark of the batch-skinny decoder kernels (gemm_skinny.hip) on the CGDecoder
shapes of BASELINE config[1] (S = 30*4*128 = 15360, batch 64), against the 256x256-tile
kernel they replace.  Prints us and effective weight-stream GB/s per layer and direction
(python tools/bench_skinny.py [--iters 20] [--batch 64])."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from opensetgaitrecognition_pcaa_amd import ops  # noqa: E402
from opensetgaitrecognition_pcaa_amd._lib import ACT_ELU, KC, PCAA_BF16, RC  # noqa: E402


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
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--S", type=int, default=15360)
    ap.add_argument("--old", action="store_true", help="also time the 256x256-tile path")
    a = ap.parse_args()
    dev = "cuda"
    M, S = a.batch, a.S
    dims = [S // 16, S // 8, S // 4, S // 2, S]
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for K, N in zip(dims[:-1], dims[1:]):
        x = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.02
        b = torch.randn(N, device=dev)
        dz = torch.randn(M, N, device=dev)
        dW = torch.empty(N, K, device=dev)
        wbytes = 4.0 * N * K
        t = {
            "fwd": timeit(lambda: ops.skinny_linear_fwd(x, W, b, ACT_ELU), a.iters),
            "dgrad": timeit(lambda: ops.skinny_linear_dgrad(dz, W, a_prev=x), a.iters),
            "wgrad": timeit(lambda: ops.skinny_linear_wgrad(dz, x, out=dW), a.iters),
        }
        line = f"K={K:6d} N={N:6d} "
        for k, v in t.items():
            tot[k] += v
            line += f"| {k} {v:7.1f} us {wbytes / v / 1e3:6.0f} GB/s "
        if a.old:
            sk = ops.pick_split_k(M, N, K, target_blocks=256, bk=64, tile=256)
            o1 = timeit(lambda: ops.gemm(x, KC, W, KC, M, N, K, split_k=sk, accumulate=True, math=PCAA_BF16), a.iters)
            sk = ops.pick_split_k(M, K, N, target_blocks=256, bk=64, tile=256)
            o2 = timeit(lambda: ops.gemm(dz, KC, W, RC, M, K, N, split_k=sk, accumulate=True, math=PCAA_BF16), a.iters)
            o3 = timeit(lambda: ops.gemm(dz, RC, x, RC, N, K, M, out=dW, math=PCAA_BF16), a.iters)
            line += f"| old {o1:7.1f} {o2:7.1f} {o3:7.1f} us"
        print(line, flush=True)
    print("total us:", {k: round(v, 1) for k, v in tot.items()}, "sum", round(sum(tot.values()), 1), flush=True)


if __name__ == "__main__":
    main()
