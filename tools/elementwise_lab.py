#!/usr/bin/env python3
"""The HBM-bound passes of the PointNet block alone on the GPU: ms and TB/s of the bytes each moves, at the bench
shape.  python tools/elementwise_lab.py        (profiles/r03_elementwise_unroll_lab.txt is this lab on a scratch build
whose column-invariant kernels took 1 / 2 / 4 / 8 quads per thread and trip; the switch was not kept)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opensetgaitrecognition_pcaa_amd import ops
P, dev = 245760, "cuda"


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for ch in (512, 1024):
    y = (torch.randn(P, ch, device=dev) * 0.7).bfloat16()
    sc = torch.rand(ch, device=dev) + 0.5
    sh = torch.randn(ch, device=dev) * 0.1
    mu = torch.randn(ch, device=dev) * 0.1
    rs = torch.rand(ch, device=dev) + 0.5
    nb = P * ch * 2
    t = timed(lambda: ops.bn_act_fwd(y, sc, sh))
    print(f"[{ch}] bn_act_fwd            {t:.3f} ms  {2 * nb / t / 1e9:.2f} TB/s (read + write)")
    t = timed(lambda: ops.bn_act_meanpool_fwd(y, sc, sh, P // 128, 128, mu, rs))
    print(f"[{ch}] bn_act_meanpool (train) {t:.3f} ms  {nb / t / 1e9:.2f} TB/s (read)")
    coef = torch.randn(3, ch, device=dev) * 0.3
    dzt = (torch.randn(P, ch, device=dev) * 0.1).bfloat16()
    outt = torch.empty_like(y)
    t = timed(lambda: ops.bn_bwd_dy(dzt, y, coef, out=outt))
    print(f"[{ch}] bn_bwd_dy             {t:.3f} ms  {3 * nb / t / 1e9:.2f} TB/s (2 reads + write)")
    t = timed(lambda: ops.bn_bwd_dy_fused(y, sc, sh, coef, da=dzt, out=outt))
    print(f"[{ch}] bn_bwd_dy_fused (da)  {t:.3f} ms  {3 * nb / t / 1e9:.2f} TB/s (2 reads + write)")
    dpool = torch.randn(P // 128, ch, device=dev) * 0.1
    t = timed(lambda: ops.bn_bwd_dy_fused(y, sc, sh, coef, dpool=dpool, group_rows=128, pool_scale=1 / 128, out=outt))
    print(f"[{ch}] bn_bwd_dy_fused (pool) {t:.3f} ms  {2 * nb / t / 1e9:.2f} TB/s (read + write)")
    del dzt, outt
    z = torch.empty_like(y)
    t = timed(lambda: z.copy_(y))
    print(f"[{ch}] torch copy            {t:.3f} ms  {2 * nb / t / 1e9:.2f} TB/s (read + write)")
    t = timed(lambda: y.sum(dim=0))
    print(f"[{ch}] torch column sum      {t:.3f} ms  {nb / t / 1e9:.2f} TB/s (read)")
x = torch.randn(P, 4, device=dev)
W = torch.randn(512, 4, device=dev) * 0.3
da = (torch.randn(P, 512, device=dev) * 0.1).bfloat16()
sc = torch.rand(512, device=dev) + 0.5; sh = torch.randn(512, device=dev) * 0.1
mu = torch.randn(512, device=dev) * 0.1; rs = torch.rand(512, device=dev) + 0.5
t = timed(lambda: ops.pointnet_in_bwd_stats(da, x, W, sc, sh, mu, rs))
print(f"pointnet_in_bwd_stats      {t:.3f} ms  {P * 512 * 2 / t / 1e9:.2f} TB/s (read)")
t = timed(lambda: ops.pointnet_in_apply(x, W, sc, sh, torch.bfloat16))
print(f"pointnet_in_apply          {t:.3f} ms  {P * 512 * 2 / t / 1e9:.2f} TB/s (write)")


class _Bn:
    weight = torch.rand(512, device=dev) + 0.5


def onepass():
    tail = ops.BnTailBwd(P, _Bn, mu, rs, 512)
    return ops.pointnet_in_bwd_onepass(da, x, W, sc, sh, mu, rs, tail, mom=mom)


mom = ops.points_moments(x)
t = timed(onepass)
print(f"pointnet_in_bwd_onepass    {t:.3f} ms  {P * 512 * 2 / t / 1e9:.2f} TB/s (read; + the small launches around it)")
