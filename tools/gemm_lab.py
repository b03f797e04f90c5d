#!/usr/bin/env python3
"""The three LDS-DMA GEMM roles (forward + BatchNorm statistics, dgrad fused with the BatchNorm/ELU backward of the
layer below, slab split-K wgrad) on the PointNet shapes of BASELINE config[1] (P = 64*30*128 = 245 760 points):
each is checked against an fp32 product of the same bf16 operands, then timed in interleaved rounds in ONE process
(cdna_hip_programming.md rule 24) next to rocBLAS/hipBLASLt through torch.matmul on the same operands.

History (round 2): this harness A/B-ed the MFMA shape (16x16x32 adopted for the KC x KC instantiations), a software L2
prefetch of the streamed operand (rejected: 3-10 % slower; profiles/r02_gemm_lab.txt) and buffer-resource addressing of
the LDS-DMA pieces (adopted: +1..10 %; profiles/r02_gemm_lab2.txt, rows "mf= 0" flat / "mf= 1" buffer).

    python tools/gemm_lab.py [--rounds 5] [--iters 10]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from opensetgaitrecognition_pcaa_amd import _lib, ops  # noqa: E402
from opensetgaitrecognition_pcaa_amd._lib import KC, PCAA_BF16, RC  # noqa: E402


def lab_set(v, *_):
    """variant 0 = the shipped default (the 4-wave tile loops, csrc/gemm_v2.h); variant 1 = pcaa_gemm_v2_enable(0): the
    loops decline every launch (round 5 removed the 8-wave loop this used to route to: plain products then run on the
    register-staged kernel, the fused roles are unsupported and skipped)"""
    _lib.load().pcaa_gemm_v2_enable(0 if int(v) == 1 else 1)


def timeit(fn, iters):
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
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--points", type=int, default=245760)
    ap.add_argument("--variants", default="0:0")
    a = ap.parse_args()
    P, dev = a.points, "cuda"
    g = torch.Generator(device=dev).manual_seed(0)

    def rnd(*shape, dtype=torch.bfloat16, s=0.5):
        return (torch.randn(*shape, device=dev, generator=g) * s).to(dtype)

    variants = [tuple(int(x) for x in v.split(":")) for v in a.variants.split(",")]
    for (cin, cout) in ((512, 512), (512, 1024), (1024, 1024)):
        x = rnd(P, cin)
        W16 = rnd(cout, cin, s=0.05)
        Wt16 = W16.t().contiguous()
        dy = rnd(P, cout)
        ybelow = rnd(P, cin)                       # pre-activation of the layer below (dgrad_bn epilogue operand)
        scale = torch.rand(cin, device=dev) + 0.5
        shift = torch.randn(cin, device=dev) * 0.1
        mean = torch.randn(cin, device=dev) * 0.1
        rstd = torch.rand(cin, device=dev) + 0.5
        y = torch.empty(P, cout, dtype=torch.bfloat16, device=dev)
        fl = 2.0 * P * cin * cout
        sk = ops.pick_split_k(cout, cin, P, target_blocks=256, bk=64, tile=256)
        dW = torch.empty(cout, cin, device=dev)
        # references on a row sample
        rows = torch.arange(0, P, 997, device=dev)
        ref_fwd = x[rows].float() @ W16.float().t()
        ref_dg = dy[rows].float() @ Wt16.float().t()
        z = ybelow[rows].float() * scale + shift
        ref_dz = ref_dg * torch.where(z > 0, torch.ones_like(z), torch.exp(z))
        ref_dW = None

        def fwd():
            stats = ops.new_stats(cout, dev)
            ops.gemm(x, KC, W16, KC, P, cout, cin, colstats=stats, out=y, out_dtype=torch.bfloat16, math=PCAA_BF16)
            return stats

        def fwd_plain():
            return ops.gemm(x, KC, W16, KC, P, cout, cin, out=y, out_dtype=torch.bfloat16, math=PCAA_BF16)

        def dgrad_bn():
            return ops.gemm_dgrad_bn(dy, Wt16, ybelow, scale, shift, mean, rstd)

        def wgrad():
            return ops.gemm_slabs(dy, RC, x, RC, cout, cin, P, sk, out=dW, math=PCAA_BF16)

        cases = {"fwd+stats": fwd, "fwd": fwd_plain, "dgrad_bn": dgrad_bn, "wgrad": wgrad}
        times = {(c, v): [] for c in cases for v in variants}
        for v in variants:                          # correctness first
            lab_set(*v)
            st = fwd()
            torch.cuda.synchronize()
            err = (y[rows].float() - ref_fwd).abs().max().item() / ref_fwd.abs().max().item()
            ssum = st.sum(0)[0].float()
            serr = ((ssum - y.float().sum(0)).abs().max() / y.float().sum(0).abs().max()).item()
            if ops.gemm_dgrad_bn_supported(P, cin, cout):
                dz, st2 = dgrad_bn()
                torch.cuda.synchronize()
                err2 = (dz[rows].float() - ref_dz).abs().max().item() / ref_dz.abs().max().item()
            else:
                err2 = 0.0          # (variant 1: the loops are off, the fused dgrad has no kernel)
            wgrad()
            torch.cuda.synchronize()
            if ref_dW is None:
                ref_dW = dW.clone()
                err3 = ((dy[:4096].float().t() @ x[:4096].float()).abs().max().item())  # scale only
                err3 = 0.0
            else:
                err3 = ((dW - ref_dW).abs().max() / ref_dW.abs().max()).item()
            ok = err < 1e-2 and err2 < 2e-2 and serr < 2e-2 and err3 < 1e-3
            print(f"check [{cin}->{cout}] mf={v[0]} pf={v[1]}: fwd {err:.2e} stats {serr:.2e} dgrad_bn {err2:.2e} "
                  f"wgrad-vs-first {err3:.2e} {'OK' if ok else 'MISMATCH'}", flush=True)
            assert ok
        # vendor reference
        tm = []
        for _ in range(a.rounds):
            tm.append(timeit(lambda: torch.matmul(x, W16.t(), out=y), a.iters))
        print(f"[{cin}->{cout}] torch.matmul (hipBLASLt)      median {sorted(tm)[len(tm) // 2]:.3f} ms  "
              f"{fl / sorted(tm)[len(tm) // 2] / 1e9:7.1f} TF   min {min(tm):.3f}", flush=True)
        for _ in range(a.rounds):
            for v in variants:
                lab_set(*v)
                for c, fn in cases.items():
                    if c == "dgrad_bn" and not ops.gemm_dgrad_bn_supported(P, cin, cout):
                        continue
                    fn()
                    times[(c, v)].append(timeit(fn, a.iters))
        for c in cases:
            for v in variants:
                t = sorted(times[(c, v)])
                if not t:
                    continue
                med = t[len(t) // 2]
                print(f"[{cin}->{cout}] {c:10s} mf={v[0]:2d} pf={v[1]}  median {med:.3f} ms  {fl / med / 1e9:7.1f} TF   "
                      f"min {t[0]:.3f}", flush=True)
    lab_set(0, 0)


if __name__ == "__main__":
    main()
