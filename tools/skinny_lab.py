#!/usr/bin/env python3
"""Decoder weight update in isolation: weight gradient -> Adam (two kernels) against the fused kernel, per decoder
layer of the bench shape, alone on the GPU: ms and GB/s of the bytes each really moves.

    python tools/skinny_lab.py [--points 128] [--batch 64]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opensetgaitrecognition_pcaa_amd import _lib, ops
from opensetgaitrecognition_pcaa_amd.train import StepCount

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=128)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
lib = _lib.load()
S = 30 * 4 * a.points
widths = [S // 16, S // 8, S // 4, S // 2, S]
M = a.batch
dev = "cuda"


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


tot = {}
for K, N in zip(widths[:-1], widths[1:]):
    g = torch.Generator(device="cpu").manual_seed(N)
    dz = (torch.randn(M, N, generator=g) * 0.1).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.02).to(dev)
    m, v, dW = torch.zeros_like(W), torch.zeros_like(W), torch.empty_like(W)
    c = StepCount(dev)
    c.advance(1e-4, 0.9, 0.99)
    n = N * K
    t_w = timed(lambda: ops.skinny_linear_wgrad(dz, x, out=dW), a.reps)
    t_a = timed(lambda: ops.adam_step_dev_(W.view(-1), dW.view(-1), m.view(-1), v.view(-1), 0.9, 0.99, 1e-8, c.coef_dev, 1.0, 256), a.reps)
    print(f"[{K}->{N}] wgrad {t_w:.3f} ms ({4 * n / t_w / 1e9:.2f} TB/s)   adam(256 blocks) {t_a:.3f} ms ({28 * n / t_a / 1e9:.2f} TB/s)"
          f"   sum {t_w + t_a:.3f}")
    tot["unfused"] = tot.get("unfused", 0) + t_w + t_a
    t_f = timed(lambda: ops.skinny_linear_wgrad_adam_(dz, x, W, m, v, 0.9, 0.99, 1e-8, c.coef_dev), a.reps)
    print(f"[{K}->{N}] fused: {t_f:.3f} ms ({24 * n / t_f / 1e9:.2f} TB/s)")
    tot["fused"] = tot.get("fused", 0) + t_f
    del dz, x, W, m, v, dW
print("totals (ms):", {k: round(t, 3) for k, t in tot.items()})

# ---- round 6: the data-parallel update from GATHERED rows, pcaa_skinny_linear_wgrad_adam_rows, at the
# row counts a world of 1 / 2 / 4 / 8 ranks x 64 sequences stacks (M = 64 .. 512), per wide decoder layer, alone on the GPU;
# beside the single-process kernel at M = 64 above.  24 B per parameter whatever M is: the GB/s column says how much of
# the HBM stream survives the longer contraction.
print("\ngathered-rows update (pcaa_skinny_linear_wgrad_adam_rows), ms per layer [TB/s of 24 B/param]")
rows_tot = {}
for K, N in zip(widths[:-1], widths[1:]):
    g = torch.Generator(device="cpu").manual_seed(N)
    W = (torch.randn(N, K, generator=g) * 0.02).to(dev)
    m, v = torch.zeros_like(W), torch.zeros_like(W)
    c = StepCount(dev)
    c.advance(1e-4, 0.9, 0.99)
    n = N * K
    cells = []
    for Mr in (64, 128, 256, 512):
        R = ops.gathered_rows_alloc(Mr)
        dz = (torch.randn(R, N, generator=g) * 0.1).to(dev)
        x = torch.randn(R, K, generator=g).to(dev)
        t = timed(lambda: ops.skinny_linear_wgrad_adam_rows_(dz, x, Mr, W, m, v, 0.9, 0.99, 1e-8, c.coef_dev, 1.0 / (Mr // 64)), a.reps)
        rows_tot[Mr] = rows_tot.get(Mr, 0) + t
        cells.append(f"M={Mr}: {t:.3f} [{24 * n / t / 1e9:.2f}]")
        del dz, x
    print(f"[{K}->{N}] " + "   ".join(cells))
    del W, m, v
print("totals over the wide layers (ms):", {f"M={k}": round(t, 3) for k, t in rows_tot.items()},
      "  single-process fused (M=64):", round(tot.get("fused", 0), 3))


# ---- round 6: the same update from PACKED operands (pcaa_pack_rows_t16 chunks, one per rank; pcaa_skinny_linear_wgrad_adam_t16)
print("\npacked-operand update (pcaa_skinny_linear_wgrad_adam_t16), ms per layer [TB/s of 24 B/param]; pack = one rank's pack launch")
t16_tot, pack_tot = {}, 0.0
for K, N in zip(widths[:-1], widths[1:]):
    g = torch.Generator(device="cpu").manual_seed(N)
    W = (torch.randn(N, K, generator=g) * 0.02).to(dev)
    m, v = torch.zeros_like(W), torch.zeros_like(W)
    c = StepCount(dev)
    c.advance(1e-4, 0.9, 0.99)
    n = N * K
    dz = (torch.randn(64, N, generator=g) * 0.1).to(dev)
    x = torch.randn(64, K, generator=g).to(dev)
    own = ops.pack_rows_t16(dz, x)
    t_p = timed(lambda: ops.pack_rows_t16(dz, x, out=own), a.reps)
    pack_tot += t_p
    cells = [f"pack {t_p * 1e3:.1f} us"]
    for chunks in (1, 2, 4, 8):
        packed = own.unsqueeze(0).repeat(chunks, 1).contiguous()
        t = timed(lambda: ops.skinny_linear_wgrad_adam_t16_(packed, chunks, W, m, v, 0.9, 0.99, 1e-8, c.coef_dev, 1.0 / chunks), a.reps)
        t16_tot[chunks] = t16_tot.get(chunks, 0) + t
        cells.append(f"{chunks} x 64: {t:.3f} [{24 * n / t / 1e9:.2f}]")
        del packed
    print(f"[{K}->{N}] " + "   ".join(cells))
    del W, m, v, dz, x
print("totals over the wide layers (ms):", {f"{k} chunks": round(t, 3) for k, t in t16_tot.items()}, " packs:", round(pack_tot, 4),
      "  rows kernel (round 5):", {f"M={k}": round(t, 3) for k, t in rows_tot.items()})
