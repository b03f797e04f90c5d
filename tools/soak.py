#!/usr/bin/env python3
"""Soak: a few thousand bf16 train steps at the bench shape (config[1]) on fresh random batches -- losses stay
finite, parameters stay finite, device memory does not grow, the step time does not drift, a state_dict round trip in
the middle changes nothing.   python tools/soak.py [--steps 3000]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3000)
a0 = ap.parse_args()
_argv, sys.argv = sys.argv, sys.argv[:1]
a = bench.parse()
sys.argv = _argv
dev = torch.device("cuda:0")
N, C, K = a.points, a.features, a.classes
tr, cfg = bench.build_trainer(a, N, dev, None, "bf16", fill="device")
from opensetgaitrecognition_pcaa_amd import constants
T, B = constants.NSTEPS, a.batch
g = torch.Generator(device=dev).manual_seed(7)


def batch():
    pcs = torch.randn(B, T, N, C, device=dev, generator=g).permute(0, 3, 1, 2)
    gt = torch.randint(0, K, (B,), device=dev, generator=g)
    z0 = torch.randn(B, 32, device=dev, generator=g)
    al = torch.rand(B, 1, device=dev, generator=g)
    return pcs, gt, z0, al


mem = {}
t_marks = {}
bad = 0
for i in range(a0.steps):
    if i in (100, a0.steps // 2, a0.steps - 200):
        torch.cuda.synchronize(); t_marks[i] = time.perf_counter()
    out = tr.step(*batch())
    if i % 250 == 0 or i == a0.steps - 1:
        vals = {k: out[k].item() for k in ("d_loss", "rec_loss", "sup_loss", "tot_loss")}
        ok = all(v == v and abs(v) < 1e6 for v in vals.values())
        bad += 0 if ok else 1
        print(f"step {i:5d}  " + "  ".join(f"{k} {v:10.4f}" for k, v in vals.items()) + f"  mem {torch.cuda.memory_allocated() / 2**20:8.1f} MiB", flush=True)
    if i in (100, a0.steps - 1):
        torch.cuda.synchronize(); mem[i] = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
    if i in (300, a0.steps // 2 + 200, a0.steps - 1):
        torch.cuda.synchronize()
        base = max(k for k in t_marks if k <= i)
        print(f"    steps {base}..{i}: {(time.perf_counter() - t_marks[base]) / (i - base + 1) * 1e3:.3f} ms/step incl. batch generation", flush=True)
    if i == a0.steps // 2:
        # state_dict round trip of every module: same bits back in place
        for m in (tr.encoder, tr.decoder, tr.discriminator):
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            m.load_state_dict(sd)
fin = all(torch.isfinite(p).all().item() for m in (tr.encoder, tr.decoder, tr.discriminator) for p in m.parameters())
print(f"parameters finite: {fin}; non-finite loss reports: {bad}")
print(f"memory allocated at step 100: {mem[100][0] / 2**20:.1f} MiB, at the end: {mem[a0.steps - 1][0] / 2**20:.1f} MiB; "
      f"reserved {mem[100][1] / 2**20:.1f} -> {mem[a0.steps - 1][1] / 2**20:.1f} MiB")
assert fin and bad == 0 and mem[a0.steps - 1][0] <= mem[100][0] * 1.02 + (1 << 20)
print("soak ok")
