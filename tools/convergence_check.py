#!/usr/bin/env python3
"""Does the HIP train step LEARN?  Variant 4 on a synthetic, class-dependent gait-like dataset (each class has its
own per-frame point-cloud shape and Doppler pattern), a few hundred steps in both precision modes; prints
the trajectory of the losses and the train / held-out accuracy.  Evidence beyond single-step parity: Adam with
the device-side step count, BatchNorm running statistics (eval accuracy), the critic game, bf16 vs fp32 mode.

    python tools/convergence_check.py [--steps 300] [--batch 64] [--points 64]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip
from opensetgaitrecognition_pcaa_amd.train import PCAATrainer

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--points", type=int, default=64)
ap.add_argument("--classes", type=int, default=6)
a = ap.parse_args()
B, N, C, K, T = a.batch, a.points, 4, a.classes, constants.NSTEPS


def make_data(n, seed):
    rng = np.random.default_rng(seed)
    y = rng.integers(0, K, n)
    t = np.arange(T)[None, :, None]
    phase = rng.uniform(0, 2 * np.pi, (n, 1, 1))
    freq = (0.5 + 0.25 * y)[:, None, None]                       # class-dependent gait frequency
    width = (0.3 + 0.1 * y)[:, None, None]                       # class-dependent body width
    x = rng.standard_normal((n, T, N, C)).astype(np.float32)
    x[..., 0] = x[..., 0] * width + 0.3 * np.sin(freq * t + phase)
    x[..., 1] *= 0.2
    x[..., 2] = x[..., 2] * 0.5 + 0.1 * y[:, None, None]
    x[..., 3] = 0.5 * x[..., 3] + np.cos(freq * t + phase) * (1 + 0.2 * y[:, None, None])     # Doppler
    x -= x.mean(axis=2, keepdims=True)
    return torch.from_numpy(x.astype(np.float32)), torch.from_numpy(y.astype(np.int64))


train_x, train_y = make_data(2048, 1)
test_x, test_y = make_data(512, 2)
for prec in ("bf16", "fp16x3", "fp32"):
    torch.manual_seed(0); np.random.seed(0)
    constants.NFEATURES = C
    cfg = dict(constants.CONFIG); cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B)
    F_hip.set_precision(prec)
    tr = PCAATrainer(cfg, device="cuda", precision=prec)
    tr.sample_prior_means(); tr.finalize(); tr.train()
    X, Y = train_x.cuda(), train_y.cuda()
    log = []
    for s in range(a.steps):
        idx = torch.randint(0, X.shape[0], (B,), device="cuda")
        z0 = torch.from_numpy(np.random.normal(0, 1, (B, 32))).float().cuda()
        al = torch.rand(B, 1).cuda()
        out = tr.step(X[idx].permute(0, 3, 1, 2), Y[idx], z0, al)
        if s % max(1, a.steps // 6) == 0 or s == a.steps - 1:
            acc = (out["preds"] == Y[idx]).float().mean().item()
            log.append((s, out["rec_loss"].item(), out["sup_loss"].item(), out["d_loss"].item(), acc))
    tr.eval()
    correct = 0
    TX, TY = test_x.cuda(), test_y.cuda()
    for i in range(0, TX.shape[0], B):
        _, _, p, _ = tr.evaluate_batch(TX[i:i + B].permute(0, 3, 1, 2), TY[i:i + B])
        correct += int((p == TY[i:i + B]).sum())
    print(f"--- {prec} mode, B={B} N={N} K={K}")
    for s, r, c, d, acc in log:
        print(f"  step {s:4d}  chamfer {r:8.3f}  CE {c:6.3f}  d_loss {d:8.3f}  batch acc {acc:.2f}")
    print(f"  held-out accuracy (eval-mode BatchNorm, {TX.shape[0]} sequences): {correct / TX.shape[0]:.3f}  (chance {1 / K:.3f})")
    del tr
    torch.cuda.empty_cache()
