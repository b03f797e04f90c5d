#!/usr/bin/env python3
"""Open-set inference throughput (BASELINE config[4]: inference_PCAA.py embedding + likelihood +
k-window vote on B=1024 sequences, one GPU): eval-mode CGEncoder -> joint likelihood under the
Gaussian-mixture prior -> k-vote.  python tools/bench_infer.py [--batch 1024] [--chunk 256]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip, ops, synthetic as syn  # noqa: E402
from opensetgaitrecognition_pcaa_amd.inference import joint_likelihood, k_vote  # noqa: E402
from opensetgaitrecognition_pcaa_amd.models import CGEncoder  # noqa: E402
from opensetgaitrecognition_pcaa_amd.utils import sample_distant_points  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--chunk", type=int, default=256, help="sequences per encoder launch")
    ap.add_argument("--points", type=int, default=128)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    K, T, C, N = 8, constants.NSTEPS, 4, a.points
    constants.NFEATURES = C
    F_hip.set_precision(a.precision)
    enc = CGEncoder(n_out_labels=K, use_projection_head=True, nmax_points=N).to(dev).float().eval()
    syn.deterministic_fill_(enc, 0)
    means = sample_distant_points(32, K, 10, 10).float().to(dev)
    pcs = syn.synthetic_pcs(a.batch, T, N, C, seed=7).to(dev).permute(0, 3, 1, 2)

    def run():
        liks, preds = [], []
        for i in range(0, a.batch, a.chunk):
            logits, sup_fv, _ = F_hip.encoder_forward(enc, pcs[i:i + a.chunk], False, a.precision)
            liks.append(joint_likelihood(sup_fv, means))
            preds.append(logits.argmax(1))
        lik = torch.cat(liks)
        pred = torch.cat(preds)
        thr = float(lik.median().item())          # stand-in for the Youden threshold of the validation set
        return k_vote(lik, pred, thr, 5, K)

    with torch.no_grad():
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            out = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
    print(f"inference B={a.batch} N={N} {a.precision}: {dt * 1e3:.2f} ms per batch, {a.batch / dt:.0f} sequences/s, "
          f"{out.numel()} windows", flush=True)


if __name__ == "__main__":
    main()
