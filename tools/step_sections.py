#!/usr/bin/env python3
"""Un-profiled section times of the train step on the MAIN stream (HIP events at ~12 section
boundaries, functional.set_marks): python tools/step_sections.py [--steps 20] [--precision bf16]."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip, synthetic as syn
from opensetgaitrecognition_pcaa_amd.train import PCAATrainer
from opensetgaitrecognition_pcaa_amd.utils import sample_distant_points

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--points", type=int, default=128)
ap.add_argument("--fused", default="on", choices=["on", "off"], help="decoder weight-gradient + Adam fusion")
ap.add_argument("--dp-force", action="store_true", help="1-rank RCCL group: the data-parallel step's collectives and hand-offs")
ap.add_argument("--grad-compress", default="bf16", choices=["none", "bf16"])
ap.add_argument("--dp-mode", default="allreduce", choices=["allreduce", "zero", "gather"])
a = ap.parse_args()
pg = None
if a.dp_force:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
    pg = dist.group.WORLD
B, N, C, K, T = a.batch, a.points, 4, 8, constants.NSTEPS
constants.NFEATURES = C
cfg = dict(constants.CONFIG); cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B)
F_hip.set_precision(a.precision)
tr = PCAATrainer(cfg, device="cuda", precision=a.precision, fused_decoder_update=("all" if a.fused != "off" else False), process_group=pg,
                 force_collectives=a.dp_force, grad_compress=None if a.grad_compress == "none" else a.grad_compress,
                 dp_zero=a.dp_mode == "zero" and pg is not None, dp_gather=a.dp_mode == "gather" and pg is not None)
for i, m in enumerate((tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head, tr.discriminator_projection_head)):
    syn.deterministic_fill_(m, i)
tr.set_prior_means(sample_distant_points(32, K, 10, 10)); tr.finalize(); tr.train()
pcs = syn.synthetic_pcs(B, T, N, C, seed=1234).cuda().permute(0, 3, 1, 2)
gt = syn.synthetic_labels(B, K, seed=1235).cuda(); z0 = syn.synthetic_z0(B, 32, seed=1236).cuda(); al = syn.synthetic_alphas(B, seed=1237).cuda()
for _ in range(3):
    tr.step(pcs, gt, z0, al)
torch.cuda.synchronize()
marks = []
F_hip.set_marks(marks)
for _ in range(a.steps):
    tr.step(pcs, gt, z0, al)
torch.cuda.synchronize()
F_hip.set_marks(None)
agg = collections.OrderedDict()
for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
    key = f"{n0} -> {n1}"
    agg.setdefault(key, []).append(e0.elapsed_time(e1))
tot = 0.0
for k, v in agg.items():
    v = sorted(v)[len(v) // 2]
    tot += v
    print(f"{k:44s} {v * 1e3:9.1f} us (median of {a.steps})")
print(f"{'sum':44s} {tot * 1e3:9.1f} us   (fused={a.fused})")
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(a.steps):
    tr.step(pcs, gt, z0, al)
t1.record()
torch.cuda.synchronize()
print(f"{'un-marked step':44s} {t0.elapsed_time(t1) / a.steps * 1e3:9.1f} us")
