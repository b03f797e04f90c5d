#!/usr/bin/env python3
"""Bisect the hipGraph capture of the data-parallel step (round 6): each case in a child process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CASES = ["single", "emu_allreduce_fp32", "emu_allreduce_bf16", "emu_gather", "rccl1_allreduce_fp32", "rccl1_gather"]
if len(sys.argv) > 1:
    import torch
    from opensetgaitrecognition_pcaa_amd import constants, synthetic as syn
    from opensetgaitrecognition_pcaa_amd.train import PCAATrainer
    from opensetgaitrecognition_pcaa_amd.utils import sample_distant_points
    case = sys.argv[1]
    B, N, C, K, T = 64, 32, 4, 8, constants.NSTEPS
    constants.NFEATURES = C
    cfg = dict(constants.CONFIG); cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B)
    kw = {}
    if case.startswith("emu"):
        kw = dict(emulate_world=1 if case.startswith("emu1") else 8, dp_gather=case.endswith("gather"), grad_compress="bf16" if not case.endswith("fp32") else None)
    if case.startswith("rccl1"):
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29519", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        kw = dict(process_group=dist.group.WORLD, force_collectives=True, dp_gather=case.endswith("gather"),
                  grad_compress="bf16" if not case.endswith("fp32") else None)
    tr = PCAATrainer(cfg, precision="bf16", **kw)
    tr.set_prior_means(sample_distant_points(32, K, 10, 10)); tr.finalize(); tr.train()
    inp = (syn.synthetic_pcs(B, T, N, C).cuda().permute(0, 3, 1, 2), syn.synthetic_labels(B, K).cuda(), syn.synthetic_z0(B, 32).cuda(),
           syn.synthetic_alphas(B).cuda())
    for _ in range(2):
        tr.step(*inp)
    torch.cuda.synchronize()
    print("eager ok", tr.dp_scheme, flush=True)
    out = tr.step_graphed(*inp, warmup=0)
    torch.cuda.synchronize()
    print("captured + replayed", float(out["tot_loss"]), flush=True)
    import time
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step_graphed(*inp, warmup=0)
    torch.cuda.synchronize()
    print("graph ms/step %.3f" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step(*inp)
    torch.cuda.synchronize()
    print("eager ms/step %.3f" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
    sys.exit(0)
for name in CASES:
    r = subprocess.run([sys.executable, __file__, name], capture_output=True, text=True, timeout=400)
    lines = [l for l in (r.stdout + r.stderr).strip().splitlines() if "amdgpu.ids" not in l and "socket.cpp" not in l]
    print(f"{name}: rc={r.returncode} :: " + " | ".join(lines[-4:]), flush=True)
