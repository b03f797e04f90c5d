#!/usr/bin/env python3
"""PCAA train-step throughput on MI355X (BASELINE.json metric: gait sequences/s
of a full V4 train step -- encoder + decoder + discriminator forward/backward,
WGAN-GP D-step, Chamfer, both Adams -- on synthetic mmGait10-shaped batches).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        # outside a torchrun environment: starts that same command as a child
                                        # (before any GPU call), forwards rank 0's line, exits with its status

``--gpus`` is binding: under a launcher WORLD_SIZE must equal it (anything else exits non-zero with the command to
run), so a line's ``n_gpus`` is always the N that was asked for.

Workload at every N: BASELINE config[1], B=64 sequences per GPU, T=30, N=128
points, C=4 features, K=8 classes, inputs resident in HBM before the timed
region; data parallel over N GPUs (weak scaling: global batch 64*N, RCCL
all-reduce of the flat gradient buffers).  One JSON line on rank 0.

At N=1 the same line also carries the other BASELINE configs as extra keys, each
measured in this run (none of them is ``value``): ``parity_mode`` (config[1] in the
parity-grade modes the 1e-4 tests run in: split-fp16 "fp16x3", and exact fp32 under ``exact_fp32``), ``sweep`` (config[3], the
point-subsampling sweep N in {32,64,128,256}), ``infer`` (config[4], open-set
inference at B=1024), ``c5`` (config[1] with the C=5 feature set BASELINE's wording
names), ``with_batcher`` (datasets.py batch collation inside the loop),
``gpu_sections`` (where the step's time goes on the main stream) and
``cpu_baseline`` (the oracle on this host's cores: 1 warm-up + 3 timed steps,
median, per-phase breakdown).
"""
import argparse
import itertools
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3       # f32-input MFMA
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5,
                    help="the timed region (exactly --steps steps between barrier + synchronize) is repeated this many "
                         "times; ms_per_step / value are the MEDIAN window, all windows are in config.windows_ms_per_step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp16x3"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--points", type=int, default=128)
    ap.add_argument("--features", type=int, default=4)
    ap.add_argument("--classes", type=int, default=8)
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="train", choices=["train", "sweep", "infer", "loop"],
                    help="train: BASELINE config[1] (default, the driver's line, with the other configs as extra keys); "
                         "sweep / infer: config[3] / config[4] as a line of their own; loop: the drop-in train_variant4 loop "
                         "alone (the default line's `loop` leg)")
    ap.add_argument("--dp-mode", default="auto", choices=["auto", "allreduce", "zero", "gather"],
                    help="data-parallel exchange of the decoder gradients: per-layer all-reduce buckets, reduce-scatter + "
                         "sharded Adam + all-gather (ZeRO-1), or gather: the wide layers all-gather their two small "
                         "weight-gradient operands and every rank forms the global gradient inside the fused "
                         "weight-gradient + Adam kernel (round 5; bf16 mode, world x batch <= 512).  auto = gather in the bf16 "
                         "throughput mode (a tenth of the bytes, no separate Adam pass), all-reduce in the parity modes")
    ap.add_argument("--dp-force", action="store_true",
                    help="N=1 only: create a 1-rank RCCL group and issue the step's collectives on it (exercises the RCCL "
                         "calls and measures their fixed cost on one GPU)")
    ap.add_argument("--dp-emulate", type=int, default=0, metavar="W",
                    help="N=1 only: time ONE RANK'S PROGRAM of a W-rank data-parallel job on this GPU (PCAATrainer "
                         "emulate_world: gradient scale 1/W, 64 W stacked rows in the gathered decoder update, every collective "
                         "replaced by a device operation of the same bytes).  The line says `emulated`; n_gpus stays 1 and value "
                         "is this one rank's sequences/s.  Without the flag the default line carries the same measurement for "
                         "W in (2, 4, 8) under `dp_emulated`, with a `scale_projection` from stated xGMI figures")
    ap.add_argument("--grad-compress", default="auto", choices=["auto", "none", "bf16"],
                    help="decoder gradient buckets cross the wire as bf16 (fp32 master gradients, moments and weights); auto "
                         "= bf16 in the bf16 throughput mode, none in the fp32 parity mode (docs/LAB_LOG.md section 6)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--decoder-update", default="fused", choices=["fused", "plain"],
                    help="single process, bf16: decoder weight gradient + Adam in one kernel (default) or as two passes")
    ap.add_argument("--no-parity-mode", action="store_true",
                    help="skip the fp32 parity-mode leg (same workload in the mode the 1e-4 parity tests run in)")
    ap.add_argument("--no-batcher-leg", action="store_true",
                    help="skip the leg that assembles every step's batch from the HBM-resident packed store")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the sweep / infer / c5 / gpu_sections legs of the default line")
    ap.add_argument("--backend", default=os.environ.get("PCAA_DIST_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path on a one-GPU box "
                         "together with PCAA_BENCH_DEVICE=0, which puts every rank on that GPU)")
    ap.add_argument("--graph", default=os.environ.get("PCAA_GRAPH", "auto"), choices=["on", "off", "auto"],
                    help="replay the step as a captured hipGraph (PCAATrainer.step_graphed): auto = where the eager step is "
                         "bound by the host's enqueues (PCAATrainer.prefers_graph)")
    a = ap.parse_args()
    if a.grad_compress == "auto":
        a.grad_compress = "bf16" if a.precision == "bf16" else "none"
    if a.dp_mode == "auto":
        a.dp_mode = "gather" if a.precision == "bf16" else "allreduce"
    a.windows = max(1, a.windows)
    return a


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (plain-PyTorch restatement of the reference, kind="port") on this host's cores
# ---------------------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _conv2d_vs_einsum(threads, b=8, T=30, N=128, cin=1024, cout=1024):
    """The oracle writes the per-point layers as einsum contractions, the reference as nn.Conv2d(1x1): the same layer
    (PointNet 4, the largest) timed both ways on this host, so that "the port is on par with ATen's conv2d" is a number
    in the line.  Best of 2 after a warm-up each, seconds."""
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(b, cin, T, N, generator=g)
    w = torch.randn(cout, cin, generator=g) * 0.03
    w4 = w.view(cout, cin, 1, 1)

    def best(fn):
        fn()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts)
    with torch.no_grad():
        t_conv = best(lambda: torch.nn.functional.conv2d(x, w4))
        t_ein = best(lambda: torch.einsum("bctn,oc->botn", x, w))
    return {"shape": f"[{b},{cin},{T},{N}] -> {cout} channels, 1x1", "aten_conv2d_s": t_conv, "oracle_einsum_s": t_ein,
            "einsum_over_conv2d": t_ein / t_conv}


def cpu_baseline(B, N, C, K, T, budget_s=150.0):
    """BASELINE.md section 3 / SURVEY 8d protocol: identical synthetic tensors, 1 warm-up + 3 timed full V4 train
    steps (loop body PCAA_ablation.py:882-1021), median, with the per-phase breakdown.  The thread count is picked
    on a small calibration batch first (torch's CPU kernels do not scale to every core of a big host).  The budget
    (round 4: 150 s, was 75) lets the WORKLOAD'S OWN batch run (4 steps of ~20 s at B=64 on the GPU box's host); only if
    even that does not fit is the batch halved -- ``batch`` says what ran, ``batch_is_workload`` whether it is B."""
    from opensetgaitrecognition_pcaa_amd import constants, models, synthetic as syn
    from oracle import pcaa_oracle as O
    ncpu = os.cpu_count() or 1
    constants.NFEATURES = C
    enc = models.CGEncoder(K, nmax_points=N, use_projection_head=True).float()
    dec = models.CGDecoder(input_dim=64, nmax_points=N).float()
    disc = models.CGDiscriminator(K).float()
    gph = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ELU()).float()
    dph = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.ELU()).float()
    sds = []
    for i, m in enumerate((enc, dec, disc, gph, dph)):
        syn.deterministic_fill_(m, i)
        sds.append({k: v.detach().clone() for k, v in m.state_dict().items()})
    means = O.sample_distant_points(32, K, 10, 10).float()
    st = O.V4State(*sds, means, C, T, N, K)
    cfg = dict(LR=1e-4, B1=0.9, B2=0.99, GP_WEIGHT=15, ADV_WEIGHT=1)
    pcs = syn.synthetic_pcs(B, T, N, C, seed=1234).permute(0, 3, 1, 2)
    gt = syn.synthetic_labels(B, K, seed=1235)
    z0 = syn.synthetic_z0(B, 32, seed=1236)
    al = syn.synthetic_alphas(B, seed=1237)

    def run(b, threads, phases=None):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        O.v4_train_step(st, pcs[:b], gt[:b], z0[:b], al[:b], cfg, phase_seconds=phases)
        return time.perf_counter() - t0

    cal_b = min(8, B)
    cands = sorted({min(ncpu, t) for t in (8, 16, 32, 64)})
    run(cal_b, cands[-1])                                   # first touch of every weight / allocator warm-up
    trials = {t: run(cal_b, t) for t in cands}
    threads = min(trials, key=trials.get)
    est_full = trials[threads] * (B / cal_b)                # s per step at the full batch (Adam does not scale with B:
    sample_b = B                                            # the estimate is an upper bound)
    while sample_b > cal_b and 4 * est_full * (sample_b / B) > budget_s:
        sample_b //= 2
    run(sample_b, threads)                                  # warm-up
    times, phases = [], []
    for _ in range(3):
        ph = {}
        times.append(run(sample_b, threads, ph))
        phases.append(ph)
    med = statistics.median(times)
    pmed = {k: statistics.median(p[k] for p in phases) for k in phases[0]}
    return {"value": sample_b / med, "unit": "sequences/s", "cores": threads, "kind": "port",
            "batch": sample_b, "batch_is_workload": sample_b == B, "threads": threads, "threads_tried": cands,
            "host_cpus": ncpu, "cpu_model": _cpu_model(), "torch": torch.__version__,
            "conv2d_vs_einsum": _conv2d_vs_einsum(threads),
            "protocol": "1 warm-up + 3 timed full V4 train steps, median",
            "seconds_per_step": times, "median_s": med,
            "phases_s": pmed,
            "phases_frac": {k: v / sum(pmed.values()) for k, v in pmed.items()},
            "sample": f"oracle (plain PyTorch fp32) at B={sample_b} of the workload's {B}, N={N}, C={C}; {threads} threads "
                      f"(best of {cands} on a B={cal_b} calibration step) of a {ncpu}-CPU host; 1 warm-up + 3 timed steps",
            "thread_calibration_s": {str(t): round(v, 2) for t, v in trials.items()}}


# ---------------------------------------------------------------------------------------------------------------
# algorithmic work
# ---------------------------------------------------------------------------------------------------------------
def pointnet_train_flops(P):
    """2 FLOP/MAC x 1.837 M MAC per point forward, x3 for forward + dgrad + wgrad."""
    return 3 * 2 * 1.837e6 * P


def step_algorithmic(tr, B, T, N):
    """(FLOPs, HBM bytes) of one train step, SURVEY section 8(d): GEMM layers 3x forward; decoder/optimizer
    parameters 40 B each (fwd read, bwd read, dW write, Adam 28), encoder activations 18.4 KB per point."""
    P = B * T * N
    n_dec = sum(p.numel() for p in tr.decoder.parameters() if p.dim() == 2)
    flops = pointnet_train_flops(P) + 3 * 2 * n_dec * B + 3 * 2 * 572928 * B * T
    nbytes = 40.0 * tr.flat_g.total + 3 * 2 * 3072 * P
    return flops, nbytes


def step_fracs(tr, B, T, N, ms):
    flops, nbytes = step_algorithmic(tr, B, T, N)
    mf = flops / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS
    hf = nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS
    return {"step_flops": flops, "step_hbm_bytes": nbytes, "mfma_frac": mf, "hbm_frac": hf,
            "bound": "hbm" if nbytes / PEAK_HBM_GBS / 1e9 > flops / PEAK_BF16_TFLOPS / 1e12 else "mfma"}


# ---------------------------------------------------------------------------------------------------------------
# construction helpers
# ---------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def device_fill_(module, seed):
    """Random-init weights of the architecture drawn on the device (the extra legs: a 627 M-parameter decoder takes
    ~10 s of numpy draws on the host; the parity-tested main leg keeps synthetic.deterministic_fill_)."""
    for i, (name, t) in enumerate(module.state_dict().items()):
        if not t.dtype.is_floating_point:
            t.zero_()
            continue
        g = torch.Generator(device=t.device)
        g.manual_seed(1000 * seed + i)
        if name.endswith("running_var"):
            t.copy_(0.5 + torch.rand(t.shape, generator=g, device=t.device))
        elif name.endswith("running_mean"):
            t.copy_(0.1 * torch.randn(t.shape, generator=g, device=t.device))
        elif t.dim() == 1:
            r = torch.randn(t.shape, generator=g, device=t.device)
            t.copy_(1.0 + 0.1 * r if name.endswith("weight") else 0.05 * r)
        else:
            fan_in = t[0].numel()
            t.copy_(torch.randn(t.shape, generator=g, device=t.device) / fan_in ** 0.5)


def build_trainer(a, N, dev, pg, precision, fill="deterministic", emulate_world=0):
    from opensetgaitrecognition_pcaa_amd import constants, synthetic as syn
    from opensetgaitrecognition_pcaa_amd.train import PCAATrainer
    from opensetgaitrecognition_pcaa_amd.utils import sample_distant_points
    cfg = dict(constants.CONFIG)
    cfg.update(NMAX=N, TRAIN_CLASSES=list(range(a.classes)), BATCH_SIZE=a.batch)
    tr = PCAATrainer(cfg, device=dev, precision=precision, process_group=pg, sync_bn=a.sync_bn,
                     dp_zero=(a.dp_mode == "zero") and pg is not None,
                     dp_gather=(a.dp_mode == "gather") and (pg is not None or bool(emulate_world)),
                     grad_compress=None if a.grad_compress == "none" else a.grad_compress,
                     force_collectives=a.dp_force, emulate_world=emulate_world,
                     fused_decoder_update=("all" if a.decoder_update == "fused" else False))
    for i, m in enumerate((tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head,
                           tr.discriminator_projection_head)):
        if fill == "deterministic":
            syn.deterministic_fill_(m, i)
        else:
            device_fill_(m, i)
    tr.set_prior_means(sample_distant_points(32, a.classes, 10, 10))
    tr.finalize()
    tr.train()
    return tr, cfg


def make_inputs(B, T, N, C, K, dev, rank=0):
    """inputs resident in HBM (point-major storage, [B,C,T,N] view), different data per rank (SURVEY 8d seeds)"""
    from opensetgaitrecognition_pcaa_amd import synthetic as syn
    pcs = syn.synthetic_pcs(B, T, N, C, seed=1234 + rank).to(dev).permute(0, 3, 1, 2)
    gt = syn.synthetic_labels(B, K, seed=1235 + rank).to(dev)
    z0 = syn.synthetic_z0(B, 32, seed=1236 + rank).to(dev)
    al = syn.synthetic_alphas(B, seed=1237 + rank).to(dev)
    return pcs, gt, z0, al


def time_single_gpu(run, steps, windows):
    """``windows`` timed regions of exactly ``steps`` calls each, synchronize on both sides -> ms per step of each."""
    out = []
    for _ in range(windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = run()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    return out, r


def spread(ms_list):
    return {"median": statistics.median(ms_list), "min": min(ms_list), "max": max(ms_list), "windows": len(ms_list)}


# ---------------------------------------------------------------------------------------------------------------
# extra legs of the default line (single GPU)
# ---------------------------------------------------------------------------------------------------------------
def leg_train_shape(a, dev, N, C, steps, windows, warmup=3, B=None, precision=None):
    """The train step at another shape (sweep points, C=5, the reference's default B=16 / N=150): ms per step (median
    window) + whole-step fractions."""
    from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip
    import copy
    B = a.batch if B is None else B
    K, T = a.classes, constants.NSTEPS
    precision = precision or a.precision
    c_saved, p_saved = constants.NFEATURES, F_hip.get_precision()
    constants.NFEATURES = C
    F_hip.set_precision(precision)
    a = copy.copy(a)
    a.batch = B
    try:
        tr, _ = build_trainer(a, N, dev, None, precision, fill="device")
        pcs, gt, z0, al = make_inputs(B, T, N, C, K, dev)
        graph = a.graph == "on" or (a.graph == "auto" and tr.prefers_graph(B, N))
        run = (lambda: tr.step_graphed(pcs, gt, z0, al, warmup=0)) if graph else (lambda: tr.step(pcs, gt, z0, al))
        for _ in range(warmup):
            tr.step(pcs, gt, z0, al)
        run()
        ms_list, out = time_single_gpu(run, steps, windows)
        ms = statistics.median(ms_list)
        ent = {"N": N, "C": C, "B": B, "precision": precision, "ms_per_step": ms, "value": B / ms * 1e3, "unit": "sequences/s",
               "windows_ms_per_step": ms_list, "steps": steps, "hip_graph": bool(graph),
               "finite_loss": bool(torch.isfinite(out["tot_loss"]).item())}
        ent.update(step_fracs(tr, B, T, N, ms))
        ent["frac"] = max(ent["mfma_frac"], ent["hbm_frac"])
        del tr
        torch.cuda.empty_cache()
        return ent
    finally:
        constants.NFEATURES = c_saved
        F_hip.set_precision(p_saved)


def leg_infer(a, dev, steps, windows, with_cpu=False):
    """BASELINE config[4]: eval-mode CGEncoder (BatchNorm + ELU [+ mean-pool] in the GEMM epilogues) -> fp64 mixture
    likelihood -> k=6 window vote on B=1024 sequences resident in HBM; one step = one batch."""
    from opensetgaitrecognition_pcaa_amd import constants, inference, models, ops, synthetic as syn
    from opensetgaitrecognition_pcaa_amd.utils import sample_distant_points
    B, N, C, K, T = 1024, a.points, a.features, a.classes, constants.NSTEPS
    enc = models.CGEncoder(K, nmax_points=N, use_projection_head=True).float()
    syn.deterministic_fill_(enc, 0)
    enc = enc.to(dev).eval()
    scorer = inference.OpenSetScorer(enc, sample_distant_points(32, K, 10, 10), batch_size=B)
    pcs = syn.synthetic_pcs(B, T, N, C, seed=5).to(dev).permute(0, 3, 1, 2)

    def step():
        preds, fv, lik = scorer.embed(pcs)
        scorer.threshold = 1e-30
        return scorer.vote(lik, preds, 6, K), lik

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    timer = ops.LaunchTimer(only_prefix="gemm_bf16_" if a.precision in ("bf16", "fp16x3") else "gemm_f32_kernel")
    ops.set_timer(timer)
    step()
    ops.set_timer(None)
    ms_list, (votes, lik) = time_single_gpu(step, steps, windows)
    ms = statistics.median(ms_list)
    agg = timer.summary()
    name, r = max(agg.items(), key=lambda kv: kv[1]["ms"])
    peak = PEAK_BF16_TFLOPS if name.startswith("gemm_bf16") else PEAK_F32_TFLOPS
    achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
    flops_seq = 2 * 1.837e6 * T * N + 2 * 572928 * T
    ent = {"workload": f"open-set inference: eval CGEncoder + joint likelihood + k=6 vote, B={B} T={T} N={N} C={C} K={K}, "
                       "BASELINE config[4]",
           "ms_per_step": ms, "value": B / ms * 1e3, "unit": "sequences/s", "windows_ms_per_step": ms_list, "steps": steps,
           "finite": bool(torch.isfinite(lik).all().item()), "windows_voted": int(votes.numel()),
           "algorithmic_gflop_per_sequence": flops_seq / 1e9,
           # whole path against the MFMA peak (14.14 GFLOP per sequence), and its dominant kernel alone
           "frac": flops_seq * B / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, "bound": "mfma",
           "roofline": {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                        "frac": achieved / peak, "traffic": None, "launches_per_step": r["launches"],
                        "avg_launch_ms": r["ms"] / r["launches"],
                        "algorithmic_flop_per_launch": r["flops"] / r["launches"]}}
    if with_cpu:
        from oracle import pcaa_oracle as O
        sd = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
        xb = pcs[:64].cpu().contiguous()
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        t1 = time.perf_counter()
        with torch.no_grad():
            O.cg_encoder_forward(xb, sd, True, training=False)
        d = time.perf_counter() - t1
        ent["cpu_baseline"] = {"value": 64 / d, "unit": "sequences/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"oracle eval-mode encoder forward on 64 of the 1024 sequences: {d:.1f} s"}
    del scorer, enc
    torch.cuda.empty_cache()
    return ent


def leg_sections(tr, inputs, steps=10):
    """HIP events at the section boundaries of the MAIN stream (functional.set_marks; tools/step_sections.py):
    median microseconds per section -- the GPU side of cpu_baseline's phase breakdown."""
    from opensetgaitrecognition_pcaa_amd import functional as F_hip
    marks = []
    F_hip.set_marks(marks)
    try:
        for _ in range(steps):
            tr.step(*inputs)
        torch.cuda.synchronize()
    finally:
        F_hip.set_marks(None)
    agg = {}
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        agg.setdefault(f"{n0} -> {n1}", []).append(e0.elapsed_time(e1) * 1e3)
    us = {k: statistics.median(v) for k, v in agg.items()}

    def tot(*keys):
        return sum(v for k, v in us.items() if any(k.startswith(p) for p in keys))
    # the CPU phases' counterparts (the critic's D-step runs on its own stream beside decoder forward / Chamfer /
    # decoder backward and costs the main stream nothing; the decoder's Adam runs beside the temporal block's backward)
    return {"us": us,
            "phases_us": {"encoder_fwd": tot("enc_fwd.begin", "enc_fwd.pointnet", "enc_fwd.dtc"),
                          "d_step": 0.0,
                          "decoder_chamfer": tot("heads_fwd", "dec_fwd"),
                          "backward": tot("chamfer", "dec_bwd", "enc_bwd.heads", "enc_bwd.dtc"),
                          "adam_g": tot("enc_bwd.pointnet", "adam+join")},
            "note": "median over %d steps of event intervals on the main stream; D-step: critic stream, hidden" % steps}


def leg_loop(a, dev, N, steps=100, valid_batches=8, epochs=3):
    """The drop-in loop itself (train.train_variant4 = the reference's PCAA_ablation.py:746-1122 call surface): `epochs`
    epochs over a synthetic split of `steps` train batches and `valid_batches` validation batches resident in HBM as the
    packed store the real-data path uses -- the device batcher's row gathers, the epoch's host RNG draws (z0, alphas: the
    reference's generators, one pinned asynchronous copy per epoch), the step, the per-epoch statistics and the validation
    pass.  Wall clock per epoch from inside the loop (its ``timing`` hook); the checkpoint writes after an improving
    epoch are outside those intervals (630 MB of torch.save for the config[1] decoder)."""
    import shutil
    import tempfile
    from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip
    from opensetgaitrecognition_pcaa_amd.datasets import SyntheticGaitDataset
    from opensetgaitrecognition_pcaa_amd.train import train_variant4
    B, C, K = a.batch, a.features, a.classes
    saved = (constants.NFEATURES, F_hip.get_precision(), os.getcwd())
    work = tempfile.mkdtemp(prefix="pcaa_loop_")
    try:
        constants.NFEATURES = C
        F_hip.set_precision(a.precision)
        os.chdir(work)
        cfg = dict(constants.CONFIG)
        cfg.update(MODEL_NAME="bench_loop", TRAIN_CLASSES=list(range(K)), NMAX=N, BATCH_SIZE=B, EPOCHS=epochs,
                   CHECKPOINT_FREQUENCY=1, SUPERVISION_FREQUENCY=1, SUBSAMPLE_FACTOR=1.0, NOTES="")
        make = lambda split: SyntheticGaitDataset(steps * B if split.value == "train" else valid_batches * B, K, N=N, C=C,
                                                  seed=4400 + (0 if split.value == "train" else 1))
        timing, records = [], []
        import contextlib
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):       # the loop prints its epoch records like the reference: stdout is the JSON line's
            trainer, hist = train_variant4(cfg, wandb_mode="disabled", dataset_factory=make, device=str(dev), timing=timing,
                                           log_fn=records.append)
        torch.cuda.synchronize()
        total_s = time.perf_counter() - t0
        del trainer
        torch.cuda.empty_cache()
    finally:
        os.chdir(saved[2])
        constants.NFEATURES = saved[0]
        F_hip.set_precision(saved[1])
        shutil.rmtree(work, ignore_errors=True)
    steady = timing[1:] or timing                       # epoch 0 carries lazy initialisation (first launches, allocator)
    tr_s = statistics.median(t["train_s"] for t in steady)
    va_s = statistics.median(t["valid_s"] for t in steady)
    n = steady[0]["train_steps"]
    return {"workload": f"train_variant4 (the drop-in loop): {epochs} epochs x {n} steps at B={B} N={N} C={C} + {valid_batches} "
                        "validation batches per epoch, synthetic split as a packed store in HBM (DeviceBatcher), epoch order / z0 / "
                        "alphas from the reference's host generators",
            "train_steps_per_epoch": n, "epochs": epochs,
            "train_ms_per_step": tr_s / n * 1e3, "value": n * B / tr_s, "unit": "sequences/s",
            "value_with_validation": n * B / (tr_s + va_s), "valid_ms_per_batch": va_s / max(1, steady[0]["valid_batches"]) * 1e3,
            "per_epoch": timing, "total_s_with_setup_and_checkpoints": total_s,
            "finite": all(v == v for v in hist[-1].values())}


XGMI_LINKS, XGMI_LINK_GBS = 7, 153.0      # per GPU: 7 point-to-point links x ~153 GB/s (MI355X_MICROARCH.md)


def leg_dp_emulated(a, dev, N, worlds=(2, 4, 8), steps=10, warmup=3):
    """One rank's program of a W-rank job on this one GPU (PCAATrainer emulate_world / dist.EmulatedExchange), for each W
    and each decoder exchange scheme, beside the single-process step built and timed the same way; and what those numbers
    say about W real GPUs under STATED assumptions (`scale_projection`).  Nothing here is a multi-GPU measurement."""
    import copy
    from opensetgaitrecognition_pcaa_amd import constants
    B, C, K, T = a.batch, a.features, a.classes, constants.NSTEPS
    inputs = make_inputs(B, T, N, C, K, dev)

    def run(world, mode, n_points=N, graph=False):
        aa = copy.copy(a)
        aa.dp_mode, aa.dp_force, aa.sync_bn = mode, False, False
        aa.grad_compress = "bf16" if a.precision == "bf16" else "none"
        tr, _ = build_trainer(aa, n_points, dev, None, a.precision, fill="device", emulate_world=world)
        inp = inputs if n_points == N else make_inputs(B, T, n_points, C, K, dev)
        for _ in range(warmup):
            tr.step(*inp)
        fn = (lambda: tr.step_graphed(*inp, warmup=0)) if graph else (lambda: tr.step(*inp))
        fn()
        ms_list, out = time_single_gpu(fn, steps, 3)
        ent = {"world": world, "N": n_points, "hip_graph": bool(graph),
               "dp_mode": tr.dp_scheme, "ms_per_step": statistics.median(ms_list), "windows_ms_per_step": ms_list,
               "steps": steps, "finite_loss": bool(torch.isfinite(out["tot_loss"]).item()),
               "collectives_per_step": tr.comm["collectives"], "payload_bytes_per_step": tr.comm["payload_bytes"],
               "gather_payload_bytes": tr.comm.get("gather_bytes", 0), "allreduce_payload_bytes": tr.comm.get("allreduce_bytes", 0)}
        del tr
        torch.cuda.empty_cache()
        return ent

    base = run(0, "allreduce")
    legs = []
    default_mode = "gather" if a.precision == "bf16" else "allreduce"
    for w in worlds:
        for mode in ((default_mode, "allreduce") if default_mode != "allreduce" else ("allreduce",)):
            if mode == "allreduce" and w != max(worlds):
                continue                       # the all-reduce rank program does not depend on W: timed once, at the largest
            ent = run(w, mode)
            # bytes this rank puts on / takes off the wire per step: an all-gather delivers the other ranks' (w-1)/w of its
            # payload, a ring all-reduce moves 2 (w-1)/w of its payload per rank and direction
            wire = (w - 1) / w * ent["gather_payload_bytes"] + 2.0 * (w - 1) / w * ent["allreduce_payload_bytes"]
            agg_ms = wire / (XGMI_LINKS * XGMI_LINK_GBS * 1e9) * 1e3
            link_ms = wire / (XGMI_LINK_GBS * 1e9) * 1e3
            ent["vs_single_process_step"] = ent["ms_per_step"] / base["ms_per_step"]
            ent["scale_projection"] = {
                "wire_bytes_per_rank": wire,
                "wire_ms_all_links": agg_ms, "wire_ms_one_link": link_ms,
                # weak scaling: W ranks each finish 64 sequences per step
                "projected_ms_per_step": ent["ms_per_step"] + agg_ms,
                "projected_value": w * B / (ent["ms_per_step"] + agg_ms) * 1e3,
                "projected_speedup_vs_1gpu": w * base["ms_per_step"] / (ent["ms_per_step"] + agg_ms),
                "pessimistic_one_link_exposed": {"projected_ms_per_step": ent["ms_per_step"] + link_ms,
                                                 "projected_speedup_vs_1gpu": w * base["ms_per_step"] / (ent["ms_per_step"] + link_ms)}}
            legs.append(ent)
    # the data-parallel step at small N, where the eager step is bound by the host's enqueues (~2.5 ms): round 6 lets
    # step_graphed capture it, collectives included (RCCL's are stream operations; here the emulated ones)
    small = {"N": 32, "world": max(worlds),
             "single_process_graph": run(0, "allreduce", 32, True)["ms_per_step"],
             "eager": run(max(worlds), default_mode, 32, False)["ms_per_step"],
             "graph": run(max(worlds), default_mode, 32, True)["ms_per_step"]}
    return {"label": "EMULATED on one GPU -- not a multi-GPU measurement", "small_n_graph_vs_eager_ms": small,
            "what": "one rank's program of a W-rank weak-scaling job (B=%d per rank, N=%d): gradient scale 1/W, the fused decoder "
                    "update from 64 W stacked rows (pcaa_skinny_linear_wgrad_adam_rows), every collective replaced by a device "
                    "operation of the same bytes on its own stream (all-reduce: in-place scale, all-gather: own rows + W-1 staged "
                    "copies); 3 warm-up + `steps` steps x 3 windows, median" % (B, N),
            "single_process_step": base, "legs": legs,
            "assumptions": ["scale_projection adds the rank's wire bytes over %d xGMI links x %.0f GB/s (all links busy, a direct "
                            "all-gather on the fully connected node) with NO overlap credited although the gathers are waited for "
                            "on the side stream only; `pessimistic_one_link_exposed` prices the same bytes over ONE link (a ring), "
                            "still fully exposed" % (XGMI_LINKS, XGMI_LINK_GBS),
                            "no latency term (4-10 collectives per step at ~20-40 us each would add 0.1-0.4 ms if none overlapped), "
                            "no rank skew, per-rank BatchNorm (no SyncBN exchange)",
                            "the emulated collectives occupy one HIP stream like RCCL's, but run as copy / scale kernels on the CUs "
                            "rather than as RCCL's kernels: CU contention of the real collectives is approximated, not reproduced"]}


def workload_sweep(a, dev):
    """BASELINE config[3] as a line of its own: the train step at N in {32,64,128,256}, B=64, one GPU."""
    from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip
    B, C, K, T = a.batch, a.features, a.classes, constants.NSTEPS
    F_hip.set_precision(a.precision)
    entries = [leg_train_shape(a, dev, N, C, a.steps, a.windows, max(a.warmup, 3)) for N in (32, 64, 128, 256)]
    tot_ms = sum(e["ms_per_step"] for e in entries)
    worst = min(entries, key=lambda e: e["frac"])
    line = {"metric": "gait sequences/sec (train step), point-subsampling sweep", "value": 4 * B / tot_ms * 1e3,
            "unit": "sequences/s", "n_gpus": 1, "steps": a.steps, "warmup": max(a.warmup, 3),
            "ms_per_step": tot_ms / 4, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if a.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"PCAA V4 train step at N in (32,64,128,256), B={B} T={T} C={C} K={K}, BASELINE config[3]; "
                                   "value = all sequences / all time", "sweep": entries},
            "roofline": {"bound": worst["bound"], "achieved": worst["step_flops"] / (worst["ms_per_step"] * 1e-3) / 1e12
                         if worst["bound"] == "mfma" else worst["step_hbm_bytes"] / (worst["ms_per_step"] * 1e-3) / 1e9,
                         "peak": PEAK_BF16_TFLOPS if worst["bound"] == "mfma" else PEAK_HBM_GBS,
                         "unit": "TFLOP/s" if worst["bound"] == "mfma" else "GB/s",
                         "frac": worst["frac"], "traffic": None,
                         "note": f"whole-step algorithmic work / step time at the sweep's worst point (N={worst['N']}); "
                                 "per-N figures in config.sweep"}}
    print(json.dumps(line), flush=True)


def workload_infer(a, dev):
    from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip
    constants.NFEATURES = a.features
    F_hip.set_precision(a.precision)
    ent = leg_infer(a, dev, a.steps, a.windows, with_cpu=not a.no_cpu_baseline)
    line = {"metric": "gait sequences/sec (open-set inference)", "value": ent["value"], "unit": "sequences/s",
            "n_gpus": 1, "steps": a.steps, "warmup": 2, "ms_per_step": ent["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if a.precision == "bf16" else "f32", "data": "synthetic",
            "config": {k: ent[k] for k in ("workload", "finite", "windows_voted", "algorithmic_gflop_per_sequence",
                                           "windows_ms_per_step")} | {"whole_path_mfma_frac": ent["frac"]},
            "roofline": ent["roofline"]}
    if "cpu_baseline" in ent:
        line["cpu_baseline"] = ent["cpu_baseline"]
    print(json.dumps(line), flush=True)


# ---------------------------------------------------------------------------------------------------------------
# --gpus N is honoured: either this process IS one of N ranks (torchrun environment, WORLD_SIZE == N), or it becomes
# the launcher of N ranks.  Decided before anything touches the GPU; the launcher never does.
# ---------------------------------------------------------------------------------------------------------------
def torchrun_command(gpus, argv, port=None):
    """The command the contract launches for N > 1 (and the one this file starts as a CHILD when asked for N > 1
    outside a torchrun environment)."""
    if port is None:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def resolve_world(a, argv, environ=None):
    """-> ("rank", world) when this process runs the bench itself, ("launch", cmd) when it must start ``--gpus`` ranks
    as child processes.  ``--gpus`` != WORLD_SIZE is always an error (SystemExit, non-zero): a line whose n_gpus is not
    the N that was asked for is worse than no line."""
    environ = os.environ if environ is None else environ
    ws = environ.get("WORLD_SIZE")
    if a.gpus < 1:
        raise SystemExit(f"bench.py: --gpus {a.gpus}: need at least one GPU")
    if ws is not None:
        if int(ws) != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={ws}: launch exactly one rank per GPU, e.g.\n  "
                             + " ".join(torchrun_command(a.gpus, argv, port=29500)))
        return "rank", int(ws)
    if a.gpus == 1:
        return "rank", 1
    if a.workload != "train":
        raise SystemExit("bench.py: --workload sweep / infer are single-GPU workloads")
    return "launch", torchrun_command(a.gpus, argv)


def launch_ranks(a, cmd):
    """Parent of an N > 1 run started as plain ``python bench.py --gpus N``: N children through torch.distributed.run
    (one per GPU, RCCL unless --backend gloo), their output forwarded as it comes (rank 0 prints the JSON line), exit
    status = theirs.  No HIP call is made in this process: device_count() does not initialise the GPU on this image,
    and nothing is exec'ed -- the children are ordinary subprocesses."""
    import subprocess
    have = torch.cuda.device_count()
    if a.backend == "nccl" and "PCAA_BENCH_DEVICE" not in os.environ and have < a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but this node shows {have} GPU(s)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    print("bench.py: starting %d ranks: %s" % (a.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT)
    try:
        rc = proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        rc = proc.wait()
    raise SystemExit(rc)


def main():
    a = parse()
    mode, what = resolve_world(a, sys.argv[1:])
    if mode == "launch":
        return launch_ranks(a, what)
    if a.workload != "train":
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("bench.py: --workload sweep / infer are single-GPU workloads")
        dev = torch.device("cuda", int(os.environ.get("PCAA_BENCH_DEVICE", "0")))
        torch.cuda.set_device(dev)
        if a.workload == "loop":
            from opensetgaitrecognition_pcaa_amd import functional as F_hip
            F_hip.set_precision(a.precision)
            print(json.dumps(leg_loop(a, dev, a.points)), flush=True)
            return None
        return workload_sweep(a, dev) if a.workload == "sweep" else workload_infer(a, dev)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip, ops

    if a.dp_emulate and (world > 1 or a.dp_force):
        raise SystemExit("bench.py: --dp-emulate W times one rank's program on ONE GPU: use it with --gpus 1 and without --dp-force")
    if a.dp_emulate and a.dp_mode == "zero":
        raise SystemExit("bench.py: --dp-emulate covers --dp-mode gather / allreduce")
    dev_index = int(os.environ.get("PCAA_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pg = None
    if world > 1 or a.dp_force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
        pg = dist.group.WORLD

    B, N, C, K, T = a.batch, a.points, a.features, a.classes, constants.NSTEPS
    constants.NFEATURES = C
    F_hip.set_precision(a.precision)
    tr, cfg = build_trainer(a, N, dev, pg, a.precision, emulate_world=a.dp_emulate)
    pcs, gt, z0, al = make_inputs(B, T, N, C, K, dev, rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world > 1:
            t = torch.tensor([seconds], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return seconds

    use_graph = (a.graph == "on" and tr.can_graph()) or (a.graph == "auto" and world == 1 and tr.prefers_graph(B, N))
    if a.graph == "on" and not use_graph:
        raise SystemExit("bench.py: --graph on, but this step cannot be captured (gloo collectives run on the host)")
    run_step = tr.step
    for _ in range(a.warmup):
        out = tr.step(pcs, gt, z0, al)
    if use_graph:
        # capture (the step is recorded, then replayed once: one more real, untimed step)
        out = tr.step_graphed(pcs, gt, z0, al, warmup=0)
        run_step = tr.step_graphed
    timer = None
    if not a.no_kernel_timing:
        # HIP events on the kernel family the roofline reports: the start/stop events ride on the launch
        # itself (hipExtLaunchKernelGGL through pcaa_time_next_gemm), i.e. they are the kernel's own begin/end
        # timestamps -- the quantity rocprofv3's kernel trace reports -- on the stream it is launched on
        timer = ops.LaunchTimer(only_prefix="gemm_bf16_" if a.precision in ("bf16", "fp16x3") else "gemm_f32_kernel")
    # the launches of the first `timed_steps` steps of the FIRST window carry the events
    # (graph mode: those steps run eagerly -- events cannot be read back from inside a replayed graph)
    timed_steps = min(a.steps, 2 if use_graph else 4) if timer is not None else 0
    windows_s, host_s = [], 0.0
    for w in range(a.windows):
        barrier()
        if w == 0 and timer is not None:
            ops.set_timer(timer)
        t0 = time.perf_counter()
        for i in range(a.steps):
            if w == 0 and i == timed_steps:
                ops.set_timer(None)
            h0 = time.perf_counter()
            out = tr.step(pcs, gt, z0, al) if (w == 0 and i < timed_steps) else run_step(pcs, gt, z0, al)
            host_s += time.perf_counter() - h0
        barrier()
        windows_s.append(max_over_ranks(time.perf_counter() - t0))
        ops.set_timer(None)
    dt = statistics.median(windows_s)
    loss_ok = bool(torch.isfinite(out["tot_loss"]).item())
    comm = dict(tr.comm)
    single = world == 1 and not a.dp_force and not a.dp_emulate
    # what the timed steps DID with the decoder's gradients (the trainer falls back from the gathered-operand scheme to
    # all-reduce buckets when world x batch > 512 rows or without its side stream: the line reports the scheme that ran)
    dp_scheme = tr.dp_scheme
    decoder_update = "fused wgrad+adam" if (tr.fused_decoder_update and dp_scheme in ("none", "gather")
                                            and tr._side is not None) else "wgrad, adam"
    if a.dp_mode == "gather" and dp_scheme not in ("gather", "none") and "--dp-mode" in sys.argv:
        raise SystemExit(f"bench.py: --dp-mode gather was asked for but the step ran the {dp_scheme!r} scheme "
                         f"(gathered operands need world x batch <= 512 rows, bf16 mode and the fused update)")

    def timed_leg(trainer, batches, steps, warmup):
        """warmup untimed + steps timed trainer steps; ``batches`` yields (pcs, gt).  Same bracket as the main
        region (barrier + synchronize on both sides); max over ranks.  -> seconds."""
        it = iter(batches)
        for _ in range(warmup):
            trainer.step(*next(it), z0, al)
        barrier()
        t_0 = time.perf_counter()
        for _ in range(steps):
            trainer.step(*next(it), z0, al)
        barrier()
        return max_over_ranks(time.perf_counter() - t_0)

    sections = None
    if single and not a.no_extra_legs and not use_graph:
        sections = leg_sections(tr, (pcs, gt, z0, al))

    batcher_leg = None
    if not a.no_batcher_leg and not use_graph and single:
        # datasets.py batch collation inside the timed loop: a packed point-major store of `pool` synthetic crops
        # resident in HBM, every step's batch gathered from it in the DataLoader's shuffled order
        # (DeviceBatcher = pcaa_gather_rows), then the same train step
        from opensetgaitrecognition_pcaa_amd import synthetic as syn
        from opensetgaitrecognition_pcaa_amd.batcher import DeviceBatcher
        pool = 64 * B
        store = syn.synthetic_pcs(pool, T, N, C, seed=4321 + rank).to(dev)
        labels = syn.synthetic_labels(pool, K, seed=4322 + rank).to(dev)
        loader = DeviceBatcher(store, labels, B, shuffle=True)

        def epochs():
            while True:
                yield from loader
        d = timed_leg(tr, epochs(), a.steps, 2)
        loader.check()
        batcher_leg = {"ms_per_step": d / a.steps * 1e3, "value": world * B * a.steps / d,
                       "store": f"{pool} crops [{pool},{T},{N},{C}] fp32 resident in HBM, shuffled epoch order, "
                                "pcaa_gather_rows per batch"}
        del store, loader

    agg = timer.summary() if timer is not None else None
    del tr
    torch.cuda.empty_cache()

    parity_leg = None
    if a.precision == "bf16" and not a.no_parity_mode and not use_graph and single:
        # the SAME workload in the two parity-grade modes -- the modes the 1e-4 / bit-exact label tests run in
        # (tests/test_round2_parity.py::test_config1_full_size_fp32_step_vs_oracle, both parametrisations):
        # "fp16x3": fp32 storage, PointNet products as three bf16 MFMA passes over [hi | lo] operand images;
        # "fp32": exact-fp32 MFMA throughout
        legs = {}
        for prec, psteps in (("fp16x3", max(1, min(a.steps, 10))), ("fp32", max(1, min(a.steps, 5)))):
            F_hip.set_precision(prec)
            trp, _ = build_trainer(a, N, dev, pg, prec)
            d = timed_leg(trp, itertools.repeat((pcs, gt)), psteps, 2)
            legs[prec] = {"precision": prec, "dtype": "f32", "steps": psteps, "warmup": 2,
                          "ms_per_step": d / psteps * 1e3, "value": world * B * psteps / d, "unit": "sequences/s"}
            del trp
            torch.cuda.empty_cache()
        parity_leg = dict(legs["fp16x3"])
        parity_leg["tolerance"] = ("1e-4 rel on losses/embeddings/logits, argmax labels bit-exact, gradients 5e-4 vs the CPU "
                                   "oracle at this size (tests/test_round2_parity.py, precision fp16x3 and fp32)")
        parity_leg["exact_fp32"] = legs["fp32"]
        F_hip.set_precision(a.precision)

    # Data-parallel legs (round 4): the driver runs `bench.py --gpus N` ONCE per N, so that one
    # invocation also times every exchange scheme the trainer has -- (all-reduce | ZeRO-1 reduce-scatter + sharded Adam +
    # all-gather) x (fp32 | bf16 gradient buckets), and SyncBN -- each with the communication time the step could not
    # hide.  ``value`` is the documented default (bf16 mode: gathered operands; parity modes: all-reduce; per-rank BatchNorm).
    dp_legs = None
    if (world > 1 or a.dp_force) and not a.no_extra_legs:
        import copy
        dp_legs = []
        combos = [("allreduce", "bf16", False), ("allreduce", "none", False), ("zero", "bf16", False),
                  ("zero", "none", False), ("allreduce", "bf16", True), ("gather", "bf16", False)]
        if a.precision != "bf16":
            combos = [("allreduce", "none", False), ("zero", "none", False), ("allreduce", "none", True)]
        lsteps = max(1, min(a.steps, 10))
        for mode_, comp, sbn in combos:
            aa = copy.copy(a)
            aa.dp_mode, aa.grad_compress, aa.sync_bn = mode_, comp, sbn
            trl, _ = build_trainer(aa, N, dev, pg, a.precision, fill="device")
            trl.time_comm = True
            for _ in range(3):
                trl.step(pcs, gt, z0, al)
            trl.comm_events.clear()
            barrier()
            t_0 = time.perf_counter()
            for _ in range(lsteps):
                trl.step(pcs, gt, z0, al)
            barrier()
            d = max_over_ranks(time.perf_counter() - t_0)
            torch.cuda.synchronize()
            exposed = sorted(trl.exposed_comm_us())
            exp_us = max_over_ranks(statistics.median(exposed)) if exposed else None
            dp_legs.append({"dp_mode": mode_, "dp_scheme_ran": trl.dp_scheme, "grad_buckets": "bf16" if comp == "bf16" else "fp32", "sync_bn": sbn,
                            "ms_per_step": d / lsteps * 1e3, "value": world * B * lsteps / d, "steps": lsteps,
                            "exposed_comm_us": exp_us,
                            "collectives_per_step": trl.comm["collectives"],
                            "payload_bytes_per_step": trl.comm["payload_bytes"],
                            "is_default": (mode_, comp, sbn) == (a.dp_mode, a.grad_compress, bool(a.sync_bn))})
            del trl
            torch.cuda.empty_cache()

    sweep = infer = c5 = ref_default = dp_emulated = loop = None
    if single and not a.no_extra_legs and a.precision == "bf16":
        esteps = max(1, min(a.steps, 10))
        # the drop-in loop itself (north_star: "keeping ... the train_AAE.py/PCAA_ablation.py training-loop API")
        loop = leg_loop(a, dev, N)
        # BASELINE config[2] (8 x MI355X, global B=512) has never had a node to run on: one rank's program of a world of
        # 2 / 4 / 8 on this GPU, and the projection that follows from it under stated link figures
        dp_emulated = leg_dp_emulated(a, dev, N, steps=esteps)
        # BASELINE config[3]: the point-subsampling sweep (train_pointsubsampling.py:19-71), 10 steps x 3 windows each
        sweep = [leg_train_shape(a, dev, n, C, esteps, 3) for n in (32, 64, 128, 256)]
        # BASELINE config[4]: open-set inference at B=1024 (inference_PCAA.py:382-469)
        infer = leg_infer(a, dev, max(1, min(a.steps, 5)), 3)
        # config[1] with the five features its wording names (x, y, z, doppler, power dB)
        c5 = leg_train_shape(a, dev, N, 5, esteps, 3)
        # the reference's own operating point (constants.py:29,55: BATCH_SIZE = 16, NMAX = 150; what a drop-in user of
        # train_variant4(CONFIG) runs; BASELINE.md section 2, first row), throughput mode and parity-grade mode
        ref_default = {"workload": "the reference's default config: V4 train step at B=16 N=150 C=4 (constants.py:29,55)",
                       "bf16": leg_train_shape(a, dev, 150, 4, esteps, 3, B=16, precision="bf16"),
                       "fp16x3": leg_train_shape(a, dev, 150, 4, max(1, min(a.steps, 5)), 3, B=16, precision="fp16x3"),
                       "parity": "tests/test_round3_parity.py::test_reference_default_shape_step_vs_oracle"}

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = world * B * a.steps / dt
        dp_on = world > 1 or a.dp_force or bool(a.dp_emulate)
        workload = (f"PCAA V4 train step (enc+dec+disc fwd/bwd, WGAN-GP, Chamfer, 2x Adam), "
                    f"B={B}/GPU T={T} N={N} C={C} K={K}, BASELINE config[1]")
        if dp_on:
            # what `value` timed under data parallelism, in the workload string itself
            scheme_words = {"gather": "dp_gather: the wide decoder layers all-gather their weight-gradient operands and every rank "
                                      "forms the global gradient inside the fused wgrad+Adam kernel; encoder / head / critic "
                                      "gradients all-reduced",
                            "allreduce": "all-reduce of every gradient (per-layer decoder buckets, "
                                         + ("bf16" if a.grad_compress == "bf16" else "fp32") + " on the wire)",
                            "zero": "ZeRO-1: decoder gradients reduce-scattered, sharded Adam, parameters all-gathered; the rest all-reduced",
                            "none": "no exchange"}[dp_scheme]
            workload += f"; data parallel over {world if not a.dp_emulate else a.dp_emulate} ranks: {scheme_words}; " + \
                        ("SyncBN" if a.sync_bn else "per-rank BatchNorm statistics")
        if a.dp_emulate:
            workload += (f"; EMULATED: this is ONE rank's program of a {a.dp_emulate}-rank job on one GPU (collectives = device "
                         "operations of the same bytes), value = that one rank's sequences/s, not a multi-GPU measurement")
        line = {
            "metric": "gait sequences/sec (train step)", "value": value, "unit": "sequences/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if a.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": B * world, "precision": a.precision,
                       "parallelism": f"dp{world}", "sync_bn": bool(a.sync_bn), "finite_loss": loss_ok,
                       "hip_graph": bool(use_graph),
                       # the timed region (exactly `steps` steps, barrier + synchronize on both sides, max over ranks)
                       # was run `windows` times back to back: ms_per_step / value are the median window
                       "windows_ms_per_step": [w / a.steps * 1e3 for w in windows_s],
                       "windows": spread([w / a.steps * 1e3 for w in windows_s]),
                       # single process, bf16: the decoder's wide weight gradients are consumed by a fused Adam kernel
                       "decoder_update": decoder_update,
                       # gradient / parameter exchanges of one step: number of collectives, payload bytes, and what a
                       # ring moves per rank and direction for them (2 (w-1)/w x payload for an all-reduce, (w-1)/w for
                       # an all-gather / reduce-scatter); `mode` is the scheme the timed steps RAN, `asked` the flag
                       "dp": {"mode": dp_scheme if dp_on else "none", "asked": a.dp_mode if dp_on else "none",
                              "grad_compress": a.grad_compress if dp_on else "none",
                              "collectives_per_step": comm["collectives"], "payload_bytes_per_step": comm["payload_bytes"],
                              "ring_wire_bytes_per_rank": ((world - 1) / world) * (
                                  2.0 * comm.get("allreduce_bytes", 0) + comm.get("gather_bytes", 0) + comm.get("scatter_bytes", 0))},
                       # time the host spends enqueueing one step (no synchronisation inside step())
                       "host_enqueue_ms_per_step": host_s / (a.steps * a.windows) * 1e3},
        }
        if a.dp_emulate:
            line["emulated"] = True
            line["config"]["emulated_world"] = a.dp_emulate
            line["config"]["parallelism"] = f"one rank of an emulated dp{a.dp_emulate}"
        if dp_legs is not None:
            line["dp_legs"] = {"note": "every exchange scheme of the data-parallel step, timed in this run (3 warm-up + "
                                       "`steps` steps, barrier + synchronize on both sides, max over ranks); exposed_comm_us = "
                                       "median over the steps (max over ranks) of the time the main stream spends waiting for "
                                       "exchanges: from the point where it needs the encoder gradients reduced to the point "
                                       "where every decoder bucket is back, + (ZeRO) the waits for the all-gathers of the "
                                       "updated decoder shards, + (SyncBN) every synchronous statistics all-reduce (HIP "
                                       "events, train.PCAATrainer.time_comm / exposed_comm_us); the line's `value` is the "
                                       "leg marked is_default"
                                       + ("; world = 1 here: these are FORCED 1-rank collectives (their fixed cost on one GPU), "
                                          "not a measurement of any exchange scheme between GPUs" if world == 1 else ""),
                               "legs": dp_legs}
            # north_star names the RCCL all-reduce of gradients: that scheme's throughput beside `value` at top level
            ar = [l for l in dp_legs if l["dp_mode"] == "allreduce" and not l["sync_bn"]
                  and l["grad_buckets"] == ("bf16" if a.grad_compress == "bf16" else "fp32")]
            if ar:
                line["value_allreduce"] = ar[0]["value"]
                line["ms_per_step_allreduce"] = ar[0]["ms_per_step"]
        if batcher_leg is not None:
            line["with_batcher"] = batcher_leg
        if sweep is not None:
            line["sweep"] = {"workload": f"BASELINE config[3]: V4 train step at N in (32,64,128,256), B={B} C={C} bf16; frac = "
                                         "whole-step algorithmic FLOPs / bytes (SURVEY 8d) over the step time against the MFMA / "
                                         "HBM peak, the larger of the two", "points": sweep}
        if infer is not None:
            line["infer"] = infer
        if c5 is not None:
            c5["workload"] = f"BASELINE config[1] at C=5 (x, y, z, doppler, power): B={B} N={N} bf16"
            line["c5"] = c5
        if ref_default is not None:
            line["ref_default"] = ref_default
        if dp_emulated is not None:
            line["dp_emulated"] = dp_emulated
        if loop is not None:
            loop["vs_value"] = loop["value"] / value
            line["loop"] = loop
        if sections is not None:
            line["gpu_sections"] = sections
        if agg:
            dom = max(agg.items(), key=lambda kv: kv[1]["ms"])
            name, r = dom
            peak = PEAK_BF16_TFLOPS if name.startswith("gemm_bf16") else PEAK_F32_TFLOPS
            achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
            # HBM bytes per launch of that kernel from the rocprofv3 PMC passes (FETCH_SIZE doubled
            # per the gfx950 correction, + WRITE_SIZE), committed as profiles/rNN_pmc_summary.json (the newest);
            # only valid for the workload it was collected on
            traffic = None
            try:
                import glob
                with open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))[-1]) as f:
                    pmc = json.load(f)
                if pmc.get("workload") == f"B={B} T={T} N={N} C={C} K={K} {a.precision}" and name in pmc["kernels"]:
                    traffic = pmc["kernels"][name]["hbm_bytes_per_launch"]
            except (OSError, ValueError, KeyError, IndexError):
                pass
            line["roofline"] = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": peak,
                                "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                                "traffic_unit": "HBM bytes per launch (PMC, from the committed rocprofv3 passes)",
                                "algorithmic_flop_per_launch": r["flops"] / r["launches"],
                                "algorithmic_bytes_per_launch": r["bytes"] / r["launches"],
                                "timed_steps": timed_steps,
                                "launches_per_step": r["launches"] / timed_steps,
                                "avg_launch_ms": r["ms"] / r["launches"],
                                "kernel_ms_per_step": r["ms"] / timed_steps,
                                "other_timed": {k: {"ms_per_step": v["ms"] / timed_steps,
                                                    "achieved": v["flops"] / (v["ms"] * 1e-3) / 1e12}
                                                for k, v in agg.items() if k != name}}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(B, N, C, K, T)
            if sections is not None:
                # the two breakdowns side by side, per phase: CPU seconds (at the sample's batch) and GPU microseconds
                cb = line["cpu_baseline"]
                cb["phases_vs_gpu"] = {k: {"cpu_s": cb["phases_s"].get(k), "gpu_us": sections["phases_us"].get(k)}
                                       for k in cb["phases_s"]}
        # order of the line's tail (a log reader that keeps only the last few KB of stdout must still see them): the CPU
        # baseline, then the dominant kernel's roofline, then the parity-grade legs -- the long legs come before
        if "roofline" in line:
            line["roofline"] = line.pop("roofline")
        if parity_leg is not None:
            line["parity_mode"] = parity_leg
        print(json.dumps(line), flush=True)
    if pg is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
