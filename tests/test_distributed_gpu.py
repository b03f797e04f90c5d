"""Data-parallel train step on the GPU: two ranks (two processes sharing cuda:0, gloo transport so
that it runs on a one-GPU box; the trainer's collectives are backend-agnostic torch.distributed
calls -- RCCL on a real node) with SyncBN must reproduce the reference's SINGLE-process step on the
global batch: the committed golden trajectory v4_B6_N32_C4_K4 (generated from the reference,
tests/golden/make_golden.py), each rank holding 3 of its 6 sequences."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, precision, zero=False, compress=None, n_override=None, gather=False):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch.distributed as dist
        from helpers import T, load_golden
        from opensetgaitrecognition_pcaa_amd import constants, dist as pdist, synthetic as syn
        from opensetgaitrecognition_pcaa_amd.train import PCAATrainer

        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        g, m = load_golden("v4_B6_N32_C4_K4")
        B, N, C, K, steps = m["B"], m["N"], m["C"], m["K"], m["steps"]
        if n_override is not None:
            N = n_override            # no golden at this width: the caller compares two modes of the same run
        constants.NFEATURES = C
        cfg = dict(constants.CONFIG)
        cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B // world, LR=1e-4, B1=0.9, B2=0.99,
                   GP_WEIGHT=15, ADV_WEIGHT=1, SUP_LATENT_DIM=32)
        tr = PCAATrainer(cfg, device="cuda:0", precision=precision, process_group=dist.group.WORLD, sync_bn=True,
                         dp_zero=zero, grad_compress=compress, dp_gather=gather)
        for mod, seed in zip((tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head,
                              tr.discriminator_projection_head), m["fill_seeds"]):
            syn.deterministic_fill_(mod, seed)
        tr.set_prior_means(torch.from_numpy(g["means"]))
        tr.finalize()
        tr.train()
        keys = ("d_loss", "gp", "rec_loss", "loss_g", "sup_loss", "tot_loss")
        out_rec = {"losses": [], "preds": []}
        for s in range(steps):
            # inputs drawn for the GLOBAL batch, sliced per rank (dist.shard_rows)
            pcs = pdist.shard_rows(syn.synthetic_pcs(B, T, N, C, seed=m["pcs_seed0"] + s), rank, world)
            gt = pdist.shard_rows(syn.synthetic_labels(B, K, seed=m["gt_seed0"] + s), rank, world)
            z0 = pdist.shard_rows(syn.synthetic_z0(B, 32, seed=m["z0_seed0"] + s), rank, world)
            al = pdist.shard_rows(syn.synthetic_alphas(B, seed=m["alpha_seed0"] + s), rank, world)
            out = tr.step(pcs.contiguous().cuda().permute(0, 3, 1, 2), gt.cuda(), z0.contiguous().cuda(),
                          al.contiguous().cuda())
            # every loss is a batch mean: the global-batch value is the mean over the equal-size shards
            lv = torch.stack([out[k].detach().double().reshape(()) for k in keys]).cpu()
            dist.all_reduce(lv)
            out_rec["losses"].append((lv / world).numpy())
            out_rec["preds"].append(out["preds"].cpu().numpy())
        torch.cuda.synchronize()
        # replicas must hold identical parameters after the steps
        flat = tr.flat_g.p.detach().cpu()
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        out_rec["replicas_equal"] = bool(torch.equal(flat, ref))
        out_rec["comm"] = dict(tr.comm)
        out_rec["g16_direct"] = len(tr._g16_direct)
        out_rec["gathered_layers"] = sorted(k[0] for k in tr._gather_bufs)
        out_rec["dec_tail"] = tr.flat_g.p[-(1 << 16):].detach().cpu().numpy()        # the end of dense5's weight
        if rank == 0:
            out_rec["params"] = {f"{nm}.{name}": v.detach().cpu().numpy() for nm, mod in tr.modules().items()
                                 for name, v in mod.state_dict().items() if v.dtype.is_floating_point
                                 and v.numel() <= 70000}
        q.put((rank, out_rec, None))
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put((rank, None, traceback.format_exc() + repr(e)))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("precision,zero,compress", [("fp32", False, None), ("fp32", True, None), ("bf16", False, None),
                                                     ("fp32", False, "bf16"), ("bf16", False, "bf16")])
def test_two_rank_syncbn_step_equals_global_batch_golden(precision, zero, compress):
    """zero=True: the sharded decoder optimizer (reduce-scatter, Adam on this rank's slice, all-gather).
    precision="bf16": the throughput mode under data parallelism, at its stated tolerance.  compress="bf16": the
    decoder gradient buckets cross the wire as bf16 -- step 0 is untouched (losses come before the exchange), the
    trajectory and the parameters stay within the bf16 tolerance, replicas stay bit-identical."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import is_pre_bn_bias, load_golden
    g, m = load_golden("v4_B6_N32_C4_K4")
    B, steps, world = m["B"], m["steps"], 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, precision, zero, compress)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, rec, err = q.get(timeout=240)
        assert err is None, f"rank {rank}: {err}"
        results[rank] = rec
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exact = precision == "fp32" and compress is None
    for s in range(steps):
        tol = (1e-4 if s == 0 else 5e-4 * s) if exact else (1e-4 if (s == 0 and precision == "fp32") else 2e-2)
        for r in range(world):
            assert np.allclose(results[r]["losses"][s], g[f"s{s}.losses"], rtol=tol, atol=1e-5 if exact else 2e-2), \
                (s, r, results[r]["losses"][s], g[f"s{s}.losses"])
        preds = np.concatenate([results[r]["preds"][s] for r in range(world)])
        if exact or (s == 0 and precision == "fp32"):
            assert np.array_equal(preds, g[f"s{s}.preds"]), "argmax labels of the sharded step must match the global batch"
    assert all(results[r]["replicas_equal"] for r in range(world))
    comm = results[0]["comm"]
    assert comm["collectives"] >= 3 and comm["payload_bytes"] > 0
    if compress == "bf16":
        # the decoder region (98 % of optimizer_G's floats) went out at 2 bytes per element
        assert comm["payload_bytes"] < 0.7 * 4 * 12_300_000      # fp32 everywhere would be ~49 MB (9.8 M decoder + 2.4 M encoder floats)
    if not exact:
        return
    # parameters after the last step against the reference's (small tensors in full; same gates as the
    # single-process golden test)
    s = steps - 1
    checked = 0
    for key, v in results[0]["params"].items():
        nm, name = key.split(".", 1)
        gk = f"s{s}.param.{nm}.{name}::full"
        if gk not in g.files or is_pre_bn_bias(name) or name.endswith("running_mean"):
            continue
        err = np.abs(v.astype(np.float64) - g[gk].astype(np.float64))
        scale = float(np.abs(g[gk]).max())
        assert err.max() <= 5e-5 * scale + 0.5e-4 * (s + 1), (key, err.max())
        assert err.mean() <= 2e-6 * max(scale, 1.0), (key, err.mean())
        checked += 1
    assert checked >= 20


def _run_two_ranks(precision, compress, n_override, gather=False):
    ctx = mp.get_context("spawn")
    port, q, world = _free_port(), ctx.Queue(), 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, precision, False, compress, n_override, gather))
             for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, rec, err = q.get(timeout=240)
        assert err is None, f"rank {rank}: {err}"
        results[rank] = rec
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.timeout(600)
def test_bf16_buckets_small_decoder_match_uncompressed():
    """bench.py's data-parallel default (precision bf16 + bf16 gradient buckets) on a decoder whose second layer is
    below the compression threshold (NMAX=16: dense2 is 240x120): that layer's gradient must cross the wire as fp32
    from the fp32 buffer -- it was once left to the bf16-direct image while the all-reduce took the fp32 path over a
    never-written range (round-2 advisor finding).  Compared with the same run without compression."""
    a = _run_two_ranks("bf16", "bf16", 16)
    b = _run_two_ranks("bf16", None, 16)
    assert all(a[r]["replicas_equal"] and b[r]["replicas_equal"] for r in range(2))
    assert a[0]["g16_direct"] >= 1, "the wide layers still use the bf16-direct wire image"
    for s in range(len(a[0]["losses"])):
        assert np.allclose(a[0]["losses"][s], b[0]["losses"][s], rtol=2e-2, atol=2e-2), (s, a[0]["losses"][s], b[0]["losses"][s])
    moved = 0
    for key, v in a[0]["params"].items():
        w = b[0]["params"][key]
        if "running_" in key:
            continue            # BatchNorm running statistics follow the activations (bf16 noise), not an optimizer step
        # three Adam steps of lr 1e-4: a parameter updated from a garbage / zero gradient differs by ~3e-4 on most
        # elements; bf16 rounding of the reduced bucket flips the sign of the update on few
        assert np.abs(v - w).mean() <= 5e-5, (key, np.abs(v - w).mean())
        moved += 1
    assert moved >= 20


@pytest.mark.timeout(600)
def test_gathered_operands_scheme_matches_the_gradient_all_reduce():
    """Round 5, dp_gather: the wide decoder layers all-gather dz [B, out] and x [B, in] instead of all-reducing the
    gradient, and every rank forms the global gradient inside the fused weight-gradient + Adam kernel
    (pcaa_skinny_linear_wgrad_adam_rows).  Same step as the all-reduce scheme (sum over ranks of dz_r^T x_r = the
    stacked-rows product): two ranks, three steps from the golden's state in the bf16 mode, against the fp32-bucket
    all-reduce run of the same mode -- losses, parameters (Adam's +-lr noise gate), replicas bit-identical, and the wire
    carries an order of magnitude less."""
    a = _run_two_ranks("bf16", None, None, gather=True)
    b = _run_two_ranks("bf16", None, None)
    assert all(a[r]["replicas_equal"] and b[r]["replicas_equal"] for r in range(2))
    assert a[0]["gathered_layers"] == [2, 3, 4, 5] and b[0]["gathered_layers"] == []
    for s in range(len(a[0]["losses"])):
        assert np.allclose(a[0]["losses"][s], b[0]["losses"][s], rtol=2e-2, atol=2e-2), (s, a[0]["losses"][s], b[0]["losses"][s])
    # three Adam steps of lr 1e-4: a layer updated from a wrong / missing gradient differs by ~3e-4 on most elements
    d5 = np.abs(a[0]["dec_tail"] - b[0]["dec_tail"])
    assert d5.mean() <= 3e-5 and d5.max() <= 6.5e-4, (d5.mean(), d5.max())
    assert np.array_equal(a[0]["dec_tail"], a[1]["dec_tail"]), "every rank formed the same global gradient"
    for key, v in a[0]["params"].items():
        if "running_" in key:
            continue
        assert np.abs(v - b[0]["params"][key]).mean() <= 5e-5, (key, np.abs(v - b[0]["params"][key]).mean())
    # fp32 buckets move ~4 B x 12.3 M floats; gathered operands 2 ranks x 3 rows x the layer widths
    assert a[0]["comm"]["payload_bytes"] < 0.3 * b[0]["comm"]["payload_bytes"], (a[0]["comm"], b[0]["comm"])


def _loop_worker(rank, world, port, q, workdir):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, ROOT)
        os.chdir(workdir)
        import torch.distributed as dist
        from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip
        from opensetgaitrecognition_pcaa_amd.datasets import SyntheticGaitDataset
        from opensetgaitrecognition_pcaa_amd.train import train_variant4
        torch.cuda.set_device(0)
        pg = None
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            pg = dist.group.WORLD
        constants.NFEATURES = 4
        F_hip.set_precision("fp32")
        cfg = dict(constants.CONFIG)
        cfg.update(MODEL_NAME=f"dp{world}", TRAIN_CLASSES=[0, 1, 2, 3], NMAX=16, BATCH_SIZE=8, EPOCHS=2,
                   CHECKPOINT_FREQUENCY=1, NOTES="")
        # the loop's host RNG: every rank seeds DIFFERENTLY, as under a plain torchrun launch of the unseeded
        # reference loop -- rank 0's initial weights, epoch order and z0 / alphas draws are the ones used
        # (PCAATrainer.sync_replicas, the batcher's and the loop's broadcasts)
        np.random.seed(5 + 1000 * rank); torch.manual_seed(5 + 1000 * rank)
        make = lambda split: SyntheticGaitDataset(48 if split.value == "train" else 16, 4, N=16, C=4, seed=11)
        trainer, hist = train_variant4(cfg, wandb_mode="disabled", dataset_factory=make, process_group=pg,
                                       sync_bn=world > 1, device="cuda:0")
        torch.cuda.synchronize()
        state = torch.cat([t.detach().double().reshape(-1).cpu() for m in trainer.modules().values()
                           for t in list(m.parameters()) + list(m.buffers())])
        q.put((rank, {"hist": hist, "p": trainer.flat_g.p.detach().cpu()[:4096].numpy(),
                      "state_sum": float(state.sum()), "state_l2": float(state.norm()),
                      "files": sorted(os.listdir(f"models/dp{world}"))}, None))
        if world > 1:
            dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, None, traceback.format_exc() + repr(e)))


@pytest.mark.timeout(300)
def test_data_parallel_loop_equals_single_process_loop(tmp_path):
    """train_variant4 on 2 ranks (global BATCH_SIZE 8 = 4 per rank, SyncBN, rank 0's epoch order / z0 / alphas
    broadcast) must log the same epoch records as the single-process loop at BATCH_SIZE 8 from the same seeds."""
    ctx = mp.get_context("spawn")
    out = {}
    for world in (1, 2):
        port = _free_port()
        q = ctx.Queue()
        procs = [ctx.Process(target=_loop_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
        for p in procs:
            p.start()
        res = {}
        for _ in range(world):
            rank, rec, err = q.get(timeout=240)
            assert err is None, f"world {world} rank {rank}: {err}"
            res[rank] = rec
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        out[world] = res
    h1, h2 = out[1][0]["hist"], out[2][0]["hist"]
    assert out[2][0]["hist"] == out[2][1]["hist"], "both ranks log the same (global) records"
    assert np.array_equal(out[2][0]["p"], out[2][1]["p"]), "replicas hold identical parameters"
    # ... every parameter AND buffer of every module (BatchNorm running statistics, the inert heads), although the
    # two ranks drew different initial weights
    assert out[2][0]["state_sum"] == out[2][1]["state_sum"] and out[2][0]["state_l2"] == out[2][1]["state_l2"]
    assert len(h1) == len(h2) == 2
    for e in range(2):
        for k in h1[e]:
            tol = 2e-3 if "Accuracy" not in k else 0.13          # one flipped argmax in 8/16 samples at most
            assert abs(h1[e][k] - h2[e][k]) <= tol * max(abs(h1[e][k]), 1.0), (e, k, h1[e][k], h2[e][k])
    assert "dp2_E.pt" in out[2][0]["files"] and "config.pkl" in out[2][0]["files"]


# ---------------------------------------------------------------------------------------------------------
# the data-parallel step AT THE BENCHMARKED SHAPE (B=64 per rank, N=128): per-layer bucket hooks at full size,
# the bf16-direct weight-gradient buckets, 16 statistics replicas under SyncBN -- against the CPU oracle's
# single-process step on the 128-sequence global batch
# ---------------------------------------------------------------------------------------------------------
SHAPE = dict(B=64, N=128, C=4, K=8, seeds=[0, 1, 2, 3, 4])
_SHAPE_KEYS = ("d_loss", "gp", "rec_loss", "loss_g", "sup_loss", "tot_loss")
_SHAPE_GRADS = ("E.pc_block.pointnet2.module.0.weight", "E.pc_block.pointnet4.module.0.weight",
                "E.tc_block.dtc3.conv1d.weight", "E.MLP_sup1.0.weight", "GPH.0.weight", "G.dense1.weight")


def _shape_cfg(B):
    from opensetgaitrecognition_pcaa_amd import constants
    cfg = dict(constants.CONFIG)
    cfg.update(NMAX=SHAPE["N"], TRAIN_CLASSES=list(range(SHAPE["K"])), BATCH_SIZE=B, LR=1e-4, B1=0.9, B2=0.99,
               GP_WEIGHT=15, ADV_WEIGHT=1, SUP_LATENT_DIM=32)
    return cfg


def _shape_inputs(world):
    from opensetgaitrecognition_pcaa_amd import constants, synthetic as syn
    Bg, N, C, K = SHAPE["B"] * world, SHAPE["N"], SHAPE["C"], SHAPE["K"]
    return (syn.synthetic_pcs(Bg, constants.NSTEPS, N, C, seed=1234), syn.synthetic_labels(Bg, K, seed=1235),
            syn.synthetic_z0(Bg, 32, seed=1236), syn.synthetic_alphas(Bg, seed=1237))


def _shape_worker(rank, world, port, q, precision, compress, means_np):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        from opensetgaitrecognition_pcaa_amd import constants, dist as pdist, synthetic as syn
        from opensetgaitrecognition_pcaa_amd.train import PCAATrainer
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        constants.NFEATURES = SHAPE["C"]
        tr = PCAATrainer(_shape_cfg(SHAPE["B"]), device="cuda:0", precision=precision, process_group=dist.group.WORLD,
                         sync_bn=True, grad_compress=compress)
        for mod, seed in zip((tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head,
                              tr.discriminator_projection_head), SHAPE["seeds"]):
            syn.deterministic_fill_(mod, seed)
        tr.set_prior_means(torch.from_numpy(means_np))
        tr.finalize()
        tr.train()
        pcs, gt, z0, al = (pdist.shard_rows(t, rank, world).contiguous() for t in _shape_inputs(world))
        out = tr.step(pcs.cuda().permute(0, 3, 1, 2), gt.cuda(), z0.cuda(), al.cuda())
        torch.cuda.synchronize()
        lv = torch.stack([out[k].detach().double().reshape(()) for k in _SHAPE_KEYS]).cpu()
        dist.all_reduce(lv)
        flat = tr.flat_g.p.detach().cpu()
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        rec = {"losses": (lv / world).numpy(), "preds": out["preds"].cpu().numpy(),
               "sup_fvs": out["sup_fvs"].cpu().numpy(), "replicas_equal": bool(torch.equal(flat, ref)),
               "comm": dict(tr.comm), "g16_direct": len(tr._g16_direct)}
        if rank == 0:
            # the reduced gradients are sums over the ranks of per-shard means: / world = the global-batch mean
            rec["grads"] = {n: (tr.flat_g.grad_views[n].detach().cpu() / world).numpy() for n in _SHAPE_GRADS}
            rec["params"] = {n: tr.flat_g.params[tr.flat_g.names.index(n)].detach().cpu().numpy()
                             for n in ("E.MLP_sup1.0.weight", "E.pc_block.pointnet2.module.0.weight", "GPH.0.weight")}
            w5 = tr.decoder.dense5.weight.detach()
            rec["dense5_rows"] = w5[:: w5.shape[0] // 16][:16].cpu().numpy()
        q.put((rank, rec, None))
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, None, traceback.format_exc() + repr(e)))


_SHAPE_ORACLE = {}


def _shape_oracle(world):
    """The single-process step on the 128-sequence global batch.  Round 5: from the full-size golden of the REFERENCE
    (tests/golden/full_B128_N128.npz, make_golden_fullsize.py: bench.py's fills and input seeds) -- through round 4 the
    CPU oracle computed it here, 50 s of host time per run of the suite.  Large tensors come as (l2, 1 024 strided
    samples); ``sampled(t)`` reads a tensor the same way."""
    if world in _SHAPE_ORACLE:
        return _SHAPE_ORACLE[world]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import full_golden
    g, m = full_golden(SHAPE["B"] * world, SHAPE["N"])
    assert m["fill_seeds"] == SHAPE["seeds"] and (m["pcs_seed"], m["gt_seed"], m["z0_seed"], m["alpha_seed"]) == (1234, 1235, 1236, 1237)

    def rec(prefix, name):
        if f"{prefix}{name}::full" in g.files:
            return ("full", g[f"{prefix}{name}::full"])
        return ("cs", float(g[f"{prefix}{name}::l2"]), g[f"{prefix}{name}::samples"])
    keep = {"losses": g["losses"], "preds": g["preds"], "sup_fvs": g["sup_fvs"], "out_labels": g["out_labels"],
            "grads": {n: rec("ggrad.", n) for n in _SHAPE_GRADS},
            "params": {n: rec("param.", n) for n in ("E.MLP_sup1.0.weight", "E.pc_block.pointnet2.module.0.weight", "GPH.0.weight")},
            "dense5_rows": g["param.dense5_rows"], "means": g["means"], "nsample": int(m["nsample"])}
    _SHAPE_ORACLE[world] = keep
    return keep


def _sampled(a, nsample):
    """the strided samples synthetic.checksum takes of a tensor (numpy in, numpy out)"""
    from opensetgaitrecognition_pcaa_amd import synthetic as syn
    return syn.checksum(torch.from_numpy(np.ascontiguousarray(a)), nsample)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("precision,compress", [("fp32", None), ("bf16", "bf16")])
def test_two_rank_step_at_bench_shape_vs_oracle(precision, compress):
    """B=64 per rank, N=128, SyncBN, 2 ranks (gloo; both on the one GPU): the step bench.py runs per rank at N GPUs,
    against the oracle's single-process step on the 128-sequence global batch.  fp32 mode: 1e-4 on losses and
    embeddings, labels bit-exact, gradients 5e-4; ("bf16", "bf16") is bench.py's data-parallel default (bf16
    PointNet, decoder buckets as bf16 with the bf16-direct weight-gradient kernels) at the bf16 tolerance."""
    world = 2
    ref = _shape_oracle(world)
    ctx = mp.get_context("spawn")
    port, q = _free_port(), ctx.Queue()
    procs = [ctx.Process(target=_shape_worker, args=(r, world, port, q, precision, compress, ref["means"]))
             for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, rec, err = q.get(timeout=900)
        assert err is None, f"rank {rank}: {err}"
        res[rank] = rec
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exact = precision == "fp32"
    assert all(res[r]["replicas_equal"] for r in range(world))
    ltol = 1e-4 if exact else 2e-2
    assert np.allclose(res[0]["losses"], ref["losses"], rtol=ltol, atol=1e-5 if exact else 2e-2), (res[0]["losses"], ref["losses"])
    preds = np.concatenate([res[r]["preds"] for r in range(world)])
    fvs = np.concatenate([res[r]["sup_fvs"] for r in range(world)])
    scale = np.abs(ref["sup_fvs"]).max()
    if exact:
        top2 = np.sort(ref["out_labels"], axis=1)[:, -2:]
        tied = (top2[:, 1] - top2[:, 0]) <= 1e-4 * np.abs(ref["out_labels"]).max()
        assert np.array_equal(preds[~tied], ref["preds"][~tied]), "argmax labels must be bit-exact"
        assert np.abs(fvs - ref["sup_fvs"]).max() <= 1e-4 * scale
    else:
        assert np.abs(fvs - ref["sup_fvs"]).max() <= 5e-2 * scale
        assert (preds == ref["preds"]).mean() >= 0.9
        assert res[0]["g16_direct"] >= 3, "the wide decoder layers must have used the bf16-direct wire image"
    gtol = 5e-4 if exact else 5e-2
    ns = ref["nsample"]
    for n in _SHAPE_GRADS:
        if compress == "bf16" and n.startswith("G."):
            continue        # compressed buckets are consumed by Adam as the reduced bf16 image: the fp32 range stays local
        a, r = res[0]["grads"][n].astype(np.float64), ref["grads"][n]
        if r[0] == "full":
            b = r[1].astype(np.float64)
            rel = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
            assert rel <= gtol, (n, rel)
        else:               # a large tensor: its l2 and 1 024 strided samples
            cs = _sampled(a, ns)
            assert abs(cs["l2"] - r[1]) <= gtol * r[1], (n, cs["l2"], r[1])
            floor = r[1] / np.sqrt(a.size)
            assert np.abs(cs["samples"] - r[2]).max() <= 4 * gtol * max(np.abs(r[2]).max(), floor), n
    # post-Adam parameters (first Adam step = +-lr per element: sign flips of rounding-noise gradients are rare)
    for n, r in list(ref["params"].items()) + [("dense5_rows", ("full", ref["dense5_rows"]))]:
        a = res[0]["dense5_rows"][:, ::16] if n == "dense5_rows" else res[0]["params"][n]
        b = r[1] if r[0] == "full" else r[2]
        if r[0] != "full":
            a = _sampled(a, ns)["samples"]
        err = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
        assert err.max() <= 5e-5 * np.abs(b).max() + 2.0e-4 * 1.001, (n, err.max())
        assert err.mean() <= (2e-6 if exact else 3e-5), (n, err.mean())
    comm = res[0]["comm"]
    assert comm["collectives"] >= 7
    if compress == "bf16":
        assert comm["payload_bytes"] < 0.6 * 4 * 159_300_000


@pytest.mark.timeout(900)
def test_bench_multi_rank_branch_runs_on_two_gloo_ranks():
    """bench.py's N>1 branch (torchrun environment, barrier, max-over-ranks windows, one JSON line on rank 0, bf16
    gradient buckets) launched the way the driver launches it, with 2 ranks sharing this box's GPU over gloo."""
    import json
    import subprocess
    env = dict(os.environ, PCAA_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--windows", "2", "--backend", "gloo", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["config"]["global_batch"] == 128
    assert d["config"]["finite_loss"] and d["config"]["parallelism"] == "dp2"
    assert d["config"]["dp"]["collectives_per_step"] >= 7 and d["config"]["dp"]["grad_compress"] == "bf16"
    assert d["config"]["dp"]["mode"] == "gather" and d["config"]["decoder_update"] == "fused wgrad+adam"
    assert len(d["config"]["windows_ms_per_step"]) == 2 and d["value"] > 0
    assert "sweep" not in d and "cpu_baseline" not in d          # single-GPU legs stay out of the N>1 line
    # round 4: the one N>1 invocation times every exchange scheme, each with its exposed communication time
    legs = d["dp_legs"]["legs"]
    kinds = {(l["dp_mode"], l["grad_buckets"], l["sync_bn"]) for l in legs}
    assert kinds == {("allreduce", "bf16", False), ("allreduce", "fp32", False), ("zero", "bf16", False),
                     ("zero", "fp32", False), ("allreduce", "bf16", True), ("gather", "bf16", False)}
    assert sum(l["is_default"] for l in legs) == 1
    for l in legs:
        assert l["ms_per_step"] > 0 and l["exposed_comm_us"] is not None and l["exposed_comm_us"] >= 0, l
        assert l["collectives_per_step"] >= 7, l
    by = {(l["dp_mode"], l["grad_buckets"], l["sync_bn"]): l for l in legs}
    assert by[("allreduce", "bf16", False)]["payload_bytes_per_step"] < 0.6 * by[("allreduce", "fp32", False)]["payload_bytes_per_step"]
    assert by[("zero", "bf16", False)]["payload_bytes_per_step"] < 0.8 * by[("zero", "fp32", False)]["payload_bytes_per_step"]
    assert by[("allreduce", "bf16", True)]["collectives_per_step"] > by[("allreduce", "bf16", False)]["collectives_per_step"]
    # round 5: gathered operands instead of gradients -- 2 ranks x 64 rows x the layer widths against 157 M gradients
    assert by[("gather", "bf16", False)]["payload_bytes_per_step"] < 0.1 * by[("allreduce", "bf16", False)]["payload_bytes_per_step"]
    assert by[("gather", "bf16", False)]["is_default"], "bf16 mode: the gathered-operands scheme is the line's value"
    # round 6: what `value` timed is said in the workload string itself, the scheme the steps RAN is
    # reported (not the flag), and the all-reduce scheme's throughput -- the exchange north_star names -- stands at top level
    assert "dp_gather" in d["config"]["workload"] and "per-rank BatchNorm" in d["config"]["workload"]
    assert d["config"]["dp"]["asked"] == "gather" and all(l["dp_scheme_ran"] == l["dp_mode"] for l in legs)
    assert d["value_allreduce"] == by[("allreduce", "bf16", False)]["value"] and d["ms_per_step_allreduce"] > 0
    assert "emulated" not in d


@pytest.mark.timeout(900)
def test_bench_dp_emulate_line_is_labelled_and_never_claims_gpus():
    """bench.py --dp-emulate 8: one rank's program of an 8-rank job on this GPU.  The line says so everywhere a reader
    could take it for a multi-GPU number: `emulated`, n_gpus 1, the workload string, value = ONE rank's sequences/s."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--dp-emulate", "8", "--steps", "3", "--warmup", "2", "--windows", "2",
           "--no-cpu-baseline", "--no-extra-legs", "--no-parity-mode", "--no-batcher-leg"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["emulated"] is True and d["n_gpus"] == 1 and d["config"]["emulated_world"] == 8
    assert "EMULATED" in d["config"]["workload"] and "not a multi-GPU measurement" in d["config"]["workload"]
    assert d["config"]["dp"]["mode"] == "gather" and d["config"]["decoder_update"] == "fused wgrad+adam"
    assert abs(d["value"] - 64 / d["ms_per_step"] * 1e3) <= 1e-6 * d["value"], "one rank's sequences, not 8 x"
    # 4 packed all-gathers (one per wide layer) + critic, encoder, decoder-rest all-reduces
    assert d["config"]["dp"]["collectives_per_step"] >= 7
    # a multi-rank launch refuses the flag
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dp-emulate", "8", "--dp-force", "--steps", "1"],
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "--dp-emulate" in bad.stderr


@pytest.mark.timeout(900)
def test_eight_rank_bf16_ring_sum_at_config1():
    """The 8-GPU run's bf16 gradient buckets at BASELINE config[1]'s size: eight "ranks" = eight
    different B=64 batches through the bf16-mode HIP step from the SAME state (LR = 0: Adam leaves the parameters where
    they are), the decoder's 157 M real gradients of each kept; per all-reduce bucket (one per decoder layer, as
    train.PCAATrainer issues them) the bf16 ring sum -- per-chunk hop order, one rounding per hop,
    helpers.ring_allreduce_bf16 -- against the fp64 sum of the fp32 gradients.  Gate: 1e-2 relative l2 (the bf16 mode's
    weight-gradient tolerance is 5e-2) and 2 % of the bucket's largest element; the measured figures are printed and
    recorded in docs/LAB_LOG.md section 6 -- they decide whether bf16 buckets stay the default at N >= 4."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import T, ring_allreduce_bf16
    from opensetgaitrecognition_pcaa_amd import constants, synthetic as syn
    from opensetgaitrecognition_pcaa_amd.train import PCAATrainer
    from oracle import pcaa_oracle as O
    DEV = "cuda:0"
    W, B, N, C, K = 8, 64, 128, 4, 8
    constants.NFEATURES = C
    cfg = dict(constants.CONFIG)
    cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B, LR=0.0)
    tr = PCAATrainer(cfg, precision="bf16", fused_decoder_update=False)
    for i, m in enumerate((tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head,
                           tr.discriminator_projection_head)):
        syn.deterministic_fill_(m, i)
    tr.set_prior_means(O.sample_distant_points(32, K, 10, 10).float())
    tr.finalize()
    tr.train()
    p0 = tr.flat_g.p.clone()
    fg = tr.flat_g
    grads = []
    for r in range(W):
        tr.step(syn.synthetic_pcs(B, T, N, C, seed=1234 + r).to(DEV).permute(0, 3, 1, 2),
                syn.synthetic_labels(B, K, seed=1235 + r).to(DEV), syn.synthetic_z0(B, 32, seed=1236 + r).to(DEV),
                syn.synthetic_alphas(B, seed=1237 + r).to(DEV))
        torch.cuda.synchronize()
        grads.append(fg.g[tr._dec_start:].clone())
    assert torch.equal(tr.flat_g.p, p0), "LR = 0 must leave the replica's parameters untouched"
    report = []
    for layer in range(2, 6):
        lo = fg.offsets[fg.names.index(f"G.dense{layer}.weight")] - tr._dec_start
        nxt = f"G.dense{layer + 1}.weight"
        hi = (fg.offsets[fg.names.index(nxt)] if nxt in fg.names else fg.total) - tr._dec_start
        gs = [g[lo:hi] for g in grads]
        exact = torch.zeros(hi - lo, dtype=torch.float64, device=DEV)
        for g in gs:
            exact += g.double()
        ring = ring_allreduce_bf16(gs).double()
        rel = float((ring - exact).norm() / exact.norm())
        mx = float((ring - exact).abs().max() / exact.abs().max())
        # one rank's own bf16 rounding, for scale (what a 1-rank "sum" already costs)
        one = float((gs[0].bfloat16().double() - gs[0].double()).norm() / gs[0].double().norm())
        report.append((layer, hi - lo, rel, mx, one))
        assert rel <= 1e-2, (layer, rel)
        assert mx <= 2e-2, (layer, mx)
        del exact, ring
    for layer, n, rel, mx, one in report:
        print(f"8-rank bf16 ring sum, decoder layer {layer} bucket ({n / 1e6:.1f} M elements): rel-l2 {rel:.2e}, max "
              f"elementwise {mx:.2e} of the largest; a single rank's bf16 rounding alone {one:.2e}")
