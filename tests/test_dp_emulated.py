"""Round 6: ONE RANK'S PROGRAM of a world of W ranks on one GPU (PCAATrainer ``emulate_world`` /
``dist.EmulatedExchange``; bench.py ``--dp-emulate`` and the ``dp_emulated`` block of the default line).

BASELINE config[2] (8 x MI355X, global batch 512) has had no node to run on in any round.  What CAN be checked on one
GPU is that the rank's program at that world size is the right program: with 7 peers' gathered rows staged, the wide
decoder layers after one step must hold Adam(W, 1/8 * sum over the 8 shards of dz_r^T x_r) -- the all-reduce scheme's
update -- formed by ``pcaa_skinny_linear_wgrad_adam_t16`` from the 8 ranks' packed chunks (512 batch rows) at the
config[1] widths.
"""
import pytest
import torch

from helpers import T, load_golden
from opensetgaitrecognition_pcaa_amd import constants, ops, synthetic as syn
from opensetgaitrecognition_pcaa_amd.train import PCAATrainer, StepCount

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cfg(B, N, K):
    cfg = dict(constants.CONFIG)
    cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B, LR=1e-4, B1=0.9, B2=0.99, GP_WEIGHT=15,
               ADV_WEIGHT=1, SUP_LATENT_DIM=32)
    return cfg


def _trainer(B, N, C, K, world, gather=True, precision="bf16"):
    constants.NFEATURES = C
    tr = PCAATrainer(_cfg(B, N, K), precision=precision, emulate_world=world, dp_gather=gather,
                     grad_compress="bf16" if precision == "bf16" else None)
    for i, mod in enumerate((tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head,
                             tr.discriminator_projection_head)):
        syn.deterministic_fill_(mod, i)
    tr.set_prior_means(torch.from_numpy(load_golden("full_B64_N128")[0]["means"]))
    tr.finalize()
    tr.train()
    return tr


def _inputs(B, N, C, K):
    return (syn.synthetic_pcs(B, T, N, C, seed=1234).to(DEV).permute(0, 3, 1, 2), syn.synthetic_labels(B, K, seed=1235).to(DEV),
            syn.synthetic_z0(B, 32, seed=1236).to(DEV), syn.synthetic_alphas(B, seed=1237).to(DEV))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [8, 2])
def test_emulated_rank_forms_the_global_decoder_update_at_config1_widths(world):
    B, N, C, K = 64, 128, 4, 8
    # (1) a probe run without staged peers: its gather buffers show this rank's own operand rows and their scale
    probe = _trainer(B, N, C, K, world)
    inp = _inputs(B, N, C, K)
    probe.step(*inp)
    torch.cuda.synchronize()
    assert probe.dp_scheme == "gather"
    layers = sorted(k[0] for k in probe._gather_bufs)
    assert layers == [2, 3, 4, 5]
    gen = torch.Generator(device="cpu").manual_seed(77)

    def unpack(chunk, n_out, n_in):
        """a packed chunk [(N + K) * 64] bf16 -> (dz [64, N], x [64, K]) fp32: the values the update kernel contracts"""
        t = chunk.view(n_out + n_in, 64).float()
        return t[:n_out].t().contiguous(), t[n_out:].t().contiguous()

    peers, shapes = {}, {}
    for layer in layers:
        packed_all, own = probe._gather_bufs[(layer, world)]
        n_out, n_in = probe._dec_fused[layer][2].shape
        shapes[layer] = (n_out, n_in)
        assert tuple(packed_all.shape) == (world, (n_out + n_in) * 64)
        # without peers every slot holds the rank's own chunk
        for r in range(world):
            assert torch.equal(packed_all[r], own)
        dz_own, x_own = unpack(own, n_out, n_in)
        # the peers' rows: this rank's own, rows permuted and perturbed per peer -- right scale, distinct shards
        rows = []
        for r in range(world - 1):
            perm = torch.randperm(B, generator=gen).to(DEV)
            ndz = 1.0 + 0.25 * torch.randn(dz_own.shape, generator=gen)
            nx = 1.0 + 0.25 * torch.randn(x_own.shape, generator=gen)
            rows.append(ops.pack_rows_t16((dz_own[perm] * ndz.to(DEV)).contiguous(), (x_own[perm] * nx.to(DEV)).contiguous()))
        peers[layer] = torch.stack(rows).contiguous()
    del probe
    torch.cuda.empty_cache()

    # (2) the emulated rank with the peers staged
    tr = _trainer(B, N, C, K, world)
    for tag, rows in peers.items():
        tr._xchg.set_peers(tag, rows)
    before = {layer: tr._dec_fused[layer][2].detach().clone() for layer in layers}
    tr.step(*inp)
    torch.cuda.synchronize()
    assert tr.dp_scheme == "gather" and tr.comm["gather_bytes"] == sum(
        2 * world * 64 * (w.shape[0] + w.shape[1]) for w in before.values())
    # no gradient of the gathered layers exists in any form
    assert sorted(tr.gradless_ranges) == sorted((tr._dec_fused[l][0], tr._dec_fused[l][1]) for l in layers)

    # (3) expectation: per-shard weight gradients summed, then the plain Adam kernel at gradient scale 1 / world
    cnt = StepCount(DEV)
    cnt.advance(1e-4, 0.9, 0.99)
    worst = {}
    for layer in layers:
        packed_all, own = tr._gather_bufs[(layer, world)]
        assert torch.equal(packed_all[0], own) and torch.equal(packed_all[1:], peers[layer]), \
            "own chunk in slot 0, the staged peers behind it"
        W0 = before[layer]
        g = torch.zeros_like(W0)
        for r in range(world):
            dz_r, x_r = unpack(packed_all[r], *shapes[layer])
            g += ops.skinny_linear_wgrad(dz_r, x_r)
        exp_w, m, v = W0.clone(), torch.zeros_like(W0), torch.zeros_like(W0)
        ops.adam_step_dev_(exp_w.view(-1), g.view(-1), m.view(-1), v.view(-1), 0.9, 0.99, 1e-8, cnt.coef_dev, 1.0 / world)
        Wv, mv, vv = tr._dec_fused[layer][2:5]
        assert not torch.equal(Wv, W0)
        err = (Wv.double() - exp_w.double()).abs()
        # Adam's first step is -lr g / (|g| + eps): where the summed gradient is rounding noise the two accumulation orders
        # (eight 64-row products added in fp32 here, one contraction over the eight chunks in the kernel) can land on opposite signs,
        # 2 lr apart at worst; such elements must be rare and everything else agrees to round-off (the gate of
        # tests/test_round2_parity.py:141-155)
        scale = float(W0.abs().max())
        assert err.max().item() <= 5e-5 * scale + 2.0e-4 * 1.001, (layer, err.max().item())
        assert (err > 5e-5 * scale + 0.5e-4).double().mean().item() <= 1e-3, layer
        assert err.mean().item() <= 2e-6 * max(scale, 1.0), (layer, err.mean().item())
        # the first moment IS the scaled gradient ((1 - b1) g / world): compared directly, relative l2
        rel = float((mv.double() - m.double()).norm() / (m.double().norm() + 1e-30))
        assert rel <= 2e-3, (layer, rel)
        worst[layer] = (err.max().item(), rel)
    print(f"emulated world {world}: per wide layer (max |dW|, rel-l2 of exp_avg):", worst)


def test_emulated_allreduce_scheme_keeps_the_single_process_step():
    """The all-reduce scheme under emulation: every bucket is scaled by W on the exchange stream and Adam applies 1 / W --
    the parameters after a step are the single-process step's (to the bf16 wire rounding of the decoder buckets)."""
    B, N, C, K = 8, 128, 4, 8
    inp = _inputs(B, N, C, K)
    single = _trainer(B, N, C, K, 0, gather=False)
    single.step(*inp)
    emu = _trainer(B, N, C, K, 4, gather=False)
    out = emu.step(*inp)
    torch.cuda.synchronize()
    assert emu.dp_scheme == "allreduce" and single.dp_scheme == "none"
    assert emu.comm["collectives"] >= 5 and emu.comm["allreduce_bytes"] == emu.comm["payload_bytes"]
    assert bool(torch.isfinite(out["tot_loss"]))
    a, b = emu.flat_g.p.double(), single.flat_g.p.double()
    err = (a - b).abs()
    assert err.max().item() <= 2.0e-4 * 1.001 + 1e-6
    assert (err > 0.5e-4).double().mean().item() <= 5e-3


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["emulated8", "rccl1"])
def test_data_parallel_step_replays_as_a_hipgraph(kind):
    """Round 6: the data-parallel step through step_graphed -- collectives inside the captured graph
    (the emulated world's device operations; a real RCCL group of one rank with forced collectives, the most a one-GPU box
    admits) -- performs the steps the eager data-parallel step performs.  N=32, where the eager step is bound by the host.
    Gate: the eager-vs-graph gate of tests/test_hip_modules.py (the two runs differ by the order of the fp64 statistics
    atomics, which bf16 roundings and Adam's sign-like first steps amplify).
    The RCCL case runs in a CHILD process: a communicator created and destroyed inside the test runner leaves RCCL's
    threads and streams behind in a process that goes on capturing and replaying graphs for two hundred more tests."""
    if kind == "rccl1":
        import os
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "rccl1"], capture_output=True, text=True, timeout=500,
                             cwd=root, env=env)
        assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
        assert "graph replay check ok" in res.stdout
        return
    _graph_replay_check(kind)


def _graph_replay_check(kind):
    import torch.distributed as dist
    B, N, C, K = 64, 32, 4, 8
    pg = None
    if kind == "rccl1":
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=torch.device("cuda", torch.cuda.current_device()))
        pg = dist.group.WORLD
    try:
        def make():
            constants.NFEATURES = C
            tr = PCAATrainer(_cfg(B, N, K), precision="bf16", dp_gather=True, grad_compress="bf16",
                             **({"emulate_world": 8} if pg is None else {"process_group": pg, "force_collectives": True}))
            for i, mod in enumerate((tr.encoder, tr.decoder, tr.discriminator, tr.decoder_projection_head,
                                     tr.discriminator_projection_head)):
                syn.deterministic_fill_(mod, i)
            tr.set_prior_means(torch.from_numpy(load_golden("full_B64_N128")[0]["means"]))
            tr.finalize()
            tr.train()
            return tr
        eager, graphed = make(), make()
        assert graphed.can_graph()
        keys = ("d_loss", "gp", "rec_loss", "loss_g", "sup_loss", "tot_loss")
        for s in range(4):
            inp = (syn.synthetic_pcs(B, T, N, C, seed=50 + s).to(DEV).permute(0, 3, 1, 2), syn.synthetic_labels(B, K, seed=60 + s).to(DEV),
                   syn.synthetic_z0(B, 32, seed=70 + s).to(DEV), syn.synthetic_alphas(B, seed=80 + s).to(DEV))
            ref = {k: v.clone() for k, v in eager.step(*inp).items() if torch.is_tensor(v)}
            out = graphed.step_graphed(*inp, warmup=1)
            got = [out[k].item() for k in keys]
            want = [ref[k].item() for k in keys]
            assert all(abs(g - w) <= 2e-2 * abs(w) + 2e-2 for g, w in zip(got, want)), (s, got, want)
        torch.cuda.synchronize()
        assert graphed.dp_scheme == eager.dp_scheme == "gather"
        assert any(e["graph"] is not None for e in graphed._graphs.values()), "the step must have been captured"
        assert graphed.flat_g.step == eager.flat_g.step == 4 and int(graphed.flat_g.step_dev.item()) == 4
        d = (graphed.flat_g.p - eager.flat_g.p).abs()
        assert float(d.max()) <= 2e-4 * 4 * 1.01 and float(d.mean()) <= 2e-5, (float(d.max()), float(d.mean()))
    finally:
        if pg is not None:
            dist.destroy_process_group()


if __name__ == "__main__":
    # child process of test_data_parallel_step_replays_as_a_hipgraph[rccl1]
    import sys
    from opensetgaitrecognition_pcaa_amd import functional as F_hip
    F_hip.set_precision("fp32")
    _graph_replay_check(sys.argv[1])
    print("graph replay check ok", flush=True)
