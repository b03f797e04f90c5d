"""Round 6: ONE RANK'S PROGRAM of a world of W ranks on one GPU (PCAATrainer ``emulate_world`` /
``dist.EmulatedExchange``; bench.py ``--dp-emulate`` and the ``dp_emulated`` block of the default line).

BASELINE config[2] (8 x MI355X, global batch 512) has had no node to run on in any round.  What CAN be checked on one
GPU is that the rank's program at that world size is the right program: with 7 peers' gathered rows staged, the wide
decoder layers after one step must hold Adam(W, 1/8 * sum over the 8 shards of dz_r^T x_r) -- the all-reduce scheme's
update -- formed by ``pcaa_skinny_linear_wgrad_adam_rows`` from 512 stacked rows at the config[1] widths.
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
    R = ops.gathered_rows_alloc(world * B)
    layers = sorted(k[0] for k in probe._gather_bufs)
    assert layers == [2, 3, 4, 5]
    gen = torch.Generator(device="cpu").manual_seed(77)
    peers = {}
    for layer in layers:
        dz_all, x_all = probe._gather_bufs[(layer, R)]
        # without peers every slot holds the rank's own rows
        for r in range(1, world):
            assert torch.equal(dz_all[r * B:(r + 1) * B], dz_all[:B]) and torch.equal(x_all[r * B:(r + 1) * B], x_all[:B])
        for which, own in (("dz", dz_all[:B]), ("x", x_all[:B])):
            # the peers' rows: this rank's own, rows permuted and perturbed per peer -- right scale, distinct shards
            rows = []
            for r in range(world - 1):
                perm = torch.randperm(B, generator=gen)
                noise = 1.0 + 0.25 * torch.randn(own.shape, generator=gen)
                rows.append(own[perm.to(DEV)] * noise.to(DEV))
            peers[(layer, which)] = torch.stack(rows).contiguous()
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
        4 * world * B * (w.shape[0] + w.shape[1]) for w in before.values())
    # no gradient of the gathered layers exists in any form
    assert sorted(tr.gradless_ranges) == sorted((tr._dec_fused[l][0], tr._dec_fused[l][1]) for l in layers)

    # (3) expectation: per-shard weight gradients summed, then the plain Adam kernel at gradient scale 1 / world
    cnt = StepCount(DEV)
    cnt.advance(1e-4, 0.9, 0.99)
    worst = {}
    for layer in layers:
        dz_all, x_all = tr._gather_bufs[(layer, R)]
        for which, buf in (("dz", dz_all), ("x", x_all)):
            got = buf[B:world * B].view(world - 1, B, -1)
            assert torch.equal(got, peers[(layer, which)]), "the staged peers must have arrived in the gather buffer"
        W0 = before[layer]
        g = torch.zeros_like(W0)
        for r in range(world):
            g += ops.skinny_linear_wgrad(dz_all[r * B:(r + 1) * B].contiguous(), x_all[r * B:(r + 1) * B].contiguous())
        exp_w, m, v = W0.clone(), torch.zeros_like(W0), torch.zeros_like(W0)
        ops.adam_step_dev_(exp_w.view(-1), g.view(-1), m.view(-1), v.view(-1), 0.9, 0.99, 1e-8, cnt.coef_dev, 1.0 / world)
        Wv, mv, vv = tr._dec_fused[layer][2:5]
        assert not torch.equal(Wv, W0)
        err = (Wv.double() - exp_w.double()).abs()
        # Adam's first step is -lr g / (|g| + eps): where the summed gradient is rounding noise the two accumulation orders
        # (eight 64-row products added in fp32 here, one 512-row contraction in the kernel) can land on opposite signs,
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
