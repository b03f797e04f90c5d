"""CPU-only checks: the C-ABI library loads and exports every symbol the
header declares (no compute calls), argument validation fails loudly, host-side
helpers (prior centroids, flat buffers, dataset contract) behave."""
import ctypes
import os

import numpy as np
import pytest
import torch

from helpers import load_golden
from opensetgaitrecognition_pcaa_amd import _lib, constants, models, synthetic as syn
from opensetgaitrecognition_pcaa_amd.utils import openness, sample_distant_points


def test_library_exports_every_declared_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 25
    lib = _lib.load()
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in include/pcaa_hip.h but not exported"
    assert lib.pcaa_abi_version() == _lib.ABI_VERSION >= 5
    assert lib.pcaa_disc_workspace_bytes(64, 8) > 0


def test_argument_validation_reports_errors_without_a_gpu():
    lib = _lib.load()
    rc = lib.pcaa_gemm(0, None, 0, 0, 0, None, 0, 0, 0, None, 0, 0, 4, 4, 4, None, None, 0, 1, 0, None)
    assert rc == 1
    assert b"null operand" in lib.pcaa_last_error()
    rc = lib.pcaa_adam_step(None, None, None, None, 0, 0.0, 0.0, 0.0, 0.0, 0, 1.0, 0, None)
    assert rc == 1


def test_modules_refuse_cpu_tensors():
    constants.NFEATURES = 4
    dec = models.CGDecoder(input_dim=32, nmax_points=8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        dec(torch.zeros(2, 32))
    disc = models.CGDiscriminator(4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        disc(torch.zeros(2, 32), torch.zeros(2, 4))


def test_prior_means_match_reference_goldens():
    g, _ = load_golden("misc")
    for K in (2, 4, 6, 8):
        got = sample_distant_points(32, K, 10, 10).numpy()
        assert np.array_equal(got, g[f"means_K{K}"])
    assert abs(openness(4, 10) - (1 - np.sqrt(8 / 14))) < 1e-12


def test_default_init_draws_match_torch_layers():
    """same torch RNG draws in the same order as the reference's constructors:
    a seeded reference run and a seeded run of this package start from identical weights"""
    constants.NFEATURES = 4
    torch.manual_seed(3)
    enc = models.CGEncoder(4, nmax_points=16, use_projection_head=True)
    torch.manual_seed(3)
    ref = torch.nn.Conv2d(4, constants.POINTNET_OUT_DIM // 2, (1, 1))
    assert torch.equal(enc.pc_block.pointnet1.module[0].weight, ref.weight)


def test_dataset_item_contract(tmp_path, monkeypatch):
    from opensetgaitrecognition_pcaa_amd.datasets import MSRadarDataset, SyntheticGaitDataset
    from opensetgaitrecognition_pcaa_amd.constants import SPLIT
    root = tmp_path / "gen"
    (root / "train").mkdir(parents=True)
    rng = np.random.default_rng(0)
    for i, (subj, sc) in enumerate([(3, "free_walk"), (7, "hands_in_pockets"), (3, "smartphone")]):
        np.save(root / "train" / f"crop{i}_subj{subj}_{sc}_track{i}.npy", rng.standard_normal((30, 12, 4)))
    monkeypatch.setattr(constants, "GEN_DATA_PATH", str(root))
    ds = MSRadarDataset(SPLIT.TRAIN)
    assert len(ds) == 3
    x, y = ds[0]
    assert x.shape == (4, 30, 12) and x.dtype == torch.float32 and y.dtype == torch.int64
    assert set(ds.labels.tolist()) == {0, 1}
    sy = SyntheticGaitDataset(5, K=3, N=12, C=4)
    x, y = sy[2]
    assert x.shape == (4, 30, 12) and x.permute(1, 2, 0).is_contiguous()


def test_deterministic_fill_is_reproducible():
    constants.NFEATURES = 4
    a = models.CGDiscriminator(4)
    b = models.CGDiscriminator(4)
    syn.deterministic_fill_(a, 5)
    syn.deterministic_fill_(b, 5)
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(va, vb), k


def test_ragged_row_tile_predicates_bound_the_padded_size():
    """a partial last row tile is served through 32-bit buffer offsets formed for every row of the
    PADDED tile; the *_supported predicates must refuse a result whose padded size reaches 4 GiB (rows past M would
    wrap into the first rows) instead of reporting it supported and failing -- or corrupting -- at launch."""
    lib = _lib.load()
    for fn in (lib.pcaa_gemm_split3_supported, lib.pcaa_gemm_dgrad_bn_supported):
        assert fn(72000, 512, 512) == 1                    # the reference's default shape: B=16, N=150 (ragged, small)
        assert fn(72000, 512, 256) == 0                    # contraction too short for the 4-wave loop's hand-off
        assert fn(4194304, 256, 512) == 1                  # whole tiles: no bound
        assert fn(4194304 - 1, 256, 512) == 0              # ragged, padded rows x 256 columns x 4 B = 4 GiB
        assert fn(4194304 - 257, 256, 512) == 1
    # the first-layer recompute variant left with the 8-wave kernel: passing x is an argument error, not a launch error
    one = ctypes.c_void_p(16)
    rc = lib.pcaa_gemm_dgrad_bn(one, 512, one, 512, None, one, 512, one, one, one, one, one, 1, 1000, 512, 512,
                                one, 4, one, None)
    assert rc == 1 and b"recompute variant" in lib.pcaa_last_error()


def test_steady_kernel_stats_keeps_only_the_timed_steps(tmp_path):
    """tools/steady_kernel_stats.py (round 5, evidence hygiene): a rocprofv3 kernel trace of bench.py holds warm-up steps
    whose launches rocprofv3's own --stats summary averages in; the reduction keeps the last ``--steps`` steps (a step =
    the launches between two cross_entropy_kernel launches) so that its AverageNs is what roofline.achieved is built from."""
    import csv
    import subprocess
    import sys
    d = tmp_path / "prof" / "runc"
    d.mkdir(parents=True)
    rows, t = [], 0
    for step in range(7):                                   # 2 warm-up + 5 timed
        dur = 900_000 if step < 2 else 250_000              # warm-up GEMMs are slower (first touch)
        for name, ns in (("gemm_bf16_v2_kernel<bf16>", dur), ("bn_act_fwd_kernel", 100_000), ("gemm_bf16_v2_kernel<bf16>", dur),
                         ("cross_entropy_kernel(float const*)", 7_000), ("adam_kernel", 50_000)):
            rows.append({"Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + ns})
            t += ns + 1000
    with open(d / "1_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "steady_kernel_stats.py"), str(tmp_path / "prof"),
                          "--steps", "5", "--flop-per-launch", "257.7e9"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    out = {r["Name"]: r for r in csv.DictReader(res.stdout.splitlines())}
    g = out["gemm_bf16_v2_kernel<bf16>"]
    assert int(g["Calls"]) == 10 and float(g["AverageNs"]) == 250000.0, "warm-up launches must not be averaged in"
    assert "1030.8 TFLOP/s" in res.stderr and "0.412" in res.stderr


def test_epoch_draws_are_the_reference_per_step_draws():
    """Round 6: the loop draws a whole epoch's z0 / alphas at its start (train._EpochDraws) -- the values and the state both
    generators are left in must be those of the reference's per-step calls (PCAA_ablation.py:915-925, 944-948):
    ``np.random.normal(0, 1, (B, L))`` cast to float32 and ``torch.rand(size=(B, 1))``, step after step."""
    from opensetgaitrecognition_pcaa_amd.train import _EpochDraws
    steps, B, L = 7, 6, 32
    np.random.seed(123)
    torch.manual_seed(456)
    ref_z = [torch.from_numpy(np.random.normal(0.0, 1.0, (B, L))).float() for _ in range(steps)]
    np.random.seed(123)
    torch.manual_seed(456)
    ref_a = [torch.rand(size=(B, 1)) for _ in range(steps)]
    # (the two generators are independent streams: the interleaving of the calls does not matter, their order within each does)
    np.random.seed(123)
    torch.manual_seed(456)
    d = _EpochDraws(L, torch.device("cpu"))
    z, a = d.draw(steps, B)
    assert z.shape == (steps, B, L) and a.shape == (steps, B, 1) and z.dtype == a.dtype == torch.float32
    for i in range(steps):
        assert torch.equal(z[i], ref_z[i]) and torch.equal(a[i], ref_a[i]), i
    # a second epoch continues both streams exactly where per-step draws would
    np.random.seed(123)
    torch.manual_seed(456)
    for _ in range(steps):
        np.random.normal(0.0, 1.0, (B, L))
        torch.rand(size=(B, 1))
    want_z, want_a = torch.from_numpy(np.random.normal(0.0, 1.0, (B, L))).float(), torch.rand(size=(B, 1))
    np.random.seed(123)
    torch.manual_seed(456)
    d2 = _EpochDraws(L, torch.device("cpu"))
    d2.draw(steps, B)
    z2, a2 = d2.draw(1, B)
    assert torch.equal(z2[0], want_z) and torch.equal(a2[0], want_a)
    assert d.draw(0, B) == (None, None)


def test_skinny_split_depth_keeps_the_grid_inside_one_round_of_resident_workgroups():
    """Round 6: pcaa_skinny_splits -- a host-side function -- picks the deepest split whose grid fits the kernel's resident
    workgroups (forward 768, dgrad 512 with bf16 products; 512 / 768 for the fp32-product forms), at least two chunks per
    workgroup, no empty split; the config[1] decoder layers as the regression cases (the 7680 -> 15360 forward ran 840
    workgroups on 768 slots through round 5)."""
    lib = _lib.load()
    S = 30 * 4 * 128
    widths = [S // 16, S // 8, S // 4, S // 2, S]
    slots = {0: 768, 1: 512, 2: 512, 3: 768}
    for K, N in zip(widths[:-1], widths[1:]):
        for kind in (0, 1, 2, 3):
            fwd = kind % 2 == 0
            groups = -(-(N if fwd else K) // (128 if fwd else 256))
            chunks = (K if fwd else N) // 64
            ns = lib.pcaa_skinny_splits(kind, 64, N, K)
            cps = -(-chunks // ns)
            assert 1 <= ns <= chunks and -(-chunks // cps) == ns, (kind, N, K, ns)          # no empty split
            assert groups * ns <= slots[kind] or ns == 1, (kind, N, K, ns, groups)
            assert cps >= 2 or chunks < 2, (kind, N, K, ns)
    assert lib.pcaa_skinny_splits(0, 64, 15360, 7680) == 6 and lib.pcaa_skinny_splits(1, 64, 15360, 7680) == 16
    assert lib.pcaa_skinny_splits(0, 64, 960, 64) == 1            # the first layer: one chunk, written by the kernel's own epilogue
    assert lib.pcaa_packed_chunk_elems(15360, 7680) == (15360 + 7680) * 64
