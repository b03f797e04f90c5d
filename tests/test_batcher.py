"""Packed crop store + device-side batch assembly (batcher.py) against the reference's batch contract:
MSRadarDataset.__getitem__ + default collate + DataLoader(shuffle=True, drop_last=True)
(datasets.py:466-479, PCAA_ablation.py:794-800), restated in oracle.collate_batch."""
import os

import numpy as np
import pytest
import torch

from opensetgaitrecognition_pcaa_amd import batcher, constants
from opensetgaitrecognition_pcaa_amd.constants import SPLIT
from opensetgaitrecognition_pcaa_amd.datasets import MSRadarDataset, SyntheticGaitDataset
from oracle import pcaa_oracle as O

T, N, C = 30, 16, 4


def _make_split(tmp_path, monkeypatch, n_per_subj=5, subjects=(0, 3, 7)):
    root = tmp_path / "generated_dataset"
    d = root / "train"
    d.mkdir(parents=True)
    rng = np.random.default_rng(5)
    raw = {}
    for s in subjects:
        for i in range(n_per_subj):
            scen = ("free_walk", "hands_in_pockets", "smartphone")[i % 3]
            name = f"crop{i}_subj{s}_{scen}_track{i % 2}.npy"
            arr = rng.standard_normal((T, N, C))          # float64, as generate_splits writes them
            np.save(d / name, arr)
            raw[name] = arr
    monkeypatch.setattr(constants, "GEN_DATA_PATH", str(root))
    return raw


def test_epoch_order_and_rng_consumption_match_torch_dataloader():
    n, bs = 53, 7
    ds = torch.utils.data.TensorDataset(torch.arange(n))
    for shuffle in (True, False):
        torch.manual_seed(1234)
        ref, ref_draws = [], []
        loader = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=shuffle, drop_last=True, num_workers=0)
        for _ in range(3):
            ref.append(torch.cat([b[0] for b in loader]))
            ref_draws.append(torch.rand(4))               # what the loop's next alphas draw would see
        torch.manual_seed(1234)
        for e in range(3):
            order = batcher.dataloader_epoch_order(n, shuffle)[: (n // bs) * bs]
            assert torch.equal(order, ref[e]), (shuffle, e)
            assert torch.equal(torch.rand(4), ref_draws[e]), "global RNG consumption differs from DataLoader's"


def test_pack_split_roundtrip_matches_dataset_items(tmp_path, monkeypatch):
    raw = _make_split(tmp_path, monkeypatch)
    ds = MSRadarDataset(SPLIT.TRAIN)
    assert len(ds) == len(raw)
    man = batcher.pack_split(ds, str(tmp_path / "packed"))
    assert man["n"] == len(ds) and man["filenames"] == list(ds.filenames) and (man["T"], man["N"], man["C"]) == (T, N, C)
    pc = batcher.PackedCrops(str(tmp_path / "packed"))
    idx = [4, 0, 11, 7]
    ref_x, ref_y = O.collate_batch([raw[f] for f in ds.filenames], ds.labels, idx)          # [B,C,T,N]
    got = torch.from_numpy(np.ascontiguousarray(pc.crops[idx])).permute(0, 3, 1, 2)
    assert torch.equal(got, ref_x)
    assert np.array_equal(pc.labels[idx], ref_y.numpy())
    # dense relabelling over the subjects present (datasets.py:455-464)
    assert sorted(set(pc.labels.tolist())) == [0, 1, 2]


@pytest.mark.gpu
def test_device_batcher_is_bit_exact_and_follows_dataloader_order(tmp_path, monkeypatch):
    raw = _make_split(tmp_path, monkeypatch, n_per_subj=9)
    ds = MSRadarDataset(SPLIT.TRAIN)
    bs = 4
    torch.manual_seed(77)
    ref = [(x.clone(), y.clone()) for x, y in torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=True,
                                                                          drop_last=True, num_workers=0)]
    torch.manual_seed(77)
    bt = batcher.batcher_for(ds, bs, "cuda", shuffle=True)
    got = list(bt)
    assert len(got) == len(ref) == len(ds) // bs
    for (gx, gy), (rx, ry) in zip(got, ref):
        assert gx.shape == rx.shape and gx.stride()[1] == 1, "batches are [B,C,T,N] views of point-major storage"
        assert torch.equal(gx.cpu(), rx) and torch.equal(gy.cpu(), ry)        # byte work: bit-exact
    bt.check()
    # second construction reuses the packed store; an out-of-range index is caught, not read
    bt2 = batcher.batcher_for(ds, bs, "cuda", shuffle=False)
    x, y = bt2.batch(torch.tensor([0, len(ds), 2, -1], device="cuda"))
    assert float(x[1].abs().max()) == 0.0 and float(x[3].abs().max()) == 0.0
    with pytest.raises(IndexError):
        bt2.check()


@pytest.mark.gpu
def test_device_batcher_full_size_properties():
    """BASELINE-shaped crops (N=128, C=4), a store of 4096 sequences: every batch of a shuffled epoch is a
    permutation slice (checksum of checksums equals the store's) and the epoch touches each crop once."""
    M, Nn, bs = 4096, 128, 64
    ds = SyntheticGaitDataset(M, 8, N=Nn, C=4, seed=3)
    bt = batcher.batcher_for(ds, bs, "cuda", shuffle=True)
    torch.manual_seed(5)
    tot = torch.zeros((), dtype=torch.float64, device="cuda")
    lab = torch.zeros(8, dtype=torch.int64, device="cuda")
    for x, y in bt:
        tot += x.double().sum()
        lab += torch.bincount(y, minlength=8)
    assert abs(float(tot) - float(ds.pcs.double().sum())) <= 1e-6 * float(ds.pcs.double().abs().sum())
    assert torch.equal(lab.cpu(), torch.bincount(ds.labels, minlength=8))


def test_packed_store_is_rebuilt_when_the_split_is_regenerated(tmp_path, monkeypatch):
    """Same file names, new contents (generate_splits with another NMAX): the signature in the manifest differs."""
    _make_split(tmp_path, monkeypatch)
    ds = MSRadarDataset(SPLIT.TRAIN)
    man = batcher.pack_split(ds, str(tmp_path / "packed"))
    assert man["source_signature"] == batcher.source_signature(ds)
    name = ds.filenames[0]
    np.save(os.path.join(ds.dataset_dir, name), np.zeros((T, N + 4, C)))       # regenerated with more points
    assert batcher.source_signature(MSRadarDataset(SPLIT.TRAIN)) != man["source_signature"]
