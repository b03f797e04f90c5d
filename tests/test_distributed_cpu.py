"""World-size-2 gloo tests (CPU) of the data-parallel logic: the compute on each rank is the
ORACLE (test infrastructure), the things under test are the package's sharding, SyncBN
statistic exchange and flat-buffer gradient all-reduce + Adam scaling."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from opensetgaitrecognition_pcaa_amd import dist as pdist, functional as F_hip, synthetic as syn
    from oracle import pcaa_oracle as O
    torch.set_num_threads(2)
    r, w, group = pdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    out = {}
    # (1) host RNG draws made for the GLOBAL batch and sliced per rank
    B = 8
    z0 = syn.synthetic_z0(B, 32, seed=5)
    mine = pdist.shard_rows(z0, rank, world)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine.contiguous())
    out["z0_ok"] = bool(torch.equal(torch.cat(gathered), z0))
    # (2) SyncBN: all-reduced fp64 (sum, sumsq) statistics == statistics of the global batch
    x = torch.from_numpy(np.random.default_rng(7).standard_normal((B * 30, 16)))
    xs = pdist.shard_rows(x, rank, world)
    stats = torch.stack([xs.sum(0), (xs * xs).sum(0)]).unsqueeze(0).contiguous()
    F_hip.set_sync_bn_group(group)
    count = F_hip._sync_stats(stats, xs.shape[0])
    F_hip.set_sync_bn_group(None)
    out["syncbn_ok"] = bool(count == x.shape[0] and torch.allclose(stats[0, 0], x.sum(0)) and
                            torch.allclose(stats[0, 1], (x * x).sum(0)))
    # (3) gradient all-reduce + Adam(grad_scale=1/world) == Adam on the mean gradient
    g_local = torch.from_numpy(np.random.default_rng(11 + rank).standard_normal(1000)).float()
    g_all = [torch.from_numpy(np.random.default_rng(11 + k).standard_normal(1000)).float() for k in range(world)]
    flat = g_local.clone()
    pdist.allreduce_sum_(flat, group)
    p1 = {"p": torch.ones(1000)}
    O.adam_step(p1, {"p": flat / world}, {}, 1e-4, 0.9, 0.99)
    p2 = {"p": torch.ones(1000)}
    O.adam_step(p2, {"p": sum(g_all) / world}, {}, 1e-4, 0.9, 0.99)
    out["grad_ok"] = bool(torch.allclose(p1["p"], p2["p"], rtol=0, atol=1e-7))
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_sharding_syncbn_and_gradient_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, out in results:
        assert all(out.values()), (rank, out)
