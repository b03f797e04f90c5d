"""bench.py's ``--gpus N`` contract: the flag is binding.  Under a launcher WORLD_SIZE must
equal it; outside one, N > 1 makes bench.py the PARENT of N ranks (torch.distributed.run as a child process, decided
before any GPU call).  A line can therefore never report an ``n_gpus`` other than the N that was asked for.

CPU part: the decision function, the error path and the parent's status forwarding.  GPU part (2 gloo ranks sharing
the box's one GPU): both launch styles end in one JSON line with n_gpus == 2."""
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("pcaa_bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _args(**kw):
    base = dict(gpus=1, workload="train", backend="nccl")
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_resolve_world_decides_rank_launch_or_error():
    b = _bench_module()
    assert b.resolve_world(_args(gpus=1), [], environ={}) == ("rank", 1)
    assert b.resolve_world(_args(gpus=8), ["--gpus", "8"], environ={"WORLD_SIZE": "8"}) == ("rank", 8)
    assert b.resolve_world(_args(gpus=1), [], environ={"WORLD_SIZE": "1"}) == ("rank", 1)
    mode, cmd = b.resolve_world(_args(gpus=8), ["--gpus", "8", "--steps", "20", "--warmup", "5"], environ={})
    assert mode == "launch"
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    i = cmd.index("--nproc-per-node")
    assert cmd[i + 1] == "8" and "--nnodes=1" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    j = cmd.index(BENCH)
    assert cmd[j + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"], "the ranks get the parent's own arguments"
    for ws, gpus in (("1", 8), ("8", 1), ("4", 8), ("2", 1)):
        with pytest.raises(SystemExit) as e:
            b.resolve_world(_args(gpus=gpus), ["--gpus", str(gpus)], environ={"WORLD_SIZE": ws})
        assert e.value.code not in (0, None) and "torch.distributed.run" in str(e.value.code)
    with pytest.raises(SystemExit):
        b.resolve_world(_args(gpus=0), [], environ={})
    with pytest.raises(SystemExit):
        b.resolve_world(_args(gpus=2, workload="sweep"), [], environ={})


def test_gpus_not_equal_world_size_exits_nonzero_before_any_gpu_call():
    """No GPU here and none needed: the check comes first.  This is the case a driver hits when it starts
    ``python bench.py --gpus 1`` inside an 8-rank launcher or the reverse."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert "--gpus 8 but WORLD_SIZE=2" in res.stderr and "--nproc-per-node 8" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")], "no JSON line may be printed"


def test_parent_forwards_the_ranks_output_and_exit_status(capfd):
    b = _bench_module()
    child = [sys.executable, "-c", "import sys; print('{\"n_gpus\": 2}', flush=True); sys.exit(7)"]
    with pytest.raises(SystemExit) as e:
        b.launch_ranks(_args(gpus=2, backend="gloo"), child)
    assert e.value.code == 7
    out = capfd.readouterr()
    assert '{"n_gpus": 2}' in out.out


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("style", ["plain", "torchrun"])
def test_both_launch_styles_print_one_line_with_the_requested_n_gpus(style):
    """2 ranks over gloo, both on cuda:0 (PCAA_BENCH_DEVICE=0): ``python bench.py --gpus 2`` (bench.py starts the
    ranks itself) and the contract's torchrun command must give the same kind of line.  Small shape: this test is about
    the launch, the full-size N>1 branch is test_distributed_gpu.py's."""
    b = _bench_module()
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "1", "--points", "32", "--batch", "8",
            "--backend", "gloo", "--no-cpu-baseline", "--no-extra-legs"]
    env = dict(os.environ, PCAA_BENCH_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, BENCH, *argv] if style == "plain" else b.torchrun_command(2, argv)
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 16
    assert d["config"]["finite_loss"] and d["value"] > 0 and d["scaling"] == "weak"
