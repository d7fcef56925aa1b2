"""ubteacher.engine.launch (the counterpart of Detectron2's `launch`, reference train_net.py:62-73) on CPU with gloo:
`launch(fn, 2)` spawns two ranks that join one world; an externally started world (torch.distributed.run's environment) is joined,
not re-spawned, and a world of the wrong size is an error; `bench.py --gpus 2` goes through the same spawn path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "unbiased-teacher-v2_amd")


def _rank_report(out_dir):
    import torch
    import torch.distributed as dist
    from ubteacher.engine.launch import dist_info
    info = dist_info()
    t = torch.tensor([float(info["rank"] + 1)])
    dist.all_reduce(t)
    with open(os.path.join(out_dir, "rank%d.json" % info["rank"]), "w") as f:
        json.dump({"info": info, "sum": float(t), "world": dist.get_world_size(), "env_rank": os.environ.get("RANK")}, f)


def test_launch_spawns_one_rank_per_gpu(tmp_path, monkeypatch):
    sys.path.insert(0, PKG)
    from ubteacher.engine.launch import launch
    monkeypatch.setenv("UTV2_DIST_BACKEND", "gloo")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    launch(_rank_report, 2, args=(str(tmp_path),))
    reps = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    for r, rep in enumerate(reps):
        assert rep["world"] == 2 and rep["sum"] == 3.0 and rep["info"]["rank"] == r and rep["info"]["backend"] == "gloo"
        assert rep["env_rank"] == str(r)


def test_launch_refuses_a_world_of_the_wrong_size(monkeypatch):
    sys.path.insert(0, PKG)
    from ubteacher.engine.launch import launch
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(RuntimeError, match="world of 2 ranks"):
        launch(lambda: None, 4)
    with pytest.raises(NotImplementedError):
        launch(lambda: None, 2, num_machines=2)


def _run_bench(extra_env, args, launcher=()):
    env = dict(os.environ, UTV2_DIST_BACKEND="gloo", UTV2_BENCH_LAUNCH_ONLY="1", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *args]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)


def test_bench_gpus_2_spawns_two_ranks():
    """`python bench.py --gpus 2` (no torchrun): the launch-only hook stops each rank right after it joined the world, before any GPU
    work, and rank 0 prints what it joined."""
    r = _run_bench({}, ["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rep = json.loads(line)
    assert rep["n_gpus"] == 2 and rep["ranks"]["world_size"] == 2 and rep["ranks"]["backend"] == "gloo"
    assert rep["ranks"]["launcher"] == "ubteacher.engine.launch" and sorted(rep["ranks"]["rank_ids"]) == [0, 1]


def test_bench_under_torchrun_joins_that_world_and_checks_its_size():
    """the driver's form: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N; a mismatching --gpus fails loudly"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    tr = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    r = _run_bench({}, ["--gpus", "2"], launcher=tr)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["n_gpus"] == 2 and rep["ranks"]["launcher"] == "torch.distributed.run" and sorted(rep["ranks"]["rank_ids"]) == [0, 1]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    tr[-1] = str(port)
    r = _run_bench({}, ["--gpus", "4"], launcher=tr)
    assert r.returncode != 0 and "world of 2 ranks" in (r.stderr + r.stdout)


def test_bench_dry_nccl_selfcheck_two_ranks():
    """`bench.py --gpus 2 --dry-nccl`: each rank joins the world, one tiny all-reduce + a barrier + an object all-gather, rank 0 prints
    how many ranks answered (gloo here; the same code path is first contact with RCCL on a GPU node)"""
    env = dict(os.environ, UTV2_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "UTV2_BENCH_LAUNCH_ONLY"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-nccl"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["n_gpus"] == 2 and rep["rccl_ranks"] == 2 and rep["rccl"]["ok"] and rep["rccl"]["sum"] == 1.0
    assert sorted(rep["rccl"]["devices"]) == [0, 1]


def _fails_after_joining(out_dir):
    from ubteacher.engine.launch import dist_info
    with open(os.path.join(out_dir, "runs.txt"), "a") as f:
        f.write("rank%d\n" % dist_info()["rank"])
    raise OSError("[Errno 98] Address already in use (a socket main_func opened itself)")


def test_address_in_use_after_the_rendezvous_is_not_retried(tmp_path, monkeypatch):
    """only the rendezvous is retried on EADDRINUSE; the same message raised by main_func after the ranks joined propagates
    (re-running main_func from scratch would repeat training progress and checkpoint writes)"""
    sys.path.insert(0, PKG)
    from ubteacher.engine.launch import launch
    monkeypatch.setenv("UTV2_DIST_BACKEND", "gloo")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(Exception, match="Address already in use"):
        launch(_fails_after_joining, 2, args=(str(tmp_path),))
    runs = open(tmp_path / "runs.txt").read().split()
    assert len(runs) <= 2 and len(set(runs)) == len(runs), runs      # every rank ran main_func at most once
