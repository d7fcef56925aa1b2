"""One process per GPU on one node: the counterpart of Detectron2's `launch` that the reference's train_net.py:62-73 calls
(`launch(main, num_gpus, num_machines, machine_rank, dist_url, args=(args,))`).

`launch(main_func, num_gpus_per_machine, ...)` spawns `num_gpus_per_machine` workers with torch.multiprocessing (start method
"spawn": HIP contexts do not survive a fork), each of which binds ITS device, joins the default process group
(backend "nccl" == RCCL over xGMI on ROCm; `UTV2_DIST_BACKEND=gloo` for CPU / single-GPU dry runs) and calls `main_func(*args)`.
The worker fails loudly when the group it joined is not the one that was asked for: a silently smaller world would report
a scaling point that was never measured.

When the process was ALREADY started as one rank of a world (torch.distributed.run exports RANK / WORLD_SIZE / LOCAL_RANK /
MASTER_*), `launch` does not spawn: it validates the environment against `num_gpus_per_machine`, joins that world and
calls `main_func` in-process.  So `python X.py --gpus N` and `python -m torch.distributed.run --nproc-per-node N X.py --gpus N`
run the same N ranks.
"""
import os
import socket

import torch
import torch.distributed as dist

__all__ = ["launch", "find_free_port", "dist_info"]

_INFO = {}


def find_free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))   # the container hostname may not resolve: always the loopback address
    port = s.getsockname()[1]
    s.close()
    return port


def dist_info():
    """what the worker actually joined: {world_size, rank, local_rank, backend, device} (filled by launch)"""
    return dict(_INFO)


def _backend():
    return os.environ.get("UTV2_DIST_BACKEND", "nccl")


def _bind_and_join(rank, local_rank, world_size, dist_url, expect_world, external=False):
    single_dev = os.environ.get("UTV2_BENCH_SINGLE_DEVICE") == "1"  # dry runs of the N > 1 path on a box with one GPU
    dev = 0 if single_dev else local_rank
    if torch.cuda.is_available():
        nvis = torch.cuda.device_count()
        if not single_dev and external and nvis == 1 and local_rank > 0:
            # an EXTERNAL launcher (torch.distributed.run) may start each rank with its own one-device view (CUDA/HIP_VISIBLE_DEVICES per
            # process).  Ranks spawned here all inherit the parent's view: there one visible GPU for N ranks stays the hard error below
            # (N ranks on one device would hang or fail inside RCCL with a duplicate-GPU error).
            dev = 0
        elif not single_dev and nvis <= dev:
            # the device THIS rank binds must exist; the world size is not a device count (ranks may see one device each)
            raise RuntimeError("launch: rank %d (local rank %d) binds device %d but only %d GPUs are visible"
                               % (rank, local_rank, dev, nvis))
        torch.cuda.set_device(dev)
    if world_size > 1 or os.environ.get("UTV2_DP_SINGLE_RANK") == "1":
        kw = {}
        if torch.cuda.is_available() and _backend() == "nccl":
            kw["device_id"] = torch.device("cuda", dev)   # RCCL communicator bound to this rank's device at init (eager, no lazy guess)
        dist.init_process_group(_backend(), init_method=dist_url, rank=rank, world_size=world_size, **kw)
        if dist.get_world_size() != expect_world:
            raise RuntimeError("launch: joined a world of %d ranks, %d were requested" % (dist.get_world_size(), expect_world))
    _INFO.update(world_size=world_size, rank=rank, local_rank=local_rank, backend=_backend() if dist.is_initialized() else None,
                 device=dev)


def _worker(local_rank, main_func, world_size, dist_url, args, joined=None):
    os.environ["RANK"] = os.environ["LOCAL_RANK"] = str(local_rank)
    os.environ["WORLD_SIZE"] = os.environ["LOCAL_WORLD_SIZE"] = str(world_size)
    _bind_and_join(local_rank, local_rank, world_size, dist_url, world_size)
    if joined is not None:
        joined[local_rank] = 1     # the rendezvous succeeded: from here on nothing is retried (see launch)
    try:
        main_func(*args)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, dist_url=None, args=()):
    if num_machines != 1 or machine_rank != 0:
        raise NotImplementedError("one node (8 x MI355X over xGMI) is the scope of this launcher")
    n = int(num_gpus_per_machine)
    if n < 1:
        raise ValueError("num_gpus_per_machine must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None and "RANK" in os.environ:
        # already one rank of an externally launched world (torch.distributed.run)
        world = int(env_world)
        if world != n:
            raise RuntimeError("launch: started inside a world of %d ranks (WORLD_SIZE) but %d GPUs were requested" % (world, n))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        url = "env://"
        _bind_and_join(int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"])), world, url, n, external=True)
        try:
            return main_func(*args)
        finally:
            if dist.is_initialized():
                dist.destroy_process_group()
    if n == 1:
        # UTV2_DP_SINGLE_RANK=1: a process group of ONE rank (the real backend on a one-GPU box; see trainer.data_parallel)
        url = "tcp://127.0.0.1:%d" % find_free_port() if os.environ.get("UTV2_DP_SINGLE_RANK") == "1" else None
        _bind_and_join(0, 0, 1, url, 1)
        try:
            return main_func(*args)
        finally:
            if dist.is_initialized():
                dist.destroy_process_group()
    import torch.multiprocessing as mp
    if dist_url not in (None, "auto"):
        mp.spawn(_worker, nprocs=n, args=(main_func, n, dist_url, args), join=True)
        return None
    # find_free_port closes its probe socket before the workers bind: another process can take the port in between.  A RENDEZVOUS
    # that dies on EADDRINUSE is retried on a fresh port; once any worker has joined the process group (it then runs main_func: training
    # progress, checkpoint writes) nothing is retried - an "address already in use" raised later by main_func itself (a metrics or
    # dataloader socket) propagates like any other error instead of re-running the job from scratch.
    ctx = mp.get_context("spawn")
    for attempt in range(4):
        dist_url = "tcp://127.0.0.1:%d" % find_free_port()
        joined = ctx.Array("i", n)     # one flag per rank, set right after init_process_group returned
        try:
            mp.spawn(_worker, nprocs=n, args=(main_func, n, dist_url, args, joined), join=True)
            return None
        except Exception as e:  # noqa: BLE001
            msg = str(e)
            in_use = "EADDRINUSE" in msg or "Address already in use" in msg or "address already in use" in msg
            if attempt == 3 or not in_use or any(joined[:]):
                raise
    return None
