"""Data-parallel UTv2 step, two ranks on one GPU (gloo process group over CUDA tensors; RCCL needs one device per rank,
which a single-GPU test box cannot give): the real backward drives the bucketed, overlapped gradient all-reduce.
Checks: (1) both ranks end the step with bit-identical students and teachers, (2) the overlapped bucket reduction gives
bit-identical weights to the single flat all-reduce, (3) every bucket was launched by the hooks DURING backward except
those that hold never-used parameters."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, overlap, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["UTV2_OVERLAP_ALLREDUCE"] = "1" if overlap else "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tests.utv2_testutil import FixedLoader, make_batch, small_fcos_cfg
    from ubteacher.engine import UBTeacherTrainer
    cfg = small_fcos_cfg(bl=2 * world, bu=2 * world)   # global batch; each rank takes 2 + 2
    torch.manual_seed(0)                               # identical initial weights on every rank
    prod, _ = make_batch(50 + rank, 2, 2, 96, 128, "cuda")   # different data per rank
    tr = UBTeacherTrainer(cfg, data_loader=FixedLoader(prod))
    assert tr.world_size == world
    # make the teacher emit pseudo boxes (same recipe as bench.py, on the device)
    import bench
    bench.tune_for_pseudo_labels(tr, prod)
    dist.broadcast(tr.model.flat_state(), 0)
    dist.broadcast(tr.model_teacher.flat_state(), 0)
    tr.iter = 1
    tr.optimizer.param_groups[0]["lr"] = 0.01
    launched_in_backward = None
    gs = getattr(tr, "_grad_sync", None)
    if gs is not None:
        orig_finish = gs.finish

        def finish():
            nonlocal launched_in_backward
            launched_in_backward = (sum(gs.launched), len(gs.bounds))
            orig_finish()
        gs.finish = finish
    tr.run_step_full_semisup()
    torch.cuda.synchronize()
    import hashlib

    def digest(t):  # bit-exact fingerprint (tensors themselves do not survive the worker's exit through an mp.Queue)
        a = t.detach().cpu().contiguous()
        return hashlib.sha1(a.numpy().tobytes()).hexdigest(), bool(torch.isfinite(a).all())
    q.put((rank, digest(tr.model.flat_state()), digest(tr.model_teacher.flat_state()), launched_in_backward))
    dist.barrier()
    dist.destroy_process_group()


def _run(overlap):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, s, t, l = q.get(timeout=600)
        res[r] = (s, t, l)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_two_rank_step_overlapped_allreduce():
    a = _run(True)
    b = _run(False)
    assert a[0][0] == a[1][0] and a[0][1] == a[1][1]                         # ranks stay in lock step (bit-identical)
    assert a[0][0] == b[0][0]                                                # overlapped buckets == flat all-reduce
    assert a[0][0][1] and a[0][1][1]                                         # finite
    launched, nb = a[0][2]
    assert nb >= 4 and launched >= nb - 1, (launched, nb)                    # the hooks fired during backward
    assert b[0][2] is None
