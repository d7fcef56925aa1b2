"""Process-group helpers (reference ubteacher/utils/comm.py:7-13 + the Detectron2 comm calls the
trainers use).  One process per GPU; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU."""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist() else 1


def get_rank():
    return dist.get_rank() if is_dist() else 0


def is_main_process():
    return get_rank() == 0


def reduce_sum(tensor):
    """SUM all-reduce (no-op for a single process), out of place like the reference."""
    if get_world_size() < 2:
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def gather(data, dst=0):
    """Gather picklable `data` from every rank to `dst` (list on dst, [] elsewhere)."""
    ws = get_world_size()
    if ws == 1:
        return [data]
    out = [None] * ws if get_rank() == dst else None
    dist.gather_object(data, out, dst=dst)
    return out if get_rank() == dst else []


def barrier():
    """dist.barrier that names this rank's device under nccl / RCCL (without device_ids the backend guesses the device from the
    rank, which is wrong whenever the rank -> device map is not the identity, and warns otherwise)."""
    if get_world_size() < 2:
        return
    if dist.get_backend() == "nccl" and torch.cuda.is_available():
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def synchronize():
    barrier()


def all_gather_object(obj):
    """dist.all_gather_object (picklable `obj` from every rank, on every rank).  Under nccl the pickled bytes travel through
    tensors on the CURRENT device: the caller's device must be bound (engine.launch does) - asserted here, not assumed."""
    ws = get_world_size()
    if ws == 1:
        return [obj]
    if dist.get_backend() == "nccl":
        assert torch.cuda.is_available() and torch.cuda.is_initialized(), "all_gather_object under nccl needs a bound device"
    out = [None] * ws
    dist.all_gather_object(out, obj)
    return out


def rccl_selfcheck():
    """One tiny all-reduce on this rank's device right after the world was joined: the sum of the ranks must come back on every
    rank.  Returns {"ok", "ranks", "sum"}; a plumbing problem (wrong device binding, IPC mode, a rank that never joined) shows here,
    in seconds, not in the first gradient bucket."""
    ws = get_world_size()
    if ws == 1:
        return {"ok": True, "ranks": 1, "sum": 0}
    dev = torch.device("cuda", torch.cuda.current_device()) if (dist.get_backend() == "nccl" or torch.cuda.is_available()) else torch.device("cpu")
    t = torch.tensor([float(get_rank()), 1.0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    got = [float(x) for x in t.cpu()]
    return {"ok": got[0] == ws * (ws - 1) / 2 and got[1] == ws, "ranks": int(got[1]), "sum": got[0]}


def shared_random_seed():
    """Detectron2 comm.shared_random_seed: a random int that is the SAME on every rank (rank 0's draw)."""
    import numpy as np
    seed = int(np.random.randint(2 ** 31))
    if get_world_size() == 1:
        return seed
    box = [seed]
    dist.broadcast_object_list(box, src=0)
    return int(box[0])
