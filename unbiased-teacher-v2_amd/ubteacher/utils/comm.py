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


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def shared_random_seed():
    """Detectron2 comm.shared_random_seed: a random int that is the SAME on every rank (rank 0's draw)."""
    import numpy as np
    seed = int(np.random.randint(2 ** 31))
    if get_world_size() == 1:
        return seed
    box = [seed]
    dist.broadcast_object_list(box, src=0)
    return int(box[0])
