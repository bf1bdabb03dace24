"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no inter-device communication at all: each device thread builds its own
graph, reads the equilibrium file itself and writes its own result file
(graph_benchmark/xrays_bench.cpp:34-108, graph_driver/xrays.cpp:419-461).  The ensemble
shards trivially, so the step loop needs NO collective.  Two collectives exist, both outside
the step loop:
  * broadcast of the work items (which carry the folded equilibrium coefficient tables,
    1-2 MB each) from rank 0, so that only one rank touches the file system / front end;
  * all-gather of the trajectory state (8 SoA arrays) at sync/output cadence.
Shards are contiguous and sized exactly as the reference sizes its per-thread batches.
"""
import os

import numpy as np

from .xrays import shard_bounds


#  force_collectives: run every collective below through the backend even in a one-rank group
#  (`bench.py --gpus 1 --backend nccl --force-collectives`): the RCCL code path — dtypes, devices,
#  API use — executes on a one-GPU box instead of for the first time on an 8-GPU node.
_FORCE = False


def _single():
    import torch.distributed as dist
    return not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE)


def init(backend=None, device_index=None, force_collectives=False):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank).
    device_index overrides LOCAL_RANK as the CUDA device (rehearsals on a one-GPU box);
    force_collectives initialises the group for one rank as well and keeps the collectives real."""
    global _FORCE
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    _FORCE = bool(force_collectives)
    if (world > 1 or force_collectives) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank if device_index is None else device_index)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def _device():
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_bytes(data, src=0):
    """Broadcast a bytes object (a serialized work item with its tables) from `src`."""
    import torch
    import torch.distributed as dist
    if _single():
        return data
    device = _device()
    length = torch.tensor([len(data) if dist.get_rank() == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(length, src)
    if dist.get_rank() == src:
        payload = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    else:
        payload = torch.empty(int(length.item()), dtype=torch.uint8, device=device)
    dist.broadcast(payload, src)
    return payload.cpu().numpy().tobytes()


def all_gather_shards(local, total):
    """All-gather contiguous shards of unequal size (reference split) into the full array.

    local: 1-D torch tensor holding this rank's shard; total: ensemble size.
    Shards are padded to the largest shard for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    if _single():
        return local.clone()
    world = dist.get_world_size()
    sizes = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
    largest = max(sizes)
    padded = torch.zeros(largest, dtype=local.dtype, device=local.device)
    padded[:local.numel()] = local
    gathered = torch.empty(largest*world, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, padded)
    return torch.cat([gathered[r*largest:r*largest + sizes[r]] for r in range(world)])


def max_over_ranks(value):
    """MAX all-reduce of a python float (timing)."""
    import torch
    import torch.distributed as dist
    if _single():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if not _single():
        dist.barrier()
