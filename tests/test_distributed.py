"""Multi-rank logic on CPU: world_size 2, gloo backend (the GPU path uses the same code with
backend nccl = RCCL).  Shard sizes follow the reference's split
(graph_benchmark/xrays_bench.cpp:38-51)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, WORKLOADS


def test_shard_bounds_match_reference_split():
    from graph_framework_amd.xrays import shard_bounds
    for total, shards in ((100000, 8), (10, 3), (7, 8), (1, 1), (1000003, 6)):
        batch, extra = total//shards, total % shards
        begin = 0
        for i in range(shards):
            size = batch + (1 if extra > i else 0)          # xrays_bench.cpp:46-47
            assert shard_bounds(total, shards, i) == (begin, begin + size)
            begin += size
        assert begin == total


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, result_queue):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from graph_framework_amd import distributed as gfd
    from graph_framework_amd.xrays import shard_bounds
    gfd.init(backend="gloo")

    path = os.path.join(WORKLOADS, "korc_initialize_gamma_f64.gfir")
    data = open(path, "rb").read() if rank == 0 else b""
    received = gfd.broadcast_bytes(data, 0)
    same = received == open(path, "rb").read()

    begin, end = shard_bounds(total, world, rank)
    local = torch.arange(begin, end, dtype=torch.float64)
    full = gfd.all_gather_shards(local, total)
    gathered_ok = bool(torch.equal(full, torch.arange(total, dtype=torch.float64)))
    slowest = gfd.max_over_ranks(float(rank + 1))
    gfd.barrier()
    result_queue.put((rank, same, gathered_ok, slowest))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("total", [11, 4096])
def test_broadcast_and_all_gather_world_size_2(total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, gathered_ok, slowest in results:
        assert same and gathered_ok and slowest == 2.0
