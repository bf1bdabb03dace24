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


def test_c_abi_shard_split_is_the_reference_split():
    """gfhip_shard_bounds (include/gf_hip.h) is what the C++ drivers split with (csrc/korc_push.cpp:
    graph_korc/xkorc.cpp:20-25; csrc/xrays_bench.cpp): the same shards as xrays.shard_bounds, which
    the ranks of bench.py use, for every thread of every case, ragged and empty shards included."""
    import ctypes
    from graph_framework_amd import _lib
    from graph_framework_amd.xrays import shard_bounds
    lib = _lib.load()
    begin, end = ctypes.c_size_t(), ctypes.c_size_t()
    for total, shards in ((10000000, 8), (10000000, 3), (7, 8), (0, 4), (1, 1), (1000003, 6), (1000, 2)):
        covered = 0
        for i in range(shards):
            assert lib.gfhip_shard_bounds(total, shards, i, ctypes.byref(begin), ctypes.byref(end)) == 0
            assert (begin.value, end.value) == shard_bounds(total, shards, i)
            assert end.value - begin.value == total//shards + (1 if total % shards > i else 0)      # xkorc.cpp:24
            assert begin.value == covered
            covered = end.value
        assert covered == total
    assert lib.gfhip_shard_bounds(10, 0, 0, ctypes.byref(begin), ctypes.byref(end)) == 1
    assert lib.gfhip_shard_bounds(10, 2, 2, ctypes.byref(begin), ctypes.byref(end)) == 1


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


def _run_bench(arguments, env=None, timeout=300):
    import subprocess
    environment = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    environment.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + arguments, env=environment,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher becomes the launcher (the reference's
    benchmark needs none: one thread per device, xrays_bench.cpp:34-108): N fresh rank processes,
    rendezvous on 127.0.0.1, stdout = rank 0's ONE JSON line.  Driven here through the CPU
    rehearsal of the rank-side plumbing (gloo): item broadcast, reference shard split,
    all-gather of unequal shards, max over ranks."""
    import json
    out = _run_bench(["--gpus", "3", "--rehearse-cpu", "--total-rays", "1000003"])
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 3 and line["total_rays"] == 1000003 and line["launcher"] == "bench.py"
    assert line["gathered_elements"] == 8*1000003 and line["slowest_rank"] == 2.0 and line["item_bytes"] > 100000


def test_bench_korc_leg_starts_its_own_ranks():
    """BASELINE configs[4] over N ranks (`--workload korc`): the four xkorc work items are broadcast
    from rank 0 and arrive byte-identical, the particles are split as graph_korc/xkorc.cpp:16-25
    splits them over its device threads, the seven fp32 particle arrays are all-gathered."""
    import json
    out = _run_bench(["--gpus", "2", "--rehearse-cpu", "--workload", "korc", "--total-rays", "10000001"])
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["workload"] == "korc" and line["n_gpus"] == 2 and line["items"] == 4
    assert line["gathered_elements"] == 7*10000001 and line["item_bytes"] > 1000000


def test_one_rank_group_runs_real_collectives():
    """--force-collectives: a ONE-rank process group whose broadcast / all-gather / max run through
    the backend (gloo here; `--backend nccl` on the GPU box is tests/test_gpu_bench.py)."""
    import json
    out = _run_bench(["--gpus", "1", "--rehearse-cpu", "--force-collectives", "--total-rays", "1001"])
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["gathered_elements"] == 8*1001


def test_bench_launcher_reports_a_failed_rank():
    """Any rank failing makes the launcher stop the others (by PID) and exit non-zero."""
    out = _run_bench(["--gpus", "2", "--rehearse-cpu", "--fail-rank", "1"])
    assert out.returncode != 0
    assert "rank 1 failed" in out.stderr


def test_bench_rank_under_an_external_launcher():
    """The same rank code under torch.distributed.run (how the round driver starts N > 1)."""
    import json
    import subprocess
    port = _free_port()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-cpu"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["launcher"] == "torchrun"
