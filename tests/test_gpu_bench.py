"""bench.py as the driver runs it, at small sizes: the one-GPU line's contract keys, and the
multi-rank path rehearsed on ONE GPU (two ranks sharing device 0 over gloo; RCCL refuses two ranks
on one device) through bench.py's own launcher."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(arguments, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + arguments, env=env,
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_one_gpu_line_has_the_contract_keys():
    line = _bench(["--rays-per-gpu", "200000", "--steps", "20", "--warmup", "2", "--warmup-seconds", "0.05",
                   "--no-cpu-baseline"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["dtype"] == "f64" and line["vs_baseline"] is None
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0.0 < roof["frac"] < 1.0
    assert abs(roof["frac"] - roof["achieved"]/roof["peak"]) < 1.0e-12
    assert roof["traffic"] is None or roof["traffic"] > 0
    assert roof["kernel_ms"] <= line["ms_per_step"]*1.05
    assert line["newton_iterations"] == 24
    extra = line["roofline_extra"]
    assert extra["loss_kernel"]["launches"] == 25 and extra["korc_step_f32"]["frac"] > 0.1
    assert line["value_with_sync_host"] < line["value"]


def test_two_ranks_on_one_gpu_through_the_own_launcher():
    line = _bench(["--gpus", "2", "--share-gpu", "--backend", "gloo", "--total-rays", "400001", "--steps", "10",
                   "--warmup", "2", "--warmup-seconds", "0.05", "--no-extra"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["total_rays"] == 400001 and line["config"]["rays_per_gpu"] == 200001   # rank 0: the larger shard
    assert line["distributed"] == {"world_size": 2, "launcher": "bench.py", "backend": "gloo",
                                   "rccl_version": line["distributed"]["rccl_version"]}
    assert line["all_gather_seconds"] > 0 and line["all_gather_bytes"] == 400001*8*8
    assert line["value"] > 1.0e8
