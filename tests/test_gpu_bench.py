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
#  25 passes of the Newton loop in launches of up to 3 passes: 9 launches, the loop ends on the first pass of the 9th, which is redone alone
    assert extra["loss_kernel"]["passes"] == 25 and extra["loss_kernel"]["passes_per_launch"] == 3 and extra["loss_kernel"]["launches"] == 10
    assert extra["korc_step_f32"]["frac"] > 0.1
    assert line["value_with_sync_host"] < line["value"]


def test_two_ranks_on_one_gpu_through_the_own_launcher():
    line = _bench(["--gpus", "2", "--share-gpu", "--backend", "gloo", "--total-rays", "400001", "--steps", "10",
                   "--warmup", "2", "--warmup-seconds", "0.05", "--no-extra"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["total_rays"] == 400001 and line["config"]["rays_per_gpu"] == 200001   # rank 0: the larger shard
    assert line["distributed"] == {"world_size": 2, "launcher": "bench.py", "backend": "gloo", "forced_for_one_rank": False,
                                   "rccl_version": line["distributed"]["rccl_version"]}
    assert line["all_gather_seconds"] > 0 and line["all_gather_bytes"] == 400001*8*8
    assert line["value"] > 1.0e8


def test_rccl_code_path_runs_with_a_one_rank_group():
    """the `nccl` (= RCCL) branches of distributed.py and bench.py had never
    executed.  `--gpus 1 --backend nccl --force-collectives` initialises a ONE-rank RCCL group and
    sends the item broadcast (uint8 on the device), the all-gather of the eight fp64 state arrays
    from the device tensors the kernels write, the max-over-ranks all-reduce and the barriers
    through RCCL: a wrong dtype, device or API use fails here, not on the first 8-GPU run."""
    line = _bench(["--gpus", "1", "--backend", "nccl", "--force-collectives", "--rays-per-gpu", "200000", "--steps", "10",
                   "--warmup", "2", "--warmup-seconds", "0.05", "--no-cpu-baseline", "--no-extra"])
    assert line["n_gpus"] == 1 and line["newton_iterations"] == 24
    assert line["distributed"]["backend"] == "nccl" and line["distributed"]["forced_for_one_rank"] is True
    assert line["distributed"]["rccl_version"]
    assert line["all_gather_seconds"] > 0 and line["all_gather_bytes"] == 200000*8*8


def test_korc_leg_over_rccl_with_a_one_rank_group():
    """The xkorc leg (BASELINE configs[4]) through RCCL: four items broadcast, seven fp32 device
    arrays all-gathered; identical particles, so the gathered checksum is 7 sums of n equal values."""
    line = _bench(["--workload", "korc", "--gpus", "1", "--backend", "nccl", "--force-collectives",
                   "--total-rays", "300000", "--steps", "10", "--warmup", "2", "--warmup-seconds", "0.05"])
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["unit"] == "particle-steps/s" and line["cpu_baseline"]["value"] > 1.0e6
    assert line["unit"] == "particle-steps/s" and line["dtype"] == "f32" and line["n_gpus"] == 1
    assert line["distributed"]["backend"] == "nccl"
    assert line["all_gather_bytes"] == 300000*4*7 and line["all_gather_checksum"] != 0.0
    assert line["config"]["axis_newton_iterations"] > 0 and 1.0 < line["config"]["b0"] < 2.0
    assert 0.0 < line["roofline"]["frac"] < 1.0


def test_korc_leg_two_ranks_on_one_gpu():
    """C5's sharded leg rehearsed on one GPU: two ranks share device 0 over gloo, 1e6+1 particles
    split 500001 + 500000 as graph_korc/xkorc.cpp:20-25 splits them; the gathered ensemble is the
    one-rank ensemble (identical particles: same checksum as the one-rank run of the same size)."""
    arguments = ["--workload", "korc", "--total-rays", "1000001", "--steps", "12", "--warmup", "3", "--warmup-seconds", "0.0",
                 "--no-cpu-baseline"]
    two = _bench(arguments + ["--gpus", "2", "--share-gpu", "--backend", "gloo"])
    one = _bench(arguments + ["--gpus", "1", "--backend", "gloo", "--force-collectives"])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["particles_per_gpu"] == 500001 and two["config"]["total_particles"] == 1000001
    assert two["distributed"]["backend"] == "gloo" and two["distributed"]["launcher"] == "bench.py"
    assert two["all_gather_bytes"] == 1000001*4*7
    assert two["all_gather_checksum"] == one["all_gather_checksum"]
    assert two["config"]["b0"] == one["config"]["b0"]
