"""The assembly body of a pass (graph_framework_amd/csrc/asm_body.hpp, GFHIP_ASM=1) checked on the CPU: the
statement the lowering writes is replayed on symbolic values (tests/asm_symbolic.py) — every definition against the
item's DAG, every load waited for, every LDS round trip returning what was sent.  The bits of the machine sequences are
the GPU tests' business (tests/test_gpu_fuzz.py::test_assembly_body_is_bit_exact, tests/test_gpu_parity.py)."""
import os
import subprocess
import sys

import pytest

import asm_symbolic
import gfir_random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = os.path.join(ROOT, "graph_framework_amd", "workloads")


def lowered(blob_path, environment):
    """(piece GFIR, kernel text) of the item in a child process: the lowering reads its options from the environment."""
    script = ("import sys, json, base64; sys.path.insert(0, %r)\n"
              "from graph_framework_amd.backend import generate_piece_sources, export_pieces\n"
              "blob = open(%r, 'rb').read()\n"
              "print(json.dumps([base64.b64encode(export_pieces(blob)[0]['gfir']).decode(), generate_piece_sources(blob)[0][0]]))" % (ROOT, blob_path))
#  (what the order search remembers goes to a scratch directory, not into the in-tree kernel cache)
    import tempfile
    scratch = tempfile.mkdtemp(prefix="gfhip_orders_")
    env = dict(os.environ, GFHIP_ASM="1", **dict(dict(GFHIP_CACHE_DIR=scratch), **environment))
    out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, check=True)
    import base64
    import json
    piece, text = json.loads(out.stdout)
    return base64.b64decode(piece), text


RANDOM_CASES = [  # seed, inputs, nodes, first register of the pool, LDS budget of the staged tables
    (42, 6, 150, 64, 65536), (42, 6, 150, 224, 0), (46, 6, 300, 200, 65536), (47, 8, 500, 160, 0), (42, 8, 700, 64, 0),
    (51, 5, 400, 176, 65536), (52, 8, 600, 128, 0), (53, 3, 250, 216, 65536),
]


@pytest.mark.parametrize("seed,inputs,nodes,pool,budget", RANDOM_CASES, ids=["%d-nodes-pool-%d-lds-%d" % c[2:] for c in RANDOM_CASES])
def test_random_items_replay(tmp_path, seed, inputs, nodes, pool, budget):
    blob, _ = gfir_random.random_item(seed, "f64", inputs, nodes, 3, 3)
    path = tmp_path/"item.gfir"
    path.write_bytes(blob)
#  (one wave per SIMD: 80 slots per lane — the small pools send most values through them)
    piece, text = lowered(str(path), dict(GFHIP_ASM_MIN_NODES="0", GFHIP_ASM_POOL_LO=str(pool), GFHIP_LDS_BUDGET=str(budget), GFHIP_ASM_WAVES="1"))
    assert "v_rcp_f64" in text, "the item kept the compiled body"
    stats = asm_symbolic.replay(piece, text)
    assert stats["definitions"] >= nodes//2
    if pool > 64:
        assert stats["spills"] > 0 and stats["fills"] > 0


@pytest.mark.parametrize("pool", [40, 64])
def test_rk4_item_replays(pool):
    """The benchmark's solver_kernel (3878 records, 680 divisions over 82 denominators, 360 gathers of 8 cells, sqrt and
    pow(x, 1.5)): 5948 vector instructions against the ~6430 hipcc issues for the same pass, no register copies."""
    piece, text = lowered(os.path.join(WORKLOADS, "solver_kernel_f64.gfir"), dict(GFHIP_ASM_POOL_LO=str(pool)))
    assert "v_rcp_f64" in text
    stats = asm_symbolic.replay(piece, text)
    assert stats["definitions"] > 3500 and stats["loads"] >= 80
    vector = sum(1 for line in asm_symbolic.statement_of(text) if line.startswith("v_"))
    assert vector < 6100


def test_a_corrupted_statement_is_caught():
    """The replay is not vacuous: dropping one wait, or swapping two operands of one instruction, fails it."""
    piece, text = lowered(os.path.join(WORKLOADS, "solver_kernel_f64.gfir"), {})
    lines = text.split("\n")
    waits = [k for k, line in enumerate(lines) if '"s_waitcnt vmcnt(' in line]
    broken = lines[:waits[3]] + lines[waits[3] + 1:]
    with pytest.raises(asm_symbolic.ReplayError):
        asm_symbolic.replay(piece, "\n".join(broken))
    target = next(k for k, line in enumerate(lines) if '"v_fma_f64 v[' in line and "; def r" in line and line.count("v[") == 4)
    import re
    registers = re.findall(r"v\[\d+:\d+\]", lines[target])
    swapped = lines[target].replace(registers[1], "@").replace(registers[3], registers[1]).replace("@", registers[3])
    assert swapped != lines[target]
    with pytest.raises(asm_symbolic.ReplayError):
        asm_symbolic.replay(piece, "\n".join(lines[:target] + [swapped] + lines[target + 1:]))


def test_the_chosen_order_is_remembered_and_reproducible(tmp_path):
    """schedule_for_assembly searches 64 tie-breaks once and leaves `<hash>.order` next to the code objects; a second
    lowering reads it and writes the same kernel text — and so does a lowering that cannot find it (the search is
    deterministic), which is what the GPU box relies on when the cache directory does not travel."""
    workload = os.path.join(WORKLOADS, "solver_kernel_f64.gfir")
    first_cache, second_cache = tmp_path/"a", tmp_path/"b"
    first_cache.mkdir()
    second_cache.mkdir()
#  (eight tie-breaks instead of 64: a key of its own, nothing remembered for it in the in-tree kernel cache)
    knobs = dict(GFHIP_ASM_TRIES="8")
    _, first = lowered(workload, dict(knobs, GFHIP_CACHE_DIR=str(first_cache)))
    remembered = [name for name in os.listdir(first_cache) if name.endswith(".order")]
    assert len(remembered) == 1 and int(open(first_cache/remembered[0]).read()) >= 0
    _, again = lowered(workload, dict(knobs, GFHIP_CACHE_DIR=str(first_cache)))
    _, fresh = lowered(workload, dict(knobs, GFHIP_CACHE_DIR=str(second_cache)))
    assert again == first and fresh == first
