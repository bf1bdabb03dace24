"""Differential fuzzing of the lowering: random work items (tests/gfir_random.py) on the CPU
oracle and on the GPU, bit for bit.

The items use only node types whose device arithmetic is IEEE-exact, so every transformation of
the lowering — pressure-aware emission order, shared reciprocals, gather index groups, exact
table compaction, LDS staging and parking, fused steps — has to leave every bit unchanged.
Kernels are built by hipRTC at run time (nothing in the kernel cache matches a random item).
"""
import numpy as np
import pytest

import gfir_random
from oracle import gfir

pytestmark = pytest.mark.gpu

CASES = [  # seed, dtype, inputs, nodes, outputs, setters, rays
    (11, "f64", 6, 120, 2, 2, 1000),
    (12, "f64", 8, 700, 3, 5, 4099),
    (13, "f64", 3, 1500, 4, 3, 777),
    (14, "f64", 8, 3000, 2, 7, 2048),
    (15, "f64", 5, 400, 5, 0, 1),
    (16, "f32", 6, 300, 3, 3, 1000),
    (17, "f32", 8, 1200, 2, 6, 3001),
    (18, "f64", 7, 5000, 3, 7, 640),
]


@pytest.mark.parametrize("seed,dtype,inputs,nodes,outputs,setters,rays", CASES,
                         ids=["%s-%d-nodes" % (c[1], c[3]) for c in CASES])
def test_random_work_item_bit_exact(seed, dtype, inputs, nodes, outputs, setters, rays):
    from graph_framework_amd import Context
    blob, _ = gfir_random.random_item(seed, dtype, inputs, nodes, outputs, setters)
    oracle_item = gfir.Item(blob)
    rng = np.random.default_rng(1000 + seed)
    initial = [rng.uniform(-1.0, 1.0, rays).astype(oracle_item.np_dtype) for _ in range(inputs)]

    context = Context(0)
    in_keys = ["in%d" % i for i in range(inputs)]
    out_keys = ["out%d" % i for i in range(outputs)]
    kernel = context.add_kernel(blob, rays)
    context.compile()
    kernel.create_kernel_call(in_keys, out_keys, initial)

    expected = [c.copy() for c in initial]
    for launch_steps in (1, 1, 3):                       # separate launches, then three fused passes
        expected_out, _ = oracle_item.run(expected, steps=launch_steps)
        kernel.run(launch_steps)
        context.wait()
        assert context.flags() == 0
        for key, want in zip(in_keys + out_keys, expected + expected_out):
            got = context.copy_to_host(key, np.empty(rays, dtype=oracle_item.np_dtype))
            assert np.array_equal(got, want), (key, launch_steps, np.flatnonzero(got != want)[:5])
    context.close()


@pytest.mark.parametrize("option,value", [("GFHIP_DIVISION", "checked"), ("GFHIP_DIVISION", "ieee"), ("GFHIP_SCHEDULE", "source"),
                                          ("GFHIP_PARK", "heavy"), ("GFHIP_LDS_BUDGET", "0"), ("GFHIP_COMPACT_TABLES", "0"),
                                          ("GFHIP_SEGMENT_NODES", "300"), ("GFHIP_SEGMENTS", "3"), ("GFHIP_CONVERGE_BATCH", "1"),
                                          ("GFHIP_WINDOW_SQRT", "1")])
def test_alternative_lowerings_are_bit_exact(monkeypatch, tmp_path, option, value):
    """Every knob options.hpp still offers computes the same bits as the default lowering.  GFHIP_SEGMENT_NODES=300
    cuts both items into segments by size (compiler's division), GFHIP_SEGMENTS=3 into three segments with the
    shared-reciprocal division and a redo launch (csrc/segments.hpp)."""
    from graph_framework_amd import Context
    monkeypatch.setenv(option, value)
    if option == "GFHIP_SEGMENTS":
        monkeypatch.setenv("GFHIP_SEGMENTS_MIN_NODES", "100")
    monkeypatch.setenv("GFHIP_CACHE_DIR", str(tmp_path))
    for seed, dtype, nodes, rays in ((31, "f64", 2200, 777), (32, "f32", 500, 130)):
        blob, _ = gfir_random.random_item(seed, dtype, 6, nodes, 3, 4)
        oracle_item = gfir.Item(blob)
        rng = np.random.default_rng(seed)
        initial = [rng.uniform(-1.0, 1.0, rays).astype(oracle_item.np_dtype) for _ in range(6)]
        context = Context(0)
        kernel = context.add_kernel(blob, rays)
        context.compile()
        in_keys, out_keys = ["in%d" % i for i in range(6)], ["out%d" % i for i in range(3)]
        kernel.create_kernel_call(in_keys, out_keys, initial)
        if option.startswith("GFHIP_SEGMENT"):
            assert kernel.info().segments >= 2
        expected = [c.copy() for c in initial]
        for launch_steps in (1, 2):
            expected_out, _ = oracle_item.run(expected, steps=launch_steps)
            kernel.run(launch_steps)
            context.wait()
#  (`checked` also sends fp64 lanes with a numerator below 2^-450 through the IEEE function)
            assert context.flags() == 0 or value == "checked"
            for key, want in zip(in_keys + out_keys, expected + expected_out):
                got = context.copy_to_host(key, np.empty(rays, dtype=oracle_item.np_dtype))
                assert np.array_equal(got, want), (option, value, key, launch_steps)
        context.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_segments_with_a_redo_launch_on_the_division_edges(monkeypatch, tmp_path, dtype):
    """csrc/segments.hpp with the redo launch where it matters: the division stress item cut into
    three segments, on the operands of tests/test_gpu_division.py (denominators of 2^+-600, zeros and infinities of
    both signs, infinite and overflowing numerators).  Lanes outside the window are flagged by the segment that sees
    them, skip their stores in the last segment, line up in the redo list and are redone by the redo kernel with the
    compiler's division: every lane bit-identical to the oracle, over several chunks of rays (the hand-over buffers hold
    16 384 rays; 40 000 are traced) and over two passes."""
    from graph_framework_amd import Context
    from test_gpu_division import _operands, _same
    monkeypatch.setenv("GFHIP_SEGMENTS", "3")
    monkeypatch.setenv("GFHIP_SEGMENTS_MIN_NODES", "10")
    monkeypatch.setenv("GFHIP_HANDOVER_BYTES", "1")
    monkeypatch.setenv("GFHIP_CACHE_DIR", str(tmp_path))
    blob = gfir_random.division_stress_item(dtype)
    oracle_item = gfir.Item(blob)
    columns = _operands(dtype, tiny_numerators=(dtype == "f32"))
    repeat = 40000//columns[0].size + 1
    columns = [np.tile(c, repeat)[:40000].copy() for c in columns]
    rays = columns[0].size
    context = Context(0)
    kernel = context.add_kernel(blob, rays)
    context.compile()
    in_keys, out_keys = ["n0", "n1", "d0", "d1", "x"], ["q0", "q1", "q2", "q3", "mix"]
    kernel.create_kernel_call(in_keys, out_keys, columns)
    assert kernel.info().segments == 3
    expected = [c.copy() for c in columns]
    with np.errstate(all="ignore"):
        for _ in range(2):
            expected_out, _ = oracle_item.run(expected)
            kernel.run(1)
            context.wait()
            for key, want in zip(in_keys + out_keys, expected + expected_out):
                got = context.copy_to_host(key, np.empty(rays, dtype=oracle_item.np_dtype))
                assert _same(got, want), key
    assert context.flags() & 1                                       # lanes did leave the window
    context.close()


ASSEMBLY_CASES = [  # seed, inputs, nodes, first register of the pool, waves per SIMD, LDS budget of the staged tables
    (42, 6, 150, 64, 1, 65536),
    (42, 6, 150, 224, 1, 65536),        # 16 register pairs: nearly every value goes through an LDS slot
    (42, 6, 150, 224, 2, 0),            # ... and the tables are read from global memory
    (46, 6, 300, 200, 1, 65536),
    (47, 8, 500, 160, 1, 0),
    (42, 8, 700, 64, 1, 65536),
    (42, 8, 700, 64, 1, 0),
]


@pytest.mark.parametrize("seed,inputs,nodes,pool,waves,budget", ASSEMBLY_CASES,
                         ids=["%d-nodes-pool-%d-waves-%d-lds-%d" % c[2:] for c in ASSEMBLY_CASES])
def test_assembly_body_is_bit_exact(monkeypatch, tmp_path, seed, inputs, nodes, pool, waves, budget):
    """csrc/asm_body.hpp (GFHIP_ASM=1): the body of a pass as gfx950 assembly with a register assignment of its own —
    values sent to LDS slots and read back, table values loaded again (from LDS-staged and from global packs),
    constants in an SGPR pool, counted waits — computes the bits of the oracle on random items, for separate launches
    and fused passes; small register pools force the traffic through the slots that the RK4 item only sees at its
    peaks."""
    from graph_framework_amd import Context
    from graph_framework_amd.backend import generate_piece_sources
    monkeypatch.setenv("GFHIP_ASM", "1")
    monkeypatch.setenv("GFHIP_ASM_MIN_NODES", "0")
    monkeypatch.setenv("GFHIP_ASM_POOL_LO", str(pool))
    monkeypatch.setenv("GFHIP_ASM_WAVES", str(waves))
    monkeypatch.setenv("GFHIP_LDS_BUDGET", str(budget))
    monkeypatch.setenv("GFHIP_CACHE_DIR", str(tmp_path))
    outputs, setters, rays = 3, 3, 1777
    blob, _ = gfir_random.random_item(seed, "f64", inputs, nodes, outputs, setters)
    text = generate_piece_sources(blob)[0][0]
    assert "v_fma_f64" in text and "v_rcp_f64" in text                  # the assembly body, not the compiled one
    if pool > 64:
        assert "ds_write_b64" in text and "ds_read_b64" in text
    if budget == 0:
        assert "global_load_dwordx2" in text
    oracle_item = gfir.Item(blob)
    rng = np.random.default_rng(2000 + seed)
    initial = [rng.uniform(-1.0, 1.0, rays).astype(np.float64) for _ in range(inputs)]
    context = Context(0)
    in_keys = ["in%d" % i for i in range(inputs)]
    out_keys = ["out%d" % i for i in range(outputs)]
    kernel = context.add_kernel(blob, rays)
    context.compile()
    kernel.create_kernel_call(in_keys, out_keys, initial)
    expected = [c.copy() for c in initial]
    for launch_steps in (1, 1, 3):
        expected_out, _ = oracle_item.run(expected, steps=launch_steps)
        kernel.run(launch_steps)
        context.wait()
        assert context.flags() == 0
        for key, want in zip(in_keys + out_keys, expected + expected_out):
            got = context.copy_to_host(key, np.empty(rays, dtype=np.float64))
            assert np.array_equal(got, want), (key, launch_steps, np.flatnonzero(got != want)[:5])
    context.close()


def test_assembly_body_on_the_division_edges(monkeypatch, tmp_path):
    """The window check of the assembly body (denominators and root arguments folded into dmax / dmin two at a time, the
    index quotients into vmax) and its redo launch, where they matter: the division stress item on the operands of
    tests/test_gpu_division.py (denominators of 2^+-600, zeros and infinities of both signs, infinite and overflowing
    numerators).  Lanes outside the window skip their stores, line up in the redo list and are redone with the
    compiler's division: every lane bit-identical to the oracle, over two passes."""
    from graph_framework_amd import Context
    from graph_framework_amd.backend import generate_piece_sources
    from test_gpu_division import _operands, _same
    monkeypatch.setenv("GFHIP_ASM_MIN_NODES", "0")
    monkeypatch.setenv("GFHIP_CACHE_DIR", str(tmp_path))
    blob = gfir_random.division_stress_item("f64")
    assert "v_rcp_f64_e32" in generate_piece_sources(blob)[0][0]
    oracle_item = gfir.Item(blob)
    columns = _operands("f64", tiny_numerators=False)
    repeat = 40000//columns[0].size + 1
    columns = [np.tile(c, repeat)[:40000].copy() for c in columns]
    rays = columns[0].size
    context = Context(0)
    kernel = context.add_kernel(blob, rays)
    context.compile()
    in_keys, out_keys = ["n0", "n1", "d0", "d1", "x"], ["q0", "q1", "q2", "q3", "mix"]
    kernel.create_kernel_call(in_keys, out_keys, columns)
    assert kernel.info().segments == 1 and kernel.info().vgprs <= 256
    expected = [c.copy() for c in columns]
    with np.errstate(all="ignore"):
        for _ in range(2):
            expected_out, _ = oracle_item.run(expected)
            kernel.run(1)
            context.wait()
            for key, want in zip(in_keys + out_keys, expected + expected_out):
                got = context.copy_to_host(key, np.empty(rays, dtype=np.float64))
                assert _same(got, want), key
    assert context.flags() & 1                                       # lanes did leave the window
    context.close()
