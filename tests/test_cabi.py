"""CPU tests of the drop-in boundary: libgf_hip.so loads, exports every symbol that
include/gf_hip.h declares, and lowers work items without a device.  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, WORKLOADS


@pytest.fixture(scope="module")
def lib():
    from graph_framework_amd import build, _lib
    build.build_library()
    return _lib.load()


def declared_symbols():
    with open(os.path.join(ROOT, "include", "gf_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gfhip_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    from graph_framework_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(s[0] for s in _lib.SYMBOLS) == names


def test_no_device_reports_cleanly(lib):
    """In the CPU container there is no GPU: creation must fail with a message, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from graph_framework_amd import Context, GfHipError
    with pytest.raises(GfHipError):
        Context(0)


def test_lowering_without_a_device(lib, monkeypatch):
    from graph_framework_amd import generate_source
    monkeypatch.setenv("GFHIP_ASM", "0")                 # the compiled body (the default is csrc/asm_body.hpp, below)
    source, source_hash = generate_source(os.path.join(WORKLOADS, "solver_kernel_f64.gfir"))
    assert "gfhip_solver_kernel" in source and source_hash != 0
    # 8 gather index groups per RK4 step (4 stages x {(r,z) cell, psi bin}) instead of 360 index expressions
    assert len(set(re.findall(r"const real \*const (g\d+) =", source))) == 8
    # tables that are exact multiples of another are not stored: 45 psi tables -> 20, 45 profile tables -> 12
    psi_groups = set(re.findall(r"const real \*const (g\d+) = pack0 \+", source))
    profile_groups = set(re.findall(r"const real \*const (g\d+) = lds1 \+", source))
    assert len(psi_groups) == 4 and len(profile_groups) == 4
    loads = re.findall(r"= (g\d+)\[(\d+)u\];", source)
    assert len({column for group, column in loads if group in psi_groups}) == 20
    assert len({column for group, column in loads if group in profile_groups}) == 12
    # one reciprocal per distinct denominator (82), 680 divisions through it (+ the 12 index quotients)
    assert len(re.findall(r"= gf_rcp\(", source)) == 82
    assert len(re.findall(r"const real r\d+ = gf_div\(r", source)) == 680
    assert len(re.findall(r"const real x\d+ = gf_div\(r", source)) == 12
    # the IEEE function: the same pass with the compiler's division, for lanes that fail a check
    assert len(re.findall(r"const real r\d+ = r\d+(?:p\d+)?/r\d+(?:p\d+)?;", source)) == 680
    assert source.count("gfhip_solver_kernel_ieee(") == 2 and source.count("if (__builtin_expect(bad || zero, 0))") == 1
    again, again_hash = generate_source(os.path.join(WORKLOADS, "solver_kernel_f64.gfir"))
    assert again == source and again_hash == source_hash


def test_default_lowering_of_the_rk4_item_is_the_assembly_body(lib):
    """solver_kernel by default: one kernel whose pass is an assembly statement (two waves per SIMD, 256 registers, no
    IEEE function inside) plus the redo kernel with the compiler's division for the lanes that leave the window."""
    from graph_framework_amd.backend import generate_piece_sources
    pieces = generate_piece_sources(os.path.join(WORKLOADS, "solver_kernel_f64.gfir"))
    assert len(pieces) == 2
    kernel, redo = pieces[0][0], pieces[1][0]
    assert "__launch_bounds__(256, 2)\ngfhip_solver_kernel(" in kernel and "gfhip_solver_kernel_redo(" in redo
    assert "_ieee(" not in kernel and "redo_list" in kernel
    statement = kernel[kernel.index("asm volatile(\n", kernel.index("float dmax")):]
    assert len(re.findall(r"; def g\d+ ", statement)) == 8                     # the 8 cells of a step
    assert len(re.findall(r"v_rcp_f64_e32 .*; def|v_fma_f64 .*; def q\d+", statement)) == 82      # one reciprocal per denominator
    assert len(re.findall(r"v_rcp_f64_e32", statement)) == 82 + 4              # ... and the four pow(x, 1.5)
    assert len(re.findall(r"v_rsq_f64_e32", statement)) == 11 + 4
#  20 psi columns x 4 stages, neighbouring columns in one 16-byte load where both are wanted soon
    narrow, wide = len(re.findall(r"global_load_dwordx2", statement)), len(re.findall(r"global_load_dwordx4", statement))
    assert 80 <= narrow + 2*wide <= 100 and wide >= 30
    assert "v_accvgpr" not in statement and "scratch_" not in statement
    vector = len(re.findall(r'^\s*"v_', statement, re.M))
    assert vector < 6000                                                        # hipcc: ~6430 for the same pass
    assert len(re.findall(r"const real r\d+ = r\d+(?:p\d+)?/r\d+(?:p\d+)?;", redo)) == 680
    assert generate_piece_sources(os.path.join(WORKLOADS, "solver_kernel_f64.gfir")) == pieces


def test_malformed_items_are_rejected(lib):
    from graph_framework_amd import generate_source, GfHipError
    with open(os.path.join(WORKLOADS, "korc_initialize_gamma_f64.gfir"), "rb") as f:
        good = f.read()
    generate_source(good)
    huge_counts = b"GFIR0001" + bytes([1, 0, 0, 0]) + bytes([255]*20) + bytes(100)
    for bad in (b"", b"not a work item", good[:40], good[:-3], huge_counts):
        with pytest.raises(GfHipError):
            generate_source(bad)
    corrupt = bytearray(good)
    corrupt[8] = 7                       # dtype
    with pytest.raises(GfHipError):
        generate_source(bytes(corrupt))


def test_table_compaction_is_exact(monkeypatch):
    """Every table the lowering derives as k*parent must equal the exported table bit for bit."""
    import struct
    from graph_framework_amd import generate_source
    path = os.path.join(WORKLOADS, "solver_kernel_f64.gfir")
    assembly, _ = generate_source(path)
    monkeypatch.setenv("GFHIP_ASM", "0")
    source, _ = generate_source(path)
    data = open(path, "rb").read()
    magic, dtype, ni, no, ns, nt, nins, nb, _r = struct.unpack_from("<8s8I", data, 0)
    pos = 40 + nb
    for _ in range(ni):
        (n,) = struct.unpack_from("<I", data, pos)
        pos += 4 + n
    tables = []
    for _ in range(nt):
        r, c = struct.unpack_from("<II", data, pos)
        pos += 8
        tables.append(np.frombuffer(data, dtype="<f8", count=r*c, offset=pos))
        pos += 8*r*c
    derived = re.findall(r"const real c(\d+)_(\d+) = (\S+)\*c\d+_(\d+);", source)
    assert len(derived) > 100
    for _group, child, factor, parent in derived:
        k = float.fromhex(factor)
        np.testing.assert_array_equal(k*tables[int(parent)], tables[int(child)])
#  the assembly body multiplies the same pairs (its annotations carry the factor's bits)
#  (table numbers are those of the piece the kernel is lowered from)
    from graph_framework_amd.backend import export_pieces
    from oracle.gfir_to_c import parse
    monkeypatch.delenv("GFHIP_ASM")
    piece_tables = [t[2] for t in parse(export_pieces(path)[0]["gfir"])["tables"]]
    multiplied = re.findall(r"; def c\d+_(\d+) = c\d+_(\d+) \* (\d+)", assembly)
    assert len(multiplied) > 100
    for child, parent, factor_bits in multiplied:
        k = struct.unpack("<d", struct.pack("<Q", int(factor_bits)))[0]
        np.testing.assert_array_equal(k*piece_tables[int(parent)], piece_tables[int(child)])


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No CPU fallback: without libgf_hip.so nothing in the package can compute."""
    from graph_framework_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libgf_hip.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
    from graph_framework_amd import generate_source
    with pytest.raises(ImportError):
        generate_source(os.path.join(WORKLOADS, "korc_initialize_gamma_f64.gfir"))


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: no module of graph_framework_amd may import it."""
    import glob
    package = os.path.join(ROOT, "graph_framework_amd")
    for path in glob.glob(os.path.join(package, "*.py")):
        with open(path) as f:
            text = f.read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), path
    for path in glob.glob(os.path.join(package, "csrc", "*")):
        with open(path) as f:
            assert "oracle/" not in f.read(), path


def test_result_file_roundtrip(tmp_path):
    """output.ResultFile writes (time, num_rays, 1) variables with an unlimited time dimension
    (output.hpp:260-273, :354-400); read back with the fixture reader."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_fixtures import H5File
    from graph_framework_amd.output import ResultFile
    path = str(tmp_path / "result0.nc")
    out = ResultFile(path, 5)
    for name in ("time", "x"):
        out.create_variable(name)
    for record in range(3):
        out.write({"time": np.full(5, 0.5*record), "x": np.arange(5.0) + record})
    out.close()
    f = H5File(path)
    x = f.read("x")
    time = f.read("_nc4_non_coord_time")          # netCDF-4's name for a variable called like a dimension
    assert f.read("time").shape == (3,)           # the dimension `time` itself: its length is the record count
    f.close()
    assert x.shape == (3, 5, 1) and time.shape == (3, 5, 1)
    np.testing.assert_array_equal(x[2, :, 0], np.arange(5.0) + 2)
    np.testing.assert_array_equal(time[1, :, 0], np.full(5, 0.5))


def test_random_items_lower_to_valid_gfx950_code(lib, tmp_path):
    """The fuzz generator's items (tests/gfir_random.py) parse, lower deterministically and
    cross-compile for gfx950 without a GPU (the GPU run is tests/test_gpu_fuzz.py)."""
    import subprocess
    import gfir_random
    from graph_framework_amd import generate_source
    from graph_framework_amd.build import HIPCC, KERNEL_FLAGS
    blob, records = gfir_random.random_item(5, "f64", num_nodes=600)
    source, source_hash = generate_source(blob)
    assert generate_source(blob) == (source, source_hash)
    # every record is emitted exactly once whatever the emission order
    assert len(set(re.findall(r"const real (r\d+) =", source))) >= records
    path = tmp_path / "fuzz.hip"
    path.write_text(source)
    subprocess.check_call([HIPCC, "--genco"] + KERNEL_FLAGS + ["-o", str(tmp_path / "fuzz.hsaco"), str(path)])


def _defined_before_use(source, kernel):
    """Every value of the pass is defined once, before its first use (the emission order is a
    topological order of the DAG whatever the scheduler did)."""
    body = source[source.index(kernel + "("):]
#  the first body of the first entry point (the IEEE second body and the other entry points
#  repeat the same pass)
    for marker in ("if (__builtin_expect(bad || zero, 0))", 'extern "C" __global__'):
        if marker in body:
            body = body[:body.index(marker)]
    defined = set()
    count = 0
    for line in body.splitlines():
        match = re.match(r"\s*const real (r\d+(?:p\d+)?) = (.*);$", line)
        if not match:
            continue
        name, expression = match.groups()
        for used in re.findall(r"\br\d+(?:p\d+)?\b", expression):
            assert used in defined, (name, used)
        assert name not in defined, name
        defined.add(name)
        count += 1
    return count


@pytest.mark.parametrize("schedule", ["greedy", "source"])
def test_emission_order_is_topological(lib, monkeypatch, schedule):
    import gfir_random
    from graph_framework_amd import generate_source
    monkeypatch.setenv("GFHIP_SCHEDULE", schedule)
    monkeypatch.setenv("GFHIP_ASM", "0")                 # the compiled body's text (tests/test_asm_body.py replays the assembly one)
    source, _ = generate_source(os.path.join(WORKLOADS, "solver_kernel_f64.gfir"))
    assert _defined_before_use(source, "gfhip_solver_kernel") >= 3878
    blob, records = gfir_random.random_item(9, "f64", num_nodes=1500)
    source, _ = generate_source(blob)
#  (a derived-table gather nobody reads is never defined: the fuzz items have a few dead nodes)
    assert _defined_before_use(source, "gfhip_fuzz") >= 0.95*records


def test_mutated_items_never_crash_the_lowering(lib):
    """tests/gfir_mutate.py: 400 truncated / bit-flipped / field-corrupted items are lowered or
    rejected; the child process must exit normally."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gfir_mutate.py")
    out = subprocess.run([sys.executable, script, "7", "400"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rejected" in out.stdout


def test_lowering_under_address_and_ub_sanitizers(tmp_path):
    """The parser, scheduler and lowering (host code, no HIP runtime) built with
    -fsanitize=address,undefined: every exported workload and 600 mutated items."""
    import glob
    import subprocess
    from conftest import ROOT
    binary = str(tmp_path / "lowering_sanitize")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-o", binary,
                           os.path.join(ROOT, "tests", "lowering_sanitize.cpp")])
    workloads = sorted(glob.glob(os.path.join(WORKLOADS, "*.gfir")))
    out = subprocess.run([binary] + workloads + ["--mutate", "5", "300", os.path.join(WORKLOADS, "loss_kernel_kx_f64.gfir"),
                                                 "--mutate", "6", "300", os.path.join(WORKLOADS, "korc_step_f32.gfir")],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("lowered")


def test_result_file_has_the_reference_netcdf4_dimensions(tmp_path):
    """graph_framework_amd/output.py writes result<n>.nc in NetCDF-4's on-disk conventions, so that
    the reference's readers find what they look up: nc_inq_dimid "time" / "num_rays"
    (output.hpp:77-78), "ray_dim" (:189-193), variables of dimensions (time, num_rays, ray_dim)
    (:260-273, :218-232).  No NetCDF library exists in the image; the header is checked against
    the structure netCDF-C itself wrote into the reference's fixtures (cf. `h5dump -H -A` of
    graph_tests/efit.nc: CLASS, NAME, _Netcdf4Dimid, DIMENSION_LIST, REFERENCE_LIST)."""
    import subprocess
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_fixtures import H5File
    from graph_framework_amd.output import ResultFile, RAY_VARIABLES
    h5dump = "/opt/conda/bin/h5dump"
    if not os.path.exists(h5dump):
        pytest.skip("h5dump not available")
    path = str(tmp_path / "result0.nc")
    n = 7
    result = ResultFile(path, n)
    for name, _ in RAY_VARIABLES:
        result.create_variable(name)
    records = [{name: np.arange(n) + 10.0*r + i for i, (name, _) in enumerate(RAY_VARIABLES)} for r in range(4)]
    for record in records:
        result.write(record)
    result.close()

    header = subprocess.run([h5dump, "-H", "-A", path], capture_output=True, text=True, check=True).stdout
    flat = " ".join(header.split())
    assert 'DATASET "time" { DATATYPE H5T_IEEE_F32BE DATASPACE SIMPLE { ( 4 ) / ( H5S_UNLIMITED ) }' in flat
    assert 'DATASET "num_rays" { DATATYPE H5T_IEEE_F32BE DATASPACE SIMPLE { ( 7 ) / ( 7 ) }' in flat
    assert 'DATASET "ray_dim" { DATATYPE H5T_IEEE_F32BE DATASPACE SIMPLE { ( 1 ) / ( 1 ) }' in flat
    assert flat.count('"DIMENSION_SCALE"') == 3
    for length in (0, 7, 1):
        assert ('"This is a netCDF dimension but not a netCDF variable.%10d"' % length) in header
    assert flat.count('ATTRIBUTE "_Netcdf4Dimid"') == 3 and '"_NCProperties"' in flat
    for name, _ in RAY_VARIABLES:
        stored = "_nc4_non_coord_time" if name == "time" else name
        assert ('DATASET "%s" { DATATYPE H5T_IEEE_F64LE DATASPACE SIMPLE { ( 4, 7, 1 ) / ( H5S_UNLIMITED, 7, 1 ) }' % stored) in flat
    assert flat.count("(DATASET") == 3*len(RAY_VARIABLES)               # every variable lists its three scales
    assert flat.count('/time ), (DATASET') == len(RAY_VARIABLES)

    f = H5File(path)
    for i, (name, _) in enumerate(RAY_VARIABLES):
        data = f.read("_nc4_non_coord_time" if name == "time" else name)
        for r in range(4):
            np.testing.assert_array_equal(data[r, :, 0], records[r][name])
    f.close()


def test_cli_distribution_draws_the_reference_samples(lib, tmp_path):
    """graph_framework_amd.xrays.cli_distribution restates std::mt19937_64 and libstdc++'s
    std::normal_distribution by hand (csrc/cli_distribution.cpp): its samples are, bit for bit,
    those of the real objects in the reference's draw order (graph_driver/xrays.cpp:397-453) —
    committed samples (tests/golden/cli_distribution_golden.npz, make_cli_fixture.cpp) and, where
    g++ is present, a fresh run of the real objects."""
    import subprocess
    from conftest import GOLDEN
    from graph_framework_amd.xrays import cli_distribution
    golden = np.load(os.path.join(GOLDEN, "cli_distribution_golden.npz"))
    shard = int(golden["shard"])
    for seed in golden["seeds"]:
        rays = cli_distribution(shard, seed=int(seed))
        phi = np.arctan2(rays["y"], rays["x"])
        for name in ("w", "ky", "kz", "z"):
            index = golden["seed%d_%s_index" % (seed, name)]
            assert np.array_equal(rays[name][index], golden["seed%d_%s" % (seed, name)]), (seed, name)
        index = golden["seed%d_phi_index" % seed]
        want = golden["seed%d_phi" % seed]
        assert np.array_equal(rays["x"][index], 2.5*np.cos(want)) and np.array_equal(rays["y"][index], 2.5*np.sin(want))
        assert np.allclose(phi[index], want, rtol=0, atol=1.0e-15)
        assert np.all(rays["kx"] == -700.0) and np.all(rays["t"] == 0.0)
    source = os.path.join(GOLDEN, "make_cli_fixture.cpp")
    binary = str(tmp_path / "make_cli_fixture")
    if subprocess.run(["g++", "-O1", "-o", binary, source]).returncode == 0:
        text = subprocess.run([binary, "5000", "5000"], capture_output=True, text=True, check=True).stdout
        rays = cli_distribution(5000, seed=1)
        live = {v: np.array([float.fromhex(l.split()[3]) for l in text.splitlines() if l.startswith("1 %d " % v)][:5000])
                for v in range(5)}
        for v, name in ((0, "w"), (1, "ky"), (2, "kz"), (3, "z")):
            assert np.array_equal(rays[name], live[v]), name
        assert np.array_equal(rays["x"], 2.5*np.cos(live[4]))


def test_result_file_is_reopened_for_the_absorption_variables(tmp_path):
    """result_file(filename) (output.hpp:77-86) + a complex data_set (:215-224): the absorption pass
    opens the trajectory file for update, adds `kamp` over (time, num_rays, ray_dim_cplx = 2), reads
    the ray variables by time index and writes kamp at that index; bin_power reads its imaginary
    part (reference_imag_variable, :321-347) and adds `power`."""
    import subprocess
    from graph_framework_amd.output import ResultFile
    path = str(tmp_path / "result0.nc")
    n = 6
    out = ResultFile(path, n)
    for name in ("time", "x"):
        out.create_variable(name)
    for record in range(4):
        out.write({"time": np.full(n, 0.25*record), "x": np.arange(n) + 10.0*record})
    out.close()

    update = ResultFile(path)
    assert (update.num_rays, update.records) == (n, 4)
    update.create_variable("kamp", parts=2)
    for record in range(4):
        x = update.read("x", record)
        np.testing.assert_array_equal(x, np.arange(n) + 10.0*record)
        update.write({"kamp": x + 1j*(record + 0.5)}, index=record)
    update.close()

    again = ResultFile(path)
    assert again.records == 4                                          # indexed writes do not grow `time`
    again.create_variable("power")
    for record in range(4):
        np.testing.assert_array_equal(again.read("kamp", record, part=1), np.full(n, record + 0.5))
        np.testing.assert_array_equal(again.read("kamp", record), np.arange(n) + 10.0*record)
        np.testing.assert_array_equal(again.read("time", record), np.full(n, 0.25*record))
        again.write({"power": np.full(n, 1.0/(record + 1))}, index=record)
    again.close()

    h5dump = "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        flat = " ".join(subprocess.run([h5dump, "-H", "-A", path], capture_output=True, text=True, check=True).stdout.split())
        assert 'DATASET "ray_dim_cplx" { DATATYPE H5T_IEEE_F32BE DATASPACE SIMPLE { ( 2 ) / ( 2 ) }' in flat
        assert 'DATASET "kamp" { DATATYPE H5T_IEEE_F64LE DATASPACE SIMPLE { ( 4, 6, 2 ) / ( H5S_UNLIMITED, 6, 2 ) }' in flat
        assert 'DATASET "power" { DATATYPE H5T_IEEE_F64LE DATASPACE SIMPLE { ( 4, 6, 1 ) / ( H5S_UNLIMITED, 6, 1 ) }' in flat
        assert flat.count('"DIMENSION_SCALE"') == 4


def _run_pieces_on_the_oracle(pieces, columns):
    """Walk the segment sequence of a split item on the CPU oracle: each piece is an ordinary GFIR item whose
    symbols are state columns or hand-over slots and whose outputs are slots or outputs of the item; only the
    last piece has setters, so every piece reads the state of the beginning of the pass."""
    from oracle import gfir
    n = columns[0].size
    slots = [None]*pieces[0]["slots"]
    outputs = {}
    for piece in pieces:
        item = gfir.Item(piece["gfir"])
        inputs = []
        for state, slot in zip(piece["symbol_state"], piece["symbol_slot"]):
            assert (state >= 0) != (slot >= 0)
            inputs.append(columns[state] if state >= 0 else slots[slot].copy())
        outs, _ = item.run(inputs)
        for value, slot, original in zip(outs, piece["output_slot"], piece["output_original"]):
            assert (slot >= 0) != (original >= 0)
            if slot >= 0:
                slots[slot] = value
            else:
                outputs[original] = value
    assert n == columns[0].size
    return [outputs[o] for o in sorted(outputs)]


@pytest.mark.parametrize("workload,segments", [("solver_kernel_f64", 4), ("solver_kernel_f64", 7), ("korc_step_f64", 3),
                                               ("adaptive_rk4_loss_kernel_f64", 5), ("solver_kernel_f32", 3)])
def test_split_items_compute_the_same_bits_on_the_oracle(lib, monkeypatch, workload, segments):
    """csrc/segments.hpp: cutting an item into consecutive segments that hand their live values over through
    memory changes no bit.  The pieces the lowering makes (GFHIP_SEGMENTS; the same code path cuts items above
    GFHIP_SEGMENT_NODES records automatically) are exported as GFIR items and run one after the other on the
    CPU oracle, against the unsplit item on the same rays: state after the pass and outputs, two passes."""
    from graph_framework_amd.backend import export_pieces
    from oracle import gfir
    from conftest import random_plasma_state, STATE
    path = os.path.join(WORKLOADS, workload + ".gfir")
    monkeypatch.setenv("GFHIP_SEGMENTS", str(segments))
    monkeypatch.setenv("GFHIP_SEGMENT_NODES", "0")
    if "korc" in workload:
        monkeypatch.setenv("GFHIP_SEGMENTS", "0")
        monkeypatch.setenv("GFHIP_SEGMENT_NODES", "80")              # a small item: cut by size instead
    pieces = export_pieces(path)
    assert len(pieces) >= 3 and pieces[0]["pieces"] == len(pieces)
    assert all(not any(s >= 0 for s in piece["output_original"]) for piece in pieces[:-1])
    whole = gfir.Item(path)
    real = np.float32 if workload.endswith("f32") else np.float64
    n = 257
    if "korc" in workload:
        rng = np.random.default_rng(4)
        columns = [rng.uniform(1.5, 1.9, n), rng.uniform(-0.2, 0.2, n), rng.uniform(-0.3, 0.3, n), rng.uniform(-3.0, 3.0, n),
                   rng.uniform(3.0, 9.0, n), rng.uniform(-3.0, 3.0, n), rng.uniform(5.0, 11.0, n)]
    else:
        state = random_plasma_state(n, seed=9)
        columns = [state[k] for k in STATE]
        if "adaptive" in workload:
            columns += [np.full(n, 1.0e-3), np.ones(n)]
    columns = [np.ascontiguousarray(c, dtype=real) for c in columns]
    split_columns = [c.copy() for c in columns]
    for _ in range(2):
        want, _ = whole.run(columns)
        got = _run_pieces_on_the_oracle(pieces, split_columns)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert np.array_equal(a, b, equal_nan=True)
        for a, b in zip(split_columns, columns):
            assert np.array_equal(a, b, equal_nan=True)
    monkeypatch.delenv("GFHIP_SEGMENTS")
    monkeypatch.setenv("GFHIP_SEGMENT_NODES", "6000")
#  by default these items run as one kernel (the fp64 RK4 item as ONE piece whose body is assembly, plus the redo kernel)
    assert len(export_pieces(path)) <= 1
