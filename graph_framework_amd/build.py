"""Build steps: libgf_hip.so (hipcc, gfx950) and the pre-built kernel cache.

Everything is built IN-TREE so that it travels with the repository snapshot:
    graph_framework_amd/libgf_hip.so
    graph_framework_amd/kernel_cache/<source hash>.hsaco   (one per workload item)
    graph_framework_amd/kernel_cache/<source hash>.hip     (the generated source, for inspection)
hipcc cross-compiles gfx950 without a GPU.
"""
import glob
import os
import subprocess

from . import _lib

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
KERNEL_FLAGS = ["-O3", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950"]


def build_library(force=False):
    csrc = os.path.join(_lib.HERE, "csrc")
    sources = glob.glob(os.path.join(csrc, "*")) + glob.glob(os.path.join(_lib.HERE, "..", "include", "*.h"))
    newest = max(os.path.getmtime(s) for s in sources)
    programs = [os.path.join(_lib.HERE, name) for name in ("xrays_bench", "solver_check", "korc_push")]
    sources.append(os.path.join(_lib.HERE, "gf_workflow.hpp"))
    newest = max(os.path.getmtime(s) for s in sources)
    if (force or not os.path.exists(_lib.LIB_PATH) or os.path.getmtime(_lib.LIB_PATH) < newest
            or any(not os.path.exists(p) or os.path.getmtime(p) < newest for p in programs)):
        subprocess.check_call(["make", "-C", csrc, "-s", "HIPCC=" + HIPCC])
    return _lib.LIB_PATH


def compile_source(source, source_hash, force=False):
    """One generated kernel text -> a gfx950 code object in the kernel cache (named by its hash)."""
    os.makedirs(_lib.CACHE_DIR, exist_ok=True)
    stem = os.path.join(_lib.CACHE_DIR, "%016x" % source_hash)
    if force or not os.path.exists(stem + ".hsaco"):
        with open(stem + ".hip", "w") as f:
            f.write(source)
        subprocess.check_call([HIPCC, "--genco"] + KERNEL_FLAGS + ["-o", stem + ".hsaco", stem + ".hip"])
    return stem + ".hsaco"


def prebuild_kernel(gfir_path, force=False):
    """Lower one workload item and compile it — every segment of it, for an item the lowering cuts
    into segments — to gfx950 code objects in the kernel cache."""
    from . import backend
    return [compile_source(source, source_hash, force) for source, source_hash in backend.generate_piece_sources(gfir_path)]


#  Kernels of non-default lowerings that tests and bench.py use on the GPU box (each would otherwise cost the box a
#  hipRTC build): (environment, workload file stem).
VARIANTS = [({"GFHIP_DIVISION": "fast"}, "solver_kernel_f64"), ({"GFHIP_DIVISION": "fast"}, "loss_kernel_kx_f64"),
            ({"GFHIP_ASM": "0"}, "solver_kernel_f64")]


def variant_sources():
    """Kernel texts of VARIANTS: lowered in a child process each, because the lowering reads its options from the
    environment."""
    import json
    import sys
    texts = []
    for overrides, stem in VARIANTS:
        env = dict(os.environ)
        env.update(overrides)
        out = subprocess.run([sys.executable, "-c",
                              "import sys, json; sys.path.insert(0, %r)\n"
                              "from graph_framework_amd.backend import generate_piece_sources\n"
                              "print(json.dumps(generate_piece_sources(%r)))"
                              % (os.path.dirname(_lib.HERE), os.path.join(_lib.WORKLOAD_DIR, stem + ".gfir"))],
                             env=env, capture_output=True, text=True, check=True)
        texts += [tuple(piece) for piece in json.loads(out.stdout)]
    return texts


def prebuild_workloads(force=False):
#  hipcc takes seconds to a minute per kernel: a few at a time.  The unit of work is one kernel text,
#  so the segments of a large item (a ray step on the 86-mode VMEC equilibrium) compile side by side.
    from concurrent.futures import ThreadPoolExecutor
    from . import backend
    paths = sorted(glob.glob(os.path.join(_lib.WORKLOAD_DIR, "*.gfir")))
    texts = [piece for path in paths for piece in backend.generate_piece_sources(path)] + variant_sources()
    workers = max(1, min(8, (os.cpu_count() or 4)))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        built = list(pool.map(lambda piece: compile_source(piece[0], piece[1], force), texts))
#  Drop code objects of earlier lowerings (the cache key is the source hash).
    keep = {os.path.splitext(b)[0] for b in built}
#  (`<hash>.order` files stay: the emission order the assembly body's search chose for an item, asm_body.hpp)
    for stale in glob.glob(os.path.join(_lib.CACHE_DIR, "*")):
        if os.path.splitext(stale)[0] not in keep and not stale.endswith(".order"):
            os.remove(stale)
    return built


def build_all(force=False):
    build_library(force)
    return prebuild_workloads(force)
