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
KERNEL_FLAGS = ["-O3", "-ffp-contract=off", "--offload-arch=gfx950"]


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


def prebuild_kernel(gfir_path, force=False):
    """Lower one workload item and compile it to a gfx950 code object in the kernel cache."""
    from . import backend
    source, source_hash = backend.generate_source(gfir_path)
    os.makedirs(_lib.CACHE_DIR, exist_ok=True)
    stem = os.path.join(_lib.CACHE_DIR, "%016x" % source_hash)
    if force or not os.path.exists(stem + ".hsaco"):
        with open(stem + ".hip", "w") as f:
            f.write(source)
        subprocess.check_call([HIPCC, "--genco"] + KERNEL_FLAGS + ["-o", stem + ".hsaco", stem + ".hip"])
    return stem + ".hsaco"


def prebuild_workloads(force=False):
#  hipcc takes seconds to a minute per item (the 86-mode VMEC field item): a few at a time.
    from concurrent.futures import ThreadPoolExecutor
    paths = sorted(glob.glob(os.path.join(_lib.WORKLOAD_DIR, "*.gfir")))
    with ThreadPoolExecutor(max_workers=4) as pool:
        built = list(pool.map(lambda path: prebuild_kernel(path, force), paths))
#  Drop code objects of earlier lowerings (the cache key is the source hash).
    keep = {os.path.splitext(b)[0] for b in built}
    for stale in glob.glob(os.path.join(_lib.CACHE_DIR, "*")):
        if os.path.splitext(stale)[0] not in keep:
            os.remove(stale)
    return built


def build_all(force=False):
    build_library(force)
    return prebuild_workloads(force)
