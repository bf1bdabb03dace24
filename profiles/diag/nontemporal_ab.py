"""A/B of the nt hint on state loads and stores (GFHIP_NONTEMPORAL=1) for the HBM-bound items, alternating
in one process, three repetitions, at 1e7 (beyond the 256 MB Infinity Cache) and 1e6 elements (inside it)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench_extra  # noqa: E402

results = {}
for repetition in range(3):
    for nt in ("0", "1"):
        os.environ["GFHIP_NONTEMPORAL"] = nt
        os.environ["GFHIP_CACHE_DIR"] = "/tmp/nt_ab_" + nt
        for n in (10000000, 1000000):
            for name, run in (("korc_f32", lambda: bench_extra.korc("f32", n=n)), ("korc_f64", lambda: bench_extra.korc("f64", n=n)),
                              ("loss", lambda: bench_extra.loss(n=n)),
                              ("stream_f64", lambda: bench_extra.stream("f64", n=n)), ("stream7_f64", lambda: bench_extra.stream7("f64", n=n)),
                              ("stream_f32", lambda: bench_extra.stream("f32", n=n))):
                out = run()
                results.setdefault((name, n, nt), []).append(out["kernel_ms"])
for (name, n, nt), values in sorted(results.items()):
    print(json.dumps({"item": name, "elements": n, "nontemporal": int(nt), "kernel_ms": values, "median": sorted(values)[1]}))
