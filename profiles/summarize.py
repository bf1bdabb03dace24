#!/usr/bin/env python3
"""Summaries of rocprofv3 output directories (gpurun_out/...) for profiles/.

    python profiles/summarize.py pmc <dir> [kernel-substring]     mean of every counter per kernel
    python profiles/summarize.py stats <dir>                      the kernel rows of *_kernel_stats.csv
    python profiles/summarize.py round <gpurun_out/rNN> <profiles/rNN>       rNN_kernel_stats.csv + rNN_pmc_summary.csv of every
                                                                           stats_* / pmc_* directory of a collection
    python profiles/summarize.py traffic <bench.json> <traffic.json> <kernel>=<FETCH_SIZE dir>,<WRITE_SIZE dir> ...
        HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; the factor 2 is gfx950's,
        MI355X_MICROARCH.md "HBM"), keyed by the code object's source hash and launch size taken from the
        bench line, which is what bench.py's roofline.traffic looks up.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def pmc(directory, only=None):
    sums = defaultdict(lambda: defaultdict(float))
    counts = defaultdict(lambda: defaultdict(int))
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                kernel = row.get("Kernel_Name", "")
                if only and only not in kernel:
                    continue
                name, value = row.get("Counter_Name"), float(row.get("Counter_Value", 0) or 0)
                sums[kernel][name] += value
                counts[kernel][name] += 1
    out = {}
    for kernel in sums:
        out[kernel] = {name: (sums[kernel][name]/counts[kernel][name], counts[kernel][name]) for name in sums[kernel]}
    return out


def stats(directory):
    rows = []
    for path in glob.glob(os.path.join(directory, "**", "*kernel_stats.csv"), recursive=True):
        with open(path) as f:
            rows += list(csv.DictReader(f))
    return rows


def traffic(bench_path, out_path, specs):
    import json
    with open(bench_path) as f:
        line = json.loads([l for l in f if l.startswith("{")][-1])
    rooflines = {line["roofline"]["kernel"]: dict(line["roofline"], units_per_launch=line["config"]["rays_per_gpu"])}
    for extra in line.get("roofline_extra", {}).values():
        rooflines[extra["kernel"]] = extra
    entries = []
    for spec in specs:
        kernel, directories = spec.split("=")
        fetch_dir, write_dir = directories.split(",")

        def mean(directory, counter):
#  dispatches that return at once (passes queued behind a converged loop's `stop` word) move no data:
#  only dispatches with at least half of the largest value count
            values = []
            for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if row.get("Kernel_Name", "").startswith(kernel) and row.get("Counter_Name") == counter:
                            values.append(float(row.get("Counter_Value", 0) or 0))
            kept = [v for v in values if v >= 0.5*max(values)]
            return sum(kept)/len(kept), len(kept)

        fetch, launches = mean(fetch_dir, "FETCH_SIZE")
        write, _ = mean(write_dir, "WRITE_SIZE")
        roof = rooflines[kernel]
        entries.append({"kernel": kernel, "source_hash": roof["source_hash"], "rays_per_launch": roof["units_per_launch"],
                        "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "fetch_correction": 2.0,
                        "traffic_bytes_per_launch": (2.0*fetch + write)*1024.0,
                        "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"], "launches_averaged": launches,
                        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (profiles/collect_r03.sh), one MI355X"})
    with open(out_path, "w") as f:
        json.dump(entries, f, indent=1)
    for e in entries:
        print(e["kernel"], e["source_hash"], e["rays_per_launch"], "traffic %.4g B" % e["traffic_bytes_per_launch"],
              "algorithmic %.4g B" % e["algorithmic_bytes_per_launch"])


def whole_round(directory, prefix):
    with open(prefix + "_kernel_stats.csv", "w") as f:
        f.write("run,kernel,calls,average_ns,percent\n")
        for run in sorted(glob.glob(os.path.join(directory, "stats_*"))):
            if not os.path.isdir(run):
                continue
            for row in stats(run):
                f.write("%s,%s,%s,%s,%s\n" % (os.path.basename(run), row.get("Name", "")[:70].replace(",", ";"), row.get("Calls"),
                                              row.get("AverageNs"), row.get("Percentage")))
    with open(prefix + "_pmc_summary.csv", "w") as f:
        f.write("run,kernel,counter,mean_per_dispatch,samples\n")
        for run in sorted(glob.glob(os.path.join(directory, "pmc_*"))):
            if not os.path.isdir(run):
                continue
            for kernel, counters in pmc(run).items():
                for name, (mean, n) in sorted(counters.items()):
                    f.write("%s,%s,%s,%.6g,%d\n" % (os.path.basename(run), kernel[:50].replace(",", ";"), name, mean, n))


if __name__ == "__main__":
    what, directory = sys.argv[1], sys.argv[2]
    if what == "round":
        whole_round(sys.argv[2], sys.argv[3])
    elif what == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4:])
    elif what == "pmc":
        for kernel, counters in pmc(directory, sys.argv[3] if len(sys.argv) > 3 else None).items():
            for name, (mean, n) in sorted(counters.items()):
                print("%s,%s,%.6g,%d" % (kernel[:60], name, mean, n))
    else:
        for row in stats(directory):
            print(",".join(str(row.get(k, "")) for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")))
