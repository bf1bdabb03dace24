#!/usr/bin/env python3
"""Summaries of rocprofv3 output directories (gpurun_out/...) for profiles/.

    python profiles/summarize.py pmc <dir> [kernel-substring]     mean of every counter per kernel
    python profiles/summarize.py stats <dir>                      the kernel rows of *_kernel_stats.csv
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


if __name__ == "__main__":
    what, directory = sys.argv[1], sys.argv[2]
    if what == "pmc":
        for kernel, counters in pmc(directory, sys.argv[3] if len(sys.argv) > 3 else None).items():
            for name, (mean, n) in sorted(counters.items()):
                print("%s,%s,%.6g,%d" % (kernel[:60], name, mean, n))
    else:
        for row in stats(directory):
            print(",".join(str(row.get(k, "")) for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")))
