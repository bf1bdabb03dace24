#!/bin/bash
# Experiment: nt hint on the state loads / stores of the HBM-bound items (GFHIP_NONTEMPORAL = 0 | 1 | 2 stores | 3 loads).
#   gpurun --timeout 900 -- 'bash profiles/diag/nontemporal_sweep.sh'
R=$(pwd)
OUT=$R/gpurun_out/nontemporal.jsonl
rm -f $OUT
for nt in 0 1 2 3; do
    for item in korc_f32 korc_f64 loss stream7_f64 stream_f64; do
        echo "{\"nontemporal\": $nt, \"item\": \"$item\", \"result\": $(GFHIP_NONTEMPORAL=$nt GFHIP_CACHE_DIR=/tmp/nt$nt python3 $R/bench_extra.py $item 2>/dev/null)}" >> $OUT
    done
    echo "{\"nontemporal\": $nt, \"item\": \"solver_1e7\", \"result\": $(GFHIP_NONTEMPORAL=$nt GFHIP_CACHE_DIR=/tmp/nt$nt python3 $R/bench.py --steps 100 --no-extra --no-cpu-baseline 2>/dev/null | tail -1)}" >> $OUT
done
python3 - <<'PY'
import json
for l in open("gpurun_out/nontemporal.jsonl"):
    d = json.loads(l); r = d["result"]
    ms = r.get("kernel_ms") or r.get("roofline", {}).get("kernel_ms")
    print(d["nontemporal"], d["item"], "%.4f ms" % ms)
PY
