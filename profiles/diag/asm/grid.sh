#!/bin/bash
#  Workgroups per CU for the RK4 kernel with the assembly body (two are resident): one box, two rounds.
out=${1:-gpurun_out/asm_grid.jsonl}
mkdir -p $(dirname $out)
: > $out
for round in 1 2; do
  for per_cu in 8 16 24 32 48 64 153; do
    if [ $per_cu = default ]; then unset GFHIP_GRID_PER_CU; else export GFHIP_GRID_PER_CU=$per_cu; fi
    python profiles/diag/segments_ab.py one 10000000 100 /tmp/state_$$.npz 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'workgroups_per_cu': '$per_cu', 'ms_per_step': d['ms_per_step'], 'event_ms': d.get('event_ms')}))" >> $out
  done
done
cat $out
