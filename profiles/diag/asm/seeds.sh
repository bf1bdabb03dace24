#!/bin/bash
#  Does the emission order matter beyond the LDS slots it needs?  The RK4 kernel with a forced tie-break seed of the list
#  schedule (GFHIP_ASM_SEED), one box, two rounds.
out=${1:-gpurun_out/asm_seeds.jsonl}
mkdir -p $(dirname $out) /tmp/asm_cache
: > $out
for round in 1 2; do
  for seed in 49 32 23 30 5 18 16 35; do
    GFHIP_CACHE_DIR=/tmp/asm_cache GFHIP_ASM_SEED=$seed python profiles/diag/segments_ab.py one 10000000 100 /tmp/s_$$.npz 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'seed': $seed, 'ms_per_step': d['ms_per_step']}))" >> $out
  done
done
cat $out
