#!/bin/bash
#  The incoherent CLI beam under deeper look-ahead of the LDS reads (table values of the staged pack and values coming
#  back from their slots) and of the global loads; identical rays beside it.  One box, two rounds.
out=${1:-gpurun_out/asm_cli_ahead.jsonl}
mkdir -p $(dirname $out) /tmp/asm_cache
: > $out
for round in 1 2; do
  for ahead in "96 24" "96 48" "96 96" "192 48" "192 96"; do
    set -- $ahead
    for distribution in bench cli; do
      GFHIP_CACHE_DIR=/tmp/asm_cache GFHIP_ASM_LOAD_AHEAD=$1 GFHIP_ASM_RELOAD_AHEAD=$2 python bench.py --distribution $distribution --no-extra --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'load_ahead': $1, 'reload_ahead': $2, 'rays': '$distribution', 'ms_per_step': d['ms_per_step']}))" >> $out
    done
  done
done
cat $out
