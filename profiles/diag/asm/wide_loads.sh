#!/bin/bash
#  16-byte table loads of the assembly body on and off: identical rays and the incoherent CLI beam, one box, two rounds.
out=${1:-gpurun_out/asm_wide_loads.jsonl}
mkdir -p $(dirname $out) /tmp/asm_cache
: > $out
for round in 1 2; do
  for wide in 1 0; do
    for distribution in bench cli; do
      GFHIP_CACHE_DIR=/tmp/asm_cache GFHIP_ASM_WIDE_LOADS=$wide python bench.py --distribution $distribution --no-extra --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'wide_loads': $wide, 'rays': '$distribution', 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms']}))" >> $out
    done
  done
done
cat $out
