#!/bin/bash
#  With the vector unit saturated a derived table value (factor x parent: one v_mul_f64) costs more than loading it:
#  the RK4 kernel with and without the exact table compaction of tables.hpp, identical rays and the CLI beam, one box.
out=${1:-gpurun_out/asm_compaction.jsonl}
mkdir -p $(dirname $out) /tmp/asm_cache
: > $out
for round in 1 2; do
  for compact in 1 0; do
    for distribution in bench cli; do
      GFHIP_CACHE_DIR=/tmp/asm_cache GFHIP_COMPACT_TABLES=$compact python bench.py --distribution $distribution --no-extra --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'compact_tables': $compact, 'rays': '$distribution', 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms'], 'lds_bytes': d['config']['lds_bytes']}))" >> $out
    done
  done
done
cat $out
