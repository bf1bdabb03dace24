#!/bin/bash
# Diagnostic: HIP-event kernel time of the HBM-side items under forced occupancies
# (second __launch_bounds__ argument, GFHIP_WAVES_PER_SIMD).   gpurun -- 'bash profiles/diag/occupancy_sweep.sh'
OUT=gpurun_out/r02_occupancy.jsonl
rm -f $OUT
for w in 0 2 3 4 5 6 8; do
  for item in loss korc_f32 korc_f64; do
    echo -n "{\"waves_per_simd\": $w, \"item\": \"$item\", \"result\": " >> $OUT
    GFHIP_WAVES_PER_SIMD=$w GFHIP_CACHE_DIR=/tmp/gfcache_$w python3 bench_extra.py $item >> $OUT 2>/dev/null || echo "null" >> $OUT
    echo "}" >> $OUT
  done
done
cat $OUT | tr -d '\n' | sed 's/}{/}\n{/g' | python3 -c "
import sys, json
for line in sys.stdin:
    try:
        d = json.loads(line)
        r = d['result']
        print(d['waves_per_simd'], d['item'], r.get('kernel_ms'), r.get('roofline', {}).get('frac'))
    except Exception as e:
        print('bad', line[:80], e)
"
