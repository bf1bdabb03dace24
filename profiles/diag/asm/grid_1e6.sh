#!/bin/bash
#  Workgroups per CU at 1e6 rays (C2), where a workgroup sees one tile: the default grid against fixed counts; the
#  compiled body last.
for per_cu in default 2 4 8 16; do
  if [ $per_cu = default ]; then unset GFHIP_GRID_PER_CU; else export GFHIP_GRID_PER_CU=$per_cu; fi
  for r in 1 2; do python profiles/diag/segments_ab.py one 1000000 400 /tmp/s.npz 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$per_cu', d['ms_per_step'], d.get('event_ms'))"; done
done
GFHIP_ASM=0 python profiles/diag/segments_ab.py one 1000000 400 /tmp/s.npz 2>/dev/null | tail -1 | cut -c1-200
