#!/bin/bash
#  Follow-up of compaction.sh: the uncompacted tables with no LDS staging and with coarser grids, and the compacted ones
#  without staging (default bench line of each, two rounds, one box).
mkdir -p /tmp/asm_cache
run() { label=$1; shift; env GFHIP_CACHE_DIR=/tmp/asm_cache "$@" python bench.py --no-extra --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['ms_per_step'], d['config']['lds_bytes'])"; }
for r in 1 2; do
run default GFHIP_ASM=1
run nocompact_lds0 GFHIP_COMPACT_TABLES=0 GFHIP_LDS_BUDGET=0
run nocompact_grid16 GFHIP_COMPACT_TABLES=0 GFHIP_GRID_PER_CU=16
run nocompact_grid8 GFHIP_COMPACT_TABLES=0 GFHIP_GRID_PER_CU=8
run compact_lds0 GFHIP_LDS_BUDGET=0
done
