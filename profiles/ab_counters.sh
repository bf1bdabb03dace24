#!/bin/bash
# A/B of two trees on one box: rocprofv3 counter passes (no trace domains) on bench.py at 1e7 rays.
#   gpurun -- 'bash profiles/ab_counters.sh <tag> <tree> [rays]'
set -o pipefail
TAG=$1; TREE=$(cd $2 && pwd); RAYS=${3:-10000000}
OUT=$(pwd)/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for counters in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
    name=$(echo $counters | cut -d' ' -f1)
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $TREE/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra --rays-per-gpu $RAYS > $OUT/pmc_$name.log 2>&1 || python3 $TREE/bench.py --steps 20 --warmup 2 --no-cpu-baseline --rays-per-gpu $RAYS > $OUT/pmc_$name.log 2>&1 || exit 1
done
