#!/bin/bash
# Round 2, after the fp32 division change (new code object for the xkorc fp32 push): the bench line and the
# push's stats / counter passes again.   gpurun --timeout 900 -- 'bash profiles/collect_r02_korc.sh'
set -o pipefail
R=$(pwd)
OUT=$R/gpurun_out/r02k
rm -rf $OUT
mkdir -p $OUT
python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || exit 1
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
for item in korc_f32; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$item -- python3 $R/bench_extra.py $item > $OUT/stats_$item.log 2>&1 || echo "stats failed: $item" >> $OUT/failed.txt
    i=0
    for counters in "FETCH_SIZE" "WRITE_SIZE" "$SQ"; do
        i=$((i+1))
        rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/pmc_${item}_$i -- python3 $R/bench_extra.py $item > $OUT/pmc_${item}_$i.log 2>&1 || echo "pmc pass failed: $item $counters" >> $OUT/failed.txt
    done
done
cd $R
for item in korc_f32 solver_f32; do
    python3 $R/bench_extra.py $item >> $OUT/extra_items.jsonl 2>> $OUT/extra.err || echo "extra failed: $item" >> $OUT/failed.txt
done
echo collected
