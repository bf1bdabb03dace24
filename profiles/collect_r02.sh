#!/bin/bash
# Round-2 evidence, one gpurun call:   gpurun --timeout 1200 -- 'bash profiles/collect_r02.sh'
# Everything lands in gpurun_out/r02/; profiles/summarize.py turns it into the files kept in profiles/.
# Counters are collected in their own passes (never together with --stats or trace domains); the
# program itself follows `--` (python3 bench.py ...), and the profiler runs from /tmp.
set -o pipefail
R=$(pwd)
OUT=$R/gpurun_out/r02
rm -rf $OUT
mkdir -p $OUT
python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || exit 1
python3 $R/bench.py --rays-per-gpu 1000000 --no-extra --no-cpu-baseline > $OUT/bench_n1_1e6.json 2>> $OUT/bench_n1.err || exit 1
python3 $R/bench.py --distribution cli --no-extra --no-cpu-baseline > $OUT/bench_cli_1e7.json 2>> $OUT/bench_n1.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
B="python3 $R/bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- $B --steps 200 --warmup 10 > $OUT/stats_bench.log 2>&1 || exit 1
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
MEM="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
for dist in bench cli; do
    i=0
    for counters in "FETCH_SIZE" "WRITE_SIZE" "$SQ" "$MEM" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
        i=$((i+1))
        rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/pmc_${dist}_$i -- $B --steps 20 --warmup 2 --distribution $dist > $OUT/pmc_${dist}_$i.log 2>&1 || echo "pmc pass failed: $dist $counters" >> $OUT/failed.txt
    done
done
for item in korc_f32 korc_f64 loss; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$item -- python3 $R/bench_extra.py $item > $OUT/stats_$item.log 2>&1 || echo "stats failed: $item" >> $OUT/failed.txt
    i=0
    for counters in "FETCH_SIZE" "WRITE_SIZE" "$SQ"; do
        i=$((i+1))
        rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/pmc_${item}_$i -- python3 $R/bench_extra.py $item > $OUT/pmc_${item}_$i.log 2>&1 || echo "pmc pass failed: $item $counters" >> $OUT/failed.txt
    done
done
cd $R
for item in korc_f32 korc_f64 loss loss_per_ray fused solver_f32 stream_f32 stream_f64 stream7_f32 stream7_f64 trajectory absorption; do
    python3 $R/bench_extra.py $item >> $OUT/extra_items.jsonl 2>> $OUT/extra.err || echo "extra failed: $item" >> $OUT/failed.txt
done
$R/graph_framework_amd/xrays_bench $R/graph_framework_amd/workloads 10000000 1000 > $OUT/xrays_bench_cpp.log 2>&1
echo collected
