#!/bin/bash
# Round-3 evidence BEFORE the assembly body became the default lowering of the RK4 item (profiles/r03_*); to reproduce it
# now, export GFHIP_ASM=0 first.  The final state of the round is collected by profiles/collect_r03b.sh (profiles/r03b_*).
# One gpurun call:   gpurun --timeout 1200 -- 'bash profiles/collect_r03.sh'
# Everything lands in gpurun_out/r03/; profiles/summarize.py turns it into the files kept in profiles/.
# Counters are collected in their own passes (never together with --stats or trace domains); the
# program itself follows `--` (python3 ...), environment variables are exported beforehand, the profiler runs from /tmp.
set -o pipefail
R=$(pwd)
OUT=$R/gpurun_out/r03
rm -rf $OUT
mkdir -p $OUT
python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || exit 1
python3 $R/bench.py --workload korc > $OUT/bench_korc_n1.json 2>> $OUT/bench_n1.err || exit 1
python3 $R/bench.py --gpus 1 --backend nccl --force-collectives --no-extra --no-cpu-baseline > $OUT/bench_n1_rccl_one_rank.json 2>> $OUT/bench_n1.err || exit 1
python3 $R/bench.py --workload korc --gpus 1 --backend nccl --force-collectives > $OUT/bench_korc_rccl_one_rank.json 2>> $OUT/bench_n1.err || exit 1
python3 $R/bench.py --distribution cli --no-extra --no-cpu-baseline > $OUT/bench_cli_1e7.json 2>> $OUT/bench_n1.err || exit 1
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- $B --steps 200 --warmup 10 > $OUT/stats_bench.log 2>&1 || exit 1
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
i=0
for counters in "FETCH_SIZE" "WRITE_SIZE" "$SQ" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/pmc_bench_$i -- $B --steps 20 --warmup 2 > $OUT/pmc_bench_$i.log 2>&1 || echo "pmc pass failed: bench $counters" >> $OUT/failed.txt
done
# The Newton init (kernels of the converge loop; bench.py runs it once per process) and the xkorc push.
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_korc -- python3 $R/bench.py --workload korc --steps 200 > $OUT/stats_korc.log 2>&1 || echo "stats failed: korc" >> $OUT/failed.txt
i=0
for counters in "FETCH_SIZE" "WRITE_SIZE" "$SQ"; do
    i=$((i+1))
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/pmc_korc_$i -- python3 $R/bench.py --workload korc --steps 20 > $OUT/pmc_korc_$i.log 2>&1 || echo "pmc pass failed: korc $counters" >> $OUT/failed.txt
done
# the RK4 item as one kernel, as 3 segments, as 3 segments at two waves per SIMD (redo launch instead
# of the IEEE function): SQ_WAVE_CYCLES and SQ_WAIT_ANY of each.
for configuration in "0 0" "3 0" "3 2" "4 2"; do
    set -- $configuration
    export GFHIP_SEGMENTS=$1
    if [ "$2" = "0" ]; then unset GFHIP_WAVES_PER_SIMD; else export GFHIP_WAVES_PER_SIMD=$2; fi
    rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_segments_$1_$2 -- python3 $R/profiles/diag/segments_ab.py one 10000000 10 /tmp/state.npz > $OUT/pmc_segments_$1_$2.log 2>&1 || echo "pmc pass failed: segments $configuration" >> $OUT/failed.txt
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_segments_$1_$2 -- python3 $R/profiles/diag/segments_ab.py one 10000000 50 /tmp/state.npz > $OUT/stats_segments_$1_$2.log 2>&1 || echo "stats failed: segments $configuration" >> $OUT/failed.txt
done
unset GFHIP_SEGMENTS GFHIP_WAVES_PER_SIMD
cd $R
python3 $R/profiles/diag/segments_ab.py run 10000000 100 > $OUT/segments_ab.jsonl 2> $OUT/segments_ab.err || echo "segments_ab failed" >> $OUT/failed.txt
python3 $R/profiles/diag/newton_batch_ab.py run 10000000 > $OUT/newton_batch_ab.jsonl 2> $OUT/newton_batch_ab.err || echo "newton_batch_ab failed" >> $OUT/failed.txt
$R/graph_framework_amd/xrays_bench $R/graph_framework_amd/workloads 10000000 1000 > $OUT/xrays_bench_cpp.log 2>&1
echo collected
