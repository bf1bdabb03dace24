#!/bin/bash
# Collect the round's evidence on the GPU box:  gpurun -- 'bash profiles/collect.sh r01'
# Everything lands in gpurun_out/<tag>_*; copy what is judged into profiles/.
# Counters are collected in their own passes (never together with --stats or trace domains).
set -o pipefail
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
python3 $R/bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_prof_stdout.log 2>&1 || exit 1
for counters in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    name=$(echo $counters | cut -d' ' -f1)
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$name -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_pmc_$name.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_korc_f32 -- python3 $R/bench_extra.py korc_f32 > $OUT/${TAG}_prof_korc_f32.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_korc_f64 -- python3 $R/bench_extra.py korc_f64 > $OUT/${TAG}_prof_korc_f64.log 2>&1 || exit 1
cd $R
for item in korc_f32 korc_f64 loss loss_per_ray fused solver_f32 stream_f32 stream_f64 stream7_f32 stream7_f64 trajectory; do
    python3 $R/bench_extra.py $item >> $OUT/${TAG}_extra_items.jsonl 2>> $OUT/${TAG}_extra.err || exit 1
done
python3 $R/bench.py --distribution cli --rays-per-gpu 10000000 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_cli_1e7.json 2>> $OUT/${TAG}_extra.err || exit 1
$R/graph_framework_amd/xrays_bench $R/graph_framework_amd/workloads 1000000 1000 > $OUT/${TAG}_xrays_bench_cpp.log 2>&1
