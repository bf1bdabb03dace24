#!/bin/bash
# Round-3 evidence for the FINAL lowering of the round (assembly body for the RK4 item, fp32 trims, no SLP), one gpurun call:
#     gpurun --timeout 1200 -- 'bash profiles/collect_r03b.sh'
# Lands in gpurun_out/r03b/; profiles/summarize.py turns it into profiles/r03b_*.csv and profiles/traffic.json.
# Counters in their own passes (never with --stats or trace domains); the program itself follows `--`; the profiler runs from /tmp.
set -o pipefail
R=$(pwd)
OUT=$R/gpurun_out/r03b
rm -rf $OUT
mkdir -p $OUT
python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || exit 1
python3 $R/bench.py --workload korc > $OUT/bench_korc_n1.json 2>> $OUT/bench_n1.err || exit 1
python3 $R/bench.py --gpus 1 --backend nccl --force-collectives --no-extra --no-cpu-baseline > $OUT/bench_n1_rccl_one_rank.json 2>> $OUT/bench_n1.err || exit 1
python3 $R/bench.py --workload korc --gpus 1 --backend nccl --force-collectives > $OUT/bench_korc_rccl_one_rank.json 2>> $OUT/bench_n1.err || exit 1
python3 $R/bench.py --distribution cli --no-extra --no-cpu-baseline > $OUT/bench_cli_1e7.json 2>> $OUT/bench_n1.err || exit 1
GFHIP_ASM=0 python3 $R/bench.py --no-extra --no-cpu-baseline > $OUT/bench_compiled_body.json 2>> $OUT/bench_n1.err || exit 1
for rays in 50000000 100000000; do      # one GPU at the per-GPU shard sizes of C4
    python3 $R/bench.py --rays-per-gpu $rays --steps 50 --warmup 5 --no-extra --no-cpu-baseline > $OUT/bench_rays_$rays.json 2>> $OUT/bench_n1.err || exit 1
done
python3 $R/bench.py --rays-per-gpu 1000000 --steps 400 --warmup 20 --no-extra --no-cpu-baseline > $OUT/bench_rays_1000000.json 2>> $OUT/bench_n1.err || exit 1      # C2
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- $B --steps 200 --warmup 10 > $OUT/stats_bench.log 2>&1 || exit 1
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
i=0
for counters in "FETCH_SIZE" "WRITE_SIZE" "$SQ" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/pmc_bench_$i -- $B --steps 20 --warmup 2 > $OUT/pmc_bench_$i.log 2>&1 || echo "pmc pass failed: bench $counters" >> $OUT/failed.txt
done
# the compiled body of the same item, for the comparison of DESIGN.md section 3
export GFHIP_ASM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_compiled_body -- $B --steps 200 --warmup 10 > $OUT/stats_compiled_body.log 2>&1 || echo "stats failed: compiled body" >> $OUT/failed.txt
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_compiled_body_3 -- $B --steps 20 --warmup 2 > $OUT/pmc_compiled_body_3.log 2>&1 || echo "pmc pass failed: compiled body" >> $OUT/failed.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_compiled_body_4 -- $B --steps 20 --warmup 2 > $OUT/pmc_compiled_body_4.log 2>&1 || echo "pmc pass failed: compiled body 4" >> $OUT/failed.txt
unset GFHIP_ASM
# the xkorc push
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_korc -- python3 $R/bench.py --workload korc --steps 200 > $OUT/stats_korc.log 2>&1 || echo "stats failed: korc" >> $OUT/failed.txt
i=0
for counters in "FETCH_SIZE" "WRITE_SIZE" "$SQ"; do
    i=$((i+1))
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/pmc_korc_$i -- python3 $R/bench.py --workload korc --steps 20 > $OUT/pmc_korc_$i.log 2>&1 || echo "pmc pass failed: korc $counters" >> $OUT/failed.txt
done
cd $R
$R/graph_framework_amd/xrays_bench $R/graph_framework_amd/workloads 10000000 1000 > $OUT/xrays_bench_cpp.log 2>&1
python3 $R/profiles/summarize.py traffic $OUT/bench_n1.json $OUT/traffic.json gfhip_solver_kernel=$OUT/pmc_bench_1,$OUT/pmc_bench_2 gfhip_loss_kernel=$OUT/pmc_bench_1,$OUT/pmc_bench_2 gfhip_step=$OUT/pmc_korc_1,$OUT/pmc_korc_2 > $OUT/traffic.log 2>&1 || echo "traffic summary failed" >> $OUT/failed.txt
python3 $R/profiles/summarize.py round $OUT $OUT/r03b || echo "round summary failed" >> $OUT/failed.txt
rm -rf $OUT/pmc_*/ $OUT/stats_*/
echo collected
