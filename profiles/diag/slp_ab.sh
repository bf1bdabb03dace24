#!/bin/bash
#  A/B on one box, interleaved: the xkorc fp32 push and the RK4 kernel built with and without clang's SLP vectorizer
#  (v_pk_*_f32 pairs and the moves that feed them), the push also with and without the sqrtf window lowering.
#  tmp_ab/{slp,noslp}/<hash>.hsaco are the same kernel texts compiled with the two flag sets (GFHIP_CACHE_DIR is
#  searched before the in-tree cache).  Output: one JSON line per run.
set -e
out=${1:-gpurun_out/slp_ab.jsonl}
mkdir -p $(dirname $out)
: > $out
for round in 1 2 3; do
  for variant in slp noslp; do
    for wsqrt in 1 0; do
      GFHIP_CACHE_DIR=$PWD/tmp_ab/$variant GFHIP_WINDOW_SQRT_F32=$wsqrt python bench.py --workload korc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'kernel': 'korc_step_f32', 'build': '$variant', 'window_sqrt_f32': $wsqrt, 'round': $round, 'kernel_ms': d['roofline']['kernel_ms'], 'ms_per_step': d['ms_per_step'], 'vgprs': d['config']['vgprs'], 'hash': d['roofline']['source_hash'], 'cached': d['config']['code_object_from_cache']}))" >> $out
    done
    GFHIP_CACHE_DIR=$PWD/tmp_ab/$variant python bench.py --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'kernel': 'solver_kernel_f64', 'build': '$variant', 'round': $round, 'kernel_ms': d['roofline']['kernel_ms'], 'ms_per_step': d['ms_per_step'], 'hash': d['roofline']['source_hash']}))" >> $out
  done
done
cat $out
