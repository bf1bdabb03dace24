#!/bin/bash
#  The RK4 kernel with the assembly body under several pools and look-ahead distances, one box, interleaved with the
#  compiled body.  Output: one JSON line per run (profiles/diag/segments_ab.py one = 1e7 rays, event-timed launches).
out=${1:-gpurun_out/asm_sweep.jsonl}
mkdir -p $(dirname $out) /tmp/asm_cache
: > $out
run() {  # label, environment assignments...
  label=$1; shift
  env GFHIP_CACHE_DIR=/tmp/asm_cache "$@" python profiles/diag/segments_ab.py one 10000000 100 /tmp/state_$$.npz 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'label': '$label', 'ms_per_step': d['ms_per_step'], 'event_ms': d.get('event_ms')}))" >> $out
}
for round in 1 2; do
  run "compiled" GFHIP_ASM=0
  run "asm pool 48 ahead 24/6" GFHIP_ASM=1 GFHIP_ASM_WAVES=2
  run "asm pool 48 ahead 48/12" GFHIP_ASM=1 GFHIP_ASM_WAVES=2 GFHIP_ASM_LOAD_AHEAD=48 GFHIP_ASM_RELOAD_AHEAD=12
  run "asm pool 48 ahead 64/16" GFHIP_ASM=1 GFHIP_ASM_WAVES=2 GFHIP_ASM_LOAD_AHEAD=64 GFHIP_ASM_RELOAD_AHEAD=16
  run "asm pool 48 ahead 96/24" GFHIP_ASM=1 GFHIP_ASM_WAVES=2 GFHIP_ASM_LOAD_AHEAD=96 GFHIP_ASM_RELOAD_AHEAD=24
  run "asm pool 48 ahead 160/40" GFHIP_ASM=1 GFHIP_ASM_WAVES=2 GFHIP_ASM_LOAD_AHEAD=160 GFHIP_ASM_RELOAD_AHEAD=40
  run "asm pool 48 ahead 48/24" GFHIP_ASM=1 GFHIP_ASM_WAVES=2 GFHIP_ASM_LOAD_AHEAD=48 GFHIP_ASM_RELOAD_AHEAD=24
  run "asm pool 40 ahead 48/12" GFHIP_ASM=1 GFHIP_ASM_WAVES=2 GFHIP_ASM_POOL_LO=40 GFHIP_ASM_LOAD_AHEAD=48 GFHIP_ASM_RELOAD_AHEAD=12
  run "asm pool 40 ahead 96/24" GFHIP_ASM=1 GFHIP_ASM_WAVES=2 GFHIP_ASM_POOL_LO=40 GFHIP_ASM_LOAD_AHEAD=96 GFHIP_ASM_RELOAD_AHEAD=24
  run "asm pool 36 ahead 96/24" GFHIP_ASM=1 GFHIP_ASM_WAVES=2 GFHIP_ASM_POOL_LO=36 GFHIP_ASM_LOAD_AHEAD=96 GFHIP_ASM_RELOAD_AHEAD=24
done
cat $out
