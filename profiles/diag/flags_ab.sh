#!/bin/bash
# A/B of hipcc back-end flags on the RK4 code object (profiles/diag/flags_ab/<variant>/<source hash>.hsaco, built in the
# CPU container from the SAME kernel text): GFHIP_CACHE_DIR makes the runtime load the variant instead of the cached one.
#   gpurun -- 'bash profiles/diag/flags_ab.sh > gpurun_out/flags_ab.jsonl'
R=$(pwd)
for round in 1 2; do
for variant in $(ls $R/profiles/diag/flags_ab); do
    [ -d $R/profiles/diag/flags_ab/$variant ] || continue
    line=$(GFHIP_CACHE_DIR=$R/profiles/diag/flags_ab/$variant python3 $R/profiles/diag/segments_ab.py one 10000000 100 /tmp/flags_state_$variant.npz | tail -1)
    echo "{\"variant\": \"$variant\", \"round\": $round, \"result\": $line, \"state_md5\": \"$(md5sum /tmp/flags_state_$variant.npz | cut -d' ' -f1)\"}"
done
done
