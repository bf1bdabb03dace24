#!/bin/bash
#  The Newton solve of the xrays_bench graph under the REFERENCE front end (oracle/_ref/hip_context_demo: hip_context's
#  create_max_call in the loop of workflow.hpp:179-205) with gfhip_run_max running ahead and without: launches of the
#  loss kernel and their total time (rocprofv3 --kernel-trace --stats), 1e7 rays.
R=$(pwd)
out=${1:-gpurun_out/run_ahead_ab.txt}
mkdir -p $(dirname $out)
python3 -c "
import sys; sys.path.insert(0, '$R')
import numpy as np
from oracle import ref
ref.write_tables(np.load('$R/tests/golden/efit_tables.npz'), '/tmp/tables.bin')"
: > $out
cd /tmp && export TMPDIR=/tmp
for ahead in 1 0; do
  export GFHIP_RUN_AHEAD=$ahead
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ra_$ahead -- $R/oracle/_ref/hip_context_demo /tmp/tables.bin 10000000 1 > /tmp/ra_$ahead.log 2>&1
  echo "GFHIP_RUN_AHEAD=$ahead: $(grep -E 'PASS|FAIL' /tmp/ra_$ahead.log | tail -1)" >> $R/$out
  python3 $R/profiles/summarize.py stats /tmp/ra_$ahead | grep -E "loss_kernel|copyBuffer" >> $R/$out
done
cat $R/$out
