#!/bin/bash
# HBM traffic per kernel from the L2 memory-side counters, one rocprofv3 --pmc pass per counter (FETCH_SIZE takes 3
# of the 4 TCC slots, WRITE_SIZE 2), over one profiled network pass at the bench batch.  Run on the GPU box from the
# repo root: bash scripts/pmc_traffic.sh [batch]
set -e
R=$GRAFT_REPO_ROOT
B=${1:-56}
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o p -- python3 $R/scripts/prof_forward.py grid $B 2 > $R/gpurun_out/pmc_$C.log 2>&1
done
python3 $R/scripts/pmc_traffic_summary.py $R/gpurun_out $R/gpurun_out/pmc_traffic.json $B
