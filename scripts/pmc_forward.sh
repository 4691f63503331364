#!/bin/bash
# SQ counter passes over one profiled network pass (run on the GPU box from the repo root): one rocprofv3 run per
# pass.  usage: bash scripts/pmc_forward.sh <batch> <kernel substring>...
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-112}; shift
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
i=1
for P in "$P1" "$P2"; do
  rm -rf /tmp/pmc_fwd_$i
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d /tmp/pmc_fwd_$i -- python3 $R/scripts/prof_forward.py grid $B 2 > $R/gpurun_out/pmc_fwd_$i.log 2>&1
  python3 $R/scripts/pmc_summary.py "/tmp/pmc_fwd_$i/**/*counter_collection.csv" "$@" > $R/gpurun_out/pmc_fwd_summary_$i.txt
  i=$((i+1))
done
echo done
