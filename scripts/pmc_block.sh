#!/bin/bash
# PMC passes over the block microbenchmark (run on the GPU box from the repo root): one rocprofv3 run per pass.
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WHICH=${1:-attn}
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_CVT SQ_CYCLES"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/pmc_${WHICH}_$i -- python3 $R/scripts/bench_block.py ${2:-16} $WHICH > $R/gpurun_out/pmc_${WHICH}_$i.log 2>&1
  i=$((i+1))
done
echo done
for i in 1 2 3; do python3 $R/scripts/pmc_summary.py "$R/gpurun_out/pmc_${WHICH}_$i/**/*counter_collection.csv" ${3:-k_wattn} > $R/gpurun_out/pmc_${WHICH}_summary_$i.txt; done
