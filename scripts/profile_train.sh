#!/bin/bash
# Train-step evidence, one GPU (run on the GPU box from the repo root): bash scripts/profile_train.sh <tag>
#   step timings of every named config, rocprofv3 kernel stats and the per-op breakdown of grid B=80 -> gpurun_out/<tag>_train_*
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp
export TMPDIR=/tmp
timeout -k 10 200 python3 $R/scripts/bench_train.py grid 80 10 > $R/gpurun_out/${TAG}_train_grid96_b80.log 2>&1
timeout -k 10 200 python3 $R/scripts/bench_train.py grid 8 10 > $R/gpurun_out/${TAG}_train_grid96_b8.log 2>&1
timeout -k 10 200 python3 $R/scripts/bench_train.py com 80 10 > $R/gpurun_out/${TAG}_train_com96_b80.log 2>&1
timeout -k 10 200 python3 $R/scripts/bench_train.py dd 4 10 > $R/gpurun_out/${TAG}_train_dd96_b4.log 2>&1
timeout -k 10 200 python3 $R/scripts/bench_train_qm9.py onehot 96 1280 > $R/gpurun_out/${TAG}_train_qm9_onehot96_b1280.log 2>&1
timeout -k 10 200 python3 $R/scripts/bench_train_qm9.py ddpm 60 1280 > $R/gpurun_out/${TAG}_train_qm9_ddpm60_b1280.log 2>&1
echo "steps done"
timeout -k 10 300 python3 $R/scripts/bench_bwd_ops.py grid 80 > $R/gpurun_out/${TAG}_train_grid96_b80_op_breakdown.log 2>&1
echo "ops done"
rm -rf /tmp/rp_train
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_train -o rp -- python3 $R/scripts/bench_train.py grid 80 10 > $R/gpurun_out/${TAG}_train_under_rocprofv3.log 2>&1
cp $(ls /tmp/rp_train/*/*kernel_stats.csv /tmp/rp_train/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/${TAG}_train_grid96_b80_rocprofv3_kernel_stats.csv
echo "rocprof done"
