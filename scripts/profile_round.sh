#!/bin/bash
# Round profile set (run on the GPU box from the repo root): bash scripts/profile_round.sh <tag> [batch]
#   1. bench.py line                      -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats   -> gpurun_out/<tag>_rocprofv3_kernel_stats.csv (+ the bench line under it)
#   3. FETCH_SIZE / WRITE_SIZE passes     -> gpurun_out/<tag>_pmc_traffic.json
#   4. SQ counter passes                  -> gpurun_out/<tag>_forward_sq_counters_b<batch>.txt
set -e
TAG=${1:-r03}; B=${2:-112}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
echo "bench done"
rm -rf /tmp/rp_stats
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -o rp -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pass-reuse --no-train-step > $R/gpurun_out/${TAG}_bench_under_rocprofv3.json 2> $R/gpurun_out/${TAG}_rocprof.err
cp $(ls /tmp/rp_stats/*/*kernel_stats.csv /tmp/rp_stats/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/${TAG}_rocprofv3_kernel_stats.csv
echo "stats done"
cd $R
bash scripts/pmc_traffic.sh $B > gpurun_out/${TAG}_pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
echo "traffic done"
bash scripts/pmc_forward.sh $B k_wattn k_mlp k_gemm k_row k_merge k_expand k_patch k_breakup k_final k_sym > gpurun_out/${TAG}_pmc_forward.log 2>&1
cat gpurun_out/pmc_fwd_summary_1.txt gpurun_out/pmc_fwd_summary_2.txt > gpurun_out/${TAG}_forward_sq_counters_b${B}.txt
echo "sq done"
