#!/bin/bash
# Other named configs, one GPU (run on the GPU box from the repo root): bash scripts/profile_configs.sh <tag>
# sampling benches (bench.py --config ...) and train-step timings (scripts/bench_train*.py) -> gpurun_out/<tag>_*
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 800 python3 bench.py --config dd --batch 112 --no-cpu-baseline > gpurun_out/${TAG}_bench_dd96_b112.json 2> gpurun_out/${TAG}_bench_dd.err
echo "dd done"
timeout -k 10 300 python3 bench.py --config community_small --batch 2048 --no-cpu-baseline > gpurun_out/${TAG}_bench_community_small96_b2048.json 2> gpurun_out/${TAG}_bench_com.err
echo "com done"
timeout -k 10 300 python3 bench.py --config ego_small --batch 2048 --no-cpu-baseline > gpurun_out/${TAG}_bench_ego_small60_b2048.json 2> gpurun_out/${TAG}_bench_ego.err
echo "ego done"
timeout -k 10 200 python3 scripts/bench_train.py grid 80 10 > gpurun_out/${TAG}_train_grid96_b80.log 2>&1
timeout -k 10 200 python3 scripts/bench_train.py grid 8 10 > gpurun_out/${TAG}_train_grid96_b8.log 2>&1
timeout -k 10 200 python3 scripts/bench_train.py com 80 10 > gpurun_out/${TAG}_train_com96_b80.log 2>&1
timeout -k 10 200 python3 scripts/bench_train.py dd 4 10 > gpurun_out/${TAG}_train_dd96_b4.log 2>&1
echo "train done"
timeout -k 10 200 python3 scripts/bench_train_qm9.py onehot 96 1280 > gpurun_out/${TAG}_train_qm9_onehot96_b1280.log 2>&1
timeout -k 10 200 python3 scripts/bench_train_qm9.py ddpm 60 1280 > gpurun_out/${TAG}_train_qm9_ddpm60_b1280.log 2>&1
echo "qm9 done"
