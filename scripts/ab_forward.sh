#!/bin/bash
# A/B of library variants on one box: bash scripts/ab_forward.sh <batch> <kernel filter> <tag>...   ('main' = regular library)
B=$1; K=$2; shift 2
for T in "$@"; do
  if [ "$T" = main ]; then unset SWG_LIB; else export SWG_LIB=$GRAFT_REPO_ROOT/swingnn_amd/libswingnn_hip_$T.so; fi
  timeout -k 10 200 python -u scripts/prof_forward.py ${CFG:-grid} $B 5 > gpurun_out/ab_$T.txt 2>&1 || exit 1
  echo "== $T: $(grep wall gpurun_out/ab_$T.txt)"; grep -E "$K" gpurun_out/ab_$T.txt
done
