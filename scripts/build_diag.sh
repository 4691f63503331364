#!/bin/bash
# Diagnostic library with in-kernel phase stamps (-DSWG_STAMPS) -> swingnn_amd/libswingnn_hip_diag.so
# usage: bash scripts/build_diag.sh wattn_res   (the other objects are taken from the regular build)
set -e
cd "$(dirname "$0")/../swingnn_amd/csrc"
F=${1:-wattn_res}
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=200000 -DSWG_STAMPS -c $F.hip -o /tmp/${F}_diag.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libswingnn_hip_diag.so /tmp/${F}_diag.o $(ls *.o | grep -v "^$F.o")
echo built ../libswingnn_hip_diag.so
