#!/bin/bash
# A/B library with extra -D switches for ONE source file -> swingnn_amd/libswingnn_hip_<tag>.so (load with SWG_LIB=...)
# usage: bash scripts/build_variant.sh <tag> <file without .hip> -DFOO=1 ...   (other objects from the regular build)
set -e
cd "$(dirname "$0")/../swingnn_amd/csrc"
TAG=$1; F=$2; shift 2
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=200000 "$@" -c $F.hip -o /tmp/${F}_${TAG}.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libswingnn_hip_${TAG}.so /tmp/${F}_${TAG}.o $(ls *.o | grep -v "^$F.o")
echo built ../libswingnn_hip_${TAG}.so
