#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_j.txt; mkdir -p gpurun_out; : > $O
run() { echo "## $*" >> $O; timeout 600 env "$@" >> $O 2>&1 || echo "(rc $?)" >> $O; }
V=$PWD/mvlpt_amd/libvar_breg.so
for i in 1 2 3; do
run MVLPT_HIP_LIB=$V MVLPT_GEMM_BREG=0 python tools/breg_probe.py
run MVLPT_HIP_LIB=$V MVLPT_GEMM_BREG=1 python tools/breg_probe.py
done
