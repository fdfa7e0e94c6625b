#!/bin/bash
# Round 6: do the launcher's geometry switches still sit at their best values on the final library (built without packed fp32)?
cd "$(dirname "$0")/.."
O=gpurun_out/r06_l.txt; mkdir -p gpurun_out; : > $O
run() { echo "## $*" >> $O; timeout 600 env "$@" 2>&1 | grep -v amdgpu.ids >> $O || echo "(rc $?)" >> $O; }
for i in 1 2; do
run python tools/text_bench.py
run MVLPT_GEMM_ONE_ROUND=0 python tools/text_bench.py
run MVLPT_GEMM_PC=0 python tools/text_bench.py
run MVLPT_GEMM_PC=2 python tools/text_bench.py
run MVLPT_GEMM_DEEP=0 python tools/text_bench.py
run MVLPT_GEMM_PCP=0 python tools/text_bench.py
run MVLPT_GEMM_ONE_ROUND_PCT=60 python tools/text_bench.py
done
for i in 1 2; do
run python tools/image_bench.py
run MVLPT_GEMM_PCP=0 python tools/image_bench.py
run MVLPT_GEMM_PHASED=0 python tools/image_bench.py
run MVLPT_GEMM_GEO=1 python tools/image_bench.py
run MVLPT_ATTN_PERSIST=0 python tools/image_bench.py
done
