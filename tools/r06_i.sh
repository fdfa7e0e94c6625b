#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_i.txt; mkdir -p gpurun_out; : > $O
run() { echo "## $*" >> $O; timeout 1500 env "$@" >> $O 2>&1 || echo "(rc $?)" >> $O; }
run tools/_build/lds_dma_bw
run MVLPT_HIP_LIB=$PWD/mvlpt_amd/libvar_attn_trace.so python tools/attn_trace.py bwd 256 205 12
run MVLPT_HIP_LIB=$PWD/mvlpt_amd/libvar_attn_trace.so python tools/attn_trace.py bwd 256 205 12
run python -m pytest tests/test_hip_bench_contract.py -q -x
run python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary --no-trim-extra
