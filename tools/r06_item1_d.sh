#!/bin/bash
# Round 6, item 1 (b): the packed residual stream under the text tower — product build vs a build with every compiler wait forced to zero
cd "$(dirname "$0")/.."
O=gpurun_out/r06_item1_d.txt; mkdir -p gpurun_out; : > $O
V=$PWD/mvlpt_amd
run() { echo "## $*" >> $O; timeout 900 env "$@" >> $O 2>&1 || echo "(rc $?)" >> $O; }
run MVLPT_RESID_PACKED=1 ITERS=12000 python tools/tower_determinism_probe.py 256 -1 3
run MVLPT_HIP_LIB=$V/libvar_fz.so MVLPT_RESID_PACKED=1 ITERS=12000 python tools/tower_determinism_probe.py 256 -1 3
echo "## pytest (new / changed tests)" >> $O
timeout 1200 python -m pytest tests/test_ctx_init.py tests/test_hip_bench_contract.py tests/test_hip_determinism.py tests/test_hip_ops.py -m gpu -q -x -k "ctx_init or bench or hazard or ragged" 2>&1 | tail -15 >> $O
