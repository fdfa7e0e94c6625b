#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_item1_e.txt; mkdir -p gpurun_out; : > $O
run() { echo "## $*" >> $O; timeout 1500 env "$@" >> $O 2>&1 || echo "(rc $?)" >> $O; }
run MVLPT_RESID_PACKED=1 ITERS=8000 python tools/tower_stage_probe.py 256
run MVLPT_RESID_PACKED=0 ITERS=1500 python tools/tower_stage_probe.py 256
