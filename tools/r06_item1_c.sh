#!/bin/bash
# Round 6, item 1, third bundle: extended micro-reproducer (13 instruction forms) + the GPU suite
cd "$(dirname "$0")/.."
O=gpurun_out/r06_item1_c.txt; mkdir -p gpurun_out; : > $O
run() { echo "## $*" >> $O; timeout 900 env "$@" >> $O 2>&1 || echo "(rc $?)" >> $O; }
run tools/_build/pkfma_hazard 400 20
echo "## pytest" >> $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 >> $O
