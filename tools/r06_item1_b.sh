#!/bin/bash
# Round 6, item 1, second bundle: micro-reproducer of the instruction sequence + regression-fit forensics on the real kernel
cd "$(dirname "$0")/.."
O=gpurun_out/r06_item1_b.txt; mkdir -p gpurun_out; : > $O
V=$PWD/mvlpt_amd
run() { echo "## $*" >> $O; timeout 600 env "$@" >> $O 2>&1 || echo "(rc $?)" >> $O; }
run tools/_build/pkfma_hazard 400 20
run MVLPT_HIP_LIB=$V/libvar_pk.so python tools/fold_consumer_probe.py 2460 3072 text 40
run MVLPT_HIP_LIB=$V/libvar_pk.so python tools/fold_consumer_probe.py 2460 3072 mfma 40
run MVLPT_HIP_LIB=$V/libvar_pk.so PARTNER_SCALE=4 python tools/fold_consumer_probe.py 2460 3072 mfma 40
echo "## pytest" >> $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 >> $O
