#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_item1_f.txt; mkdir -p gpurun_out; : > $O
run() { echo "## $*" >> $O; timeout 1500 env "$@" >> $O 2>&1 || echo "(rc $?)" >> $O; }
run python tools/assemble_packed_probe.py none 100
run python tools/assemble_packed_probe.py text 1500
run python tools/assemble_packed_probe.py mfma 600
run python tools/assemble_packed_probe.py mem 300
run python tools/assemble_packed_probe.py valu 300
