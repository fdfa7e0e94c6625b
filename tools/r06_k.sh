#!/bin/bash
# Round 6, final library: determinism soak of the shipped defaults (packed residual stream, LayerNorm folding from 1 024 rows) under a concurrent text tower
cd "$(dirname "$0")/.."
O=gpurun_out/r06_k.txt; mkdir -p gpurun_out; : > $O
run() { echo "## $*" >> $O; timeout 2400 env "$@" 2>&1 | grep -v amdgpu.ids >> $O || echo "(rc $?)" >> $O; }
python -c "
import ctypes;l=ctypes.CDLL('mvlpt_amd/libmvlpt_hip.so');l.mvlpt_version.restype=ctypes.c_char_p;print('library:', l.mvlpt_version().decode())" >> $O
run ITERS=30000 python tools/tower_determinism_probe.py 256 -1 3
run ITERS=6000 python tools/text_determinism_probe.py 256
run python tools/assemble_packed_probe.py mfma 300
run python tools/fold_consumer_probe.py 2460 3072 mfma 60
