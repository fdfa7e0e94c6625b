#!/bin/bash
# Round 6: (1) the packed-fp32-free build under the partners that break the packed tower entry; (2) persistent image attention A/B; (3) bench A/B
cd "$(dirname "$0")/.."
O=gpurun_out/r06_g.txt; mkdir -p gpurun_out; : > $O
V=$PWD/mvlpt_amd
run() { echo "## $*" >> $O; timeout 1500 env "$@" >> $O 2>&1 || echo "(rc $?)" >> $O; }
run MVLPT_HIP_LIB=$V/libvar_nopk.so python tools/assemble_packed_probe.py mfma 300
run MVLPT_HIP_LIB=$V/libvar_nopk.so python tools/assemble_packed_probe.py text 600
run python -m pytest tests/test_hip_ops.py -q -x -k "attention" 
run MVLPT_ATTN_PERSIST=0 python tools/attn_bench.py
run MVLPT_ATTN_PERSIST=1 python tools/attn_bench.py
for i in 1 2; do
run MVLPT_ATTN_PERSIST=0 python tools/image_bench.py
run MVLPT_ATTN_PERSIST=1 python tools/image_bench.py
run MVLPT_HIP_LIB=$V/libvar_nopk.so MVLPT_ATTN_PERSIST=1 python tools/image_bench.py
done
for i in 1 2; do
run MVLPT_ATTN_PERSIST=0 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-trim-extra
run MVLPT_ATTN_PERSIST=1 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-trim-extra
run MVLPT_HIP_LIB=$V/libvar_nopk.so MVLPT_ATTN_PERSIST=1 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-trim-extra
done
run MVLPT_HIP_LIB=$V/libvar_nopk.so MVLPT_RESID_PACKED=1 ITERS=6000 python tools/tower_stage_probe.py 256
