#!/bin/bash
# Round 6: the build without packed fp32 — micro-reproducer (17 forms), GPU suite, determinism campaign of the packed stream, A/Bs of the two switches it unblocks
cd "$(dirname "$0")/.."
O=gpurun_out/r06_h.txt; mkdir -p gpurun_out; : > $O
run() { echo "## $*" >> $O; timeout 2400 env "$@" >> $O 2>&1 || echo "(rc $?)" >> $O; }
run tools/_build/pkfma_hazard 400 20
run python tools/fold_consumer_probe.py 2460 3072 mfma 60
run python tools/fold_consumer_probe.py 2460 3072 text 60
run python tools/assemble_packed_probe.py mfma 300
echo "## pytest" >> $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 >> $O
run MVLPT_RESID_PACKED=1 ITERS=20000 python tools/tower_determinism_probe.py 256 -1 3
for i in 1 2; do
run MVLPT_RESID_PACKED=0 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-trim-extra
run MVLPT_RESID_PACKED=1 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-trim-extra
run MVLPT_RESID_PACKED=0 python tools/image_bench.py
run MVLPT_RESID_PACKED=1 python tools/image_bench.py
run MVLPT_LN_FOLD_MIN_ROWS=4096 python bench.py --arch ViT-B/32 --batch 32 --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --no-trim-extra
run MVLPT_LN_FOLD_MIN_ROWS=1024 python bench.py --arch ViT-B/32 --batch 32 --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --no-trim-extra
done
