#!/bin/bash
# Round-3 A/B: mixed pair (default) vs 16-bit pairs (MVLPT_SPLIT_LO8=0): parity report + bench lines of every config.
cd $GRAFT_REPO_ROOT
TAG=${1:-r03b}; O=gpurun_out/$TAG; mkdir -p $O
python tools/parity_report.py fp16 > $O/parity_lo8.txt 2>&1
MVLPT_SPLIT_LO8=0 python tools/parity_report.py fp16 > $O/parity_pair16.txt 2>&1
python bench.py --steps 30 --warmup 8 2>$O/bench.err | tail -1 > $O/bench_line.json
MVLPT_SPLIT_LO8=0 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_pair16.json
bash tools/config_sweep.sh > $O/config_sweep_lo8.txt 2>&1
MVLPT_SPLIT_LO8=0 bash tools/config_sweep.sh > $O/config_sweep_pair16.txt 2>&1
