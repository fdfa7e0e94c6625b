#!/bin/bash
# Round-3: headline step + the two towers alone: wall times and rocprofv3 kernel stats.  bash tools/r03_profile_headline.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r03}; O=gpurun_out/$TAG; mkdir -p $O
python tools/text_bench.py > $O/towers_alone.txt 2>&1
python tools/image_bench.py >> $O/towers_alone.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_text -o t -- python tools/text_bench.py > /dev/null 2>&1
python tools/rocpd_summary.py $(ls $O/trace_text/*.db | head -1) > $O/text_tower_kernel_stats.md
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_b -o t -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-trim-extra --no-secondary > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric" $O/bench_under_rocprof.log | tail -1 > $O/bench_line_under_rocprof.json
python tools/rocpd_summary.py $(ls $O/trace_b/*.db | head -1) > $O/bench_kernel_stats.md
rm -rf $O/trace_text $O/trace_b
cat $O/towers_alone.txt
