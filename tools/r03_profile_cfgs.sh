#!/bin/bash
# Round-3: rocprofv3 kernel stats of the configurations that carry an image backward (BASELINE configs[2..4]) and cfg1.
# Usage (GPU box): bash tools/r03_profile_cfgs.sh <tag>     -> gpurun_out/<tag>/cfgN_kernel_stats.md + cfgN_line.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r03}; O=gpurun_out/$TAG; mkdir -p $O
prof() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_$name -o t -- python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary "$@" > $O/${name}.log 2>&1
  grep "^{\"metric" $O/${name}.log | tail -1 > $O/${name}_line_under_rocprof.json
  python tools/rocpd_summary.py $(ls $O/trace_$name/*.db | head -1) > $O/${name}_kernel_stats.md
  rm -rf $O/trace_$name
}
prof cfg1 --arch ViT-B/32 --batch 32 --steps 20 --warmup 5
prof cfg3 --method vpt --classes 1000 --steps 6 --warmup 2
prof cfg4 --method upt --classes 2191 --steps 4 --warmup 2
prof cfg5 --arch ViT-L/14@336px --method upt --classes 1151 --batch 128 --steps 3 --warmup 1
ls -la $O
