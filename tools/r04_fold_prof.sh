#!/bin/bash
# image tower alone under rocprofv3, LayerNorm folding off / on: per-kernel table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_fold; mkdir -p $O
for m in 0 1; do
  MVLPT_LN_FOLD=$m timeout 600 rocprofv3 --kernel-trace --stats -d $O/tr_$m -o t -- python tools/image_bench.py > /dev/null 2>&1
  python tools/rocpd_summary.py $(ls $O/tr_$m/*.db | head -1) | head -16 > $O/image_tower_kernels_fold$m.md
  rm -rf $O/tr_$m
done
cat $O/image_tower_kernels_fold0.md $O/image_tower_kernels_fold1.md
