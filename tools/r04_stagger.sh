#!/bin/bash
# Experiment: workgroup start stagger in the persistent GEMM (build with -DMVLPT_GEMM_STAGGER_EXP into exp_build/libmvlpt_stagger.so).
cd $GRAFT_REPO_ROOT
SH="50432,2304,768,0;50432,768,768,2;50432,3072,768,1;50432,768,3072,2"
for st in 0 8000 16000 32000 64000 0; do
  echo "== stagger $st cycles"
  MVLPT_HIP_LIB=$PWD/exp_build/libmvlpt_stagger.so MVLPT_GEMM_STAGGER=$st python tools/gemm_bench.py "$SH" 40 2>&1 | grep -v amdgpu.ids
done
