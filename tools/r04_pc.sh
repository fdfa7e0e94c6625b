#!/bin/bash
# Text-tower GEMM A/B (GPU box): data-movement waves (MVLPT_GEMM_PC) x padded pair rows (MVLPT_WIDE_PITCH); tower alone and the headline step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_pc; mkdir -p $O
for rep in 1 2; do for cfgv in "0 0" "1 0" "1 1"; do
  set -- $cfgv
  echo "PC=$1 WIDE_PITCH=$2" >> $O/text.txt
  MVLPT_GEMM_PC=$1 MVLPT_WIDE_PITCH=$2 timeout 300 python tools/text_bench.py 2>&1 | grep -v amdgpu.ids >> $O/text.txt
  MVLPT_GEMM_PC=$1 MVLPT_WIDE_PITCH=$2 timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-trim-extra 2>>$O/bench.err | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('pc $1 pitch $2', l['value'], l['ms_per_step'], l['step_mfma_fraction'], l['config']['loss'])" >> $O/bench.txt 2>&1
done; done
cat $O/text.txt $O/bench.txt; tail -3 $O/bench.err
