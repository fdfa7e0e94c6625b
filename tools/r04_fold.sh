#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_fold; mkdir -p $O
timeout 900 python tools/fold_check.py parity 2>&1 | grep -v amdgpu.ids > $O/parity.txt
timeout 300 python tools/fold_check.py time 2>&1 | grep -v amdgpu.ids > $O/time.txt
for m in 0 1 2; do
  MVLPT_LN_FOLD=$m timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-trim-extra 2>>$O/bench.err | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('fold $m', l['value'], l['ms_per_step'], l['step_mfma_fraction'], l['config']['loss'])" >> $O/bench.txt 2>&1
done
cat $O/parity.txt $O/time.txt $O/bench.txt; tail -3 $O/bench.err
