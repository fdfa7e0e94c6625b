#!/bin/bash
# Softmax VALU trimming in the attention forwards (mask only where needed, scale folded into the exponent FMA): A/B against a library
# linked with HEAD's attention objects (mvlpt_amd/libvar_old.so), same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_attn2; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_mixed_pair.py tests/test_hip_model.py -m gpu -x -q -k "attention or full or tiny" 2>&1 | tail -4 > $O/pytest.txt
for rep in 1 2; do for v in old new; do
  L=""; [ $v = old ] && L=$PWD/mvlpt_amd/libvar_old.so
  echo "== $v" >> $O/attn.txt
  MVLPT_HIP_LIB=$L timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids >> $O/attn.txt
  MVLPT_HIP_LIB=$L timeout 300 python tools/image_bench.py 2>&1 | grep -v amdgpu.ids >> $O/attn.txt
  MVLPT_HIP_LIB=$L timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-trim-extra 2>>$O/bench.err | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$v', l['value'], l['ms_per_step'], l['step_mfma_fraction'], l['config']['loss'])" >> $O/bench.txt 2>&1
done; done
cat $O/pytest.txt $O/attn.txt $O/bench.txt; tail -3 $O/bench.err
