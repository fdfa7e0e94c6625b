#!/bin/bash
# 256x128 GEMM with data-movement waves (gemm_pcp_kernel) A/B: parity tests, kernel timings, towers alone, headline and cfg3 steps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_pcp; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_mixed_pair.py tests/test_hip_fold.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
for rep in 1 2; do for k in 0 1; do
  echo "PCP=$k" >> $O/gemm.txt
  MVLPT_GEMM_PCP=$k timeout 300 python tools/gemm_bench.py "50432,768,768,2;50432,768,768,0;20000,768,3072,2;16640,2304,768,0" 2>&1 | grep -v amdgpu.ids >> $O/gemm.txt
  MVLPT_GEMM_PCP=$k timeout 300 python tools/gemm_mixed_bench.py "7700,2048,512,5;7700,2048,512,6;7700,1536,512,7;50432,768,768,2" 2>&1 | grep -v amdgpu.ids >> $O/gemm.txt
  MVLPT_GEMM_PCP=$k timeout 300 python tools/text_bench.py 2>&1 | grep -v amdgpu.ids >> $O/gemm.txt
  MVLPT_GEMM_PCP=$k timeout 300 python tools/image_bench.py 2>&1 | grep -v amdgpu.ids >> $O/gemm.txt
  MVLPT_GEMM_PCP=$k timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-trim-extra 2>>$O/bench.err | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('pcp $k', l['value'], l['ms_per_step'], l['step_mfma_fraction'], l['config']['loss'])" >> $O/bench.txt 2>&1
done; done
for k in 0 1; do
  MVLPT_GEMM_PCP=$k timeout 600 python bench.py --method vpt --classes 1000 --steps 10 --warmup 3 --no-cpu-baseline --no-trim-extra 2>>$O/bench.err | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('cfg3 pcp $k', l['value'], l['ms_per_step'], l['config']['loss'])" >> $O/bench.txt 2>&1
done
cat $O/pytest.txt $O/gemm.txt $O/bench.txt; tail -3 $O/bench.err
