#!/bin/bash
# Persistent resident pair attention A/B (GPU box): parity tests, kernel timings with MVLPT_ATTN32_PERSIST = 0 / 1, cfg3 step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_attn; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_mixed_pair.py -m gpu -x -q -k "attention32 or mixed" 2>&1 | tail -5 > $O/pytest.txt
for rep in 1 2; do for k in 0 1; do
  echo "PERSIST=$k" >> $O/attn.txt
  MVLPT_ATTN32_PERSIST=$k timeout 300 python tools/attn_bench.py 2>&1 | grep -A5 "split-precision" >> $O/attn.txt
done; done
for k in 0 1; do
  MVLPT_ATTN32_PERSIST=$k timeout 600 python bench.py --method vpt --classes 1000 --steps 10 --warmup 3 --no-cpu-baseline --no-trim-extra 2>>$O/bench.err | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('cfg3 persist $k', l['value'], l['ms_per_step'], l['config']['loss'])" >> $O/bench.txt 2>&1
done
cat $O/pytest.txt $O/attn.txt $O/bench.txt; tail -3 $O/bench.err
