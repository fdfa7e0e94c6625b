#!/bin/bash
# Round 4: the two towers on disjoint compute units (CustomCLIP.set_cu_partition): headline bench over the text partition size.
#   bash tools/r04_cu_partition_sweep.sh [tag]   -> gpurun_out/<tag>/cu_partition_sweep.txt
cd $GRAFT_REPO_ROOT
TAG=${1:-r04}; O=gpurun_out/$TAG; mkdir -p $O
: > $O/cu_partition_sweep.txt
for t in ${SWEEP:-0 32 48 64 80 96 128 0}; do
  echo "== text_cus $t" >> $O/cu_partition_sweep.txt
  timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-trim-extra --text-cus $t 2>>$O/cu_partition_sweep.err | tail -1 |
    python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l.get('roofline',{}); print(json.dumps({k: l[k] for k in ('value','ms_per_step','step_mfma_fraction')} | {'clock': l.get('clock',{}).get('sclk_mhz_avg'), 'power': l.get('clock',{}).get('power_w_avg'), 'gemm_frac': r.get('frac'), 'cu_partition': l['config']['cu_partition']}))" >> $O/cu_partition_sweep.txt 2>&1
done
cat $O/cu_partition_sweep.txt
