#!/bin/bash
# HBM-traffic PMC passes only (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 runs) for the headline and BASELINE configs [2], [3], [4]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
pmc() {
  local name=$1; shift
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf_$name -o p -- python bench.py --no-cpu-baseline --no-kernel-timing --no-trim-extra "$@" > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw_$name -o p -- python bench.py --no-cpu-baseline --no-kernel-timing --no-trim-extra "$@" > /dev/null 2>&1
  python tools/traffic_from_pmc.py $(ls $O/pf_$name/*.db | head -1) $(ls $O/pw_$name/*.db | head -1) "$name: bench.py $*" > $O/gemm_hbm_traffic_$name.json
  rm -rf $O/pf_$name $O/pw_$name
}
pmc headline --steps 3 --warmup 2
pmc cfg3 --method vpt --classes 1000 --steps 2 --warmup 1
pmc cfg4 --method upt --classes 2191 --steps 2 --warmup 1
pmc cfg5 --arch ViT-L/14@336px --method upt --classes 1151 --batch 128 --steps 2 --warmup 1
cat $O/gemm_hbm_traffic_headline.json
