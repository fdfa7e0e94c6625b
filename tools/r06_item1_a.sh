#!/bin/bash
# Round 6, item 1, first bundle: forensics + resource-poisoning partners on the round-5 failing kernel form (one gpurun call)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_item1_a.txt; mkdir -p gpurun_out; : > $O
V=$PWD/mvlpt_amd
run() { echo "## $*" >> $O; timeout 300 env "$@" >> $O 2>&1 || echo "(rc $?)" >> $O; }
probe() { lib=$1; shift; run MVLPT_HIP_LIB=$V/$lib python tools/fold_consumer_probe.py 2460 3072 "$@"; }
probe libvar_pk.so text
probe libvar_pk.so none
probe libvar_pkl.so none 20
probe libvar_pkv.so none 20
probe libvar_pkl.so text 30
probe libvar_pkv.so text 30
for p in vgpr64 vgpr128 vgpr256 vgpr512 sgpr lds40960 lds81920 lds163840 mem valu mfma ldsrw; do probe libvar_pk.so $p 30; done
probe libvar_pk2.so text
probe libvar_pk3.so text
probe libvar_scl.so none 20
probe libvar_scv.so none 20
probe libmvlpt_hip.so text
echo "## bench" >> $O
python bench.py --steps 20 --warmup 5 --no-secondary > gpurun_out/r06_bench_start.json 2>> $O
tail -c 1500 gpurun_out/r06_bench_start.json >> $O
