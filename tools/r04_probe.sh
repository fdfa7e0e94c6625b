#!/bin/bash
# pitch probe of the under-filled text GEMMs (power-of-two row pitch vs neighbours) and tile timelines of the image tower's GEMMs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_probe; mkdir -p $O
SH="7700,512,1920,2;7700,512,2048,2;7700,512,2176,2;7700,512,384,2;7700,512,512,2;7700,512,640,2;7700,512,1536,4;7700,512,1664,4"
for k in 0 1; do echo "PC=$k" >> $O/pitch.txt; MVLPT_GEMM_PC=$k python tools/gemm_mixed_bench.py "$SH" 2>&1 | grep -v amdgpu.ids >> $O/pitch.txt; done
for sh in "50432 2304 768 0" "50432 3072 768 1" "50432 768 3072 2" "50432 768 768 2"; do
  echo "== $sh" >> $O/trace.txt
  MVLPT_HIP_LIB=$PWD/mvlpt_amd/libvar_trace.so python tools/gemm_trace.py $sh 2>&1 | grep -v amdgpu.ids >> $O/trace.txt
done
cat $O/pitch.txt $O/trace.txt
