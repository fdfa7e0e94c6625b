#!/bin/bash
# A second copy of the library with extra compiler flags (debug / ablation builds), same ABI: tools/build_variant.sh <name> <flags...>
# -> mvlpt_amd/libvar_<name>.so (git-ignored, travels with gpurun); load it with MVLPT_HIP_LIB=$PWD/mvlpt_amd/libvar_<name>.so
# NOPK= (empty): WITH packed fp32 VALU ops (the product build disables them: Makefile).  ONLY_GEMM=1: recompile gemm.hip / gemm_duo.hip only and link the other objects of the product build (build/obj)
set -e
name=$1; shift
O=build/var_$name; mkdir -p $O
files="gemm gemm_duo norm attention attention_stream attention32 glue preprocess engine"
if [ -n "$ONLY_GEMM" ]; then
  files="gemm"
  for f in gemm_duo norm attention attention_stream attention32 glue preprocess engine; do cp build/obj/$f.o $O/$f.o; done
fi
for f in $files; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-inline-asm ${NOPK--Xclang -target-feature -Xclang -packed-fp32-ops} "$@" -c mvlpt_amd/csrc/$f.hip -o $O/$f.o 2> >(grep -v "is not a recognized feature" >&2) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mvlpt_amd/libvar_$name.so $O/*.o
ls -la mvlpt_amd/libvar_$name.so
