#!/bin/bash
# A second copy of the library with extra compiler flags (debug / ablation builds), same ABI: tools/build_variant.sh <name> <flags...>
# -> mvlpt_amd/libvar_<name>.so (git-ignored, travels with gpurun); load it with MVLPT_HIP_LIB=$PWD/mvlpt_amd/libvar_<name>.so
set -e
name=$1; shift
O=build/var_$name; mkdir -p $O
for f in gemm gemm_duo norm attention attention_stream attention32 glue preprocess engine; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-inline-asm "$@" -c mvlpt_amd/csrc/$f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mvlpt_amd/libvar_$name.so $O/*.o
ls -la mvlpt_amd/libvar_$name.so
