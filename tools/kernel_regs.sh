#!/bin/bash
# VGPR / spill table of every kernel in one .hip file:  tools/kernel_regs.sh mvlpt_amd/csrc/gemm.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-inline-asm "$@" \
  -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 |
  grep -E "Function Name|    VGPRs:|AGPRs:|ScratchSize|VGPRs Spill|Occupancy" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' |
  awk '/Function Name/ {if (n) print n, v, a, s, sp, o; n=$3; next} /^VGPRs:/ {v="vgpr="$2} /^AGPRs:/ {a="agpr="$2} /ScratchSize/ {s="scratch="$4} /VGPRs Spill/ {sp="spill="$3} /Occupancy/ {o="occ="$4} END {print n, v, a, s, sp, o}' |
  while read n rest; do echo "$(echo $n | c++filt | sed -e 's/mvlpt:://g' -e 's/(.*//' -e 's/void //') $rest"; done
