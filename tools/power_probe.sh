#!/bin/bash
# Clock / power of the GPU while a command runs (rocm-smi polled twice a second): `bash tools/power_probe.sh <out-file> <command...>`
# Evidence for DESIGN.md "power cap": under sustained MFMA load the MI355X sits at its 1400 W cap and sclk drops from 2400 to ~1760 MHz.
out=$1; shift
("$@" > ${out}.cmd.txt 2>&1) &
pid=$!
: > $out
t=0
while kill -0 $pid 2>/dev/null; do
  echo -n "t=$t " >> $out
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/GPU\[0\]//; s/clock level//' | tr -s '\t ' ' ' | tr '\n' ' ' >> $out
  echo >> $out
  sleep 0.5; t=$((t+1))
done
wait $pid
