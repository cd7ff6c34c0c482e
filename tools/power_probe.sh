#!/bin/bash
# Samples the card's power draw and clocks (rocm-smi) while a command runs: is a kernel's rate set by its schedule or by the power cap?
#   tools/power_probe.sh OUTFILE -- python tools/bench_gemm.py
out=$1; shift; shift
"$@" > "$out.cmd.log" 2>&1 &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
  { date +%s.%N; rocm-smi --showpower --showclocks --showuse 2>&1 | grep -i "power\|sclk\|mclk\|GPU use" ; } >> "$out"
  sleep 0.5
done
wait $pid
