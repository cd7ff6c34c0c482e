#!/bin/bash
# Kernel trace of the default bench command (run on the GPU box): per-kernel totals per step -> OUT/kernel_stats.csv, top 60 -> stdout.
#   bash tools/ktrace_step.sh OUTDIR [extra bench args]
set -e
OUT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline "$@" > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
python3 tools/kstats.py $(ls $OUT/kt/*.db | head -1) 12 $OUT/kernel_stats.csv | head -70
python3 tools/stream_stats.py $(ls $OUT/kt/*.db | head -1) 12 > $OUT/streams.txt 2>&1 || true
rm -rf $OUT/kt
