#!/bin/bash
# Alternating runs of the metric's command from two source trees on ONE box (e.g. ab/r05 = `git archive <round-5 commit>` + make, against HEAD):
#   bash tools/ab_trees.sh ROUNDS TREE_A TREE_B      (prints ms/step, p50 per run)
R=$1; A=$2; B=$3
for i in $(seq 1 $R); do
  for t in "$A" "$B"; do
    line=$(python3 $t/bench.py --steps 8 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | head -1)
    echo "$t $(echo "$line" | python3 -c 'import json,sys; o=json.loads(sys.stdin.readline()); print(o["ms_per_step"], o["step_ms_p50"], o["loss"])')"
  done
done
