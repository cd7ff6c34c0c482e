#!/bin/bash
# Run GPU steps one after another on the box; stop at the first one that times out or is killed (exit 124 / 137): no further GPU step is started
# after a hang.  Usage: tools/gpu_steps.sh OUTDIR "name|seconds|command" ...   (stdout + stderr of each step -> OUTDIR/name.log)
out="$1"; shift
mkdir -p "$out"
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "== $name (limit ${secs}s)"
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "rc=$rc" >> "$out/$name.log"
  echo "   rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out: stopping"; exit $rc; fi
done
exit 0
