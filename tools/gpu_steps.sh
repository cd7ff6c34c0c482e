#!/bin/bash
# Run GPU steps one after another on the box; stop at the first one that times out, is killed or dies on a signal / GPU fault (exit 124 or >= 128):
# no further GPU step is started after a hang or a faulting kernel.  Usage: tools/gpu_steps.sh OUTDIR "name|seconds|command" ...   (stdout + stderr of each step -> OUTDIR/name.log)
out="$1"; shift
mkdir -p "$out"
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "== $name (limit ${secs}s)"
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "rc=$rc" >> "$out/$name.log"
  echo "   rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "step $name timed out or died on a signal (rc $rc): stopping"; exit $rc; fi
done
exit 0
