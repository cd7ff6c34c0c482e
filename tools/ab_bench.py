"""Alternating A/B of the metric's own command on ONE box: every variant is a fresh `python bench.py` process (graph replay, all streams, as
the driver runs it), variants interleaved round by round so that box-to-box and drift effects cancel.
usage: python tools/ab_bench.py [--rounds 2] [--steps 8] [--args "--precomputed-te"] VARIANT [VARIANT ...]
       a VARIANT is ENV=VALUE[,ENV=VALUE...] or the word base.   (The parent never touches the GPU.)"""
import argparse
import json
import os
import subprocess
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--args", default="")
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {v: [] for v in a.variants}
for rnd in range(a.rounds):
    for v in a.variants:
        env = dict(os.environ)
        if v != "base":
            for kv in v.split(","):
                k, val = kv.split("=", 1)
                env[k] = val
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", str(a.steps), "--no-cpu-baseline", "--no-roofline"] + a.args.split(),
                           env=env, capture_output=True, text=True)
        line = next((l for l in p.stdout.splitlines() if l.startswith("{")), None)
        if line is None:
            print(f"{v}: FAILED rc={p.returncode}\n{p.stderr[-1500:]}", flush=True)
            res[v].append((float("nan"), float("nan")))
            continue
        o = json.loads(line)
        res[v].append((o["ms_per_step"], o["step_ms_p50"]))
        print(f"round {rnd} {v:40s} ms/step {o['ms_per_step']:8.2f}  p50 {o['step_ms_p50']:8.2f}  loss {o['loss']}", flush=True)
print("---- ms/step (p50) per round")
for v, r in res.items():
    print(f"{v:40s} " + "  ".join(f"{m:7.2f} ({p:7.2f})" for m, p in r))
