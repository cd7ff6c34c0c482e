"""Summarise a rocprofv3 --pmc results db: per kernel name, dispatch count and the per-dispatch mean of each counter.
usage: pmc_summary.py <results.db> [out.csv]"""
import collections, csv, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, avg(value), count(*), sum(value) from counters_collection group by 1, 2").fetchall()
agg = collections.defaultdict(dict)
cnt = {}
for k, c, v, n, s in rows:
    agg[k][c] = v
    cnt[k] = n
counters = sorted({c for d in agg.values() for c in d})
out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
out.writerow(["kernel", "dispatches"] + [c + "_per_dispatch" for c in counters])
for k in sorted(agg, key=lambda k: -cnt[k] * max(agg[k].values())):
    out.writerow([k[:120], cnt[k]] + [f"{agg[k].get(c, 0):.6g}" for c in counters])
