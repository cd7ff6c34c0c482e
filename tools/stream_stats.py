"""Per-stream busy time and the main stream's kernel mix / idle gaps from a rocprofv3 --kernel-trace db.
usage: stream_stats.py <db> <steps>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2])
rows = db.execute("select stream_id, name, start, end from kernels order by start").fetchall()
by = collections.defaultdict(list)
for s, n, a, b in rows: by[s].append((a, b, n))
tot = {s: sum(b - a for a, b, _ in v) for s, v in by.items()}
main = max(tot, key=tot.get)
for s, v in sorted(by.items(), key=lambda kv: -tot[kv[0]]):
    span = v[-1][1] - v[0][0]
    print(f"stream {s}: {len(v):6d} kernels, busy {tot[s]/steps/1e6:8.2f} ms/step, first-to-last span {span/steps/1e6:8.2f} ms/step")
v = by[main]
gaps = [max(0, v[i + 1][0] - v[i][1]) for i in range(len(v) - 1)]
big = sum(g for g in gaps if g > 20000)
print(f"main stream idle between kernels: {sum(gaps)/steps/1e6:.2f} ms/step (gaps > 20 us: {big/steps/1e6:.2f} ms/step, count {sum(1 for g in gaps if g > 20000)/steps:.0f}/step)")
agg = collections.defaultdict(lambda: [0, 0])
for a, b, n in v: agg[n[:60]][0] += 1; agg[n[:60]][1] += b - a
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]: print(f"  {n:60s} {c/steps:7.1f}/step {t/steps/1e6:8.2f} ms/step {t/c/1e3:8.1f} us")
