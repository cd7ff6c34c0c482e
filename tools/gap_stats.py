"""Idle time between consecutive kernels of the busiest stream, by gap size (rocprofv3 --kernel-trace db). usage: gap_stats.py <db> <steps>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2])
rows = db.execute("select stream_id, start, end from kernels order by start").fetchall()
by = collections.defaultdict(list)
for s, a, b in rows: by[s].append((a, b))
main = max(by, key=lambda s: sum(b - a for a, b in by[s]))
v = by[main]
gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
# global union across streams: time when NO kernel runs on any stream (true GPU idle), inside windows without long pauses
allk = sorted((a, b) for s in by for a, b in by[s])
idle = []; cur_end = allk[0][1]
for a, b in allk[1:]:
    if a > cur_end: idle.append(a - cur_end)
    cur_end = max(cur_end, b)
for name, g in (("main-stream gaps", gaps), ("whole-GPU idle", idle)):
    print(name)
    for lo, hi in [(0, 2e3), (2e3, 5e3), (5e3, 10e3), (10e3, 30e3), (30e3, 200e3), (200e3, 1e12)]:
        sel = [x for x in g if lo < x <= hi]
        print(f"  {lo/1e3:6.0f}-{hi/1e3:<8.0f} us: {len(sel)/steps:8.1f}/step {sum(sel)/steps/1e6:8.2f} ms/step")
rows2 = db.execute("select start, end, name from kernels order by start").fetchall()
cur_end = rows2[0][1]; last = rows2[0][2]
pairs = collections.Counter(); tsum = collections.Counter()
for a, b, n in rows2[1:]:
    if a > cur_end and 10e3 < a - cur_end <= 200e3:
        key = (last[:44], n[:44]); pairs[key] += 1; tsum[key] += a - cur_end
    if b > cur_end: cur_end = b; last = n
print("whole-GPU idle 10-200 us, by (kernel before -> kernel after):")
for k, t in tsum.most_common(22): print(f"  {t/steps/1e6:6.2f} ms/step {pairs[k]/steps:6.1f}/step  {k[0]}  ->  {k[1]}")
