"""Per-kernel totals from a rocprofv3 --kernel-trace results db.  usage: kstats.py <db> [steps] [out.csv]"""
import csv, math, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), avg((end-start)*(end-start)) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
if len(sys.argv) > 3:
    w = csv.writer(open(sys.argv[3], "w", newline=""), quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for n, c, t, a, mn, mx, sq in rows: w.writerow([n, c, t, round(a, 3), round(100 * t / tot, 2), mn, mx, round(math.sqrt(max(sq - a * a, 0)), 3)])
for r in rows[:40]: print(f"{r[0][:64]:64s} {r[1]:6d} {r[2]/steps/1e6:8.2f} ms/step {r[3]/1e3:8.1f} us")
print("total ms/step", tot / steps / 1e6)
