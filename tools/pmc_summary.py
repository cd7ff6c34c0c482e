"""Summarise a rocprofv3 --pmc results db: per kernel name x grid, mean of each counter."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()][:10])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
print(cols)
rows = db.execute("select kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name, avg(value), count(*) from counters_collection group by 1,2,3,4,5").fetchall()
agg = collections.defaultdict(dict)
for k, gx, gy, gz, c, v, n in rows:
    agg[(k[:48], gx, gy, gz)][c] = v
for key, d in agg.items():
    if "nk_gemm" not in key[0]: continue
    print(key, " ".join(f"{c}={v:.4g}" for c, v in sorted(d.items())))
