import os; os.environ.setdefault("NK_GRAPH", "0")   # this tool watches / flips the Python-side launches: keep the eager chain
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select stream_id, name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
prev = collections.Counter(); nxt = collections.Counter(); sizes = collections.Counter()
for i, r in enumerate(rows):
    if "copyBuffer" in r[1] or "FillFunctor" in r[1] or "reduce_kernel" in r[1]:
        key = r[1][:40]
        sizes[(key, r[4])] += 1
        j = i - 1
        while j >= 0 and ("copyBuffer" in rows[j][1]): j -= 1
        prev[(key, rows[j][1][:50])] += 1
        j = i + 1
        while j < len(rows) and ("copyBuffer" in rows[j][1]): j += 1
        if j < len(rows): nxt[(key, rows[j][1][:50])] += 1
print("sizes:"); [print("  ", k, v) for k, v in sizes.most_common(14)]
print("preceded by:"); [print("  ", k, v) for k, v in prev.most_common(14)]
print("followed by:"); [print("  ", k, v) for k, v in nxt.most_common(14)]
