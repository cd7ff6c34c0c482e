"""Per-kernel, per-grid average durations of the attention kernels from a rocprofv3 --kernel-trace results db.
usage: attn_kernel_times.py <results.db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, grid_x, grid_y, grid_z, count(*), avg(end-start), min(end-start) from kernels where name like '%attn%' "
                 "group by name, grid_x, grid_y, grid_z order by name, grid_z * grid_y, grid_x").fetchall()
for r in rows:
    print(f"{r[0][:46]:46s} grid {r[1]:6d} x {r[2]:3d} x {r[3]:3d}  n {r[4]:4d}  avg {r[5] / 1e3:8.1f} us  min {r[6] / 1e3:8.1f} us")
