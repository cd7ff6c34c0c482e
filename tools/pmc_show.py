"""Per-kernel means of the counters of one `rocprofv3 --pmc ... --output-format csv` pass.  usage: pmc_show.py <dir> [name-filter]"""
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if flt and flt not in n:
            continue
        a = agg[n[:70]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for n, cs in agg.items():
    print(n)
    for c, (k, v) in sorted(cs.items()):
        print(f"    {c:32s} launches {k:5d}  mean {v / k:16.1f}")
