"""Merge the three PMC passes and the kernel-trace stats of the bench command into profiles/<round>_pmc_summary.csv:
per kernel HBM bytes per dispatch (2 x FETCH_SIZE KB on gfx950 + WRITE_SIZE KB), achieved GB/s (bytes / mean duration of
the kernel-trace run) and MFMA busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128))."""
import csv, os, sys
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
R = sys.argv[1] if len(sys.argv) > 1 else "r03"      # round prefix of the files under profiles/
def load(name):
    r = list(csv.reader(open(os.path.join(P, name)))); return r[0], {row[0][:100]: row for row in r[1:]}
hf, F = load(f"{R}_pmc_fetch_size.csv"); hw, W = load(f"{R}_pmc_write_size.csv"); hm, M = load(f"{R}_pmc_mfma.csv"); hs, S = load(f"{R}_bench_kernel_stats.csv")
rows = []
for k in F:
    f = float(F[k][2]) * 2 * 1024; w = float(W[k][2]) * 1024 if k in W else 0.0; n = int(F[k][1])
    d = float(S[k][3]) if k in S else None
    u = None
    if k in M:
        dd = dict(zip(hm[2:], map(float, M[k][2:])))
        if dd.get("GRBM_GUI_ACTIVE_per_dispatch", 0) > 0: u = dd["SQ_VALU_MFMA_BUSY_CYCLES_per_dispatch"] / (dd["GRBM_GUI_ACTIVE_per_dispatch"] * 128)
    rows.append((k, n, f, w, d, u))
rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
with open(os.path.join(P, f"{R}_pmc_summary.csv"), "w", newline="") as fo:
    o = csv.writer(fo)
    o.writerow(["kernel", "dispatches(pmc run)", "hbm_read_bytes_per_dispatch(2xFETCH_SIZE KB)", "hbm_write_bytes_per_dispatch(WRITE_SIZE KB)", "avg_duration_ns(kernel-trace run)", "achieved_GBps", "mfma_busy_frac"])
    for k, n, f, w, d, u in rows:
        o.writerow([k, n, f"{f:.4g}", f"{w:.4g}", f"{d:.0f}" if d else "", f"{(f + w) / d:.0f}" if d else "", f"{u:.3f}" if u is not None else ""])
for k, n, f, w, d, u in rows[:26]:
    print(f"{k[:58]:58s} n={n:5d} rd={f/1e6:9.1f}MB wr={w/1e6:8.1f}MB dur={(d or 0)/1e3:8.1f}us {((f+w)/d if d else 0):6.0f} GB/s mfma={u if u is not None else -1:.3f}")
te = lambda r: r[0].startswith(("void nk_gemm", "nk_gemm", "void nk_conv3x3_halo"))
tn = sum(r[1] for r in rows if te(r)); tb = sum(r[1] * (r[2] + r[3]) for r in rows if te(r))
print("tile engine: launches", tn, "avg HBM bytes/launch %.1f MB" % (tb / tn / 1e6))
