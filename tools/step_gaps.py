"""Where one steady-state step's wall time goes on the main stream, from a rocprofv3 --kernel-trace db: the window between the last two
launches of a marker kernel (default: the VAE's conv_in, once per step), its busy / idle split, the idle time attributed to the kernel that
FOLLOWS each gap (top entries), and the launch count per kernel name.
usage: step_gaps.py <db> [marker substring]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "conv3x3_few_channels"
rows = db.execute("select stream_id, name, start, end from kernels order by start").fetchall()
marks = [a for s, n, a, b in rows if marker in n]
if marker == "conv3x3_few_channels":     # twice per step (the VAE's conv_in, then the UNet's): every other one
    marks = marks[len(marks) % 2::2]
t0, t1 = marks[-3], marks[-2]
win = [(s, n, a, b) for s, n, a, b in rows if t0 <= a < t1]
print(f"window: {(t1 - t0) / 1e6:.2f} ms, {len(win)} kernels")
by = collections.defaultdict(list)
for s, n, a, b in win: by[s].append((a, b, n))
main = max(by, key=lambda s: sum(b - a for a, b, _ in by[s]))
for s, v in sorted(by.items(), key=lambda kv: -sum(b - a for a, b, _ in kv[1])):
    print(f"stream {s}: {len(v):5d} kernels, busy {sum(b - a for a, b, _ in v) / 1e6:7.2f} ms, from {(v[0][0] - t0) / 1e6:7.2f} to {(v[-1][1] - t0) / 1e6:7.2f} ms")
# union of busy intervals over ALL streams: time when the GPU runs nothing at all
iv = sorted((a, b) for s, n, a, b in win)
cov, cur_a, cur_b = 0, iv[0][0], iv[0][1]
for a, b in iv[1:]:
    if a > cur_b: cov += cur_b - cur_a; cur_a, cur_b = a, b
    else: cur_b = max(cur_b, b)
cov += cur_b - cur_a
print(f"some kernel running (any stream): {cov / 1e6:.2f} ms; nothing running: {(t1 - t0 - cov) / 1e6:.2f} ms")
v = by[main]
gap_by = collections.defaultdict(lambda: [0, 0])
hist = collections.Counter()
for i in range(len(v) - 1):
    g = max(0, v[i + 1][0] - v[i][1])
    gap_by[v[i + 1][2][:50]][0] += 1; gap_by[v[i + 1][2][:50]][1] += g
    hist[min(int(g / 1000).bit_length(), 12)] += g
tot = sum(x[1] for x in gap_by.values())
print(f"main stream {main}: idle between its kernels {tot / 1e6:.2f} ms; by gap size (us, upper bound: ms):",
      {(1 << k): round(t / 1e6, 2) for k, t in sorted(hist.items())})
print("idle time in front of (kernel): count, total ms, mean us")
for n, (c, t) in sorted(gap_by.items(), key=lambda kv: -kv[1][1])[:25]: print(f"  {n:50s} {c:5d} {t / 1e6:7.2f} {t / c / 1e3:7.1f}")
cnt = collections.Counter(n[:50] for a, b, n in v)
print("main-stream launches by kernel:", dict(cnt.most_common(40)))
print("largest single gaps on the main stream: at ms, length us, after -> before, side-stream kernels overlapping the gap")
big = sorted(((v[i + 1][0] - v[i][1], i) for i in range(len(v) - 1)), reverse=True)[:30]
others = [(a, b, n, s) for s, vv in by.items() if s != main for a, b, n in vv]
for g, i in sorted(big, key=lambda x: x[1]):
    ga, gb = v[i][1], v[i + 1][0]
    ov = [f"s{s}:{n[:28]}" for a, b, n, s in others if a < gb and b > ga]
    print(f"  {(ga - t0) / 1e6:8.2f} {g / 1e3:8.1f}  {v[i][2][:34]:34s} -> {v[i + 1][2][:34]:34s} | {len(ov)} {ov[:3]}")
