"""Which torch (ATen) operators launch device work inside one steady-state training step, and from which line of this repo:
torch.profiler around ONE step after warm-up; every aten:: operator that has device time is listed with its call count, device time
and the innermost frame under neurosis_amd/ or bench.py that issued it.  (The HIP kernels of the C-ABI are not ATen operators and do not
appear: this lists what is NOT ours on the streams.)"""
import os, sys, collections, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
eng = bench.build_engine(dev)
gen = torch.Generator(device=dev).manual_seed(42)
def step():
    batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen)
    sig = bench.draw_sigmas(4, gen, dev)
    loss = eng.training_step(batch, 0, sigmas=sig); loss.backward(); eng.optimizer_step(lr=1e-6)
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::"): continue
    dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if dt <= 0: continue
    where = "?"
    for fr in (ev.stack or []):
        if ("neurosis_amd/" in fr or "bench.py" in fr) and "tools/" not in fr:
            where = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr
            break
    rows[(ev.name, where)][0] += 1
    rows[(ev.name, where)][1] += dt
tot_n = sum(v[0] for v in rows.values()); tot_t = sum(v[1] for v in rows.values())
print(f"ATen operators with device time in one step: {tot_n} launches, {tot_t / 1e3:.2f} ms of device time")
for (name, where), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
    print(f"{n:5d} x {name:28s} {t / 1e3:7.3f} ms   {where}")
# device-side view of the same step: kernels / copies that are not this library's (graph replays included)
dev_rows = collections.defaultdict(lambda: [0, 0.0])
ours = 0
for ev in prof.events():
    if str(getattr(ev, "device_type", "")).endswith("CUDA") and ev.name and not ev.name.startswith("aten::"):
        mine = not any(k in ev.name for k in ("at::native", "rocclr", "Memcpy", "Memset", "memcpy", "memset"))
        if mine:
            ours += 1
            continue
        dev_rows[ev.name[:90]][0] += 1
        dev_rows[ev.name[:90]][1] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
print(f"\ndevice activities in that step: {ours} launches of this library's kernels; foreign ones:")
for name, (n, t) in sorted(dev_rows.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"{n:5d} x {t / 1e3:7.3f} ms  {name}")
