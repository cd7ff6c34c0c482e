"""Host time to ENQUEUE one training step against the GPU time to run it: each measured step starts from an idle GPU (synchronise), the
host clock stops when step() returns (everything queued), the GPU clock when the device is idle again.  Phases of the host time from
perf_counter marks around the engine's calls.  usage: host_time.py  (NK_GRAPH=unet | 1 | unet,vae | ... as the environment says)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

dev = torch.device("cuda", 0)
eng = bench.build_engine(dev, (1024, 1024), bench.build_conditioner(dev))
eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
gen = torch.Generator(device=dev).manual_seed(42)

def step(marks=None):
    t = [time.perf_counter()]
    batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen, False)
    sig = bench.draw_sigmas(4, gen, dev)
    t.append(time.perf_counter())
    loss = eng.training_step(batch, 0, sigmas=sig)
    t.append(time.perf_counter())
    loss.backward()
    t.append(time.perf_counter())
    eng.optimizer_step(lr=1e-6)
    t.append(time.perf_counter())
    if marks is not None:
        marks.append([1e3 * (b - a) for a, b in zip(t[:-1], t[1:])])

for _ in range(5):
    step()
torch.cuda.synchronize()
import gc
gc.collect(); gc.freeze()
host, gpu, marks = [], [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(marks)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(1e3 * (t1 - t0)); gpu.append(1e3 * (t2 - t0))
med = lambda v: sorted(v)[len(v) // 2]
print(f"NK_GRAPH={os.environ.get('NK_GRAPH', 'unet')}: host enqueue {med(host):.1f} ms/step, step from idle to idle {med(gpu):.1f} ms")
print("  host phases (median ms): batch %.1f | training_step (VAE + conditioner + UNet forward) %.1f | backward %.1f | optimizer_step %.1f"
      % tuple(med([m[i] for m in marks]) for i in range(4)))
