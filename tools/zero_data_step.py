"""Is the step bound by the clock the chip holds under load?  Times the same training step on the normal random data / weights and
again with every UNet / VAE weight and every input zeroed (same kernels, same launch sequence, same cycles; MFMA and data paths
toggle far less, so DVFS holds a higher clock: cdna guide section 5.4 rule 25, MI355X_MICROARCH 'DVFS give-back')."""
import os; os.environ.setdefault("NK_GRAPH", "0")   # this tool watches / flips the Python-side launches: keep the eager chain
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

dev = torch.device("cuda", 0)
eng = bench.build_engine(dev)
eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
gen = torch.Generator(device=dev).manual_seed(42); gen_cpu = torch.Generator().manual_seed(42)
zero = False
def step():
    batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen)
    if zero:
        batch = {k: torch.zeros_like(v) for k, v in batch.items()}
    sig = bench.draw_sigmas(4, gen_cpu, dev)
    noise = torch.zeros(4, 4, 128, 128, device=dev) if zero else None
    loss = eng.training_step(batch, 0, sigmas=sig, **({"noise": noise} if zero else {})); loss.backward(); eng.optimizer_step(lr=1e-6)
def timeit(n=6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): step()
import gc; gc.collect(); gc.freeze()
r1 = timeit()
zero = True
with torch.no_grad():
    eng.store.master.zero_(); eng.store.refresh()
    for p in eng.vae_encoder.parameters(): p.zero_()
for _ in range(2): step()
z1 = timeit()
print(f"random data {r1:.1f} ms/step   all-zero data {z1:.1f} ms/step   ratio {r1 / z1:.3f}")
