"""How far does the weight-gradient stream lag behind the dgrad chain when backward ends?  Events on both streams at every
ops.join_wgrad_stream(); prints, per step, main-arrival -> side-done for the joins (the last one is the end of the UNet backward)."""
import os; os.environ.setdefault("NK_GRAPH", "0")   # this tool watches / flips the Python-side launches: keep the eager chain
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from neurosis_amd import ops
import neurosis_amd.nn as nkn
import neurosis_amd.modules.diffusion.openaimodel as om

dev = torch.device("cuda", 0)
eng = bench.build_engine(dev)
eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
gen = torch.Generator(device=dev).manual_seed(42); gen_cpu = torch.Generator().manual_seed(42)
pairs = []
orig = ops.join_wgrad_stream
def join(owner=None):
    side = eng.store.state.wgrad_stream
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(torch.cuda.current_stream()); b.record(side)
    pairs.append((a, b))
    orig(owner)
ops.join_wgrad_stream = join
def step():
    batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen)
    sig = bench.draw_sigmas(4, gen_cpu, dev)
    s0 = torch.cuda.Event(enable_timing=True); s0.record()
    loss = eng.training_step(batch, 0, sigmas=sig)
    f = torch.cuda.Event(enable_timing=True); f.record()
    loss.backward()
    e = torch.cuda.Event(enable_timing=True); e.record()
    eng.optimizer_step(lr=1e-6)
    return s0, f, e
for _ in range(3): step()
import gc; gc.collect(); gc.freeze()
for i in range(4):
    pairs.clear()
    s0, f, e = step()
    torch.cuda.synchronize()
    lags = [a.elapsed_time(b) for a, b in pairs]
    print(f"step {i}: forward {s0.elapsed_time(f):6.1f} ms  backward {f.elapsed_time(e):6.1f} ms  joins {len(pairs)}  side-stream lag at each join (ms): " + " ".join(f"{x:.2f}" for x in lags[-6:]))
