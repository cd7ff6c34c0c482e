"""How long does the host take to ENQUEUE one training step (no device sync inside)?"""
import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda", 0)
eng = bench.build_engine(dev)
gen = torch.Generator(device=dev).manual_seed(42); gen_cpu = torch.Generator().manual_seed(42)
def step():
    batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen)
    sig = bench.draw_sigmas(4, gen_cpu, dev)
    loss = eng.training_step(batch, 0, sigmas=sig); loss.backward(); eng.optimizer_step(lr=1e-6)
eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
for _ in range(3): step()      # warm-up, graph capture, first replay
torch.cuda.synchronize()
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms   total {1e3*(t2-t0):.1f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
