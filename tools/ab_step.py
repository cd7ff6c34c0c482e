"""In-process A/B of training-step variants (same device, interleaved rounds): prints ms/step per variant."""
import os, sys, time, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from neurosis_amd import ops

def main():
    dev = torch.device("cuda", 0)
    eng = bench.build_engine(dev)
    gen = torch.Generator(device=dev).manual_seed(42); gen_cpu = torch.Generator().manual_seed(42)
    def step():
        batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen)
        sig = bench.draw_sigmas(4, gen_cpu, dev)
        loss = eng.training_step(batch, 0, sigmas=sig); loss.backward(); eng.optimizer_step(lr=1e-6)
    variants = {}
    est = eng.store.state
    side = est.wgrad_stream
    variants["base"] = lambda: None
    def setenv(lo, hi):
        os.environ["NK_SPLIT_LO"] = str(lo); os.environ["NK_SPLIT_HI"] = str(hi)
    variants["ln_side"] = lambda: setattr(est, "norm_params_on_side_stream", True)
    variants["no_ring"] = lambda: os.environ.__setitem__("NK_GEMM_RING", "0")
    variants["no_sk"] = lambda: os.environ.__setitem__("NK_GEMM_SK", "0")
    def restore(): est.wgrad_stream = side; setenv(96, 192); os.environ["NK_GEMM_NW"] = "8"; os.environ["NK_GEMM_SK"] = "4"; os.environ["NK_SK_GRID"] = "512"; est.norm_params_on_side_stream = False; os.environ["NK_GEMM_RING"] = "1"
    for _ in range(2): step()
    res = {k: [] for k in variants}
    for rnd in range(3):
        for name, setup in variants.items():
            restore(); setup()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): step()
            torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 3 * 1e3)
    restore()
    for k, v in res.items(): print(f"{k:18s} " + " ".join(f"{x:7.1f}" for x in v) + f"   min {min(v):.1f} ms")
main()
