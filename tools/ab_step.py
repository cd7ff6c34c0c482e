"""In-process A/B of training-step variants (same device, interleaved rounds): prints ms/step per variant.
usage: ab_step.py [variant ...]   a variant is a built-in name ("no g2", "no batch", "no halo conv", "hp main stream", ...) or ENV=VALUE[,ENV=VALUE...];
"base" always runs.  With no arguments every built-in runs."""
import os; os.environ.setdefault("NK_GRAPH", "0")   # this tool watches / flips the Python-side launches: keep the eager chain
import os, sys, time, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from neurosis_amd import ops

def main():
    dev = torch.device("cuda", 0)
    eng = bench.build_engine(dev)
    eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
    gen = torch.Generator(device=dev).manual_seed(42); gen_cpu = torch.Generator().manual_seed(42)
    def step():
        batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen)
        sig = bench.draw_sigmas(4, gen, dev)
        loss = eng.training_step(batch, 0, sigmas=sig); loss.backward(); eng.optimizer_step(lr=1e-6)
    variants = {}
    est = eng.store.state
    side = est.wgrad_stream
    variants["base"] = lambda: None
    variants["no g2"] = lambda: os.environ.__setitem__("NK_GEMM_G2", "0")
    variants["no batch"] = lambda: setattr(est, "batch_wgrads", False)
    variants["ln params on side stream"] = lambda: setattr(est, "norm_params_on_side_stream", True)
    variants["no halo conv"] = lambda: os.environ.__setitem__("NK_CONV_HALO", "0")
    variants["no stream-K"] = lambda: os.environ.__setitem__("NK_GEMM_SK", "0")
    variants["streamed optimizer"] = lambda: setattr(eng, "stream_optimizer", True)
    variants["hp main stream"] = lambda: None
    asked = [a for a in sys.argv[1:] if "=" in a]
    for name in asked:              # extra variants from the command line: ENV=VALUE[,ENV=VALUE...]
        kv = [a.split("=", 1) for a in name.split(",")]
        variants[name] = lambda kv=kv: [os.environ.__setitem__(k, v) for k, v in kv]
    extra_env = sorted({a.split("=", 1)[0] for name in asked for a in name.split(",")})
    if sys.argv[1:]:
        variants = {k: v for k, v in variants.items() if k == "base" or k in sys.argv[1:]}
    def restore():
        est.wgrad_stream = side; est.batch_wgrads = True; eng.stream_optimizer = False; est.norm_params_on_side_stream = None
        for k in ("NK_GEMM_G2", "NK_CONV_HALO", "NK_GEMM_SK"): os.environ.pop(k, None)
        for k in extra_env: os.environ.pop(k, None)
    for _ in range(3): step()
    import gc; gc.collect(); gc.freeze()      # (the cyclic GC's full collections otherwise show up as 200+ ms steps: DESIGN section 7)
    hp = torch.cuda.Stream(priority=-1)       # "main work on a high-priority stream" variants: name starts with "hp"
    res = {k: [] for k in variants}
    for rnd in range(5):
        for name, setup in variants.items():
            restore(); setup()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if name.startswith("hp"):
                hp.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(hp):
                    for _ in range(3): step()
                torch.cuda.current_stream().wait_stream(hp)
            else:
                for _ in range(3): step()
            torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 3 * 1e3)
    restore()
    for k, v in res.items(): print(f"{k:18s} " + " ".join(f"{x:7.1f}" for x in v) + f"   min {min(v):.1f}  median {sorted(v)[len(v) // 2]:.1f} ms")
main()
