"""Per-shape time of the tile engine inside one real training step (serialized launches, HIP events)."""
import os; os.environ.setdefault("NK_GRAPH", "0")   # this tool watches / flips the Python-side launches: keep the eager chain
import sys, torch, collections
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from neurosis_amd import ops
dev = torch.device("cuda", 0)
eng = bench.build_engine(dev)
gen = torch.Generator(device=dev).manual_seed(42); gen_cpu = torch.Generator().manual_seed(42)
def step():
    batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen)
    sig = bench.draw_sigmas(4, gen, dev)
    loss = eng.training_step(batch, 0, sigmas=sig); loss.backward(); eng.optimizer_step(lr=1e-6)
for _ in range(2): step()
timer = bench.GemmTimer()
orig_flops = timer.flops
shapes = []
def flops(name, args):
    if name == "nk_linear_fwd_batched":      # key carries the count: "x3 per launch"
        key = (f"{name[:-8]}[x{args[3]}]", args[4], args[5], args[6])
    elif name == "nk_linear_wgrad_batched":  # (ptrs, ptrs, ptrs, bias ptrs, count, M, N, K, ...)
        key = (f"{name[:-8]}[x{args[4]}]", args[5], args[6], args[7])
    elif name == "nk_linear_wgrad_bias":     # (dy, x, dw, dbias, M, N, K, ...)
        key = (name, args[4], args[5], args[6])
    elif name in ("nk_linear_fwd_geglu", "nk_linear_fwd_geglu_s"):
        key = (name, args[5], 2 * args[6], args[7])
    elif name in ("nk_linear_dgrad_geglu", "nk_linear_dgrad_geglu_s"):
        key = (name, args[4], args[6], args[5])
    elif name.startswith("nk_linear"):
        key = (name, args[5], args[6], args[7]) if name == "nk_linear_fwd" else ((name, args[4], args[5], args[6]) if name == "nk_linear_dgrad" else (name, args[3], args[4], args[5]))
    else:
        d = args[0]._obj
        key = (name, d.N * d.Ho * d.Wo, d.Cout, d.KH * d.KW * d.Cin)
    shapes.append(key)
    return orig_flops(name, args)
timer.flops = flops
timer.install()
side, eng.store.state.wgrad_stream = eng.store.state.wgrad_stream, None
step()
eng.store.state.wgrad_stream = side
timer.uninstall()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, (name, f, s, e) in zip(shapes, timer.records):
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += f; a[2] += s.elapsed_time(e)
tot = sum(a[2] for a in agg.values())
print(f"total {tot:.1f} ms over {len(timer.records)} launches")
import json
json.dump([[list(k), a] for k, a in agg.items()], open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'gemm_shapes.json'), 'w'))
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:80]:
    print(f"{key[0]:16s} M/N/K={key[1]:>8d} {key[2]:>6d} {key[3]:>6d}  x{a[0]:4d}  {a[2]:7.2f} ms  {a[1]/a[2]/1e9:7.0f} TF/s  ({a[2]/a[0]*1e3:7.1f} us each)")
