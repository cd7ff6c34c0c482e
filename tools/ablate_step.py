"""Where does the step's wall time go?  Skips one family of kernel launches at a time (the entry point returns at once, its
outputs stay uninitialised) and times the real two-stream training step, interleaved rounds in one process.  What a
family "costs" here is what the step would gain if it were free -- overlap with the other stream included -- which is the
bound on what optimising it can buy.  (Outputs are garbage while something is skipped; only the clock is read.)"""
import os; os.environ.setdefault("NK_GRAPH", "0")   # this tool watches / flips the Python-side launches: keep the eager chain
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from neurosis_amd import lib, ops
import neurosis_amd.nn as nkn
import neurosis_amd.modules.diffusion.loss as nkl

FAMILIES = {
    "base": (),
    "wgrad": ("nk_linear_wgrad", "nk_conv2d_wgrad", "nk_linear_wgrad_batched"),
    "colsum": ("nk_colsum",),
    "wgrad+colsum": ("nk_linear_wgrad", "nk_conv2d_wgrad", "nk_colsum"),
    "attn_bwd": ("nk_attention_bwd",),
    "attn_fwd": ("nk_attention_fwd",),
    "ln_bwd": ("nk_layernorm_bwd_dx", "nk_layernorm_bwd_params", "nk_layernorm_bwd"),
    "ln_bwd_params": ("nk_layernorm_bwd_params",),
    "gn_bwd": ("nk_groupnorm_bwd",),
    "gn_fwd": ("nk_groupnorm_fwd",),
    "ln_fwd": ("nk_layernorm_fwd",),
    "geglu": ("nk_geglu_fwd", "nk_geglu_bwd"),
    "dgrad": ("nk_linear_dgrad", "nk_conv2d_dgrad"),
    "linear_fwd": ("nk_linear_fwd",),
    "conv_fwd": ("nk_conv2d_fwd",),
    "optimizer": ("nk_adafactor_chunk", "nk_adafactor_init"),
    "ALL(host floor)": ("*",),
}


def main():
    only = sys.argv[1:]
    dev = torch.device("cuda", 0)
    eng = bench.build_engine(dev, conditioner=bench.build_conditioner(dev))
    eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
    gen = torch.Generator(device=dev).manual_seed(42); gen_cpu = torch.Generator().manual_seed(42)

    def step():
        batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen, False)
        sig = bench.draw_sigmas(4, gen_cpu, dev)
        loss = eng.training_step(batch, 0, sigmas=sig); loss.backward(); eng.optimizer_step(lr=1e-6)

    orig = lib.call
    skip = set()

    def call(name, *args):
        if name in skip or "*" in skip:
            return None
        return orig(name, *args)

    for m in (lib, ops, nkn, nkl):
        m.call = call
    import neurosis_amd.optim as nko
    nko.call = call
    for _ in range(3): step()
    import gc; gc.collect(); gc.freeze()
    fams = {k: v for k, v in FAMILIES.items() if not only or k in only or k == "base"}
    res = {k: [] for k in fams}
    for rnd in range(3):
        for name, names in fams.items():
            skip.clear(); skip.update(names)
            step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): step()
            torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 3 * 1e3)
    base = min(res["base"])
    for k, v in res.items():
        print(f"{k:16s} " + " ".join(f"{x:7.1f}" for x in v) + f"   min {min(v):7.1f} ms   saves {base - min(v):6.1f} ms", flush=True)


main()
