"""Hunt for a rare non-finite value in the metric's step: bench.py's own step (graph replay, both backward streams, the optimizer stream) run for
many steps; after every backward the loss and the flat gradient buffer are checked, and at the first non-finite one the parameters whose
gradients hold it are listed in module order (the LAST module in forward order with a bad gradient is where the backward went wrong).
    python tools/nan_hunt.py [steps] [sync|nosync] [mixed]
mixed: every step draws one of BASELINE config 4's aspect buckets (bench.MIXED_BUCKETS) instead of 1024 x 1024.
sync (default): the check is read every step (the host waits for the GPU once per step); nosync: flags stay on the device until the end.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
SYNC = (sys.argv[2] if len(sys.argv) > 2 else "sync") == "sync"
MIXED = len(sys.argv) > 3 and sys.argv[3] == "mixed"
gen_cpu = torch.Generator().manual_seed(7)
args = bench.parse_args([])
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
from neurosis_amd import lib  # noqa: E402

lib.load()
eng = bench.build_engine(device, (args.res, args.res), bench.build_conditioner(device))
unet = eng.model.diffusion_model
eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
gen = torch.Generator(device=device).manual_seed(42)
flags = torch.zeros(STEPS, 3, device=device, dtype=torch.int32)


def report(step):
    torch.cuda.synchronize()
    names = [(n, p) for n, p in unet.named_parameters() if p.grad is not None]
    bad = [(i, n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for i, (n, p) in enumerate(names) if not bool(torch.isfinite(p.grad).all())]
    print(f"step {step}: {len(bad)} of {len(names)} parameters have non-finite gradients", flush=True)
    for i, n, c, tot in bad[:12]:
        print(f"   first: #{i:4d} {n:80s} {c} / {tot}")
    for i, n, c, tot in bad[-12:]:
        print(f"   last:  #{i:4d} {n:80s} {c} / {tot}")
    wbad = [n for n, p in unet.named_parameters() if not bool(torch.isfinite(p).all())]
    print(f"   parameters (shadows) with non-finite values: {len(wbad)} {wbad[:5]}", flush=True)


for s in range(STEPS):
    hw = bench.MIXED_BUCKETS[int(torch.randint(len(bench.MIXED_BUCKETS), (1,), generator=gen_cpu))] if MIXED else (args.res, args.res)
    batch = bench.synthetic_batch(device, args.batch, hw, gen, False)
    sig = bench.draw_sigmas(args.batch, gen, device)
    eng.accumulate(0, None, last=True)
    loss = eng.training_step(batch, 0, sigmas=sig)
    loss.backward()
    flags[s, 0] = torch.isnan(loss.detach()).any()
    flags[s, 1] = ~torch.isfinite(eng.store.grad).all()
    if SYNC and bool(flags[s].any()):
        print(f"step {s}: loss non-finite {int(flags[s, 0])}  gradient buffer non-finite {int(flags[s, 1])}  loss {float(loss):.6f}", flush=True)
        report(s)
        break
    eng.optimizer_step(lr=1e-6, weight_decay=1e-2, grad_scale=1.0, dp=None)
    if SYNC:
        torch.cuda.synchronize()
        if not bool(torch.isfinite(eng.store.master).all()):
            print(f"step {s}: the optimizer left non-finite parameters (gradients were finite)", flush=True)
            for n, p in unet.named_parameters():
                if bool(torch.isfinite(p).all()):
                    continue
                g = p.grad.float()
                print(f"   {n} {tuple(p.shape)}: non-finite weights {int((~torch.isfinite(p)).sum())} / {p.numel()};  gradient max |g| {float(g.abs().max()):.4g}  "
                      f"mean |g| {float(g.abs().mean()):.4g}  zeros {int((g == 0).sum())}", flush=True)
                g2 = g.reshape(g.shape[0], -1)
                big = (g2.abs() > 100 * g2.abs().mean()).nonzero()
                print(f"   entries above 100 x mean: {big.shape[0]}; rows {big[:, 0].min().item() if big.numel() else -1} .. {big[:, 0].max().item() if big.numel() else -1}, "
                      f"columns {big[:, 1].min().item() if big.numel() else -1} .. {big[:, 1].max().item() if big.numel() else -1}", flush=True)
                print("   first few:", [(int(r), int(c), float(g2[r, c])) for r, c in big[:12].tolist()], flush=True)
                w2 = p.detach().float().reshape(g.shape[0], -1)
                wb = (~torch.isfinite(w2)).nonzero()
                print(f"   non-finite weights: rows {wb[:, 0].min().item()} .. {wb[:, 0].max().item()}, columns {wb[:, 1].min().item()} .. {wb[:, 1].max().item()}; "
                      f"distinct rows {wb[:, 0].unique().numel()}, distinct columns {wb[:, 1].unique().numel()}", flush=True)
                torch.save({"grad": g.cpu(), "name": n}, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "nan_grad.pt"))
            break
    if s % 50 == 49:
        print(f"   ... step {s + 1}", flush=True)
else:
    torch.cuda.synchronize()
    f = flags.cpu()
    bad = f.any(dim=1).nonzero().flatten().tolist()
    print(f"{STEPS} steps: non-finite at steps {bad[:10]} (loss, gradient) {[f[b].tolist() for b in bad[:4]]}" if bad else f"{STEPS} steps: all finite", flush=True)
