"""Eager vs hipGraph-replayed training steps of the full SDXL engine on the same seeded inputs: per-step loss, gradient norm,
and the parameters whose gradient differs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda", 0)
ref = {}
for mode in ["0", "1"]:
    os.environ["NK_GRAPH"] = mode
    eng = bench.build_engine(dev)
    eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
    gen = torch.Generator(device=dev).manual_seed(42); gen_cpu = torch.Generator().manual_seed(42)
    names = [n for n, p in eng.model.diffusion_model.named_parameters() if p.requires_grad]
    for i in range(4):
        batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen)
        sig = bench.draw_sigmas(4, gen_cpu, dev)
        loss = eng.training_step(batch, 0, sigmas=sig); loss.backward()
        torch.cuda.synchronize()
        g = eng.store.grad
        norms = torch.stack([p.grad.float().norm() for p in eng.store.params]).cpu()
        print(mode, i, float(loss.detach()), float(g.norm()), flush=True)
        if mode == "0":
            ref[i] = norms
        else:
            rel = (norms - ref[i]).abs() / (ref[i] + 1e-12)
            bad = [(names[j] if j < len(names) else j, float(ref[i][j]), float(norms[j])) for j in torch.nonzero(rel > 1e-2).flatten().tolist()]
            print("   parameters whose gradient norm differs by > 1 %:", len(bad), bad[:30], flush=True)
        eng.optimizer_step(lr=1e-6)
    del eng; torch.cuda.empty_cache()
