"""Eager vs hipGraph-replayed training steps WITH gradient accumulation (two signatures: overwrite / add): per micro-batch loss
and gradient norm."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda", 0)
ACC = 3
for mode in ["0", "unet"]:
    os.environ["NK_GRAPH"] = mode
    eng = bench.build_engine(dev)
    eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
    gen = torch.Generator(device=dev).manual_seed(42); gen_cpu = torch.Generator().manual_seed(42)
    for i in range(4):
        for mb in range(ACC):
            batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen)
            sig = bench.draw_sigmas(4, gen_cpu, dev)
            eng.accumulate(mb, None, last=mb == ACC - 1)
            loss = eng.training_step(batch, 0, sigmas=sig); (loss / ACC).backward()
            torch.cuda.synchronize()
            g = eng.store.grad
            print(mode, i, mb, float(loss.detach()), float(g.norm()), int((~torch.isfinite(g)).sum()), flush=True)
        eng.optimizer_step(lr=1e-6)
    del eng; torch.cuda.empty_cache()
