"""Under hipGraph replay: how long after the main chain's last segment does the weight-gradient stream finish (the exposed tail of
the segmented backward), and how long is a whole backward?"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from neurosis_amd import graphs
dev = torch.device("cuda", 0)
eng = bench.build_engine(dev)
eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
gen = torch.Generator(device=dev).manual_seed(42); gen_cpu = torch.Generator().manual_seed(42)
marks = []
orig = graphs.ChainGraphs._replay_backward
def timed(self, pair):
    from neurosis_amd import ops
    side = ops.state_of(self.owner).wgrad_stream
    main = torch.cuda.current_stream()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(main)
    hook = self.hook()
    aux = ops.state_of(self.owner).aux_stream
    for g_m, g_w, g_a, module in pair.segments:
        g_m.replay()
        if g_w is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                g_w.replay()
        if g_a is not None:
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                g_a.replay()
        if hook is not None and module is not None:
            hook(module)
    e1.record(main)
    e2.record(side)
    main.wait_stream(side)
    if aux is not None:
        main.wait_stream(aux)
    marks.append((e0, e1, e2))
graphs.ChainGraphs._replay_backward = timed
def step():
    batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen)
    sig = bench.draw_sigmas(4, gen_cpu, dev)
    loss = eng.training_step(batch, 0, sigmas=sig); loss.backward(); eng.optimizer_step(lr=1e-6)
for _ in range(8): step()
torch.cuda.synchronize()
for e0, e1, e2 in marks[-4:]:
    print(f"backward main chain {e0.elapsed_time(e1):7.2f} ms   side stream ends {e1.elapsed_time(e2):+6.2f} ms after it   segments {len(eng.model.diffusion_model._nk_graphs.pairs and next(iter(eng.model.diffusion_model._nk_graphs.pairs.values())).segments)}")
