"""(experiment) order of hipGraph launches in the replayed backward: M_k, W_k, M_k+1, ... (as shipped) against M_k, M_k+1, W_k, ... (the next
main segment is submitted BEFORE the side-stream graph of the segment just finished).  Alternating rounds in one process; also prints the
host time spent inside each replay() call of one steady step."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from neurosis_amd import ops
from neurosis_amd.graphs import ChainGraphs

host_log = []


def replay_shipped(self, pair):
    st = ops.state_of(self.owner)
    side = st.wgrad_stream
    main = torch.cuda.current_stream()
    hook = self.hook()
    for g_m, g_w, module in pair.segments:
        t0 = time.perf_counter(); g_m.replay(); t1 = time.perf_counter()
        if g_w is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                g_w.replay()
        t2 = time.perf_counter()
        host_log.append((t1 - t0, t2 - t1))
        if hook is not None and module is not None:
            hook(module)
        self.replays += 1
    if side is not None:
        main.wait_stream(side)


def replay_main_first(self, pair):
    st = ops.state_of(self.owner)
    side = st.wgrad_stream
    main = torch.cuda.current_stream()
    hook = self.hook()
    pending = None

    def flush(p):
        g_w, ev, module = p
        if g_w is not None:
            side.wait_event(ev)
            with torch.cuda.stream(side):
                g_w.replay()
        if hook is not None and module is not None:
            hook(module)

    for g_m, g_w, module in pair.segments:
        g_m.replay()
        ev = torch.cuda.Event()
        ev.record(main)
        if pending is not None:
            flush(pending)
        pending = (g_w, ev, module)
        self.replays += 1
    flush(pending)
    if side is not None:
        main.wait_stream(side)


def main():
    dev = torch.device("cuda", 0)
    eng = bench.build_engine(dev)
    eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
    gen = torch.Generator(device=dev).manual_seed(42); gen_cpu = torch.Generator().manual_seed(42)

    phases = []

    def step():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen)
        sig = bench.draw_sigmas(4, gen, dev)
        ev[1].record()
        loss = eng.training_step(batch, 0, sigmas=sig)
        ev[2].record()
        loss.backward()
        ev[3].record()
        eng.optimizer_step(lr=1e-6)
        ev[4].record()
        phases.append(ev)

    for _ in range(4): step()
    import gc; gc.collect(); gc.freeze()
    variants = {"shipped (M W M W)": replay_shipped, "main first (M M W)": replay_main_first}
    res = {k: [] for k in variants}
    for rnd in range(4):
        for name, fn in variants.items():
            ChainGraphs._replay_backward = fn
            step(); torch.cuda.synchronize()
            host_log.clear()
            t0 = time.perf_counter()
            for _ in range(4): step()
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize(); res[name].append(((time.perf_counter() - t0) / 4 * 1e3, t_host / 4 * 1e3))
            if rnd == 0 and host_log:
                n = len(host_log) // 4
                print("host ms inside replay() per segment of one step (main graph, side graph):", [(round(a * 1e3, 2), round(b * 1e3, 2)) for a, b in host_log[-n:]], flush=True)
    torch.cuda.synchronize()
    last = phases[-4:]
    names = ["batch synthesis", "training_step (VAE + conditioner + UNet forward + loss)", "backward", "optimizer_step (main-stream part)"]
    for i, n in enumerate(names):
        print(f"  {n:60s} {sum(e[i].elapsed_time(e[i + 1]) for e in last) / len(last):8.2f} ms")
    for k, v in res.items():
        print(f"{k:22s} step ms " + " ".join(f"{a:7.1f}" for a, _ in v) + "   host-enqueue ms " + " ".join(f"{b:7.1f}" for _, b in v))
main()
