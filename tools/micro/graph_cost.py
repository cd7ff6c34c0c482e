"""What does a hipGraph replay cost per kernel node on this ROCm?  N dependent small kernels (a) launched eagerly through the
C-ABI, (b) replayed from a captured graph; host time to enqueue and GPU time per launch."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from neurosis_amd import ops
dev = torch.device("cuda", 0)
a = torch.zeros(4096, 320, device=dev, dtype=torch.bfloat16); b = torch.ones_like(a)
N = 2000
def body():
    x = a
    for _ in range(N):
        x = ops.add(x, b)
    return x
body(); torch.cuda.synchronize()
for tag in ("eager", "graph"):
    if tag == "graph":
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            body()
        torch.cuda.current_stream().wait_stream(s)
        t0 = time.perf_counter()
        with torch.cuda.graph(g):
            out = body()
        print(f"capture + instantiate: {(time.perf_counter() - t0) * 1e3:.1f} ms")
        fn = g.replay
    else:
        fn = body
    fn(); torch.cuda.synchronize()
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(); fn(); e1.record(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{tag:6s} host enqueue {1e6 * (t1 - t0) / N:6.2f} us/node   gpu {1e3 * e0.elapsed_time(e1) / N:6.2f} us/node   wall {1e6 * (t2 - t0) / N:6.2f} us/node", flush=True)
