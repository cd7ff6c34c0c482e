"""Is a short-K GEMM slower inside the step than in a loop over the same buffers because its operands are cold?  4096 x 1280 x 1280 forward (and a
second shape), timed (a) over ONE set of buffers (everything L2 / Infinity-Cache resident after the first pass), (b) with the WEIGHTS rotating over
more distinct buffers than the Infinity Cache holds, (c) with activations and outputs rotating too (what a training step does)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
def rb(*shape): return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)
def timed(fns, iters):
    for f in fns[:8]: f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fns[i % len(fns)]()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for M, N, K in [(4096, 1280, 1280), (4096, 3840, 1280), (4096, 1280, 5120), (16384, 640, 640)]:
    R = 96
    xs, ws = [rb(M, K) for _ in range(R)], [rb(N, K) for _ in range(R)]
    dys = [rb(M, N) for _ in range(R)]
    fl = 2.0 * M * N * K
    for kind, mk in (("fwd", lambda i, j: (lambda: ops.gemm_nt(xs[i], ws[j]))), ("dgrad", lambda i, j: (lambda: ops.gemm_nn(dys[i], ws[j])))):
        for krot in ("0", "1"):      # NK_GEMM_KROT: every XCD starts its k loop at 0 / an eighth of K apart (gemm_g2.h OpG2::rotate)
            os.environ["NK_GEMM_KROT"] = krot
            hot = min(timed([mk(0, 0)], 192) for _ in range(2))
            coldw = min(timed([mk(0, j) for j in range(R)], 192) for _ in range(2))
            cold = min(timed([mk(j, j) for j in range(R)], 192) for _ in range(2))
            print(f"{kind:5s} {M} x {N} x {K} krot={krot}:  same buffers {hot:6.1f} us ({fl/hot/1e6:5.0f} TF/s) | weights rotating {coldw:6.1f} ({fl/coldw/1e6:5.0f}) | everything rotating {cold:6.1f} ({fl/cold/1e6:5.0f})", flush=True)

# Linear weight gradients dW[N, K] = dy[M, N]^T x[M, K] (128 x 128 double-buffer kernel): both operands are activations, every tile walks the tokens
for M, N, K in [(4096, 10240, 1280), (4096, 1280, 5120), (4096, 3840, 1280)]:
    R = 24
    xs, dys = [rb(M, K) for _ in range(R)], [rb(M, N) for _ in range(R)]
    dw = torch.zeros(N, K, device="cuda")
    fl = 2.0 * M * N * K
    for krot in ("0", "1"):
        os.environ["NK_GEMM_KROT"] = krot
        hot = min(timed([lambda: ops.gemm_tn_f32(dys[0], xs[0], dw, False)], 96) for _ in range(2))
        cold = min(timed([(lambda j=j: ops.gemm_tn_f32(dys[j], xs[j], dw, False)) for j in range(R)], 96) for _ in range(2))
        print(f"wgrad {M} x {N} x {K} krot={krot}:  same buffers {hot:6.1f} us ({fl/hot/1e6:5.0f} TF/s) | activations rotating {cold:6.1f} ({fl/cold/1e6:5.0f})", flush=True)

# 3 x 3 convolutions of the UNet's 32^2 / 64^2 levels (halo-tile kernel, conv_halo.h): weights of 7-59 MB, rotating over 12 buffers
from neurosis_amd.ops import Img
for (N, H, W, Ci, Co) in [(4, 32, 32, 1280, 1280), (4, 32, 32, 2560, 1280), (4, 64, 64, 640, 640), (4, 64, 64, 1280, 640)]:
    R = 12
    x = Img(rb(N * H * W, Ci), N, H, W)
    ws = [torch.nn.Parameter(ops.conv_weight_param(Co, Ci, 3, 3).data.normal_(0, (9 * Ci) ** -0.5).cuda(), requires_grad=False) for _ in range(R)]
    bias = torch.randn(Co, device="cuda")
    fl = 2.0 * N * H * W * Ci * Co * 9
    for w in ws: ops.conv2d_fwd(x, w, bias, need_dx=False)        # (bf16 shadows made here, outside the timing)
    for krot in ("0", "1"):
        os.environ["NK_GEMM_KROT"] = krot
        hot = min(timed([lambda: ops.conv2d_fwd(x, ws[0], bias, need_dx=False)], 48) for _ in range(2))
        cold = min(timed([(lambda w=w: ops.conv2d_fwd(x, w, bias, need_dx=False)) for w in ws], 48) for _ in range(2))
        print(f"conv  {N} x {H}x{W} {Ci} -> {Co} krot={krot}:  same weights {hot:6.1f} us ({fl/hot/1e6:5.0f} TF/s) | weights rotating {cold:6.1f} ({fl/cold/1e6:5.0f})", flush=True)
