"""Race screen of the ring GEMMs with POISONED local memory and ALTERNATING operands.  A screen that runs one operand set over and over cannot see
a kernel that reads a staging buffer (or a register) before its data arrived: the bytes it finds are the same ones, left by the launch before.
Here tools/poison_lds.hip fills every CU's LDS with NaN patterns before every launch, consecutive launches use different operand sets, a second
stream loads the chip, and every result is compared on the device with that set's first result (with a tolerance where fp32 atomics reorder).
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/poison_lds.hip -o tools/bin/libpoison_lds.so
    python tools/race_stress.py [iters]
What it cannot see either: registers.  A kernel that computes on a vector register before its load returned gets the register's previous contents,
which a poisoned LDS does not reach (round 6's rare garbage weight gradient; tools/check_async_reads.py screens the assembly for that, and
tools/nan_hunt.py finds it in the step itself).
"""
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from neurosis_amd import ops  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
poison = C.CDLL(os.path.join(HERE, "bin", "libpoison_lds.so"))
poison.poison_lds.argtypes = [C.c_void_p, C.c_void_p]
sink = torch.zeros(1024, device="cuda", dtype=torch.int32)
side = torch.cuda.Stream()
NS = 3


def rb(*shape, scale=0.5):
    return (torch.randn(*shape, device="cuda") * scale).to(torch.bfloat16)


hog_a, hog_b = rb(8192, 2048), rb(2048, 2048)


def screen(name, run, tol_rel):
    refs = [run(i).clone() for i in range(NS)]
    tols = [tol_rel * float(r.float().abs().max()) for r in refs]
    bad = torch.zeros((), device="cuda", dtype=torch.int64)
    for it in range(ITERS):
        if it % 4 == 1:
            with torch.cuda.stream(side):
                os.environ["NK_GEMM_G2"] = "0"
                ops.gemm_nt(hog_a, hog_b)
                os.environ.pop("NK_GEMM_G2")
        assert poison.poison_lds(sink.data_ptr(), ops._stream()) == 0
        s = it % NS
        d = (run(s).float() - refs[s].float()).abs().max()
        bad += ~(d <= tols[s])          # (NaN counts as beyond)
    torch.cuda.synchronize()
    print(f"{name:64s} launches {ITERS}  beyond tolerance {int(bad)}", flush=True)


torch.manual_seed(1)
# the exact-round weight gradient (csrc/gemm_w160.h) as the step launches it: K split by plan, bias gradient from the same launch, zeroed destination
for (M, N, K) in [(16384, 640, 2560), (4096, 1280, 5120), (16384, 640, 640), (4096, 1280, 1280), (4096, 10240, 1280)]:
    dys, xs = [rb(M, N) for _ in range(NS)], [rb(M, K) for _ in range(NS)]
    dw, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")

    def run_w(i):
        dw.zero_()
        db.zero_()
        ops.gemm_tn_f32(dys[i], xs[i], dw, 2, dbias=db)
        return torch.cat([dw.flatten(), db])

    screen(f"w160 wgrad + bias (K split by plan) {M} x {N} x {K}", run_w, 1e-3)

# the two-group producer-wave kernel (csrc/gemm_g2.h): forward and input gradient, bit for bit
for (M, N, K) in [(4096, 1280, 1280), (4096, 1280, 5120), (16384, 640, 2560), (4096, 3840, 1280)]:
    xs, dys, w = [rb(M, K) for _ in range(NS)], [rb(M, N) for _ in range(NS)], rb(N, K, scale=K ** -0.5)
    screen(f"g2p fwd   {M} x {N} x {K}", lambda i: ops.gemm_nt(xs[i], w), 0.0)
    screen(f"g2p dgrad {M} x {N} x {K}", lambda i: ops.gemm_nn(dys[i], w), 0.0)

# ---- the other staged kernels (round 6, late): halo-tile convolutions (the 128-column tile with its producer waves, the 160-column tile), their
# weight gradients, the 256 x 256 two-group kernel with the GEGLU epilogue, the GEGLU-fused input gradient, attention of head dim 64 and 512 ----
import ctypes as C  # noqa: E402

from neurosis_amd.ops import Img  # noqa: E402


def conv_weight(co, ci):
    return torch.nn.Parameter((torch.randn(co, 3, 3, ci, device="cuda") * 0.02).permute(0, 3, 1, 2))


for (N, H, W, Ci, Co, stats) in [(2, 256, 256, 128, 128, 32), (2, 128, 128, 256, 256, 32), (4, 64, 64, 512, 512, None), (4, 64, 64, 640, 640, 32), (4, 32, 32, 1280, 1280, None)]:
    xs = [Img(rb(N * H * W, Ci), N, H, W) for _ in range(NS)]
    wt = conv_weight(Co, Ci)
    bias = torch.randn(Co, device="cuda")

    def run_c(i):
        y = ops.conv2d_fwd(xs[i], wt, bias, stride=1, padding=1, stats_groups=stats)[0]
        return y.t if y.sums is None else torch.cat([y.t.float().flatten(), y.sums.flatten()])

    screen(f"halo conv fwd{' + stats' if stats else ''} {N} x {H}x{W} {Ci} -> {Co}", run_c, 0.0)
    dys = [rb(N * H * W, Co) for _ in range(NS)]
    dw = torch.zeros(Co * 9 * Ci, device="cuda")
    d = ops._conv_desc(N, H, W, Ci, Co, 3, 3, 1, 1, 1, H, W, False)

    def run_cw(i):
        ops.call("nk_conv2d_wgrad", C.byref(d), dys[i].data_ptr(), xs[i].t.data_ptr(), dw.data_ptr(), 0, ops._stream())
        return dw

    screen(f"halo conv wgrad {N} x {H}x{W} {Ci} -> {Co}", run_cw, 1e-3)

for (M, I, K) in [(4096, 5120, 1280), (16384, 2560, 640)]:
    xs = [rb(M, K) for _ in range(NS)]
    w = torch.nn.Parameter(rb(2 * I, K, scale=K ** -0.5).float())
    b = torch.nn.Parameter(torch.randn(2 * I, device="cuda") * 0.1)

    def run_g(i):
        s_, h_, _ = ops.linear_geglu_fwd(xs[i], w, b, save_derivative=True)
        return torch.cat([s_.flatten(), h_.flatten()])

    screen(f"GEGLU-fused projection (256 x 256 two-group) {M} x {2 * I} x {K}", run_g, 0.0)
    ss, dhs = [rb(M, 2 * I) for _ in range(NS)], [rb(M, K) for _ in range(NS)]
    w2 = rb(K, I, scale=I ** -0.5)          # the output projection [K][I]: dh = dy w2, du through the saved-derivative GEGLU backward
    du = torch.empty(M, 2 * I, device="cuda", dtype=torch.bfloat16)

    def run_gd(i):
        ops.call("nk_linear_dgrad_geglu_s", dhs[i].data_ptr(), w2.data_ptr(), ss[i].data_ptr(), du.data_ptr(), M, K, I, K, I, 2 * I, 2 * I, ops._stream())
        return du

    screen(f"GEGLU-fused input gradient {M} x {I} x {K}", run_gd, 0.0)

for (B, Hh, L, D) in [(4, 20, 1024, 64), (2, 10, 4096, 64), (2, 1, 1024, 512)]:
    qs, ks, vs, dos = ([rb(B * L, Hh * D) for _ in range(NS)] for _ in range(4))

    def run_a(i):
        o, bwd = ops.attention_fwd(qs[i], ks[i], vs[i], B, Hh, D)
        dq, dk, dv = bwd(dos[i])
        return torch.cat([o.flatten(), dq.flatten(), dk.flatten(), dv.flatten()])

    screen(f"attention fwd + bwd B {B} H {Hh} L {L} D {D}", run_a, 1e-3)
