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
