import sys, torch
sys.path.insert(0, ".")
from tests.test_modules_gpu import _build_unet, _loss
from tests.util import rel_err, cosine

fx, net, st = _build_unet("unet_sdxl_tiny", True)
def grads(n, s):
    s.zero_grad(); _loss(n, fx).mean().backward(); torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in n.named_parameters()}
a = grads(net, st); b = grads(net, st)
gmax = max(float(g.abs().max()) for g in a.values())
def report(x, y, tag):
    rows = []
    for k in x:
        scale = max(float(x[k].abs().max()), 1e-2 * gmax)
        rows.append((float((y[k]-x[k]).abs().max())/scale, cosine(y[k], x[k]), float(x[k].norm()), k))
    rows.sort(reverse=True)
    print(tag, "gmax", gmax)
    for r in rows[:8]: print("   ", r)
report(a, b, "run-to-run")
fx, net2, st2 = _build_unet("unet_sdxl_tiny", True, use_checkpoint=True)
c = grads(net2, st2)
report(a, c, "checkpoint")
