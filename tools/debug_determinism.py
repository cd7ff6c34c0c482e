import sys, torch
sys.path.insert(0, ".")
from tests.test_modules_gpu import _build_unet, _loss
from tests.util import rel_err, cosine

fx, net, st = _build_unet("unet_sdxl_tiny", True)
def grads(n, s):
    s.zero_grad(); _loss(n, fx).mean().backward(); torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in n.named_parameters()}
a = grads(net, st); b = grads(net, st)
worst = sorted(((rel_err(b[k], a[k]), k) for k in a), reverse=True)[:6]
print("run-to-run:", worst)
fx, net2, st2 = _build_unet("unet_sdxl_tiny", True, use_checkpoint=True)
c = grads(net2, st2)
worst = sorted(((rel_err(c[k], a[k]), k) for k in a), reverse=True)[:6]
print("checkpoint vs plain:", worst)
print("cos time_embed.0.weight", cosine(c["time_embed.0.weight"], a["time_embed.0.weight"]))
