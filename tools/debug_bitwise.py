import sys, torch
sys.path.insert(0, ".")
from neurosis_amd import ops
from tests.test_modules_gpu import _build_unet, _loss
torch.manual_seed(0)
def rb(*s): return (torch.randn(*s, device="cuda")).to(torch.bfloat16)
def same(a, b): return bool(torch.equal(a, b))
def twice(fn, tag):
    a = fn(); b = fn(); torch.cuda.synchronize()
    if not isinstance(a, (tuple, list)): a, b = [a], [b]
    print(f"{tag:28s}", [same(x, y) for x, y in zip(a, b)], [float((x.float()-y.float()).abs().max()) for x, y in zip(a, b)])
x, w, dy = rb(1000, 320), rb(640, 320), rb(1000, 640)
twice(lambda: ops.gemm_nt(x, w), "gemm_nt")
twice(lambda: ops.gemm_nn(dy, w), "gemm_nn")
def wg():
    d = torch.zeros(640, 320, device="cuda"); ops.gemm_tn_f32(dy, x, d, 0); return d
twice(wg, "gemm_tn (maybe splitk)")
q, k, v, do = rb(2*256, 128), rb(2*77, 128), rb(2*77, 128), rb(2*256, 128)
def att():
    o, b = ops.attention_fwd(q, k, v, 2, 2, 64); return (o,) + tuple(b(do))
twice(att, "attention cross")
q2 = rb(2*256, 128)
def att2():
    o, b = ops.attention_fwd(q2, q2, q2, 2, 2, 64); return (o,) + tuple(b(do))
twice(att2, "attention self")
img = ops.Img(rb(2*16*16, 64), 2, 16, 16)
gw, gb = torch.nn.Parameter(torch.randn(64, device="cuda")), torch.nn.Parameter(torch.randn(64, device="cuda"))
dyi = rb(2*16*16, 64)
def gn():
    gw.grad = None; gb.grad = None
    o, b = ops.groupnorm_fwd(img, gw, gb, 32, 1e-5, True); dx = b(dyi); return o.t, dx, gw.grad.clone(), gb.grad.clone()
twice(gn, "groupnorm")
xl = rb(512, 128); lw, lb = torch.nn.Parameter(torch.randn(128, device="cuda")), torch.nn.Parameter(torch.randn(128, device="cuda"))
def ln():
    lw.grad = None; lb.grad = None
    o, b = ops.layernorm_fwd(xl, lw, lb); dx = b(rb(512,128)*0+1); return o, dx, lw.grad.clone(), lb.grad.clone()
twice(ln, "layernorm")
cw = torch.nn.Parameter((torch.randn(64, 3, 3, 64, device="cuda")*0.05).permute(0,3,1,2)); cb = torch.nn.Parameter(torch.randn(64, device="cuda"))
def cv():
    cw.grad=None; cb.grad=None
    o, b = ops.conv2d_fwd(img, cw, cb, 1, 1); dx, _ = b(dyi); return o.t, dx.t, cw.grad.clone(), cb.grad.clone()
twice(cv, "conv")
fx, net, st = _build_unet("unet_sdxl_tiny", True)
with torch.no_grad():
    l1 = _loss(net, fx).clone(); l2 = _loss(net, fx).clone()
print("loss fwd bitwise", same(l1, l2), l1.tolist(), l2.tolist())
