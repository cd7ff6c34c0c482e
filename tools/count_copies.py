"""Which Python call sites make device-to-device copies in one (eager) training step?  Patches Tensor.clone / contiguous /
copy_ / to and torch.cat, counts by caller (file:line) and bytes."""
import os; os.environ["NK_GRAPH"] = "0"
import sys, collections, traceback, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda", 0)
eng = bench.build_engine(dev, conditioner=bench.build_conditioner(dev))
eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
gen = torch.Generator(device=dev).manual_seed(42); gen_cpu = torch.Generator().manual_seed(42)
def step():
    batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen, False)
    sig = bench.draw_sigmas(4, gen_cpu, dev)
    loss = eng.training_step(batch, 0, sigmas=sig); loss.backward(); eng.optimizer_step(lr=1e-6)
for _ in range(2): step()
torch.cuda.synchronize()
count = collections.Counter(); nbytes = collections.Counter()
def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "neurosis_amd" in f.filename or "bench.py" in f.filename:
            return f"{os.path.relpath(f.filename)}:{f.lineno} {f.name}"
    return "?"
def wrap(obj, name, is_copy):
    orig = getattr(obj, name)
    def w(*a, **k):
        t = a[0] if a and torch.is_tensor(a[0]) else None
        r = orig(*a, **k)
        try:
            if is_copy(a, k, r):
                s = f"{name} @ {site()}"
                count[s] += 1
                nbytes[s] += (r.numel() * r.element_size()) if torch.is_tensor(r) else 0
        except Exception:
            pass
        return r
    setattr(obj, name, w)
wrap(torch.Tensor, "clone", lambda a, k, r: a[0].is_cuda)
wrap(torch.Tensor, "contiguous", lambda a, k, r: a[0].is_cuda and r.data_ptr() != a[0].data_ptr())
wrap(torch.Tensor, "copy_", lambda a, k, r: a[0].is_cuda)
wrap(torch.Tensor, "to", lambda a, k, r: torch.is_tensor(r) and r.is_cuda and r.data_ptr() != a[0].data_ptr())
wrap(torch.Tensor, "float", lambda a, k, r: r.is_cuda and r.data_ptr() != a[0].data_ptr())
wrap(torch, "cat", lambda a, k, r: r.is_cuda)
from neurosis_amd import ops as _ops, lib as _lib
import neurosis_amd.nn as _nkn
_orig_call = _ops.call
def _call(name, *a):
    if name == "nk_cast_f32_to_bf16":
        st = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in reversed(traceback.extract_stack()[:-1][-6:]))
        count["cast " + st] += 1
    return _orig_call(name, *a)
_ops.call = _call; _nkn.call = _call
step()
torch.cuda.synchronize()
for s, c in count.most_common(40):
    print(f"{c:5d} x  {nbytes[s] / max(c, 1) / 1e6:9.3f} MB each   {s}")
