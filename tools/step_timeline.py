"""Where the replayed segments of one training step lie in time, UNTRACED: device timestamps (nk_debug_stamp, the 100 MHz constant clock)
captured into the forward graph and into every backward segment M_k (main stream) / W_k (weight-gradient stream), plus eager stamps around
the VAE encoder, the backward as a whole and the optimizer.  rocprofv3's kernel trace perturbs how the two streams overlap (DESIGN section
5), so this is the measurement of that overlap.   usage (GPU box): python tools/step_timeline.py [--steps 6]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from neurosis_amd import graphs, ops

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--serialize", action="store_true", help="no weight-gradient side stream, optimizer in line (as bench.py --serialize)")
ap.add_argument("--drop-wgrads", action="store_true", help="TIMING ONLY (wrong gradients): nothing is issued on the weight-gradient stream -- how long the main chain takes alone")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
eng = bench.build_engine(dev, (1024, 1024), None)
eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)      # as bench.py: configs/sdxl/sdxl.example.yaml:158-164
if args.drop_wgrads:
    from neurosis_amd import lib as _lib
    import neurosis_amd.nn as _nn
    import neurosis_amd.modules.diffusion.loss as _loss
    _orig_call = _lib.call
    _DROP = ("nk_linear_wgrad", "nk_linear_wgrad_bias", "nk_linear_wgrad_batched", "nk_conv2d_wgrad", "nk_conv2d_wgrad_bias", "nk_colpart_reduce_batch",
             "nk_layernorm_bwd_params")

    def _call(name, *a):
        if name in _DROP:
            return None
        return _orig_call(name, *a)

    for _m in (_lib, ops, _nn, _loss):
        if hasattr(_m, "call"):
            _m.call = _call
if args.serialize:
    eng.store.state.wgrad_stream = None
    eng.overlap_optimizer = False
graphs.stamps = st = graphs.Stamps(dev)
gen = torch.Generator(device=dev).manual_seed(42)
_opt, _enc, _join = eng.adafactor.step, eng.encode_first_stage, eng.join_optimizer


def opt_step(gs=1.0):          # (runs inside optimizer_step's stream scope: these two stamps are on the optimizer stream)
    st.mark("opt.begin")
    _opt(gs)
    st.mark("opt.end")


def encode(x):
    st.mark("vae.begin")
    z = _enc(x)
    st.mark("vae.end")
    return z


def join():
    _join()
    st.mark("optimizer.joined")      # main stream, behind the wait for the previous step's update


eng.adafactor.step, eng.encode_first_stage, eng.join_optimizer = opt_step, encode, join


def step(i):
    st.mark("step.begin")
    batch = bench.synthetic_batch(dev, args.batch, (1024, 1024), gen, True)
    sig = bench.draw_sigmas(args.batch, gen, dev)
    st.mark("batch.end")
    eng.accumulate(0, None, last=True)
    loss = eng.training_step(batch, 0, sigmas=sig)
    st.mark("forward.end")
    loss.backward()
    st.mark("backward.end")
    eng.optimizer_step(lr=1e-6, weight_decay=1e-2, grad_scale=1.0, dp=None)
    st.mark("optimizer.issued")
    return loss


for i in range(args.steps):
    step(i)
st.mark("next.begin")
t = st.read()
# the last step's stamps: everything was overwritten by it (every label is stamped once per step)
base = t["step.begin"]
rows = sorted(((v - base) / 1e3, k) for k, v in t.items())
print(f"step (step.begin -> next.begin): {(t['next.begin'] - base) / 1e3:.2f} ms")
for ms, k in rows:
    print(f"  {ms:9.3f} ms  {k}")
segs = sorted({int(k[1:].split('.')[0]) for k in t if k[0] == 'M' and k[1].isdigit()})
print("segment: M begin..end (len) | gap since previous M end | W begin..end (len) | W begin - M end")
prev_end = None
tot_gap = tot_m = tot_w = 0.0
for k in segs:
    mb, me = (t[f"M{k}.begin"] - base) / 1e3, (t[f"M{k}.end"] - base) / 1e3
    gap = 0.0 if prev_end is None else mb - prev_end
    tot_gap += gap
    tot_m += me - mb
    line = f"  {k:3d}: M {mb:8.3f}..{me:8.3f} ({me - mb:6.3f}) | gap {gap:6.3f}"
    if f"W{k}.begin" in t:
        wb, we = (t[f"W{k}.begin"] - base) / 1e3, (t[f"W{k}.end"] - base) / 1e3
        tot_w += we - wb
        line += f" | W {wb:8.3f}..{we:8.3f} ({we - wb:6.3f}) | {wb - me:6.3f}"
    print(line)
    prev_end = me
print(f"sum of M segments {tot_m:.2f} ms, of the gaps between them {tot_gap:.2f} ms, of W segments {tot_w:.2f} ms")
