"""Where the frozen VAE encoder's time goes at the metric's size (batch 4, 1024^2): every top-level piece of Encoder.fwd timed with HIP
events (each piece run alone, back to back, inputs resident)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
from neurosis_amd.modules.diffusion.model import Encoder
from neurosis_amd.ops import Img


def t(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


torch.manual_seed(0)
enc = Encoder(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3, resolution=1024,
              z_channels=4, double_z=True, attn_type="vanilla-xformers", standalone=True, embed_dim=4).cuda().requires_grad_(False)
B, R = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.rand(B, 3, R, R, device="cuda") * 2 - 1
with torch.no_grad():
    total = t(lambda: enc(x, regularize=True))
    print(f"whole encoder ({B} x {R}^2): {total:8.2f} ms")
    img = Img(ops.nchw_to_tokens(x, 8), B, R, R)
    print(f"  nchw_to_tokens          {t(lambda: ops.nchw_to_tokens(x, 8)):8.3f} ms")
    h = enc.conv_in.fwd(img, need_dx=False)[0]
    print(f"  conv_in                 {t(lambda: enc.conv_in.fwd(img, need_dx=False)):8.3f} ms")
    for i_level in range(enc.num_resolutions):
        level = enc.down[i_level]
        for i_block in range(enc.num_res_blocks):
            blk = level.block[i_block]
            hin = h
            ms = t(lambda: blk.fwd(hin, want_sums=True))
            h = blk.fwd(hin, want_sums=True)
            print(f"  down.{i_level}.block.{i_block} ({hin.C:3d}->{h.C:3d} @ {h.H:4d})  {ms:8.3f} ms")
        if i_level != enc.num_resolutions - 1:
            hin = Img(h.t, h.N, h.H, h.W)
            print(f"  down.{i_level}.downsample        {t(lambda: level.downsample.fwd(hin)):8.3f} ms")
            h = level.downsample.fwd(hin)
    hin = Img(h.t, h.N, h.H, h.W)
    print(f"  mid.block_1             {t(lambda: enc.mid.block_1.fwd(hin, want_sums=True)):8.3f} ms")
    h = enc.mid.block_1.fwd(hin, want_sums=True)
    hin2 = h
    print(f"  mid.attn_1              {t(lambda: enc.mid.attn_1.fwd(hin2)):8.3f} ms")
    h = enc.mid.attn_1.fwd(hin2)
    hin3 = Img(h.t, h.N, h.H, h.W)
    print(f"  mid.block_2             {t(lambda: enc.mid.block_2.fwd(hin3, want_sums=True)):8.3f} ms")
    h = enc.mid.block_2.fwd(hin3, want_sums=True)
    hin4 = h
    print(f"  norm_out + conv_out     {t(lambda: enc.conv_out.fwd(ops.groupnorm_fwd(hin4, enc.norm_out.weight, enc.norm_out.bias, 32, 1e-6, True)[0], need_dx=False)):8.3f} ms")
