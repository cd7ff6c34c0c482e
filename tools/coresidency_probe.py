"""Does an HBM-streaming kernel (the fused optimizer's class) cost a co-running halo-tile convolution (the frozen VAE encoder's class) less
when its waves are small enough to share a CU with a convolution workgroup?  The convolution kernel takes a CU's LDS and 2 x 235 of the
512 registers per lane of every SIMD: a streaming wave of <= 40 registers fits beside it, one of 72 (af_apply today) does not -- the two
kernels then share the chip CU by CU.  Probe kernels are built here with hipcc (tools only; not part of the library):
  light : persistent grid of G workgroups x 256 threads, <= 40 VGPRs, 4 x 16-byte loads in flight per lane
  heavy : the same traffic from a conventional big grid with ~72 VGPRs
Prints: convolution alone, stream alone (GB/s), and both together on two streams.   usage (GPU box): python tools/coresidency_probe.py"""
import ctypes, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neurosis_amd import ops
from neurosis_amd.ops import Img

SRC = r'''
#include <hip/hip_runtime.h>
typedef float float4_t __attribute__((ext_vector_type(4)));
// p = p * d - s * g ; 12 B touched per 4-byte element like a slim optimizer apply (read g, read p, write p)
template <int UNROLL>
__device__ __forceinline__ void body(float* __restrict__ p, const float* __restrict__ g, long n4, long start, long stride) {
  float4_t* p4 = (float4_t*)p; const float4_t* g4 = (const float4_t*)g;
  for (long i = start; i < n4; i += stride * UNROLL) {
    float4_t a[UNROLL], b[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { long j = i + u * stride; if (j < n4) { a[u] = __builtin_nontemporal_load(p4 + j); b[u] = __builtin_nontemporal_load(g4 + j); } }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { long j = i + u * stride; if (j < n4) { p4[j] = a[u] * 0.999f - b[u] * 1e-3f; } }
  }
}
extern "C" __global__ __launch_bounds__(256) void stream_light(float* p, const float* g, long n4) {
  body<2>(p, g, n4, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}
extern "C" __global__ __launch_bounds__(256) void stream_heavy(float* p, const float* g, long n4) {
  body<8>(p, g, n4, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}
extern "C" int launch(int heavy, int grid, float* p, const float* g, long n4, void* stream) {
  if (heavy) hipLaunchKernelGGL(stream_heavy, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, n4);
  else hipLaunchKernelGGL(stream_light, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, n4);
  return (int)hipGetLastError();
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "probe.hip"), "w").write(SRC)
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-Rpass-analysis=kernel-resource-usage", os.path.join(d, "probe.hip"), "-o", os.path.join(d, "probe.so")],
                   capture_output=True, text=True)
for l in r.stderr.splitlines():
    if "Function Name" in l or " VGPRs:" in l:
        print(l.split("remark:")[1].split("[-R")[0].strip())
lib = ctypes.CDLL(os.path.join(d, "probe.so"))
lib.launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]

N, H, W, Ci, Co = 4, 1024, 1024, 128, 128
x = Img((torch.randn(N * H * W, Ci, device="cuda")).to(torch.bfloat16), N, H, W)
w = torch.nn.Parameter(ops.conv_weight_param(Co, Ci, 3, 3).data.normal_(0, (9 * Ci) ** -0.5).cuda(), requires_grad=False)
bias = torch.randn(Co, device="cuda")
n = 1 << 30                                  # 4 GiB of p + 4 GiB of g per pass: far beyond the Infinity Cache
p = torch.zeros(n, device="cuda"); g = torch.randn(n, device="cuda")
side = torch.cuda.Stream()
NCONV = 6                                    # ~9 ms of convolutions


def conv():
    for _ in range(NCONV):
        ops.conv2d_fwd(x, w, bias, need_dx=False)


def stream(heavy, grid):
    lib.launch(heavy, grid, p.data_ptr(), g.data_ptr(), n // 4, side.cuda_stream)


def timed(fn_main, fn_side):
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    side.wait_stream(torch.cuda.current_stream())
    e[0].record()
    if fn_side is not None:
        e[2].record(side); fn_side(); e[3].record(side)
    if fn_main is not None:
        fn_main()
    e[1].record()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]), (e[2].elapsed_time(e[3]) if fn_side is not None else 0.0)


conv(); torch.cuda.synchronize()
tc = min(timed(conv, None)[0] for _ in range(3))
print(f"convolutions alone ({NCONV} x 4x1024^2 128->128): {tc:.2f} ms")
for name, heavy, grid in (("light, 256 WGs (1 wave/SIMD)", 0, 256), ("light, 512 WGs", 0, 512), ("light, 1024 WGs", 0, 1024), ("heavy, 256 WGs", 1, 256), ("heavy, 2048 WGs", 1, 2048), ("heavy, 8192 WGs", 1, 8192)):
    stream(heavy, grid); torch.cuda.synchronize()
    ts = min(timed(None, lambda: stream(heavy, grid))[1] for _ in range(3))
    both = [timed(conv, lambda: stream(heavy, grid)) for _ in range(3)]
    bm, bs = min(b[0] for b in both), min(b[1] for b in both)
    gb = 12.0 * n / 1e9
    print(f"{name:30s} stream alone {ts:6.2f} ms ({gb / ts:5.2f} TB/s) | together: convolutions {bm:6.2f} ms (+{bm - tc:5.2f}), stream {bs:6.2f} ms | "
          f"serial sum {tc + ts:6.2f}, overlapped end {max(bm, bs):6.2f}")
