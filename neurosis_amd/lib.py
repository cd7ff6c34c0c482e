"""ctypes binding of libneurosis_hip.so (the C-ABI declared in include/neurosis_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback behind these calls.  If the
shared object is missing or a kernel launch fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch  # noqa: F401  -- must come first: PyTorch-ROCm bundles the HIP runtime (libamdhip64) this library binds to;
# loading libneurosis_hip.so before torch would pull in /opt/rocm's copy and leave two runtimes in one process

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = CSRC / "libneurosis_hip.so"

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_long, C.c_float


class NkConvDesc(C.Structure):
    _fields_ = [(n, i32) for n in ("N", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad_t", "pad_l", "Ho", "Wo", "upsample")]


class NkAttnDesc(C.Structure):
    _fields_ = (
        [(n, i32) for n in ("B", "H", "Lq", "Lk", "D")]
        + [(n, i64) for n in ("sq", "sk", "sv", "so", "bq", "bk", "bv", "bo", "sdq", "sdk", "sdv", "sdo", "bdq", "bdk", "bdv", "bdo")]
        + [("scale", f32), ("causal", i32)]
    )


NK_COLPART_MAX = 32


class NkColpartBatch(C.Structure):
    _fields_ = [("part", vp * NK_COLPART_MAX), ("dgamma", vp * NK_COLPART_MAX), ("dbeta", vp * NK_COLPART_MAX), ("nrows", i32 * NK_COLPART_MAX),
                ("C", i32 * NK_COLPART_MAX), ("accumulate", i32 * NK_COLPART_MAX), ("n", i32)]


cdp, adp = C.POINTER(NkConvDesc), C.POINTER(NkAttnDesc)

# name -> argtypes; every entry point returns int (0 = ok).  Mirrors include/neurosis_hip.h one to one.
SIGNATURES: dict[str, list] = {
    "nk_linear_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, f32, vp],
    "nk_linear_dgrad": [vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, vp],
    "nk_linear_dgrad_geglu": [vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, vp],
    "nk_linear_fwd_geglu": [vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, vp],
    "nk_linear_fwd_geglu_s": [vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, vp],
    "nk_linear_dgrad_geglu_s": [vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, vp],
    "nk_linear_wgrad": [vp, vp, vp, i32, i32, i32, i64, i64, i64, i32, vp],
    "nk_linear_fwd_batched": [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i32, i32, i32, i64, i64, i64, vp],
    "nk_linear_wgrad_batched": [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i32, i32, i32, i64, i64, i64, i32, vp],
    "nk_linear_wgrad_bias": [vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i32, vp],
    "nk_conv2d_wgrad_bias": [cdp, vp, vp, vp, vp, i32, vp],
    "nk_conv2d_fwd": [cdp, vp, vp, vp, vp, vp, vp, vp],
    "nk_conv2d_fwd_stats": [cdp, vp, vp, vp, vp, vp, vp, vp, i32, vp],
    "nk_conv_weight_flip": [vp, vp, i32, i32, i32, vp],
    "nk_conv3x3_few_channels_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "nk_conv2d_dgrad": [cdp, vp, vp, vp, vp],
    "nk_conv2d_dgrad_flipped": [cdp, vp, vp, vp, vp],
    "nk_conv2d_wgrad": [cdp, vp, vp, vp, i32, vp],
    "nk_attention_fwd": [adp, vp, vp, vp, vp, vp, vp],
    "nk_attention_bwd": [adp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "nk_softmax_rows": [vp, i64, i32, vp],
    "nk_softmax_rows_bwd": [vp, vp, i64, i32, f32, vp],
    "nk_groupnorm_fwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "nk_groupnorm_sums": [vp, vp, vp, i32, i32, i32, i32, vp],
    "nk_groupnorm_sums_from_parts": [vp, vp, vp, i32, i32, i32, vp],
    "nk_groupnorm_apply": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "nk_groupnorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "nk_layernorm_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, f32, vp],
    "nk_layernorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "nk_layernorm_bwd_dx": [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "nk_layernorm_bwd_rows": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "nk_colpart_reduce_batch": [C.POINTER(NkColpartBatch), vp],
    "nk_layernorm_bwd_params": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "nk_geglu_fwd": [vp, vp, i64, i32, vp],
    "nk_geglu_bwd": [vp, vp, vp, i64, i32, vp],
    "nk_geglu_fwd_s": [vp, vp, vp, i64, i32, vp],
    "nk_geglu_bwd_s": [vp, vp, vp, i64, i32, vp],
    "nk_silu_fwd": [vp, vp, i64, vp],
    "nk_gelu_fwd": [vp, vp, i64, i32, vp],
    "nk_leaky_relu_fwd": [vp, vp, i64, f32, vp],
    "nk_leaky_relu_bwd": [vp, vp, vp, i64, f32, vp],
    "nk_maxpool2x2_fwd": [vp, vp, i32, i32, i32, i32, vp],
    "nk_maxpool2x2_bwd": [vp, vp, vp, i32, i32, i32, i32, vp],
    "nk_maxpool_fwd": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "nk_maxpool_bwd": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "nk_lpips_layer_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp],
    "nk_lpips_layer_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "nk_batchnorm_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, f32, vp],
    "nk_batchnorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp],
    "nk_batchnorm_eval": [vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, vp],
    "nk_silu_bwd": [vp, vp, vp, i64, vp],
    "nk_add": [vp, vp, vp, i64, vp],
    "nk_cat_channels": [vp, vp, vp, i64, i32, i32, vp],
    "nk_split_channels": [vp, vp, vp, i64, i32, i32, vp],
    "nk_upsample2x_bwd": [vp, vp, i32, i32, i32, i32, vp],
    "nk_nchw_to_nhwc": [vp, i32, vp, i32, i32, i32, i32, f32, vp],
    "nk_nhwc_to_nchw": [vp, vp, i32, i32, i32, i32, i32, vp],
    "nk_cast_f32_to_bf16": [vp, vp, i64, vp],
    "nk_cast_bf16_to_f32": [vp, vp, i64, vp],
    "nk_colsum": [vp, vp, vp, i64, i32, i64, i32, vp],
    "nk_colsum_batched": [vp, vp, vp, i64, i32, i64, i32, i32, vp],
    "nk_timestep_embedding": [vp, vp, i32, i32, f32, vp],
    "nk_edm_prepare": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "nk_edm_loss": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "nk_sample_prepare": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "nk_sample_denoise": [vp, vp, vp, vp, f32, vp, i32, i32, i32, i32, i32, vp],
    "nk_sample_euler_step": [vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, vp],
    "nk_adamw_flat": [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp],
    "nk_ema_flat": [vp, vp, i64, f32, vp],
    "nk_debug_raise_health": [vp],
    "nk_debug_stamp": [vp, vp],
    "nk_health_clear": [],
    "nk_health_export": [vp, vp],
    "nk_health_import": [vp, vp],
    "nk_adafactor_init": [vp, vp],
    "nk_adafactor_chunk": [vp, vp],
}

# entry points that return a size (long) instead of a status
SIZE_QUERIES: dict[str, list] = {
    "nk_groupnorm_ws_floats": [i32, i32, i32, i32],
    "nk_groupnorm_sums_ws_floats": [i32, i32, i32],
    "nk_conv2d_stats_tiles": [cdp, i32],
    "nk_conv2d_dgrad_flipped_ok": [cdp],
    "nk_linear_fwd_geglu_ok": [i32, i32, i32],
    "nk_layernorm_ws_floats": [i32, i32],
    "nk_layernorm_part_rows": [i32],
    "nk_colsum_ws_floats": [i64, i32],
    "nk_batchnorm_ws_floats": [i64, i32],
    "nk_lpips_layer_ws_floats": [i32, i32],
    "nk_attention_bwd_ws_floats": [adp],
    "nk_adafactor_tensor_bytes": [],
    "nk_gemm_sk_status": [],
    "nk_health_status": [],
}

_lib = None


class NkError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("NEUROSIS_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise NkError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C neurosis_amd/csrc`). There is no CPU fallback for the product path."
        )
    lib = C.CDLL(str(path))
    lib.nk_last_error.restype = C.c_char_p
    lib.nk_last_error.argtypes = []
    lib.nk_abi_version.restype = i32
    lib.nk_abi_version.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = i32
    for name, argtypes in SIZE_QUERIES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = i64
    _lib = lib
    return lib


def query(name: str, *args) -> int:
    return int(getattr(load(), name)(*args))


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise NkError(f"{name} failed (rc={rc}): {lib.nk_last_error().decode()}")
