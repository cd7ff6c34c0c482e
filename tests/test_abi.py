"""The C-ABI library loads without a GPU and exports every symbol include/neurosis_hip.h declares
(no compute calls here)."""
import os
import re
import shutil
from pathlib import Path

import pytest

from neurosis_amd import lib

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "neurosis_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    l = lib.load()
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(l, s), f"{s} declared in include/neurosis_hip.h but not exported"
    assert l.nk_abi_version() == 5


def test_binding_table_matches_header():
    syms = set(declared_symbols()) - {"nk_last_error", "nk_abi_version"}
    bound = set(lib.SIGNATURES) | set(lib.SIZE_QUERIES)
    assert syms == bound, (syms ^ bound)


def test_argument_counts_match_header():
    text = (ROOT / "include" / "neurosis_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, argtypes in {**lib.SIGNATURES, **lib.SIZE_QUERIES}.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", text, flags=re.S)
        assert m, name
        n = len([a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"])
        assert n == len(argtypes), f"{name}: header has {n} parameters, binding {len(argtypes)}"


def test_engine_restores_from_safetensors_checkpoint(tmp_path):
    """DiffusionEngine.init_from_ckpt (reference models/diffusion.py:127-144): non-strict restore, `first_stage_model.*` leftovers
    of a full checkpoint ignored, UNet keys under model.diffusion_model.*"""
    import torch
    from safetensors.torch import save_file

    import neurosis_amd.modules.diffusion as D
    from neurosis_amd.models.diffusion import DiffusionEngine

    cfg = dict(in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=[2], channel_mult=[1, 2], num_head_channels=16,
               use_linear_in_transformer=True, transformer_depth=1, context_dim=32, use_checkpoint=False)
    torch.manual_seed(0)
    src = DiffusionEngine(D.UNetModel(**cfg), D.Denoiser(D.EpsPreconditioning()), None)
    sd = {k: v.contiguous() for k, v in src.state_dict().items()}
    sd["first_stage_model.encoder.conv_in.weight"] = torch.zeros(3)
    sd["some.other.key"] = torch.zeros(1)
    save_file(sd, str(tmp_path / "m.safetensors"))
    torch.manual_seed(1)
    dst = DiffusionEngine(D.UNetModel(**cfg), D.Denoiser(D.EpsPreconditioning()), None, ckpt_path=tmp_path / "m.safetensors")
    assert all(torch.equal(a, b) for a, b in zip(src.model.state_dict().values(), dst.model.state_dict().values()))
    missing, unexpected = dst.init_from_ckpt(tmp_path / "m.safetensors")
    assert missing == [] and unexpected == ["some.other.key"]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_register_of_an_asm_lds_read_is_touched_before_its_wait():
    """The kernels that keep LDS reads in flight behind hand-counted waits (gemm_w160.h, gemm_g2.h, attention) issue those reads from inline asm;
    to the compiler such a statement delivers its outputs on the spot, so the allocator may place a copy of them in front of the wait.  Round 6
    found one (the phi copies on the edge into gemm_w160.h's bias loop: one 16 x 64 block of one weight gradient computed on stale registers
    about once per hundred steps).  The screen walks the generated assembly of every such translation unit (no GPU needed; cached by source hash);
    its own self-test feeds it a listing with the hazard."""
    from tools.check_async_reads import check, screen_kernel

    hazard = """
	;;#ASMSTART
	ds_read_b64_tr_b16 v[10:11], v3 offset:0
	;;#ASMEND
	v_mov_b64_e32 v[20:21], v[10:11]
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mfma_f32_16x16x32_bf16 v[4:7], v[20:23], v[24:27], v[4:7]
	s_endpgm
"""
    lines, in_asm = [], False
    for no, text in enumerate(hazard.splitlines(), 1):
        if "#ASMSTART" in text:
            in_asm = True
        elif "#ASMEND" in text:
            in_asm = False
        else:
            lines.append((no, text, in_asm))
    found = screen_kernel("synthetic", lines)
    assert len(found) == 1 and "v_mov_b64" in found[0][1]
    clean = [(no, t, a) for no, t, a in lines if "v_mov_b64" not in t]
    assert screen_kernel("synthetic", clean) == []
    check(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neurosis_amd", "csrc"))
