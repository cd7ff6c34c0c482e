"""The ISA check behind attn512.h's physically addressed accumulator file: the kernel keeps its O^T accumulator in ALL
256 AGPRs from inline asm, which is only sound while hipcc allocates no AGPR of its own behind the zero fill, spills nothing on the hot path
of the key loop, and the loop body is exactly the 64 MFMAs that were written.  A compiler bump can change any of that silently, so
`__graft_entry__.build()` runs this on every build and tests/test_attn512_gpu.py runs it in the CPU suite (hipcc cross-compiles; no GPU).
usage: python tools/check_attn512_isa.py"""
import os
import re
import shutil
import subprocess
import sys


def _sources_hash(csrc: str) -> str:
    import hashlib

    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.startswith(("attention", "attn512", "nk_common")) and name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    h.update(open(os.path.abspath(__file__), "rb").read())
    return h.hexdigest()


def check(csrc: str, use_cache: bool = True) -> None:
    """Compiles attention.hip to ISA (~10 s) and checks it; the verdict is cached beside the objects, keyed by a hash of the attention sources and of
    this script, so a build that changed nothing in them does not pay for it again."""
    stamp = os.path.join(csrc, ".attn512_isa_ok")
    digest = _sources_hash(csrc)
    if use_cache and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-function", "-Wno-inline-asm", "-S",
                          "--cuda-device-only", os.path.join(csrc, "attention.hip"), "-o", "-"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    _check_fwd(lines)
    _check_bwd(lines)
    try:
        open(stamp, "w").write(digest)
    except OSError:
        pass


def _check_bwd(lines) -> None:
    """attn512_bwd_kernel<0 / 1> run at launch_bounds(256, 1) with 128 accumulator + 64 prefetch registers per lane; an earlier form spilled 155
    registers (header comment of csrc/attn512_bwd.h).  Gate: no scratch traffic anywhere in either instantiation."""
    for tag in ("_Z18attn512_bwd_kernelILi0E", "_Z18attn512_bwd_kernelILi1E"):
        start = next(i for i, l in enumerate(lines) if l.startswith(tag))
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        spills = [lines[i].strip() for i in range(start, end) if "scratch_" in lines[i]]
        assert not spills, (tag, spills[:4])


def _check_fwd(lines) -> None:
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z18attn512_fwd_kernel"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    in_asm, first_acc = False, None
    loop_marks, foreign, scratch = [], [], []
    for i in range(start, end):
        l = lines[i]
        if "ASMSTART" in l:
            in_asm = True
            continue
        if "ASMEND" in l:
            in_asm = False
            continue
        if "Depth=1" in l:                       # block labels of the key loop (its layout may be rotated: header not first)
            loop_marks.append(i)
        if in_asm and first_acc is None and "v_accvgpr_write_b32 a[0*16+0]" in l:
            first_acc = i
        if not in_asm and first_acc is not None and re.search(r"accvgpr|\ba\[|\ba\d+\b", l):
            foreign.append(l.strip())
        if "scratch_" in l:
            scratch.append(i)
    assert first_acc is not None and loop_marks
    assert not foreign, foreign[:5]
    loop_head = min(loop_marks)
    loop_end = next(i for i in range(max(loop_marks) + 1, end) if re.match(r"^\.LBB", lines[i]) or "s_endpgm" in lines[i])
    # scratch traffic in the key loop is tolerated only on the ragged-tail path (the blocks that clamp rows with v_min_i32 / s_min_i32): a reload
    # waits for vmcnt(0), i.e. for the tile DMA in flight
    hot = [i for i in scratch if loop_head <= i <= loop_end]
    for i in hot:
        block = "\n".join(lines[max(loop_head, i - 60):i + 60])
        assert "v_min_i32" in block or "s_min_i32" in block, (i - start, lines[i])
    body = [l for l in lines[loop_head:loop_end] if "v_mfma_f32_32x32x16_bf16" in l]
    assert len(body) == 64, len(body)



if __name__ == "__main__":
    check(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neurosis_amd", "csrc"), use_cache=False)
    print("attn512 ISA check ok (forward: AGPR file, hot-path scratch, 64 MFMAs; backward <0> / <1>: no scratch)")
