"""ISA screen for the hazard behind round 6's rare garbage weight gradient: an LDS read issued from inline asm returns its
data asynchronously, but to the compiler the asm statement DEFINES its output registers there and then -- so a register copy the allocator places
between the read and the hand-written `s_waitcnt lgkmcnt(N)` (typically the phi copies on the edge into a loop whose body keeps reads in flight
across iterations) copies whatever the register held before.  The kernel then computes on stale registers, almost always unnoticed.

For every kernel of an assembly listing (hipcc -S) this walks the instructions in order, keeps the destination registers of the LDS reads that are
still in flight (LDS operations return in order: `s_waitcnt lgkmcnt(N)` retires all but the N youngest), follows each backward branch once (the
second trip through a loop starts from the state the first one left), and reports every instruction that touches a register a read still in
flight is going to write.
    python tools/check_async_reads.py file.s [file.s ...]        # exit status 1 if anything is reported
    python tools/check_async_reads.py --build                    # compile neurosis_amd/csrc/*.hip to assembly (hipcc -S) and screen them all
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^\s*s_(?:c?branch\w*)\s+(\.LBB\d+_\d+)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def screen_kernel(name, lines):
    """lines: [(line number, text, in_asm)] of one kernel.  Walks the control-flow graph: the state is the ordered list of LDS reads in flight;
    every (block start, state) pair is visited once."""
    labels = {}
    for idx, (_, text, _) in enumerate(lines):
        m = LABEL.match(text)
        if m:
            labels[m.group(1)] = idx
    findings = {}
    seen = set()
    work = [(0, ())]            # (pc, in-flight reads oldest first: (line number, from asm))
    dsts = {}                   # line number of a read -> its destination registers
    smem = set()                # line numbers of scalar memory reads (out-of-order returns on lgkmcnt)
    budget = 400 * len(lines) + 10000
    while work and budget > 0:
        pc, state = work.pop()
        key = (pc, tuple(r for r, _ in state))
        if key in seen:
            continue
        seen.add(key)
        inflight = list(state)
        while pc < len(lines) and budget > 0:
            budget -= 1
            no, text, in_asm = lines[pc]
            ins = text.split(";")[0].strip()
            if LABEL.match(text) and (pc, tuple(r for r, _ in inflight)) != key:
                work.append((pc, tuple(inflight)))          # a block start: continue from the work list (deduplicated)
                break
            pc += 1
            if not ins or ins.endswith(":") or ins.startswith("."):
                continue
            op = ins.split()[0]
            if op == "s_endpgm":
                break
            if op == "s_waitcnt":
                n = None
                m = LGKM.search(ins)
                if m:
                    n = int(m.group(1))
                elif "cnt" not in ins.split(None, 1)[1]:                      # raw immediate: lgkmcnt is bits 11:8
                    n = (int(ins.split()[1], 0) >> 8) & 15
                if n is not None and n > 0 and in_asm:
                    # scalar memory reads share lgkmcnt with LDS and return OUT OF ORDER: one that is YOUNGER than an asm LDS read and returns
                    # before it lets a hand-counted lgkmcnt(N > 0) pass with that read still outstanding.  (An OLDER one only makes the wait
                    # longer: the compiler's hoisted kernel-argument loads in front of a loop are fine.)
                    seen_asm = False
                    for r, a in inflight:
                        if a:
                            seen_asm = True
                        elif r in smem and seen_asm:
                            findings.setdefault((no, -1), (no, ins + "      [counted wait with a younger scalar memory read in flight]", r, [0]))
                            break
                if n is not None and n < len(inflight):
                    inflight = inflight[len(inflight) - n:] if n else []
                continue
            rest = ins.split(None, 1)[1] if len(ins.split(None, 1)) > 1 else ""
            touched = regs_of(rest)
            for rno, asm_read in inflight:
                if asm_read and touched & dsts[rno]:
                    findings.setdefault((no, rno), (no, ins, rno, sorted(touched & dsts[rno])))
                    break
            if op.startswith("ds_read") or op.startswith("ds_load"):
                dsts[no] = regs_of(rest.split(",")[0])
                inflight.append((no, in_asm))
            elif op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_mem") or op.startswith("s_buffer_load"):
                dsts[no] = set()
                inflight.append((no, False))                   # counts on lgkmcnt without a vector destination to protect
                if op.startswith("s_"):
                    smem.add(no)
            if len(inflight) > 40:
                inflight = inflight[-40:]
            m = BRANCH.match(text)
            if m and m.group(1) in labels:
                work.append((labels[m.group(1)], tuple(inflight)))
                if op == "s_branch":
                    break
    return [findings[k] for k in sorted(findings)]


def _kernels_of(path):
    kernels, cur, name, in_asm = [], None, None, False
    with open(path) as fh:
        for no, raw in enumerate(fh, 1):
            text = raw.rstrip("\n")
            m = re.match(r"^(_Z\w+|nk_\w+):\s", text + " ")
            if m and cur is None:
                name, cur, in_asm = m.group(1), [], False
                continue
            if cur is None:
                continue
            if "#ASMSTART" in text:
                in_asm = True
            elif "#ASMEND" in text:
                in_asm = False
            elif text.startswith(".Lfunc_end"):
                kernels.append((name, cur))
                cur = None
            else:
                cur.append((no, text, in_asm))
    return kernels


def screen_file(path):
    kernels = _kernels_of(path)
    total = 0
    for kname, lines in kernels:
        fs = screen_kernel(kname, lines)
        if fs:
            print(f"{os.path.basename(path)}: {kname}: {len(fs)} instruction(s) touch a register an inline-asm LDS read still in flight will write")
            for no, ins, rno, hit in fs[:8]:
                print(f"    line {no}: {ins}    <- v{hit[0]}{'..' if len(hit) > 1 else ''} is the destination of the read at line {rno}")
            total += len(fs)
    return total, len(kernels)


def _sources_hash(csrc: str) -> str:
    import hashlib

    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    h.update(open(os.path.abspath(__file__), "rb").read())
    return h.hexdigest()


def check(csrc: str, use_cache: bool = True) -> None:
    """What `__graft_entry__.build()` and the CPU suite run: every translation unit of csrc/ that reads LDS from inline asm (itself or through
    its headers) compiled to assembly and screened; raises AssertionError with the findings.  The verdict is cached beside the objects, keyed
    by a hash of all sources and of this script."""
    stamp = os.path.join(csrc, ".async_reads_ok")
    digest = _sources_hash(csrc)
    if use_cache and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return
    with_asm_reads = {n for n in os.listdir(csrc) if n.endswith((".hip", ".h")) and re.search(r"ds_read\w* %", open(os.path.join(csrc, n)).read())}
    units = []
    for n in sorted(os.listdir(csrc)):
        if n.endswith(".hip"):
            text = open(os.path.join(csrc, n)).read()
            if n in with_asm_reads or any(f'#include "{h}"' in text for h in with_asm_reads):
                units.append(n)
    assert units, "no translation unit with inline-asm LDS reads found: the screen's source pattern is stale"
    tmp = tempfile.mkdtemp(prefix="nk_isa_")
    report = []
    for n in units:
        out = os.path.join(tmp, n[:-4] + ".s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-function", "-Wno-inline-asm",
                            "-S", "--cuda-device-only", os.path.join(csrc, n), "-o", out], capture_output=True, text=True, cwd=csrc, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        kernels = _kernels_of(out)
        assert kernels, f"{n}: no kernels found in the assembly: the screen's parser is stale"
        for kname, lines in kernels:
            for no, ins, rno, hit in screen_kernel(kname, lines):
                report.append(f"{n}: {kname}: line {no}: {ins}  <- v{hit[0]} is the destination of the asm LDS read at line {rno}, still in flight")
    assert not report, "registers of inline-asm LDS reads touched before their wait:\n" + "\n".join(report[:20])
    try:
        open(stamp, "w").write(digest)
    except OSError:
        pass


def main(argv):
    files = [a for a in argv if not a.startswith("--")]
    tmp = None
    if "--build" in argv:
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "neurosis_amd", "csrc")
        tmp = tempfile.mkdtemp(prefix="nk_isa_")
        for src in sorted(glob.glob(os.path.join(here, "*.hip"))):
            out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S", "--cuda-device-only", src, "-o", out],
                           check=True, cwd=here)
            files.append(out)
    bad = 0
    for f in files:
        n, k = screen_file(f)
        print(f"{os.path.basename(f)}: {k} kernels screened, {n} finding(s)")
        bad += n
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
