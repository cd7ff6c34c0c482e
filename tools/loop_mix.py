"""Instruction mix of the MFMA loops of a kernel in an AMDGPU assembly listing (hipcc -S --cuda-device-only): for every backward
branch whose body holds MFMAs, the counts of MFMA / VALU / SALU / LDS / VMEM instructions (the innermost such loop per kernel).
usage: loop_mix.py file.s [substring of the mangled kernel name ...]"""
import collections, re, sys
s = open(sys.argv[1]).read()
want = sys.argv[2:]
for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M):
    name = m.group(1)
    if "gemm" not in name and "attn" not in name:
        continue
    if want and not any(w in name for w in want):
        continue
    i = m.end(); j = s.find("s_endpgm", i)
    raw = [l.strip() for l in s[i:j].split("\n")]
    lines = [l for l in raw if l and not l.startswith((";", ".s", ".p", ".a", ".t", ".g", ".w"))]
    labels = {}
    for idx, l in enumerate(lines):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm: labels[mm.group(1)] = idx
    loops = []
    for idx, l in enumerate(lines):
        mm = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < idx:
            seg = lines[labels[mm.group(1)]:idx]
            n = sum(x.startswith("v_mfma") for x in seg)
            if n: loops.append((len(seg), n, seg))
    if not loops: continue
    loops.sort()
    ln, n, seg = loops[0]                      # the shortest loop that holds MFMAs = the innermost
    c = collections.Counter()
    for x in seg:
        op = x.split()[0]
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "buffer_")): c["vmem"] += 1
    top = collections.Counter(x.split()[0] for x in seg if x.startswith("v_") and not x.startswith("v_mfma")).most_common(6)
    print(f"{name[4:52]:48s} mfma {c['mfma']:3d} valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d}  valu/mfma {c['valu'] / c['mfma']:.1f}  {top}")
