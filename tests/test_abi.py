"""The C-ABI library loads without a GPU and exports every symbol include/neurosis_hip.h declares
(no compute calls here)."""
import re
from pathlib import Path

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
    assert l.nk_abi_version() == 1


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
