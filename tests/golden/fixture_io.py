"""Golden fixtures without pickle: a fixture is a nested structure of dicts / lists / tuples whose leaves are tensors or plain scalars.
The tensors go into `<name>.safetensors` (keyed by their position in the structure), everything else into `<name>.json`, which mirrors the
structure with markers for what JSON cannot say (tensors, tuples, non-string dict keys).  Loading needs no code execution."""
from __future__ import annotations

import json
from pathlib import Path

import torch
from safetensors.torch import load_file, save_file

HERE = Path(__file__).resolve().parent


def _enc(o, path: str, tensors: dict):
    if torch.is_tensor(o):
        key = path or "_"
        tensors[key] = o.detach().contiguous().clone()
        return {"__tensor__": key}
    if isinstance(o, dict):
        if all(isinstance(k, str) and not k.startswith("__") for k in o):
            return {k: _enc(v, f"{path}/{k}" if path else k, tensors) for k, v in o.items()}
        return {"__items__": [[_enc(k, f"{path}/key{i}", tensors), _enc(v, f"{path}/{i}", tensors)] for i, (k, v) in enumerate(o.items())]}
    if isinstance(o, tuple):
        return {"__tuple__": [_enc(v, f"{path}/{i}", tensors) for i, v in enumerate(o)]}
    if isinstance(o, list):
        return [_enc(v, f"{path}/{i}", tensors) for i, v in enumerate(o)]
    if o is None or isinstance(o, (bool, int, float, str)):
        return o
    raise TypeError(f"fixture leaf of type {type(o).__name__} at {path!r}")


def _dec(o, tensors: dict):
    if isinstance(o, dict):
        if "__tensor__" in o:
            return tensors[o["__tensor__"]]
        if "__tuple__" in o:
            return tuple(_dec(v, tensors) for v in o["__tuple__"])
        if "__items__" in o:
            return {_dec(k, tensors): _dec(v, tensors) for k, v in o["__items__"]}
        return {k: _dec(v, tensors) for k, v in o.items()}
    if isinstance(o, list):
        return [_dec(v, tensors) for v in o]
    return o


def save_fixture(obj, name: str, directory: Path = HERE) -> None:
    tensors: dict = {}
    meta = _enc(obj, "", tensors)
    save_file(tensors, str(directory / f"{name}.safetensors"))
    (directory / f"{name}.json").write_text(json.dumps(meta))


def load_fixture(name: str, directory: Path = HERE):
    tensors = load_file(str(directory / f"{name}.safetensors"))
    return _dec(json.loads((directory / f"{name}.json").read_text()), tensors)
