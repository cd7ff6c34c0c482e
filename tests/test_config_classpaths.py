"""Drop-in boundary: the class paths of the reference's example configs under the `neurosis.` -> `neurosis_amd.` prefix
swap.  tests/golden/config_class_paths.json is data derived from
/root/reference/configs/sdxl/sdxl.example.yaml and configs/sd15/sd15.example.yml by make_golden.py::config_case: every
node of the `model:` tree that names a class, its init_args names (and plain values), and whether the path resolves in the
reference itself.  Here: every path the reference can resolve resolves after the swap, each class accepts the init_args
the YAML gives it, and the whole tree instantiates bottom-up (on the meta device: no weights, no GPU) the way LightningCLI /
jsonargparse builds it."""
import contextlib
import importlib
import inspect
import json
from functools import partial
from pathlib import Path

import pytest
import torch

NODES = json.loads((Path(__file__).parent / "golden" / "config_class_paths.json").read_text())
# class paths the reference's own example configs get wrong (they do not import in the reference either: the module is
# modules/diffusion/sampling/sigma_generators.py and the class DiscreteSigmaGenerator); kept as data, mapped to the real class
BROKEN_IN_REFERENCE = {"neurosis.modules.diffusion.sigma_sampling.DiscreteSampling": "neurosis_amd.modules.diffusion.DiscreteSigmaGenerator"}
# Lightning hands these two to the engine as callables (OptimizerCallable / LRSchedulerCallable), not instances
CALLABLE_SLOTS = ("optimizer", "scheduler")


def swap(cp: str) -> str:
    cp = BROKEN_IN_REFERENCE.get(cp, cp)
    return "neurosis_amd." + cp[len("neurosis."):] if cp.startswith("neurosis.") else cp


def resolve(cp: str):
    mod, _, name = cp.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def all_nodes():
    return [(cfg, n) for cfg, v in NODES.items() for n in v["nodes"]]


@pytest.mark.parametrize("cfg,node", all_nodes(), ids=lambda x: x if isinstance(x, str) else x["where"])
def test_class_path_resolves_and_accepts_its_init_args(cfg, node):
    cp = node["class_path"]
    if not node["resolves_in_reference"]:
        assert cp in BROKEN_IN_REFERENCE, f"{cp} does not resolve in the reference and is not a known-broken path"
    cls = resolve(swap(cp))
    assert inspect.isclass(cls), cp
    sig = inspect.signature(cls.__init__)
    params = sig.parameters
    has_kwargs = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())
    for name in node["init_arg_names"]:
        assert name in params or has_kwargs, f"{swap(cp)}.__init__ has no argument {name!r} (config {cfg}, {node['where']})"


def coerce(cls, kwargs: dict) -> dict:
    """what jsonargparse does from the type hints: PyYAML (YAML 1.1) reads `4e-7` as a string"""
    hints = {n: p.annotation for n, p in inspect.signature(cls.__init__).parameters.items()}
    out = {}
    for k, v in kwargs.items():
        if isinstance(v, str) and hints.get(k) in (float, "float"):
            v = float(v)
        out[k] = v
    return out


def build(tree_nodes, where: str):
    """instantiate the node at `where`, children first (what jsonargparse does with class_path / init_args)"""
    node = next(n for n in tree_nodes if n["where"] == where)
    cls = resolve(swap(node["class_path"]))
    kwargs = dict(node["plain_init_args"])
    for k in ("ckpt_path",):                        # checkpoint files are not in the repo
        if k in kwargs:
            kwargs[k] = None
    kwargs = {k: (None if isinstance(v, str) and v.startswith("${") else v) for k, v in kwargs.items()}   # ${data...} interpolations
    for name in node["init_arg_names"]:
        prefix = f"{where}.init_args.{name}"
        kids = [n for n in tree_nodes if n["where"] == prefix]
        if kids:
            slot = name in CALLABLE_SLOTS and where == "model"
            kwargs[name] = build_callable(tree_nodes, prefix) if slot else build(tree_nodes, prefix)
        else:
            items = sorted((n["where"] for n in tree_nodes if n["where"].startswith(prefix + "[") and n["where"].count(".init_args.") == prefix.count(".init_args.")),
                           key=lambda w: int(w[len(prefix) + 1:w.index("]", len(prefix))]))
            if items:
                kwargs[name] = [build(tree_nodes, w) for w in items]
    for k in ("input_key",):
        if k in kwargs and kwargs[k] is None:
            kwargs[k] = "image" if where == "model" else "caption"
    heavy = cls.__name__ in ("UNetModel", "AutoencoderKL", "FrozenCLIPEmbedder", "FrozenOpenCLIPEmbedder2")   # weights on the meta device
    with torch.device("meta") if heavy else contextlib.nullcontext():
        return cls(**coerce(cls, kwargs))


def build_callable(tree_nodes, where: str):
    node = next(n for n in tree_nodes if n["where"] == where)
    cls = resolve(swap(node["class_path"]))
    return partial(cls, **coerce(cls, node["plain_init_args"]))


@pytest.mark.parametrize("cfg", list(NODES))
def test_model_tree_instantiates_under_the_prefix_swap(cfg):
    from neurosis_amd.models import DiffusionEngine
    from neurosis_amd.optimizers import Adafactor, AdafactorScheduler

    eng = build(NODES[cfg]["nodes"], "model")
    assert isinstance(eng, DiffusionEngine)
    n_unet = sum(p.numel() for p in eng.model.diffusion_model.parameters())
    assert n_unet == (2_567_463_684 if "sdxl" in cfg else 859_520_964), n_unet      # SDXL-base / SD1.5 UNet sizes
    assert eng.vae_encoder is not None and hasattr(eng.vae_encoder, "quant_conv")
    assert eng.sampler is not None and eng.loss_fn is not None
    # the optimizer / scheduler slots are honoured, not dropped: configure_optimizers builds the config's classes
    out = eng.configure_optimizers()
    assert isinstance(out["optimizer"], Adafactor) and isinstance(out["lr_scheduler"]["scheduler"], AdafactorScheduler)
    g = out["optimizer"].param_groups[0]
    assert g["scale_parameter"] and g["relative_step"] and g["warmup_init"] and g["name"] == "UNet"
    assert out["lr_scheduler"]["scheduler"].get_last_lr() == [4e-7 if "sdxl" in cfg else out["lr_scheduler"]["scheduler"].initial_lr]


def test_engine_refuses_an_optimizer_it_cannot_fuse():
    import neurosis_amd.modules.diffusion as D
    from neurosis_amd.models import DiffusionEngine
    from tests.golden.make_golden import UNET_TINY

    eng = DiffusionEngine(model=D.UNetModel(**UNET_TINY), denoiser=D.Denoiser(preconditioning=D.EpsPreconditioning()), first_stage_model=None,
                          optimizer=partial(torch.optim.SGD, lr=0.1), loss_fn=None)
    with pytest.raises(TypeError, match="cannot be fused"):
        eng.configure_optimizers()
