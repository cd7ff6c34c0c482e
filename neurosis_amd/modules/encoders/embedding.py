"""Conditioner glue (the part that is not a text encoder): `AbstractEmbModel` and
`GeneralConditioner` as in the reference (`modules/encoders/embedding.py:17-57,59-149`): embedders are routed by
`input_key` / `input_keys`, their outputs by tensor rank (2 -> "vector", 3 -> "crossattn", 4/5 -> "concat") and
concatenated along `KEY2CATDIM`; `ucg_rate` drops captions / zeroes embeddings per sample as the reference does.
Frozen text encoders are out of scope: their OUTPUTS enter through `PrecomputedEmbedder`."""
from __future__ import annotations

from contextlib import nullcontext
from typing import Optional

import numpy as np
import torch
from torch import Tensor, nn


class AbstractEmbModel(nn.Module):
    """embedding.py:17-57"""

    def __init__(self, name: Optional[str] = None, input_key: Optional[str] = None, ucg_rate: Optional[float] = 0.0,
                 is_trainable: Optional[bool] = None, base_lr: Optional[float] = None):
        super().__init__()
        if not hasattr(self, "is_trainable"):
            self.is_trainable = is_trainable or False
        if not hasattr(self, "ucg_rate"):
            self.ucg_rate = ucg_rate
        if not hasattr(self, "input_key") and input_key is not None:
            self.input_key = input_key
        if not hasattr(self, "base_lr") and base_lr is not None:
            self.base_lr = base_lr
        if not hasattr(self, "name"):
            self.name = name or str(self.__class__.__name__)

    def freeze(self) -> None:
        self.eval()
        self.requires_grad_(False)

    @property
    def context(self):
        return nullcontext if self.is_trainable else torch.no_grad


class PrecomputedEmbedder(AbstractEmbModel):
    """Passes a tensor the batch already carries (the output of a frozen text encoder computed elsewhere) through the
    conditioner's routing: (B, 77, C) lands in "crossattn", (B, C) in "vector"."""

    def forward(self, x: Tensor) -> Tensor:
        return x


class GeneralConditioner(nn.Module):
    """embedding.py:59-149"""

    OUTPUT_DIM2KEYS = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}

    def __init__(self, emb_models: list[AbstractEmbModel]):
        super().__init__()
        embedders = []
        for idx, embedder in enumerate(emb_models):
            if not isinstance(embedder, AbstractEmbModel):
                raise ValueError(f"embedder model #{idx} {embedder.__class__.__name__} is not a subclass of AbstractEmbModel")
            if not any((hasattr(embedder, "input_key"), hasattr(embedder, "input_keys"))):
                raise KeyError(f"need either 'input_key' or 'input_keys' for embedder #{idx} {embedder.__class__.__name__}")
            embedders.append(embedder)
        if len(embedders) == 0:
            raise ValueError("no embedders were added! what is my purpose? why am I here? check your config!")
        self.embedders = nn.ModuleList(embedders)
        self.rng = np.random.default_rng()

    def forward(self, batch: dict, force_zero_embeddings: Optional[list] = None) -> dict:
        output = dict()
        force_zero_embeddings = force_zero_embeddings or []
        for embedder in self.embedders:
            with embedder.context():
                if getattr(embedder, "input_key", None) is not None:
                    inputs = batch[embedder.input_key]
                    if isinstance(inputs, list) and embedder.__class__.__name__ == "ConcatTimestepEmbedderND":
                        ref = batch["image"] if "image" in batch else next(v for v in batch.values() if torch.is_tensor(v))
                        inputs = torch.tensor(inputs, device=ref.device, dtype=torch.float32)
                    if embedder.ucg_rate > 0.0 and embedder.input_key == "caption" and self.rng.random() < embedder.ucg_rate:
                        inputs = [" "] * len(inputs)
                    emb_out = embedder(inputs)
                elif getattr(embedder, "input_keys", None) is not None:
                    emb_out = embedder(*[batch[k] for k in embedder.input_keys])
                else:
                    raise KeyError(f"embedder {embedder.__class__.__name__} has neither input_key nor input_keys")
            if not isinstance(emb_out, (Tensor, list, tuple)):
                raise ValueError(f"encoder outputs must be tensors or a sequence, but got {type(emb_out)}")
            if not isinstance(emb_out, (list, tuple)):
                emb_out = [emb_out]
            for emb in emb_out:
                out_key = self.OUTPUT_DIM2KEYS[emb.dim()]
                if hasattr(embedder, "input_key") and embedder.input_key in force_zero_embeddings:
                    emb = torch.zeros_like(emb)
                elif embedder.ucg_rate > 0.0 and embedder.input_key != "caption":
                    keep = torch.bernoulli(torch.full((emb.shape[0],), 1.0 - embedder.ucg_rate, device=emb.device))
                    emb = emb.mul(keep.reshape((-1,) + (1,) * (emb.dim() - 1)).to(emb.dtype))
                if out_key in output:
                    output[out_key] = torch.cat((output[out_key], emb.to(output[out_key].dtype)), self.KEY2CATDIM[out_key])
                else:
                    output[out_key] = emb
        return output
