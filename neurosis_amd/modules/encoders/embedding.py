"""Conditioner glue (everything except the text encoders themselves).

Public surface of `neurosis.modules.encoders.embedding` (reference :17-149): `AbstractEmbModel` (an embedder declares
`input_key` or `input_keys`, optionally `ucg_rate`, `is_trainable`) and `GeneralConditioner`, which runs every embedder on
its batch entries, files each output under "vector" / "crossattn" / "concat" by its rank and concatenates outputs that
share a key.  Frozen text encoders are not rebuilt here: their outputs enter through `PrecomputedEmbedder`.
"""
from __future__ import annotations

from contextlib import nullcontext
from typing import Optional, Sequence

import numpy as np
import torch
from torch import Tensor, nn

# output rank -> (conditioning key, dim along which several embedders' outputs are joined)
_ROUTE = {2: ("vector", 1), 3: ("crossattn", 2), 4: ("concat", 1), 5: ("concat", 1)}


class AbstractEmbModel(nn.Module):
    def __init__(self, name: Optional[str] = None, input_key: Optional[str] = None, ucg_rate: Optional[float] = 0.0,
                 is_trainable: Optional[bool] = None, base_lr: Optional[float] = None):
        super().__init__()
        # a subclass may have fixed any of these as class attributes; constructor values only fill the gaps
        defaults = {"is_trainable": is_trainable or False, "ucg_rate": ucg_rate, "name": name or type(self).__name__}
        optional = {"input_key": input_key, "base_lr": base_lr}
        for attr, value in defaults.items():
            if not hasattr(self, attr):
                setattr(self, attr, value)
        for attr, value in optional.items():
            if value is not None and not hasattr(self, attr):
                setattr(self, attr, value)

    def freeze(self) -> None:
        self.requires_grad_(False)
        self.eval()

    @property
    def context(self):
        """context manager factory the conditioner runs the embedder under"""
        return nullcontext if self.is_trainable else torch.no_grad


class PrecomputedEmbedder(AbstractEmbModel):
    """A tensor the batch already carries (a frozen text encoder's output computed elsewhere): (B, 77, C) is filed under
    "crossattn", (B, C) under "vector"."""

    def forward(self, x: Tensor) -> Tensor:
        return x


class GeneralConditioner(nn.Module):
    OUTPUT_DIM2KEYS = {rank: key for rank, (key, _) in _ROUTE.items()}
    KEY2CATDIM = {key: dim for key, dim in _ROUTE.values()}

    def __init__(self, emb_models: Sequence[AbstractEmbModel]):
        super().__init__()
        for number, model in enumerate(emb_models):
            kind = type(model).__name__
            if not isinstance(model, AbstractEmbModel):
                raise ValueError(f"embedder model #{number} {kind} is not a subclass of AbstractEmbModel")
            if not (hasattr(model, "input_key") or hasattr(model, "input_keys")):
                raise KeyError(f"need either 'input_key' or 'input_keys' for embedder #{number} {kind}")
        if len(emb_models) == 0:
            raise ValueError("no embedders were added! what is my purpose? why am I here? check your config!")
        self.embedders = nn.ModuleList(emb_models)
        self.rng = np.random.default_rng()

    # -- pieces of forward ------------------------------------------------------------------------
    def _gather(self, model: AbstractEmbModel, batch: dict):
        """positional arguments for one embedder, taken from the batch"""
        key = getattr(model, "input_key", None)
        if key is None:
            return [batch[k] for k in model.input_keys]
        value = batch[key]
        if isinstance(value, list) and type(model).__name__ == "ConcatTimestepEmbedderND":
            # size / crop tuples arrive as python lists: one fp32 row per sample, on the device of the batch's tensors
            device = batch["image"].device if "image" in batch else next(v.device for v in batch.values() if torch.is_tensor(v))
            value = torch.tensor(value, device=device, dtype=torch.float32)
        if key == "caption" and model.ucg_rate > 0.0 and self.rng.random() < model.ucg_rate:
            value = [" "] * len(value)          # unconditional-guidance dropout of the whole caption batch
        return [value]

    @staticmethod
    def _drop_rows(emb: Tensor, rate: float) -> Tensor:
        """zero whole samples with probability `rate` (non-caption embedders)"""
        keep = torch.bernoulli(torch.full((emb.shape[0],), 1.0 - rate, device=emb.device))
        return emb * keep.reshape((-1,) + (1,) * (emb.dim() - 1)).to(emb.dtype)

    def forward(self, batch: dict, force_zero_embeddings: Optional[list] = None) -> dict:
        zeroed = set(force_zero_embeddings or ())
        cond: dict[str, Tensor] = {}
        for model in self.embedders:
            with model.context():
                produced = model(*self._gather(model, batch))
            if torch.is_tensor(produced):
                produced = (produced,)
            elif not isinstance(produced, (list, tuple)):
                raise ValueError(f"encoder outputs must be tensors or a sequence, but got {type(produced)}")
            key_of_model = getattr(model, "input_key", None)
            for emb in produced:
                slot, cat_dim = _ROUTE[emb.dim()]
                if key_of_model is not None and key_of_model in zeroed:
                    emb = torch.zeros_like(emb)
                elif model.ucg_rate > 0.0 and key_of_model != "caption":
                    emb = self._drop_rows(emb, model.ucg_rate)
                cond[slot] = emb if slot not in cond else torch.cat((cond[slot], emb.to(cond[slot].dtype)), cat_dim)
        return cond
