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

    def _run_embedders(self, batch: dict) -> list:
        """Every embedder's output, in order.  The frozen text towers are chains of small launches (308 rows: a few dozen tiles each) that
        leave most of the chip idle, and they do not depend on each other: with NK_TE_OVERLAP != 0 (default) every second frozen embedder that has
        weights runs on a side stream beside the previous one and is joined before the outputs are routed."""
        import os

        heavy = [i for i, m in enumerate(self.embedders) if not m.is_trainable and (p0 := next(m.parameters(), None)) is not None and p0.is_cuda]
        from ...graphs import graphs_enabled

        # (not when the towers are replayed from hipGraphs, NK_GRAPH=1 / "te": a replay belongs to the stream it was captured on)
        on_side = set(heavy[0::2]) if len(heavy) >= 2 and os.environ.get("NK_TE_OVERLAP", "1") != "0" and not graphs_enabled("te") else set()
        outs, side = [], None
        if on_side:
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream()
            side = self._side_stream
            side.wait_stream(torch.cuda.current_stream())
        for i, model in enumerate(self.embedders):
            args = self._gather(model, batch)
            with model.context():
                if i in on_side:
                    with torch.cuda.stream(side):
                        produced = model(*args)
                else:
                    produced = model(*args)
            outs.append(produced)
        if on_side:
            main = torch.cuda.current_stream()
            main.wait_stream(side)
            for i in on_side:
                for t in ((outs[i],) if torch.is_tensor(outs[i]) else outs[i]):
                    if torch.is_tensor(t):
                        t.record_stream(main)         # allocated on the side stream, consumed on this one
        return outs

    def forward(self, batch: dict, force_zero_embeddings: Optional[list] = None) -> dict:
        zeroed = set(force_zero_embeddings or ())
        cond: dict[str, Tensor] = {}
        for model, produced in zip(self.embedders, self._run_embedders(batch)):
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
