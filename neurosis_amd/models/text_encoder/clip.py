"""CLIP text towers as frozen embedders: `FrozenCLIPEmbedder` (OpenAI CLIP ViT-L/14 text model in the HuggingFace layout) and
`FrozenOpenCLIPEmbedder2` (OpenCLIP ViT-bigG/14 text model in the open_clip layout), with the constructor arguments, layer
selection and return conventions of `neurosis.models.text_encoder.clip` (:22-388).

The reference delegates the transformer itself to third-party code (transformers' CLIPTextModel, open_clip's CLIP); here both
towers are one pre-LN causal transformer over the HIP kernels -- LayerNorm, QKV / out / MLP GEMMs with fused bias and
residual, the fused attention kernel with its causal flag (77 tokens: one key tile), GELU -- that differ only in parameter
names and activation, so that either checkpoint layout loads unchanged:

    HF        text_model.embeddings.{token,position}_embedding.weight, text_model.encoder.layers.N.{self_attn.{q,k,v,out}_proj,
              layer_norm1, layer_norm2, mlp.fc1, mlp.fc2}, text_model.final_layer_norm          (transformers 4.x naming;
              the un-prefixed 5.x names load too)
    open_clip token_embedding.weight, positional_embedding, transformer.resblocks.N.{ln_1, attn.in_proj_{weight,bias},
              attn.out_proj, ln_2, mlp.c_fc, mlp.c_proj}, ln_final, text_projection, logit_scale

Forward only (the encoders are frozen in the SDXL configs; training them is configs/sdxl/sdxl-te.example.yaml, not built).
Tokenisation needs the CLIP vocabulary files: they are looked up locally through transformers' CLIPTokenizer when text is
passed; token ids (LongTensor [B, 77]) are accepted directly.
"""
from __future__ import annotations

import logging
from typing import Optional, Sequence, Union

import numpy as np
import torch
from torch import Tensor, nn

from ... import ops
from ...graphs import ForwardGraphs, frozen_stamp, graphs_enabled
from ...modules.encoders.embedding import AbstractEmbModel

logger = logging.getLogger(__name__)
LN_EPS = 1e-5


# ---------------------------------------------------------------------------------------------------
# the transformer both towers share
# ---------------------------------------------------------------------------------------------------
def _linear(x: Tensor, layer: nn.Linear, residual: Optional[Tensor] = None) -> Tensor:
    return ops.gemm_nt(x, ops.w2d(layer.weight), layer.bias, residual=residual)


def _norm(x: Tensor, layer: nn.LayerNorm) -> Tensor:
    return ops.layernorm_fwd(x, layer.weight, layer.bias, layer.eps)[0]


def _residual_block(x: Tensor, batch: int, heads: int, ln_1, qkv, out_proj, ln_2, fc_in, fc_out, quick_gelu: bool) -> Tensor:
    """x + attn(ln_1 x), then x + mlp(ln_2 x), on a dense bf16 token matrix [batch * L, width].  `qkv` maps the normed
    tokens to the (q, k, v) token matrices (column slices of one fused projection where the checkpoint stores it fused)."""
    width = x.shape[1]
    q, k, v = qkv(_norm(x, ln_1))
    attended = ops.attention_fwd(q, k, v, batch, heads, width // heads, causal=True)[0]
    x = _linear(attended, out_proj, residual=x)
    hidden = ops.gelu(_linear(_norm(x, ln_2), fc_in), quick=quick_gelu)
    return _linear(hidden, fc_out, residual=x)


def _embed(ids: Tensor, table: Tensor, positions: Tensor) -> Tensor:
    """token + position embeddings as bf16 tokens [B * L, width]"""
    if ids.dim() != 2 or ids.shape[1] > positions.shape[0]:
        raise ValueError(f"token ids must be [batch, <= {positions.shape[0]}], got {tuple(ids.shape)}")
    summed = table[ids] + positions[: ids.shape[1]]
    return ops.cast_bf16(summed.reshape(-1, summed.shape[-1]).float())


def _graphable(tower: nn.Module, ids: Tensor) -> bool:
    """A frozen tower on the GPU runs its ~250-450 dependent small launches from a hipGraph (neurosis_amd/graphs.py): they are
    launch-latency, not work (4-5 ms of the SDXL step for both towers)."""
    return ids.is_cuda and graphs_enabled("te") and not any(p.requires_grad for p in tower.parameters())


def _forward_graphs(tower: nn.Module) -> ForwardGraphs:
    fg = tower.__dict__.get("_nk_fgraphs")
    if fg is None:
        fg = tower.__dict__["_nk_fgraphs"] = ForwardGraphs(next(tower.parameters()).device)
    return fg


def _check_ids(ids: Tensor, device) -> Tensor:
    if ids.dtype not in (torch.int64, torch.int32):
        raise TypeError(f"token ids must be an integer tensor, got {ids.dtype}")
    return ids.to(device=device, dtype=torch.int64)


# ---------------------------------------------------------------------------------------------------
# HuggingFace layout (CLIPTextModel)
# ---------------------------------------------------------------------------------------------------
class _HFAttention(nn.Module):
    def __init__(self, width: int):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(width, width) for _ in range(4))
        self._packed = None

    def qkv(self, h: Tensor):
        """q, k, v of the normed tokens from ONE GEMM: the three projections packed [q; k; v] (bf16 weight, fp32 bias), rebuilt
        only when one of the six tensors changes -- 24 launches fewer per CLIP-L forward than three GEMMs per layer"""
        parts = (self.q_proj, self.k_proj, self.v_proj)
        stamp = tuple(ops._param_stamp(t) for lin in parts for t in (lin.weight, lin.bias))
        if self._packed is None or self._packed[0] != stamp:
            weight = torch.cat([ops.w2d(lin.weight) for lin in parts], dim=0).contiguous()
            bias = torch.cat([lin.bias.detach().float() for lin in parts]).contiguous()
            self._packed = (stamp, weight, bias)
        width = self.q_proj.weight.shape[0]
        fused = ops.gemm_nt(h, self._packed[1], self._packed[2])
        return fused[:, :width], fused[:, width:2 * width], fused[:, 2 * width:]


class _HFMLP(nn.Module):
    def __init__(self, width: int, inner: int):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(width, inner), nn.Linear(inner, width)


class _HFLayer(nn.Module):
    def __init__(self, width: int, inner: int):
        super().__init__()
        self.self_attn = _HFAttention(width)
        self.layer_norm1 = nn.LayerNorm(width, eps=LN_EPS)
        self.mlp = _HFMLP(width, inner)
        self.layer_norm2 = nn.LayerNorm(width, eps=LN_EPS)


class CLIPTextTower(nn.Module):
    """transformers' CLIPTextModel for the text side of openai/clip-vit-large-patch14 (defaults) -- same state_dict."""

    def __init__(self, vocab_size: int = 49408, hidden_size: int = 768, intermediate_size: int = 3072, num_hidden_layers: int = 12,
                 num_attention_heads: int = 12, max_position_embeddings: int = 77, hidden_act: str = "quick_gelu", eos_token_id: int = 2, **unused):
        super().__init__()
        if hidden_act not in ("quick_gelu", "gelu"):
            raise ValueError(f"hidden_act {hidden_act!r}: the CLIP text models use 'quick_gelu' or 'gelu'")
        self.heads, self.quick_gelu, self.eos_token_id = num_attention_heads, hidden_act == "quick_gelu", eos_token_id
        tm = self.text_model = nn.Module()
        tm.embeddings = nn.Module()
        tm.embeddings.token_embedding = nn.Embedding(vocab_size, hidden_size)
        tm.embeddings.position_embedding = nn.Embedding(max_position_embeddings, hidden_size)
        tm.encoder = nn.Module()
        tm.encoder.layers = nn.ModuleList(_HFLayer(hidden_size, intermediate_size) for _ in range(num_hidden_layers))
        tm.final_layer_norm = nn.LayerNorm(hidden_size, eps=LN_EPS)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # transformers 5.x dropped the "text_model." level from CLIPTextModel's parameter names
        inner = prefix + "text_model."
        if not any(k.startswith(inner) for k in state_dict):
            for key in [k for k in state_dict if k.startswith(prefix)]:
                state_dict[inner + key[len(prefix):]] = state_dict.pop(key)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @torch.no_grad()
    def forward(self, input_ids: Tensor, output_hidden_states: bool = False) -> dict:
        """{"last_hidden_state" [B, L, C], "pooler_output" [B, C], "hidden_states": tuple of L+1 [B, L, C] or None}, fp32;
        what CLIPTextModel returns (hidden_states[0] = embeddings, [i] = output of layer i, none of them final-normed)."""
        tm = self.text_model
        ids = _check_ids(input_ids, tm.embeddings.token_embedding.weight.device)
        if _graphable(self, ids):
            return _forward_graphs(self).run(lambda t: self._forward_ids(t, output_hidden_states), [ids], extra_key=(output_hidden_states, frozen_stamp(self)))
        return self._forward_ids(ids, output_hidden_states)

    def _forward_ids(self, ids: Tensor, output_hidden_states: bool) -> dict:
        tm = self.text_model
        B, L = ids.shape
        x = _embed(ids, tm.embeddings.token_embedding.weight, tm.embeddings.position_embedding.weight)
        states = [x] if output_hidden_states else None
        for layer in tm.encoder.layers:
            attn = layer.self_attn
            x = _residual_block(x, B, self.heads, layer.layer_norm1, attn.qkv, attn.out_proj, layer.layer_norm2, layer.mlp.fc1, layer.mlp.fc2, self.quick_gelu)
            if states is not None:
                states.append(x)
        last = _norm(x, tm.final_layer_norm).float().reshape(B, L, -1)
        # the pooled vector is the end-of-text position: the highest id in the legacy vocabulary (eos_token_id == 2 configs),
        # else the first occurrence of eos_token_id
        eos = ids.argmax(-1) if self.eos_token_id == 2 else (ids == self.eos_token_id).int().argmax(-1)
        return {"last_hidden_state": last, "pooler_output": last[torch.arange(B, device=last.device), eos],
                "hidden_states": None if states is None else tuple(s.float().reshape(B, L, -1) for s in states)}


# ---------------------------------------------------------------------------------------------------
# open_clip layout (the text half of open_clip.CLIP)
# ---------------------------------------------------------------------------------------------------
class _PackedAttention(nn.Module):
    """parameter names of nn.MultiheadAttention (what open_clip's ResidualAttentionBlock holds)"""

    def __init__(self, width: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)
        nn.init.normal_(self.in_proj_weight, std=width**-0.5)


class _OpenCLIPMLP(nn.Module):
    def __init__(self, width: int, inner: int):
        super().__init__()
        self.c_fc, self.c_proj = nn.Linear(width, inner), nn.Linear(inner, width)


class _ResBlock(nn.Module):
    def __init__(self, width: int, inner: int):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width, eps=LN_EPS)
        self.attn = _PackedAttention(width)
        self.ln_2 = nn.LayerNorm(width, eps=LN_EPS)
        self.mlp = _OpenCLIPMLP(width, inner)


class OpenCLIPTextTower(nn.Module):
    """The text tower of open_clip's CLIP (defaults: ViT-bigG-14 -- width 1280, 32 layers, 20 heads, projection to 1280)."""

    def __init__(self, vocab_size: int = 49408, width: int = 1280, layers: int = 32, heads: int = 20, context_length: int = 77, embed_dim: int = 1280,
                 mlp_ratio: float = 4.0, quick_gelu: bool = False):
        super().__init__()
        self.heads, self.quick_gelu, self.context_length = heads, quick_gelu, context_length
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width).normal_(std=0.01))
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList(_ResBlock(width, int(width * mlp_ratio)) for _ in range(layers))
        self.transformer.grad_checkpointing = False
        self.ln_final = nn.LayerNorm(width, eps=LN_EPS)
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim).normal_(std=width**-0.5))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self._projection_cache = None

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        state_dict.pop(prefix + "attn_mask", None)          # a buffer in open_clip; the causal mask is a kernel flag here
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def _projection_t(self) -> Tensor:
        """text_projection^T as the [embed_dim, width] bf16 matrix gemm_nt reads"""
        cached = self._projection_cache
        if cached is None or cached[0] != (ops.state.param_epoch, self.text_projection.data_ptr(), self.text_projection._version):
            w = ops.cast_bf16(self.text_projection.detach().t().contiguous().float())
            cached = self._projection_cache = ((ops.state.param_epoch, self.text_projection.data_ptr(), self.text_projection._version), w)
        return cached[1]

    @torch.no_grad()
    def forward(self, text: Tensor) -> dict:
        """{"last", "penultimate": [B, L, width] (neither final-normed), "pooled": [B, embed_dim]}, fp32 -- the dictionary the
        reference's encode_with_transformer builds (models/text_encoder/clip.py:311-343)."""
        ids = _check_ids(text, self.token_embedding.weight.device)
        if _graphable(self, ids):
            return _forward_graphs(self).run(self._forward_ids, [ids], extra_key=(frozen_stamp(self),))
        return self._forward_ids(ids)

    def _forward_ids(self, ids: Tensor) -> dict:
        B, L = ids.shape
        x = _embed(ids, self.token_embedding.weight, self.positional_embedding)
        width = x.shape[1]
        penultimate = x
        for block in self.transformer.resblocks:
            penultimate = x
            attn = block.attn

            def packed_qkv(h, a=attn):
                fused = ops.gemm_nt(h, ops.w2d(a.in_proj_weight), a.in_proj_bias)
                return fused[:, :width], fused[:, width:2 * width], fused[:, 2 * width:]

            x = _residual_block(x, B, self.heads, block.ln_1, packed_qkv, attn.out_proj, block.ln_2, block.mlp.c_fc, block.mlp.c_proj, self.quick_gelu)
        normed = _norm(x, self.ln_final).reshape(B, L, width)
        eot = normed[torch.arange(B, device=ids.device), ids.argmax(dim=-1)].contiguous()       # highest id = end of text
        pooled = ops.gemm_nt(eot, self._projection_t())
        return {"last": x.float().reshape(B, L, width), "penultimate": penultimate.float().reshape(B, L, width), "pooled": pooled.float()}


# ---------------------------------------------------------------------------------------------------
# embedders
# ---------------------------------------------------------------------------------------------------
def _decode_text(text) -> list:
    if isinstance(text, (str, bytes, np.bytes_)):
        text = [text]
    return [t.decode("utf-8") if isinstance(t, (bytes, np.bytes_)) else t for t in text]


class _TokenizingEmbedder(AbstractEmbModel):
    """tokenisation shared by both embedders (reference :156-202 and :348-388 are the same code twice)"""

    tokenizer = None
    max_length = 77
    extended_chunks = 0

    def _load_tokenizer(self, repo: str) -> None:
        try:
            from transformers import CLIPTokenizer

            self.tokenizer = CLIPTokenizer.from_pretrained(repo, local_files_only=True)
        except Exception as err:   # no vocabulary on this machine: token ids can still be passed to forward()
            logger.warning("CLIP tokenizer %s is not available locally (%s); pass token ids instead of text", repo, type(err).__name__)
            self.tokenizer = None

    def _tokenizer_or_raise(self):
        if self.tokenizer is None:
            raise RuntimeError("this embedder has no tokenizer (CLIP vocabulary files not found locally): pass LongTensor token ids [B, 77]")
        return self.tokenizer

    def tokenize(self, text: Sequence[str]) -> dict:
        enc = self._tokenizer_or_raise()(text, truncation=True, max_length=self.max_length, return_length=True, return_overflowing_tokens=False,
                                         padding="max_length", return_tensors="pt")
        return {"input_ids": enc["input_ids"].to(self.device)}

    def tokenize_extended(self, text: Sequence[str]) -> dict:
        """[B, chunks, 77]: the prompt is cut into `extended_chunks` pieces of 75 tokens, each wrapped in BOS / EOS"""
        tok = self._tokenizer_or_raise()
        body = tok.model_max_length - 2
        ids = tok(text, truncation=True, add_special_tokens=False, max_length=self.extended_chunks * body, padding="max_length",
                  return_tensors="pt")["input_ids"].to(self.device)
        ids = ids.view(len(text), self.extended_chunks, body)
        edge = ids.new_ones(ids.shape[:2] + (1,))
        return {"input_ids": torch.cat((edge * tok.bos_token_id, ids, edge * tok.eos_token_id), dim=2)}

    def _ids_for(self, text) -> Tensor:
        """token ids for forward(): [B, 77], or [B, chunks, 77] in extended mode; tensors pass through"""
        if torch.is_tensor(text):
            return text.to(self.device)
        text = _decode_text(text)
        if self.ucg_rate > 0.0 and self.ucg_rate < np.random.rand():       # (the reference's comparison, kept as is)
            text = [""] * len(text)
        return (self.tokenize_extended(text) if self.extended_chunks > 1 else self.tokenize(text))["input_ids"]

    def encode(self, text):
        return self(text)


class FrozenCLIPEmbedder(_TokenizingEmbedder):
    """reference :22-202.  `config` (a dict of CLIPTextTower arguments) replaces the hub lookup of `version`'s config when given."""

    LAYERS = ["last", "pooled", "hidden", "penultimate"]

    def __init__(self, version: str = "openai/clip-vit-large-patch14", device="cuda", max_length: int = 77, freeze: bool = True, layer: str = "last",
                 layer_idx: Optional[int] = None, always_return_pooled: bool = False, extended_chunks: int = 0, load_pretrained: bool = False,
                 config: Optional[dict] = None, **kwargs):
        super().__init__(**kwargs)
        if layer not in self.LAYERS:
            raise ValueError(f"layer must be one of {self.LAYERS}, got {layer=}")
        if load_pretrained:
            raise NotImplementedError("load_pretrained: there is no hub access here; load a state_dict into .transformer instead")
        self.transformer = CLIPTextTower(**(config or {}))
        self._load_tokenizer(version)
        self.device, self.max_length = torch.device(device), max_length
        self.layer, self.return_pooled, self.extended_chunks = layer, always_return_pooled, extended_chunks
        self.output_hidden_states = layer in ("hidden", "penultimate")
        depth = len(self.transformer.text_model.encoder.layers)
        if layer == "hidden":
            if layer_idx is None:
                raise ValueError("layer_idx must be specified for hidden layer")
            if not (0 <= abs(layer_idx) <= depth):
                raise ValueError(f"layer_idx must be between -{depth} and {depth}")
            self.layer_idx = layer_idx + depth if layer_idx < 0 else layer_idx
        elif layer == "penultimate":
            self.layer_idx = depth - 2
        else:
            # the reference raises here for "last" and "pooled" (its match statement has no such arms) although its forward
            # handles both; they are accepted
            self.layer_idx = None
        if not self.is_trainable:
            self.freeze()

    def _select(self, out: dict) -> Tensor:
        if self.layer == "last":
            return out["last_hidden_state"]
        if self.layer == "pooled":
            return out["pooler_output"][:, None, :]
        return out["hidden_states"][self.layer_idx + 1]

    @torch.no_grad()
    def forward(self, text: Union[str, list, Tensor]):
        ids = self._ids_for(text)
        if ids.dim() == 2:
            out = self.transformer(ids, output_hidden_states=self.output_hidden_states)
            z = self._select(out)
            return (z, out["pooler_output"]) if self.return_pooled else z
        # extended mode: each prompt's chunks run as one mini-batch and are laid end to end along the token axis
        per_prompt, pooled = [], []
        for chunks in ids:
            out = self.transformer(chunks, output_hidden_states=self.output_hidden_states)
            per_prompt.append(self._select(out).reshape(-1, out["last_hidden_state"].shape[-1]))
            pooled.append(out["pooler_output"][0])
        z = torch.stack(per_prompt, dim=0)
        return (z, pooled) if self.return_pooled else z


OPENCLIP_TOKENIZERS = {"default": "laion/CLIP-ViT-bigG-14-laion2B-39B-b160k", "ViT-bigG-14": "laion/CLIP-ViT-bigG-14-laion2B-39B-b160k"}
OPENCLIP_ARCHS = {"ViT-bigG-14": dict(width=1280, layers=32, heads=20, embed_dim=1280), "ViT-H-14": dict(width=1024, layers=24, heads=16, embed_dim=1024)}


class FrozenOpenCLIPEmbedder2(_TokenizingEmbedder):
    """reference :205-388.  `arch` selects the text tower's shape from OPENCLIP_ARCHS, or pass `config` (OpenCLIPTextTower arguments)."""

    LAYERS = ["pooled", "last", "penultimate"]

    def __init__(self, arch: str = "ViT-bigG-14", version: Optional[str] = "laion2b_s39b_b160k", device="cuda", max_length: int = 77, layer: str = "last",
                 always_return_pooled: bool = False, legacy: bool = False, extended_chunks: int = 0, config: Optional[dict] = None, freeze: bool = True, **kwargs):
        super().__init__(**kwargs)
        if layer not in self.LAYERS:
            raise ValueError(f"layer must be one of {self.LAYERS}, got {layer=}")
        if always_return_pooled and legacy:
            raise ValueError("legacy mode does not support returning pooled embeddings!")
        if extended_chunks > 1 and legacy:
            raise ValueError("legacy mode does not support extended chunks!")
        if config is None and arch not in OPENCLIP_ARCHS:
            raise ValueError(f"unknown arch {arch!r}: pass config= or one of {sorted(OPENCLIP_ARCHS)}")
        self.model = OpenCLIPTextTower(**(config if config is not None else OPENCLIP_ARCHS[arch]))
        repo = OPENCLIP_TOKENIZERS.get(arch)
        if repo is None:
            logger.warning(f"Could not find tokenizer for {arch=} and {version=}, using default")
            repo = OPENCLIP_TOKENIZERS["default"]
        self._load_tokenizer(repo)
        self.device, self.max_length = torch.device(device), max_length
        self.layer, self.return_pooled, self.legacy, self.extended_chunks = layer, always_return_pooled, legacy, extended_chunks
        self.embed_dim = self.model.text_projection.shape[-1]
        if not self.is_trainable:
            self.freeze()

    def encode_with_transformer(self, text: Tensor):
        out = self.model(text)
        if not self.legacy:
            return out
        # legacy: the chosen layer's tokens through ln_final, nothing else
        chosen = out[self.layer]
        B, L, width = chosen.shape
        return _norm(ops.cast_bf16(chosen.reshape(B * L, width)), self.model.ln_final).float().reshape(B, L, width)

    @torch.no_grad()
    def forward(self, text: Union[str, list, Tensor]):
        ids = self._ids_for(text)
        if ids.dim() == 2:
            out = self.encode_with_transformer(ids)
            if self.legacy:
                return out
            return (out[self.layer], out["pooled"]) if self.return_pooled else out[self.layer]
        per_prompt, pooled = [], []
        for chunks in ids:
            out = self.encode_with_transformer(chunks)
            per_prompt.append(out[self.layer].reshape(-1, out[self.layer].shape[-1]))
            pooled.append(out["pooled"][0:1])
        z = torch.stack(per_prompt, dim=0)
        return (z, torch.cat(pooled, dim=0)) if self.return_pooled else z
