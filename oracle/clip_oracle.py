"""CPU oracle for the frozen CLIP text encoders -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The reference delegates both transformers to third-party packages, so this restates THEIR published algorithms and the
reference's call sites around them (paths relative to /root/reference/src/neurosis/):

* `hf_text_model`   -- transformers `CLIPTextModel` (reference requires transformers >= 4.36.0; 5.15.0 is installed in this
  image and produced the fixtures): embeddings = token + position; per layer  x += attn(layer_norm1 x) with a causal mask,
  x += fc2(act(fc1(layer_norm2 x))), act = quick_gelu x*sigmoid(1.702x) for openai/clip-vit-large-patch14; final_layer_norm
  gives last_hidden_state; pooler_output = the row of the highest token id (eos_token_id == 2 legacy rule); hidden_states =
  [embeddings, layer outputs...] without the final norm.  Call site: FrozenCLIPEmbedder.forward, models/text_encoder/clip.py:88-151.
* `openclip_text_tower` -- open_clip (`open-clip-torch >= 2.2.0`, NOT installed here) `CLIP` text half: token_embedding +
  positional_embedding; ResidualAttentionBlock = ln_1 -> nn.MultiheadAttention (packed in_proj, additive -inf causal mask)
  -> residual; ln_2 -> c_fc -> GELU(erf) -> c_proj -> residual; and the reference's own encode_with_transformer /
  text_transformer_forward / pool (clip.py:311-343): "penultimate" = input of the last block, "last" = its output,
  "pooled" = ln_final(last)[argmax id] @ text_projection.

Pinned by tests/golden/text_encoders_tiny.pt: the reference's FrozenCLIPEmbedder.forward over the real CLIPTextModel, and its
FrozenOpenCLIPEmbedder2.forward over a stand-in of open_clip's tower built from torch.nn.MultiheadAttention (the package
being absent, that half is anchored on the reference's call sites and PyTorch's MultiheadAttention, not on open_clip itself).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor

SD = dict


def _causal_attention(q: Tensor, k: Tensor, v: Tensor, heads: int) -> Tensor:
    B, L, C = q.shape
    d = C // heads
    q, k, v = (t.reshape(B, L, heads, d).transpose(1, 2) for t in (q, k, v))
    scores = (q @ k.transpose(-1, -2)) * d ** -0.5
    scores = scores + torch.full((L, L), float("-inf")).triu(1)
    return (scores.softmax(-1) @ v).transpose(1, 2).reshape(B, L, C)


def _lin(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def _ln(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def hf_text_model(sd: SD, cfg: dict, ids: Tensor) -> dict:
    """sd: CLIPTextModel state_dict with or without the 4.x "text_model." prefix."""
    if any(k.startswith("text_model.") for k in sd):
        sd = {k[len("text_model."):]: v for k, v in sd.items()}
    heads = cfg["num_attention_heads"]
    quick = cfg.get("hidden_act", "quick_gelu") == "quick_gelu"
    x = sd["embeddings.token_embedding.weight"][ids] + sd["embeddings.position_embedding.weight"][: ids.shape[1]]
    states = [x]
    for i in range(cfg["num_hidden_layers"]):
        p = f"encoder.layers.{i}"
        h = _ln(sd, p + ".layer_norm1", x)
        a = _causal_attention(_lin(sd, p + ".self_attn.q_proj", h), _lin(sd, p + ".self_attn.k_proj", h), _lin(sd, p + ".self_attn.v_proj", h), heads)
        x = x + _lin(sd, p + ".self_attn.out_proj", a)
        u = _lin(sd, p + ".mlp.fc1", _ln(sd, p + ".layer_norm2", x))
        u = u * torch.sigmoid(1.702 * u) if quick else F.gelu(u)
        x = x + _lin(sd, p + ".mlp.fc2", u)
        states.append(x)
    last = _ln(sd, "final_layer_norm", x)
    rows = torch.arange(ids.shape[0])
    eos = ids.argmax(-1) if cfg.get("eos_token_id", 2) == 2 else (ids == cfg["eos_token_id"]).int().argmax(-1)
    return {"last_hidden_state": last, "pooler_output": last[rows, eos], "hidden_states": states}


def frozen_clip_embedder(sd: SD, cfg: dict, ids: Tensor, layer: str, layer_idx, return_pooled: bool):
    """FrozenCLIPEmbedder.forward, standard (non-extended) mode, clip.py:126-151"""
    out = hf_text_model(sd, cfg, ids)
    if layer == "last":
        z = out["last_hidden_state"]
    elif layer == "pooled":
        z = out["pooler_output"][:, None, :]
    else:
        z = out["hidden_states"][layer_idx + 1]
    return (z, out["pooler_output"]) if return_pooled else z


def openclip_text_tower(sd: SD, cfg: dict, ids: Tensor) -> dict:
    heads, width = cfg["heads"], cfg["width"]
    x = sd["token_embedding.weight"][ids] + sd["positional_embedding"]
    penultimate = x
    for i in range(cfg["layers"]):
        p = f"transformer.resblocks.{i}"
        penultimate = x
        h = _ln(sd, p + ".ln_1", x)
        q, k, v = F.linear(h, sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"]).split(width, dim=-1)
        x = x + _lin(sd, p + ".attn.out_proj", _causal_attention(q, k, v, heads))
        x = x + _lin(sd, p + ".mlp.c_proj", F.gelu(_lin(sd, p + ".mlp.c_fc", _ln(sd, p + ".ln_2", x))))
    normed = _ln(sd, "ln_final", x)
    pooled = normed[torch.arange(ids.shape[0]), ids.argmax(dim=-1)] @ sd["text_projection"]
    return {"last": x, "penultimate": penultimate, "pooled": pooled}


def frozen_openclip_embedder2(sd: SD, cfg: dict, ids: Tensor, layer: str, return_pooled: bool, legacy: bool):
    """FrozenOpenCLIPEmbedder2.forward, standard mode, clip.py:297-327"""
    out = openclip_text_tower(sd, cfg, ids)
    if legacy:
        return _ln(sd, "ln_final", out[layer])
    return (out[layer], out["pooled"]) if return_pooled else out[layer]
