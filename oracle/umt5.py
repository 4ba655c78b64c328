"""ORACLE (test infrastructure).  CPU fp32 restatement of the UMT5 text *encoder* that turns a prompt into the
`encoder_hidden_states` of the DiT: `WanPipeline.encode_prompt` -> `_get_t5_prompt_embeds`
(diffusers==0.33.1, called inside `pipe(prompt=..., negative_prompt=...)` at /root/reference/inference_t23d.py:94-103) and the
in-tree twin /root/reference/utils/wan_utils.py:25-60 (`compute_wan_text_embeddings`): tokenizer(max_length=512, padded) ->
`UMT5EncoderModel(input_ids, attention_mask).last_hidden_state` -> rows past the prompt length replaced by zeros.

The model itself lives in the third-party wheel `transformers` (reference pin: transformers==4.46.3, requirements.txt:21; absent
from /root/reference).  PARITY PINNED against the copy of that dependency present in this image (transformers 5.15.0 — same
UMT5 architecture): tests/golden/umt5_tiny.safetensors is the output of `transformers.UMT5EncoderModel` itself on weights from
`make_weights` below (generator: tests/golden/make_golden.py::umt5_tiny).  Tokenisation needs the sentencepiece vocabulary of
google/umt5-xxl (a checkpoint asset, not reachable offline) and is therefore outside the oracle: it starts from token ids.

Architecture restated (modeling_umt5.py): T5LayerNorm (RMS, no bias) -> self-attention with q,k,v,o bias-free, NO 1/sqrt(d)
scaling, additive relative-position bias from a per-LAYER 32-bucket table (UMT5 gives every block its own table; T5 shares the
first), padding keys masked -> residual; T5LayerNorm -> gated GELU(tanh) FFN wo(gelu_new(wi_0 x) * wi_1 x) -> residual; final
T5LayerNorm."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F


@dataclass
class UMT5Config:
    vocab_size: int = 256384
    d_model: int = 4096
    d_kv: int = 64
    d_ff: int = 10240
    num_layers: int = 24
    num_heads: int = 64
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6


def relative_position_bucket(rel: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """Bidirectional T5 bucketing of rel = key_pos - query_pos: half the buckets per sign, exact up to 8, then log-spaced."""
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(small, rel, large)


def position_bias(table: torch.Tensor, L: int, cfg: UMT5Config) -> torch.Tensor:
    """table [num_buckets, H] -> bias [H, L(query), L(key)]."""
    pos = torch.arange(L)
    rel = pos[None, :] - pos[:, None]
    b = relative_position_bucket(rel, cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance)
    return table[b].permute(2, 0, 1)


def rms_norm(x, w, eps):
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def encode(sd: Dict[str, torch.Tensor], cfg: UMT5Config, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """-> last_hidden_state [B, L, d_model] (all L positions, like the HF module; padding rows are not meaningful)."""
    sd = {k: v.float() for k, v in sd.items()}
    B, L = input_ids.shape
    H, dk = cfg.num_heads, cfg.d_kv
    h = sd["shared.weight"][input_ids]
    neg = (1.0 - attention_mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}.layer."
        a = p + "0.SelfAttention."
        n = rms_norm(h, sd[p + "0.layer_norm.weight"], cfg.layer_norm_epsilon)
        q = (n @ sd[a + "q.weight"].T).view(B, L, H, dk).transpose(1, 2)
        k = (n @ sd[a + "k.weight"].T).view(B, L, H, dk).transpose(1, 2)
        v = (n @ sd[a + "v.weight"].T).view(B, L, H, dk).transpose(1, 2)
        scores = q @ k.transpose(-1, -2) + position_bias(sd[a + "relative_attention_bias.weight"], L, cfg)[None] + neg
        o = (F.softmax(scores, dim=-1) @ v).transpose(1, 2).reshape(B, L, H * dk)
        h = h + o @ sd[a + "o.weight"].T
        f = p + "1.DenseReluDense."
        n = rms_norm(h, sd[p + "1.layer_norm.weight"], cfg.layer_norm_epsilon)
        h = h + (gelu_new(n @ sd[f + "wi_0.weight"].T) * (n @ sd[f + "wi_1.weight"].T)) @ sd[f + "wo.weight"].T
    return rms_norm(h, sd["encoder.final_layer_norm.weight"], cfg.layer_norm_epsilon)


def prompt_embeds(sd, cfg, input_ids, attention_mask, max_sequence_length: int) -> torch.Tensor:
    """_get_t5_prompt_embeds / compute_wan_text_embeddings tail: keep the first seq_len rows, zero rows up to max length."""
    hs = encode(sd, cfg, input_ids, attention_mask)
    lens = attention_mask.gt(0).sum(dim=1).long()
    return torch.stack([torch.cat([u[:v], u.new_zeros(max_sequence_length - int(v), u.size(1))]) for u, v in zip(hs, lens)], 0)


def make_weights(cfg: UMT5Config, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded weights under the HF UMT5EncoderModel state-dict names."""
    g = torch.Generator().manual_seed(seed)
    d, inner = cfg.d_model, cfg.num_heads * cfg.d_kv
    sd = {"shared.weight": torch.randn(cfg.vocab_size, d, generator=g)}
    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}.layer."
        a = p + "0.SelfAttention."
        for n in ("q", "k", "v"):
            sd[a + n + ".weight"] = torch.randn(inner, d, generator=g) * (d ** -0.5) * (0.6 if n != "v" else 1.0)
        sd[a + "o.weight"] = torch.randn(d, inner, generator=g) * (inner ** -0.5)
        sd[a + "relative_attention_bias.weight"] = torch.randn(cfg.relative_attention_num_buckets, cfg.num_heads, generator=g)
        sd[p + "0.layer_norm.weight"] = 1 + 0.1 * torch.randn(d, generator=g)
        f = p + "1.DenseReluDense."
        sd[f + "wi_0.weight"] = torch.randn(cfg.d_ff, d, generator=g) * (d ** -0.5)
        sd[f + "wi_1.weight"] = torch.randn(cfg.d_ff, d, generator=g) * (d ** -0.5)
        sd[f + "wo.weight"] = torch.randn(d, cfg.d_ff, generator=g) * (cfg.d_ff ** -0.5)
        sd[p + "1.layer_norm.weight"] = 1 + 0.1 * torch.randn(d, generator=g)
    sd["encoder.final_layer_norm.weight"] = 1 + 0.1 * torch.randn(d, generator=g)
    return sd
