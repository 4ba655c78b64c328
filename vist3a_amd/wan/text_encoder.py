"""UMT5 text encoder on MI355X — the `text_encoder` half of diffusers' `WanPipeline.encode_prompt`
(`_get_t5_prompt_embeds`, driven by /root/reference/inference_t23d.py:94-103) and of the in-tree twin
/root/reference/utils/wan_utils.py:25-60 (`compute_wan_text_embeddings`).

    ids, mask  --UMT5EncoderModel-->  last_hidden_state  --keep seq_len rows, zero the rest-->  prompt_embeds [B, 512, 4096]

MI355X-first choices (same arithmetic per valid token as the HF module):
  * only the VALID tokens are computed: padding keys are masked out of every softmax and padding rows are overwritten with
    zeros afterwards, so a 40-token prompt runs M = 40 rows through the 24 blocks instead of 512 — the encoder becomes a pure
    weight-streaming problem (11.4 GB of bf16 weights per prompt);
  * per block: RMS norm (fp32 residual stream -> bf16) -> fused [q;k] GEMM + V^T GEMM -> the flash kernel's RELB variant, which
    adds the relative-position bias from a per-layer [H, 2*512-1] table (bucket lookup done once at load) and uses scale 1.0
    (T5 does not scale by 1/sqrt(d)) -> o-proj with the fp32 residual in the epilogue -> RMS norm -> wi_0 (GELU-tanh fused) and
    wi_1 GEMMs -> product -> wo with the residual in the epilogue;
  * tokenisation stays on the host and outside this module (the sentencepiece vocabulary is a checkpoint asset): any HF-style
    tokenizer callable can be passed to `compute_wan_text_embeddings`."""
from __future__ import annotations

import html
import math
import re
from dataclasses import dataclass
from typing import Dict, List, Optional, Union

import torch

from .. import lib as L
from .. import ops

bf16, f32 = torch.bfloat16, torch.float32


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
    max_positions: int = 512


def _bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < max_exact, rel, large)


class UMT5TextEncoder:
    """`text_encoder(input_ids, attention_mask).last_hidden_state` for right-padded prompts (forward only)."""

    def __init__(self, cfg: UMT5Config, state_dict: Dict[str, torch.Tensor], device="cuda"):
        L.load()
        if cfg.d_kv != 64:
            raise NotImplementedError("the relative-bias attention kernel is built for d_kv = 64 (every UMT5 size)")
        self.cfg, self.device, self.dtype = cfg, torch.device(device), bf16
        sd, dev = state_dict, self.device
        W = lambda k: sd[k].to(device=dev, dtype=bf16).contiguous()
        Fv = lambda k: sd[k].to(device=dev, dtype=f32).contiguous()
        self.embed = W("shared.weight")
        Lm = cfg.max_positions
        rel = torch.arange(-(Lm - 1), Lm)  # key - query
        bucket = _bucket(rel, cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance)
        self.blocks = []
        for i in range(cfg.num_layers):
            p = f"encoder.block.{i}.layer."
            a = p + "0.SelfAttention."
            f = p + "1.DenseReluDense."
            tab = sd[a + "relative_attention_bias.weight"].float().cpu()  # [buckets, H]
            self.blocks.append(dict(
                ln1=Fv(p + "0.layer_norm.weight"), ln2=Fv(p + "1.layer_norm.weight"),
                wqk=torch.cat([W(a + "q.weight"), W(a + "k.weight")], 0).contiguous(), wv=W(a + "v.weight"), wo=W(a + "o.weight"),
                relb=tab[bucket].t().contiguous().to(dev),  # [H, 2*Lm-1], entry (key - query + Lm - 1)
                wi0=W(f + "wi_0.weight"), wi1=W(f + "wi_1.weight"), wo2=W(f + "wo.weight")))
        self.final_ln = Fv("encoder.final_layer_norm.weight")

    @torch.no_grad()
    def encode_valid(self, ids: torch.Tensor) -> torch.Tensor:
        """ids [n] (the n valid tokens of ONE prompt) -> hidden states [n, d_model] fp32."""
        cfg = self.cfg
        n = ids.numel()
        if n < 1 or n > cfg.max_positions:
            raise ValueError(f"prompt length {n} outside [1, {cfg.max_positions}]")
        d, H, dk, eps = cfg.d_model, cfg.num_heads, cfg.d_kv, cfg.layer_norm_epsilon
        inner = H * dk
        Mp = (n + 7) // 8 * 8   # GEMM row granularity; the filler rows are never attended to (Nk = n) and are dropped at the end
        dev = self.device
        idp = torch.zeros(Mp, dtype=torch.long, device=dev)
        idp[:n] = ids.to(dev)
        h = self.embed[idp].float()  # fp32 residual stream (fp16 embeddings + bf16 autocast outputs promote to fp32 in the reference)
        nb = torch.empty(Mp, d, device=dev, dtype=bf16)
        qk = torch.empty(Mp, 2 * inner, device=dev, dtype=bf16)
        vt = torch.zeros(inner, (Mp + 63) // 64 * 64, device=dev, dtype=bf16)
        ao = torch.empty(Mp, inner, device=dev, dtype=bf16)
        g0 = torch.empty(Mp, cfg.d_ff, device=dev, dtype=bf16)
        g1 = torch.empty(Mp, cfg.d_ff, device=dev, dtype=bf16)
        center = cfg.max_positions - 1
        for b in self.blocks:
            ops.layernorm(h, out=nb, weight=b["ln1"], eps=eps, rms=True)
            ops.gemm(nb, b["wqk"], out=qk)
            ops.gemm(b["wv"], nb, out=vt[:, :Mp])
            ops.attention(qk[:, :inner], qk[:, inner:], vt, ao, B=1, H=H, Nq=Mp, Nk=n, D=dk, q_batch_stride=0, k_batch_stride=0,
                          vt_batch_stride=0, o_batch_stride=0, scale=1.0, rel_bias=b["relb"], rel_bias_center=center)
            ops.gemm(ao, b["wo"], out=h, residual=h, out_f32=True)
            ops.layernorm(h, out=nb, weight=b["ln2"], eps=eps, rms=True)
            ops.gemm(nb, b["wi0"], out=g0, act=L.ACT_GELU_TANH)
            ops.gemm(nb, b["wi1"], out=g1)
            g0.mul_(g1)
            ops.gemm(g0, b["wo2"], out=h, residual=h, out_f32=True)
        out = ops.layernorm(h, weight=self.final_ln, eps=eps, rms=True, out_dtype=f32)
        return out[:n]

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None):
        """HF surface: returns an object with `.last_hidden_state` [B, L, d_model]; rows at padding positions are zero
        (the reference discards them: wan_utils.py:52-59)."""
        B, Lt = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        out = torch.zeros(B, Lt, self.cfg.d_model, device=self.device, dtype=f32)
        for i in range(B):
            m = attention_mask[i].bool().cpu()
            n = int(m.sum())
            if n == 0:
                continue
            if not bool(m[:n].all()):
                raise NotImplementedError("attention_mask must be right-padded (ones then zeros), as the Wan tokenizer call produces")
            out[i, :n] = self.encode_valid(input_ids[i, :n])
        return _Out(out)


class _Out:
    def __init__(self, last_hidden_state):
        self.last_hidden_state = last_hidden_state


def prompt_clean(text: str) -> str:
    """diffusers.pipelines.wan.pipeline_wan.prompt_clean = whitespace_clean(basic_clean(text)).  `ftfy.fix_text` is applied when
    the package is importable (it is not in this image: mojibake repair is then skipped; plain prompts are unaffected)."""
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


@torch.no_grad()
def compute_wan_text_embeddings(prompt: Union[str, List[str]], text_encoders, tokenizers, max_sequence_length: int = 226, device=None):
    """utils/wan_utils.py:25-60 with the same arguments: clean -> tokenize (padded to max_sequence_length, truncated, special
    tokens) -> encoder -> keep the seq_len valid rows, zero the rest.  `tokenizers` is any HF-style tokenizer callable."""
    dtype = text_encoders.dtype
    prompt = [prompt] if isinstance(prompt, str) else prompt
    prompt = [prompt_clean(u) for u in prompt]
    ti = tokenizers(prompt, padding="max_length", max_length=max_sequence_length, truncation=True, add_special_tokens=True,
                    return_attention_mask=True, return_tensors="pt")
    ids, mask = ti.input_ids, ti.attention_mask
    seq_lens = mask.gt(0).sum(dim=1).long()
    hs = text_encoders(ids.to(device), mask.to(device)).last_hidden_state.to(dtype=dtype, device=device)
    hs = [u[:v] for u, v in zip(hs, seq_lens)]
    return torch.stack([torch.cat([u, u.new_zeros(max_sequence_length - u.size(0), u.size(1))]) for u in hs], dim=0)


def make_pipeline_text_encoder(encoder: UMT5TextEncoder, tokenizer):
    """Adapter for WanT2VPipeline(text_encoder=...): (prompts, max_len) -> prompt_embeds [B, max_len, d_model]."""
    def enc(prompts: List[str], max_len: int) -> torch.Tensor:
        return compute_wan_text_embeddings(prompts, encoder, tokenizer, max_sequence_length=max_len, device=encoder.device)
    return enc
