"""Weight sources for the generator: seeded synthetic weights of the exact production shapes (no checkpoint is
reachable offline: SURVEY.md §0.4) and loaders for the real diffusers / peft layouts."""
from __future__ import annotations

import json
import math
from pathlib import Path
from typing import Dict

import torch

from .dit import WanDiTConfig, merge_lora_into_state_dict


def random_dit_state_dict(cfg: WanDiTConfig, seed: int = 0, device="cpu", dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """diffusers-named state dict: Linear/Conv ~ N(0, 0.02), bias ~ N(0,0.02), RMSNorm gamma ~ 1, norm2 affine ~ (1,0),
    scale_shift_table ~ N(0,1)/sqrt(d)  (SURVEY.md §8d synthetic-input spec)."""
    g = torch.Generator(device=device).manual_seed(seed)
    d, ffn = cfg.dim, cfg.ffn_dim
    sd: Dict[str, torch.Tensor] = {}
    rn = lambda *s, std=1.0: (torch.randn(*s, generator=g, device=device) * std).to(dtype)

    def lin(name, o, i):
        sd[name + ".weight"] = rn(o, i, std=0.02)
        sd[name + ".bias"] = rn(o, std=0.02)

    pt, ph, pw = cfg.patch_size
    sd["patch_embedding.weight"] = rn(d, cfg.in_channels, pt, ph, pw, std=0.05)
    sd["patch_embedding.bias"] = rn(d, std=0.02)
    ce = "condition_embedder."
    lin(ce + "time_embedder.linear_1", d, cfg.freq_dim)
    lin(ce + "time_embedder.linear_2", d, d)
    lin(ce + "time_proj", 6 * d, d)
    lin(ce + "text_embedder.linear_1", d, cfg.text_dim)
    lin(ce + "text_embedder.linear_2", d, d)
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        sd[p + "scale_shift_table"] = rn(1, 6, d, std=1 / math.sqrt(d)).float()
        for a in ("attn1", "attn2"):
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                lin(p + f"{a}.{n}", d, d)
            sd[p + f"{a}.norm_q.weight"] = (1 + rn(d, std=0.1).float()).to(dtype)
            sd[p + f"{a}.norm_k.weight"] = (1 + rn(d, std=0.1).float()).to(dtype)
        sd[p + "norm2.weight"] = (1 + rn(d, std=0.1).float()).float()
        sd[p + "norm2.bias"] = rn(d, std=0.05).float()
        lin(p + "ffn.net.0.proj", ffn, d)
        lin(p + "ffn.net.2", d, ffn)
    sd["scale_shift_table"] = rn(1, 2, d, std=1 / math.sqrt(d)).float()
    lin("proj_out", cfg.out_channels * pt * ph * pw, d)
    return sd


def load_dit_config(model_dir: str, default: WanDiTConfig) -> WanDiTConfig:
    """`<model_dir>/transformer/config.json` the way diffusers' `from_pretrained` reads it (WanTransformer3DModel's registered config:
    patch_size, num_attention_heads, attention_head_dim, in/out_channels, text_dim, freq_dim, ffn_dim, num_layers, eps,
    rope_max_seq_len); `default` when the folder has no config.  Options this path does not implement raise instead of being ignored."""
    p = Path(model_dir)
    if (p / "transformer").is_dir():
        p = p / "transformer"
    f = p / "config.json"
    if not f.exists():
        return default
    c = json.loads(f.read_text())
    if c.get("image_dim") is not None or c.get("added_kv_proj_dim") is not None:
        raise NotImplementedError("image-to-video Wan transformers (image_dim / added_kv_proj_dim) are outside the text->3DGS path")
    if c.get("qk_norm", "rms_norm_across_heads") != "rms_norm_across_heads" or not c.get("cross_attn_norm", True):
        raise NotImplementedError("only qk_norm='rms_norm_across_heads' with cross_attn_norm=True (Wan 2.1 T2V) is implemented")
    names = ("num_attention_heads", "attention_head_dim", "in_channels", "out_channels", "text_dim", "freq_dim", "ffn_dim", "num_layers",
             "eps", "rope_max_seq_len")
    kw = {k: c[k] for k in names if k in c}
    if "patch_size" in c:
        kw["patch_size"] = tuple(c["patch_size"])
    return WanDiTConfig(**kw)


def load_dit_state_dict(model_dir: str) -> Dict[str, torch.Tensor]:
    """Read a diffusers `transformer/` folder (sharded or single .safetensors)."""
    from safetensors.torch import load_file
    p = Path(model_dir)
    if (p / "transformer").is_dir():
        p = p / "transformer"
    idx = p / "diffusion_pytorch_model.safetensors.index.json"
    sd: Dict[str, torch.Tensor] = {}
    if idx.exists():
        for shard in sorted(set(json.loads(idx.read_text())["weight_map"].values())):
            sd.update(load_file(str(p / shard)))
    else:
        sd.update(load_file(str(p / "diffusion_pytorch_model.safetensors")))
    return sd


def load_peft_lora(lora_dir: str, sd: Dict[str, torch.Tensor]) -> int:
    """Merge a peft adapter folder (adapter_config.json + adapter_model.safetensors; the `lora_ema/` layout written by
    /root/reference/train_vdm.py:32-97 and read at /root/reference/inference_t23d.py:74-77) into `sd`."""
    from safetensors.torch import load_file
    p = Path(lora_dir)
    cfg = json.loads((p / "adapter_config.json").read_text())
    lsd = load_file(str(p / "adapter_model.safetensors"))
    return merge_lora_into_state_dict(sd, lsd, float(cfg["lora_alpha"]), int(cfg["r"]))
