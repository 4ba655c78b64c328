"""Golden vector for the UMT5 text encoder, produced by the `transformers` package itself (the third-party dependency the
reference's WanPipeline drives; it is not part of /root/reference, so this generator does not use the stub importer of
make_golden.py and lives in its own process).

    python tests/golden/make_golden_umt5.py"""
from __future__ import annotations

import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))

import torch
from safetensors.torch import save_file


def _save(name, tensors):
    out = HERE / f"{name}.safetensors"
    save_file({k: v.contiguous() for k, v in tensors.items()}, str(out))
    print(f"wrote {out} ({out.stat().st_size / 1024:.0f} KiB)")


def umt5_tiny():
    """transformers.UMT5EncoderModel (the dependency the reference's WanPipeline drives) on a 2-layer, 2-head (d_kv=64) config:
    two right-padded prompts of 37 and 9 tokens in a 48-token window."""
    import transformers
    from transformers import UMT5Config as HFConfig, UMT5EncoderModel
    from oracle import umt5 as OU
    cfg = OU.UMT5Config(vocab_size=120, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2)
    sd = OU.make_weights(cfg, seed=13)
    hf = UMT5EncoderModel(HFConfig(vocab_size=120, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2,
                                   relative_attention_num_buckets=32, relative_attention_max_distance=128,
                                   feed_forward_proj="gated-gelu", dropout_rate=0.0)).eval()
    full = dict(sd)
    full["encoder.embed_tokens.weight"] = sd["shared.weight"]
    res = hf.load_state_dict(full, strict=False)
    assert not res.unexpected_keys and not [k for k in res.missing_keys if "embed_tokens" not in k], res
    g = torch.Generator().manual_seed(23)
    ids = torch.randint(2, 120, (2, 48), generator=g)
    mask = torch.zeros(2, 48, dtype=torch.long)
    mask[0, :37] = 1
    mask[1, :9] = 1
    ids = ids * mask  # pad id 0
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=mask).last_hidden_state
        mine = OU.encode(sd, cfg, ids, mask)
    valid = mask.bool()
    err = (ref - mine)[valid].abs().max().item()
    print(f"umt5_tiny: transformers {transformers.__version__}; oracle vs HF max abs err on valid rows {err:.2e}; |out| {ref[valid].abs().mean():.3f}")
    assert err < 2e-5
    _save("umt5_tiny", {"input_ids": ids, "attention_mask": mask, "out": ref * valid[..., None]})


if __name__ == "__main__":
    umt5_tiny()
