"""Time the full-size UMT5-XXL text encoder (24 blocks, d_model 4096) for one prompt of n valid tokens."""
import sys, json, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.wan.text_encoder import UMT5Config, UMT5TextEncoder

cfg = UMT5Config(vocab_size=4096)  # the 256384-row embedding (2.1 GB) is a gather, not compute: a small table stands in
g = torch.Generator(device="cuda").manual_seed(0)
d, inner = cfg.d_model, cfg.num_heads * cfg.d_kv
sd = {"shared.weight": torch.randn(cfg.vocab_size, d, device="cuda", generator=g), "encoder.final_layer_norm.weight": torch.ones(d, device="cuda")}
for i in range(cfg.num_layers):
    p = f"encoder.block.{i}.layer."
    a, f = p + "0.SelfAttention.", p + "1.DenseReluDense."
    for n_ in ("q", "k", "v"):
        sd[a + n_ + ".weight"] = torch.randn(inner, d, device="cuda", generator=g) * d ** -0.5 * 0.6
    sd[a + "o.weight"] = torch.randn(d, inner, device="cuda", generator=g) * inner ** -0.5
    sd[a + "relative_attention_bias.weight"] = torch.randn(32, cfg.num_heads, device="cuda", generator=g)
    sd[p + "0.layer_norm.weight"] = torch.ones(d, device="cuda")
    sd[p + "1.layer_norm.weight"] = torch.ones(d, device="cuda")
    sd[f + "wi_0.weight"] = torch.randn(cfg.d_ff, d, device="cuda", generator=g) * d ** -0.5
    sd[f + "wi_1.weight"] = torch.randn(cfg.d_ff, d, device="cuda", generator=g) * d ** -0.5
    sd[f + "wo.weight"] = torch.randn(d, cfg.d_ff, device="cuda", generator=g) * cfg.d_ff ** -0.5
enc = UMT5TextEncoder(cfg, sd)
del sd
wbytes = cfg.num_layers * (4 * d * inner + 3 * d * cfg.d_ff) * 2
for n in (40, 128, 512):
    ids = torch.randint(1, 4096, (n,), device="cuda")
    for _ in range(2):
        o = enc.encode_valid(ids)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5):
        o = enc.encode_valid(ids)
    torch.cuda.synchronize(); ms = (time.time() - t0) / 5 * 1e3
    print(json.dumps(dict(tokens=n, ms=round(ms, 2), weight_GB=round(wbytes / 1e9, 2), weight_stream_TBps=round(wbytes / ms / 1e9, 2),
                          finite=bool(torch.isfinite(o).all()))))
