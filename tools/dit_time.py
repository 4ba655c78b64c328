"""Time one full-size Wan-1.3B DiT forward (batch-2 CFG pair, 13 views @512) on the GPU box."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.wan.dit import WanDiT, WAN_1_3B
from vist3a_amd.wan.weights import random_dit_state_dict

cfg = WAN_1_3B
sd = random_dit_state_dict(cfg, seed=0, device="cuda")
m = WanDiT(cfg, sd)
import os
if os.environ.get('V3A_FUSED_QKV') == '1':
    m.fused_qkv = True   # A/B: q | k | V^T from one launch (transposed-tail tile)
if os.environ.get('V3A_CTX_VO') == '0':
    m.ctx_vo = False   # A/B: flash cross-attention + to_out GEMM instead of the cached-context form
del sd
Tl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lat = torch.randn(2, 16, Tl, 64, 64, device="cuda").bfloat16()
text = torch.randn(2, 512, 4096, device="cuda") * 0.1
text[0, 64:] = 0; text[1, 80:] = 0
t = torch.tensor([900, 900], device="cuda")
for _ in range(2):
    o = m(lat, t, text)[0]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
t0 = time.time(); e0.record()
for _ in range(n):
    o = m(lat, t, text)[0]
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
N = Tl * 1024
d, ffn, L, ctx = cfg.dim, cfg.ffn_dim, cfg.num_layers, 512
fl = 2 * L * (8 * N * d * d + 4 * N * N * d + 4 * N * d * d + 4 * N * ctx * d + 4 * N * d * ffn)  # ctx K/V cached: excluded
print(json.dumps(dict(ms_per_cfg_pair=ms, wall_ms=(time.time() - t0) / n * 1e3, tflops=fl / ms / 1e9, finite=bool(torch.isfinite(o.float()).all()))))
