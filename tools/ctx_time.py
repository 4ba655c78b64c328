"""Time of the per-prompt cross-attention context (text embedder MLP, K | V projections of every block, V.Wo^T of the cached-context form)."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.wan.dit import WanDiT, WAN_1_3B
from vist3a_amd.wan.weights import random_dit_state_dict
m = WanDiT(WAN_1_3B, random_dit_state_dict(WAN_1_3B, seed=0, device="cuda"))
text = torch.randn(2, 512, 4096, device="cuda") * 0.1
text[0, 64:] = 0; text[1, 80:] = 0
for _ in range(2):
    m._context(text.clone())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for _ in range(n):
    m._context(text.clone())
e1.record(); torch.cuda.synchronize()
print(json.dumps(dict(context_ms_per_prompt=round(e0.elapsed_time(e1) / n, 3), ctx_vo=m.ctx_vo)))
