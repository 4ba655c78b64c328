"""Experiment: the V^T projection on a second stream beside the q|k RMSNorm + RoPE pass (WanDiT.overlap_vt) - CFG step time with / without."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.wan.dit import WAN_1_3B, WanDiT
from vist3a_amd.wan.weights import random_dit_state_dict
from vist3a_amd.t23d import synthetic_text_embeddings
dev = "cuda"
cfg = WAN_1_3B
dit = WanDiT(cfg, random_dit_state_dict(cfg, seed=0, device=dev), device=dev)
pe, ne = synthetic_text_embeddings(dev)
text2 = torch.cat([pe, ne], 0).contiguous()
shape = (16, 4, 64, 64)
ts = torch.full((50,), 500, device=dev, dtype=torch.int64)
tab2 = dit.time_tables(ts, 2)
x2 = torch.randn((2,) + shape, device=dev).bfloat16()
ref = dit.forward(x2, ts[:1].expand(2), text2)[0].clone()
def run(n):
    for i in range(n):
        dit.forward(None, None, text2, tokens_in=True, tokens_out=True, latent_shape=(2,) + shape, time_table=tab2[i % 50])
for mode in (False, True, False, True):
    dit.overlap_vt = mode
    out = dit.forward(x2, ts[:1].expand(2), text2)[0]
    torch.cuda.synchronize()
    same = bool(torch.equal(out, ref))
    run(5); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(30); b.record(); torch.cuda.synchronize()
    print(json.dumps({"overlap_vt": mode, "ms_per_cfg_step": round(a.elapsed_time(b) / 30, 3), "bit_identical_to_serial": same}), flush=True)
