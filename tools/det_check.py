"""Run-to-run determinism of the full-size DiT forward (must be bit-identical)."""
import sys, os
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.wan.dit import WanDiT, WAN_1_3B
from vist3a_amd.wan.weights import random_dit_state_dict
from vist3a_amd.t23d import synthetic_text_embeddings
cfg = WAN_1_3B
m = WanDiT(cfg, random_dit_state_dict(cfg, seed=0, device="cuda"))
pe, ne = synthetic_text_embeddings("cuda")
text2 = torch.cat([pe, ne], 0).contiguous()
lat = torch.randn(2, 16, 4, 64, 64, device="cuda").bfloat16()
t = torch.tensor([900, 900], device="cuda")
tag = {k: os.environ[k] for k in os.environ if k.startswith("V3A_")}
for nl in (1, 30):
    outs = [m(lat, t, text2, num_layers=nl)[0].clone() for _ in range(4)]
    torch.cuda.synchronize()
    print(tag, "layers", nl, "identical runs:", [bool(torch.equal(outs[0], o)) for o in outs[1:]],
          "max|d|", max(float((outs[0].float() - o.float()).abs().max()) for o in outs[1:]))
