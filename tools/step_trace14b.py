"""rocprofv3 --kernel-trace --stats target: three Wan-14B CFG forwards in the fp8 mode (attention + GEMMs)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.wan.dit import WAN_14B, WanDiT
from vist3a_amd.wan.weights import random_dit_state_dict
m = WanDiT(WAN_14B, random_dit_state_dict(WAN_14B, seed=0, device="cuda"))
m.attn_dtype = "fp8"
m.enable_fp8_gemm()
text = torch.zeros(2, 512, 4096, device="cuda")
text[0, :64] = torch.randn(64, 4096, device="cuda") * 0.1
text[1, :80] = torch.randn(80, 4096, device="cuda") * 0.1
t = torch.tensor([900, 900], device="cuda")
lat = torch.randn(2, 16, 4, 64, 64, device="cuda").bfloat16()
for _ in range(3):
    m(lat, t, text)
torch.cuda.synchronize()
