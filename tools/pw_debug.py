import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import lib, ops
L = lib.load()
B, H, N, D = 1, 2, 512, 128
Nk = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d = H * D
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B * N, d, device="cuda", generator=g) * 0.5).bfloat16()
k = (torch.randn(B * Nk, d, device="cuda", generator=g) * 0.5).bfloat16()
vt = torch.randn(d, B * Nk, device="cuda", generator=g).bfloat16()
outs = {}
for which in (1, 3):
    L.v3a_attention_set_kernel(which)
    o = torch.zeros(B * N, d, device="cuda", dtype=torch.bfloat16)
    ops.attention(q, k, vt, o, B=B, H=H, Nq=N, Nk=Nk, D=D, q_batch_stride=N * d, k_batch_stride=Nk * d, vt_batch_stride=Nk, o_batch_stride=N * d)
    torch.cuda.synchronize()
    outs[which] = o.float()
diff = (outs[1] - outs[3]).abs()
print("Nk", Nk, "max diff", diff.max().item(), "rel", (diff.norm() / outs[1].norm()).item())
blk = diff.view(N // 32, 32, H, D).amax(dim=(1, 3))
print("per 32-row block x head max diff:\n", blk)
print("per d-tile (32) max diff:", diff.view(N, H, 4, 32).amax(dim=(0, 1, 3)))
ratio = (outs[3] / outs[1].clamp_min(1e-6))
print("ratio o3/o1 median per 32-row block (head 0):", ratio.view(N // 32, 32, H, D)[:, :, 0].flatten(1).median(dim=1).values)
