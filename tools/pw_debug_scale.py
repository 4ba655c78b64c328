import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import lib, ops
L = lib.load()
B, H, N, D = (int(x) for x in sys.argv[1].split("x")) if len(sys.argv) > 1 else (2, 16, 4096, 128)
Nk = N
d = H * D
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B * N, d, device="cuda", generator=g) * 0.5).bfloat16()
k = (torch.randn(B * Nk, d, device="cuda", generator=g) * 0.5).bfloat16()
vt = torch.randn(d, B * Nk, device="cuda", generator=g).bfloat16()
def run(which):
    L.v3a_attention_set_kernel(which)
    o = torch.zeros(B * N, d, device="cuda", dtype=torch.bfloat16)
    ops.attention(q, k, vt, o, B=B, H=H, Nq=N, Nk=Nk, D=D, q_batch_stride=N * d, k_batch_stride=Nk * d, vt_batch_stride=Nk, o_batch_stride=N * d)
    torch.cuda.synchronize()
    return o.float()
o1 = run(1)
o3a, o3b = run(3), run(3)
print("k3 run-to-run identical:", torch.equal(o3a, o3b))
diff = (o1 - o3a).abs().view(B, N // 32, 32, H, 4, 32)
print("max diff", diff.max().item(), " fraction of elements differing", (diff > 0).float().mean().item())
print("by q-block parity (A, B):", diff.view(B, N // 64, 2, 32, H, 4, 32).amax(dim=(0, 1, 3, 4, 5, 6)).tolist())
print("by d-tile:", diff.amax(dim=(0, 1, 2, 3, 5)).tolist())
print("by wave in workgroup (4):", diff.view(B, N // 256, 4, 2, 32, H, 4, 32).amax(dim=(0, 1, 3, 4, 5, 6, 7)).tolist())
bad = (diff.amax(dim=(2, 4, 5)) > 0)   # [B, N/32, H]
print("bad 32-row blocks:", int(bad.sum()), "of", bad.numel(), "; per head:", bad.sum(dim=(0, 1)).tolist())
print("bad blocks by query-block position (first 16 of 64-row units):", bad.view(B, N // 64, 2, H).sum(dim=(0, 2, 3))[:16].tolist())
