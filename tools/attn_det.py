"""Run-to-run determinism of the attention launches of one DiT block (self: Nk=4096, cross: Nk=512)."""
import sys, os
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops
B, H, N, D = 2, 12, 4096, 128
d = H * D
g = torch.Generator(device="cuda").manual_seed(0)
for Nk in (4096, 512, 1000):
    q = (torch.randn(B * N, d, device="cuda", generator=g) * 0.5).bfloat16()
    k = (torch.randn(B * Nk, d, device="cuda", generator=g) * 0.5).bfloat16()
    Lp = (Nk + 63) // 64 * 64
    vt = torch.zeros(d, B * Lp, device="cuda", dtype=torch.bfloat16)
    for b in range(B):
        vt[:, b * Lp: b * Lp + Nk] = torch.randn(d, Nk, device="cuda", generator=g).bfloat16()
    outs = []
    for _ in range(6):
        o = torch.empty(B * N, d, device="cuda", dtype=torch.bfloat16)
        ops.attention(q, k, vt, o, B=B, H=H, Nq=N, Nk=Nk, D=D, q_batch_stride=N * d, k_batch_stride=Nk * d, vt_batch_stride=Lp, o_batch_stride=N * d)
        outs.append(o)
    torch.cuda.synchronize()
    bad = [(outs[0] != o).any(dim=1).nonzero().flatten() for o in outs[1:]]
    print({k_: os.environ[k_] for k_ in os.environ if k_.startswith("V3A_")}, "Nk", Nk, "identical:", [len(x) == 0 for x in bad],
          "first bad rows:", [x[:6].tolist() for x in bad if len(x)][:2], "n bad rows:", [len(x) for x in bad])
