"""Time + check the DiT self-attention launch (B=2, H=12, N=4096, D=128)."""
import sys, json, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops
B, H, N, D = 2, 12, 4096, 128
d = H * D
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B * N, d, device="cuda", generator=g) * 0.5).bfloat16()
k = (torch.randn(B * N, d, device="cuda", generator=g) * 0.5).bfloat16()
v = torch.randn(B * N, d, device="cuda", generator=g).bfloat16()
vt = torch.empty(d, B * N, device="cuda", dtype=torch.bfloat16)
for b in range(B):
    vt[:, b * N:(b + 1) * N] = v[b * N:(b + 1) * N].t()
o = torch.empty(B * N, d, device="cuda", dtype=torch.bfloat16)
run = lambda: ops.attention(q, k, vt, o, B=B, H=H, Nq=N, Nk=N, D=D, q_batch_stride=N * d, k_batch_stride=N * d, vt_batch_stride=N, o_batch_stride=N * d)
for _ in range(20): run()
torch.cuda.synchronize()
times = []
for _ in range(6):   # the first rounds run on cold clocks: report all, quote the best
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    times.append(round(e0.elapsed_time(e1) / 50 * 1e3, 1))
us = min(times)
# reference on one (b, h)
qf, kf, vf = q[:N, :D].float(), k[:N, :D].float(), v[:N, :D].float()
ref = torch.softmax(qf @ kf.t() * D ** -0.5, -1) @ vf
rel = ((o[:N, :D].float() - ref).norm() / ref.norm()).item()
print(json.dumps(dict(mode="occ2" if os.environ.get("V3A_ATTN_OCC2") else "occ3", us=round(us, 1), rounds_us=times, tflops=round(4 * B * H * N * N * D / us / 1e6), rel=rel)))
