"""Cross-attention launch of one DiT block: 512 keys vs merged padding keys (88 keys + key bias)."""
import sys, json, math
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops
B, H, N, D = 2, 12, 4096, 128
d = H * D
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B * N, d, device="cuda", generator=g) * 0.5).bfloat16()
o = torch.empty(B * N, d, device="cuda", dtype=torch.bfloat16)
for Nk, kb in ((512, False), (88, True), (88, False)):
    Lt, Lp = 512, 512
    k = (torch.randn(B * Lt, d, device="cuda", generator=g) * 0.5).bfloat16()
    vt = torch.randn(d, B * Lp, device="cuda", generator=g).bfloat16()
    bias = torch.zeros(B, Lp, device="cuda")
    bias[:, Nk - 1] = math.log(512 - Nk + 1)
    run = lambda: ops.attention(q, k, vt, o, B=B, H=H, Nq=N, Nk=Nk, D=D, q_batch_stride=N * d, k_batch_stride=Lt * d, vt_batch_stride=Lp,
                                o_batch_stride=N * d, key_bias=bias if kb else None, key_bias_first=Nk - 1)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(json.dumps(dict(Nk=Nk, key_bias=kb, us=round(e0.elapsed_time(e1) / 50 * 1e3, 1))))
