"""fp8 vs bf16 self-attention launch at the Wan-1.3B (B=2, H=12) and Wan-14B (B=2, H=40) shapes, N = 4096, D = 128."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops
D = 128
g = torch.Generator(device="cuda").manual_seed(0)

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (B, H, N) in ((2, 12, 4096), (2, 40, 4096), (2, 12, 6144)):
    d = H * D
    q = (torch.randn(B * N, d, device="cuda", generator=g)).bfloat16()
    k = (torch.randn(B * N, d, device="cuda", generator=g)).bfloat16()
    vt = torch.randn(d, B * N, device="cuda", generator=g).bfloat16()
    o = torch.empty(B * N, d, device="cuda", dtype=torch.bfloat16)
    kw = dict(B=B, H=H, Nq=N, Nk=N, q_batch_stride=N * d, k_batch_stride=N * d, vt_batch_stride=N, o_batch_stride=N * d)
    us16 = t(lambda: ops.attention(q, k, vt, o, D=D, **kw))
    q8, k8, vt8 = ops.quantize_fp8(q), ops.quantize_fp8(k), ops.quantize_fp8(vt)
    us8 = t(lambda: ops.attention_fp8(q8, k8, vt8, o, **kw))
    usq = t(lambda: (ops.quantize_fp8(q, out=q8), ops.quantize_fp8(k, out=k8), ops.quantize_fp8(vt, out=vt8)))
    fl = 4 * B * H * N * N * D
    print(json.dumps(dict(B=B, H=H, N=N, bf16_us=round(us16, 1), bf16_tf=round(fl / us16 / 1e6), fp8_us=round(us8, 1), fp8_tf=round(fl / us8 / 1e6),
                          quantize_qkv_us=round(usq, 1))), flush=True)
