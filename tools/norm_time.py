"""Time the two row-normalisation launches of a DiT block at production size (HBM-bound: bytes / time against ~6.3 TB/s achievable)."""
import sys, json, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops
M, d = 8192, 1536
x = torch.randn(M, d, device="cuda").bfloat16(); y = torch.empty_like(x)
sc = torch.randn(2, d, device="cuda") * 0.1; sh = torch.randn(2, d, device="cuda") * 0.1
qk = torch.randn(M, 2 * d, device="cuda").bfloat16()
w = torch.ones(d, device="cuda")
def t(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3
us = t(lambda: ops.layernorm(x, out=y, scale=sc, shift=sh, rows_per_batch=M // 2, eps=1e-6))
print(json.dumps(dict(op="layernorm_adaln", M=M, d=d, us=round(us, 2), TBps=round(2 * M * d * 2 / us / 1e6, 2), lds=os.environ.get("V3A_NORM_LDS", "0"))))
