"""The HBM-bound row passes of a DiT block at production size, each beside the COPY YARDSTICK of the same box: a plain `copy_` of the same number of
bytes (read + written), the achievable rate of this chip for a one-read-one-write pass (6.3-6.8 TB/s; the 8 TB/s peak is quoted by `frac_of_8TBps`).
  layernorm (AdaLN form)            in 25 MB + out 25 MB
  q|k RMSNorm across heads + RoPE   in 50 MB + out 50 MB (in place)
  cross-attention probabilities     q 25 MB + row statistics 1.6 MB in, P [8192, 12 x 96] 18.9 MB out (tools/xprobs_time.py times it alone)"""
import json
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vist3a_amd import ops

bf16 = torch.bfloat16
M, d, H = 8192, 1536, 12


def t(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


def copy_us(nbytes_moved):
    """a torch copy_ that reads nbytes_moved / 2 and writes nbytes_moved / 2"""
    a = torch.empty(nbytes_moved // 2, device="cuda", dtype=torch.uint8)
    b = torch.empty_like(a)
    return t(lambda: b.copy_(a))


def row(op, us, mb):
    cu = copy_us(int(mb * 1e6))
    print(json.dumps(dict(op=op, us=round(us, 2), MB=round(mb, 1), TBps=round(mb / us, 2), frac_of_8TBps=round(mb / us / 8, 3),
                          copy_same_bytes_us=round(cu, 2), copy_TBps=round(mb / cu, 2), vs_copy=round(us / cu, 3))), flush=True)


x = torch.randn(M, d, device="cuda").to(bf16)
y = torch.empty_like(x)
sc, sh = torch.randn(2, d, device="cuda") * 0.1, torch.randn(2, d, device="cuda") * 0.1
row("layernorm_adaln", t(lambda: ops.layernorm(x, out=y, scale=sc, shift=sh, rows_per_batch=M // 2, eps=1e-6)), 2 * M * d * 2 / 1e6)
w, b = torch.randn(d, device="cuda"), torch.randn(d, device="cuda")
row("layernorm_affine", t(lambda: ops.layernorm(x, out=y, weight=w, bias=b)), 2 * M * d * 2 / 1e6)
qk = torch.randn(M, 2 * d, device="cuda").to(bf16)
rope = torch.randn(4096, 64, 2, device="cuda")
row("qk_rmsnorm_rope", t(lambda: ops.rmsnorm_rope(qk, w, out=qk, rope=rope, head_dim=128, tokens_per_batch=4096, eps=1e-6, weight2=b)), 2 * M * 2 * d * 2 / 1e6)
Nk, Lkp = 88, 96
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(M, d, device="cuda", generator=g) * 0.7).to(bf16)
qsq = (q.float() ** 2).view(M, d // 32, 32).sum(-1).contiguous()
k = (torch.randn(2 * Nk, d, device="cuda", generator=g) * 0.5).to(bf16)
bias = torch.zeros(2, Nk, device="cuda")
bias[:, Nk - 1] = math.log(512 - Nk + 1)
p = torch.zeros(M, H * Lkp, device="cuda", dtype=bf16)
us = t(lambda: ops.xattn_probs(q, k, p, B=2, H=H, Nq=4096, Nk=Nk, Lkp=Lkp, q_batch_stride=4096 * d, k_batch_stride=Nk * d, p_batch_stride=4096 * H * Lkp,
                               key_bias=bias, key_bias_first=Nk - 1, q_row_sumsq=qsq, q_eps=1e-6))
row("xattn_probs_88_keys", us, (q.numel() * 2 + p.numel() * 2 + k.numel() * 2 + qsq.numel() * 4) / 1e6)
