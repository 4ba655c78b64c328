"""Launch one kernel shape a few times (for rocprofv3 --pmc passes).  usage: kprof.py gemm M N K tile | attn B H N D"""
import sys, math
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops
bf16 = torch.bfloat16
what = sys.argv[1]
g = torch.Generator(device="cuda").manual_seed(0)
if what == "gemm":
    M, N, K, tile = map(int, sys.argv[2:6])
    a = torch.randn(M, K, device="cuda", generator=g).to(bf16)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(bf16)
    b = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda", dtype=bf16)
    for _ in range(5):
        ops.gemm(a, w, b, out=out, tile=tile)
elif what == "attn":
    B, H, N, D = map(int, sys.argv[2:6])
    q = torch.randn(B * N, H * D, device="cuda", generator=g).to(bf16)
    k = torch.randn(B * N, H * D, device="cuda", generator=g).to(bf16)
    vt = torch.randn(H * D, B * N + 64, device="cuda", generator=g).to(bf16)
    o = torch.empty(B * N, H * D, device="cuda", dtype=bf16)
    for _ in range(5):
        ops.attention(q, k, vt, o, B=B, H=H, Nq=N, Nk=N, D=D, q_batch_stride=N * H * D, k_batch_stride=N * H * D, vt_batch_stride=N, o_batch_stride=N * H * D)
torch.cuda.synchronize()
