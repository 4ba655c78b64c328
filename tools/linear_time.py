"""The camera head's fp32 skinny linears (csrc/elementwise.hip linear_f32_kernel) at their production shapes: time per launch, weight-stream
bandwidth and a checksum of the output (kernel variants must reproduce it bit for bit).  usage: [V3A_LIB=...] python tools/linear_time.py"""
import sys, json, math, hashlib
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops, lib as L
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K, act in ((13, 6144, 2048, L.ACT_NONE), (13, 2048, 2048, L.ACT_NONE), (13, 8192, 2048, L.ACT_GELU_ERF), (13, 2048, 8192, L.ACT_NONE),
                     (21, 6144, 2048, L.ACT_NONE), (13, 2048, 12, L.ACT_SILU)):
    x = torch.randn(M, K, device="cuda", generator=g)
    ws = [torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K) for _ in range(max(1, int(600e6 // (N * K * 4))))]   # > Infinity Cache: weights stream from HBM as in the trunk
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    y = ops.linear_f32(x, ws[0], b, act=act)
    torch.cuda.synchronize()
    sha = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for w in ws: ops.linear_f32(x, w, b, act=act)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / len(ws) * 1e3)
    print(json.dumps(dict(M=M, N=N, K=K, us=round(best, 1), weight_TBps=round(N * K * 4 / best / 1e6, 2), sha=sha)))
