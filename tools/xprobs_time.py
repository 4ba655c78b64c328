"""The cross-attention probability launch of one DiT block (csrc/xattn_probs.hip) at the production shape: time per launch and a checksum of
the output (variants of the kernel must reproduce it bit for bit).  usage: [V3A_LIB=...] python tools/xprobs_time.py [Nk ...]"""
import sys, json, math, hashlib
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops
B, H, N, D = 2, 12, 4096, 128
d = H * D
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B * N, d, device="cuda", generator=g) * 0.7).bfloat16()
qsq = torch.empty(B * N, d // 32, device="cuda")
qsq.copy_((q.float() ** 2).view(B * N, d // 32, 32).sum(-1))
for Nk in ([int(a) for a in sys.argv[1:]] or [73, 88, 128]):
    Lkp = min(128, -(-Nk // 16) * 16)
    k = (torch.randn(B * Nk, d, device="cuda", generator=g) * 0.5).bfloat16()
    bias = torch.zeros(B, Nk, device="cuda")
    bias[:, Nk - 1] = math.log(512 - Nk + 1)
    p = torch.zeros(B * N, H * Lkp, device="cuda", dtype=torch.bfloat16)
    run = lambda: ops.xattn_probs(q, k, p, B=B, H=H, Nq=N, Nk=Nk, Lkp=Lkp, q_batch_stride=N * d, k_batch_stride=Nk * d, p_batch_stride=N * H * Lkp,
                                  key_bias=bias, key_bias_first=Nk - 1, q_row_sumsq=qsq, q_eps=1e-6)
    for _ in range(3): run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    mb = (q.numel() * 2 + p.numel() * 2 + k.numel() * 2 + qsq.numel() * 4) / 1e6
    print(json.dumps(dict(Nk=Nk, Lkp=Lkp, us=round(best, 2), MB=round(mb, 1), TBps=round(mb / best, 2),
                          rowsum=round(float(p.float().view(B * N, H, Lkp).sum(-1).mean()), 5),
                          sha=hashlib.sha256(p.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12])))
