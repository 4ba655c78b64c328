"""gemm_w4_kernel (tile 14) against the ping-pong tile 6 / lockstep tile 0: bit-equality on edge shapes (K = 64 / 128 / 192, ragged M and N,
epilogue variants) and time on the DiT's shapes.  Development instrument (gpurun)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vist3a_amd import lib as L
from vist3a_amd import ops

bf16 = torch.bfloat16
W4 = 14
g = torch.Generator(device="cuda").manual_seed(0)


def one(M, N, K, **kw):
    a = torch.randn(M, K, device="cuda", generator=g).to(bf16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(bf16)
    b = torch.randn(N, device="cuda", generator=g)
    ref = ops.gemm(a, w, b, tile=0, **kw)
    out = ops.gemm(a, w, b, tile=W4, **kw)
    torch.cuda.synchronize()
    eq = bool(torch.equal(ref, out))
    d = (ref.float() - out.float()).abs().max().item()
    print(json.dumps(dict(M=M, N=N, K=K, kw=sorted(kw), bit_equal=eq, max_abs_diff=d)), flush=True)
    return eq


ok = True
for (M, N, K) in ((256, 192, 64), (256, 192, 128), (256, 192, 192), (256, 192, 1536), (512, 384, 256), (8192, 1536, 1536), (300, 200, 320), (13416, 1024, 1024),
                  (8192, 8960, 1536)):
    ok &= one(M, N, K)
M, N, K = 8192, 1536, 1536
res = torch.randn(M, N, device="cuda", generator=g).to(bf16)
scale = torch.randn(2, N, device="cuda", generator=g)
ok &= one(M, N, K, residual=res, scale=scale, rows_per_batch=4096)
ok &= one(M, N, K, act=L.ACT_GELU_TANH)
print("ALL BIT-EQUAL" if ok else "MISMATCH", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "time" and ok:
    import subprocess
    subprocess.run([sys.executable, str(Path(__file__).parent / "gemm_sweep.py"), "6,14", "0,1,2,3,8"])
