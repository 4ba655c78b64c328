"""Every GEMM tile on the DiT's single-round shapes (per-tile fixed cost study): us per launch."""
import sys, json
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import lib as L, ops
bf16 = torch.bfloat16
lib = L.load()
g = torch.Generator(device="cuda").manual_seed(0)
def t(fn, n=40):
    for _ in range(8): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / n)
    return round(best * 1e3, 1)
for (M, N, K) in ((8192, 1536, 1536), (8192, 1536, 960), (8192, 3072, 1536), (8192, 8960, 1536), (8192, 1536, 8960)):
    x = torch.randn(M, K, device="cuda", generator=g).to(bf16)
    w = (torch.randn(N, K, device="cuda", generator=g) / 39).to(bf16)
    b = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda", dtype=bf16)
    res = {}
    for ti in range(lib.v3a_gemm_num_tiles()):
        try:
            res[lib.v3a_gemm_tile_name(ti).decode()] = t(lambda: ops.gemm(x, w, b, out=out, tile=ti))
        except RuntimeError:
            pass
    print(json.dumps(dict(M=M, N=N, K=K, tflops_best=round(2e-6 * M * N * K / min(res.values()), 0), us=res)), flush=True)
