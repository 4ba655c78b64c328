"""GEMM experiments: fixed-overhead model (time vs K) and epilogue ablation via debug flags."""
import sys, math, json, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops, lib as L
bf16 = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)

def run(M, N, K, tile, flags=0, res=False, iters=30):
    a = torch.randn(M, K, device="cuda", generator=g).to(bf16)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(bf16)
    b = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda", dtype=bf16)
    r = torch.randn(M, N, device="cuda", generator=g).to(bf16) if res else None
    args = L.GemmArgs(a.data_ptr(), w.data_ptr(), out.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, None, M, N, K, K, K, N, N if res else 0,
                      0, 0, 0, flags, tile, None, 0, 0, 0, 0, 0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = L.load().v3a_gemm_bf16_nt
    for _ in range(3): f(C.byref(args), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f(C.byref(args), st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us



for tile in (0, 2):
    M, N, K = 8192, 1536, 8960
    base = run(M, N, K, tile)
    nodma = run(M, N, K, tile, flags=1 << 28)
    hot = run(M, N, K, tile, flags=1 << 26)
    f = lambda t: round(2*M*N*K/t/1e6)
    print(json.dumps(dict(tile=tile, us=round(base, 1), tf=f(base), tf_no_refill_dma=f(nodma), tf_cache_hot_dma=f(hot))), flush=True)
