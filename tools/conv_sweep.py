"""Implicit-GEMM convolution tile sweep on the VAE decoder's two heaviest layers (ms per tile, bit-equality against the first tile)."""
import sys; sys.path.insert(0, "/root/repo")
import torch, json
from vist3a_amd import ops, lib as L
lib = L.load()
names = [lib.v3a_gemm_tile_name(t).decode() for t in range(lib.v3a_gemm_num_tiles())]
bf16 = torch.bfloat16
def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best
g = torch.Generator().manual_seed(0)
for (T, H, W, Cin, Cout) in ((13, 512, 512, 96, 96), (13, 256, 256, 192, 192)):
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.02
    cw = ops.ConvWeight(w, torch.randn(Cout, generator=g))
    x = torch.randn(T, H, W, Cin, generator=g).to(bf16).cuda()
    ref = None
    row = {}
    for t, nm in enumerate(names):
        if nm.startswith("pp"): continue
        try:
            out = ops.conv(x, cw, pad=(2, 1, 1), tile=t)
        except Exception as e:
            continue
        ms = timeit(lambda: ops.conv(x, cw, pad=(2, 1, 1), tile=t))
        if ref is None: ref = out
        row[nm.split("_stg")[0]] = (round(ms, 3), bool(torch.equal(out, ref)))
    auto = timeit(lambda: ops.conv(x, cw, pad=(2, 1, 1)))
    print(T, H, W, Cin, Cout, "auto", round(auto, 3), json.dumps(row))
