"""GEMM tile sweep on the DiT's production shapes: TFLOP/s per tile (back-to-back launches, random data) and bit-equality of every
tile against tile 0.  Run on the MI355X box through gpurun; a development instrument, not part of the product path."""
import ctypes as C
import json
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vist3a_amd import lib as L

bf16 = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
lib = L.load()
names = [lib.v3a_gemm_tile_name(t).decode() for t in range(lib.v3a_gemm_num_tiles())]


def run(M, N, K, tile, res=False, iters=20, rounds=3, act=0):
    a = torch.randn(M, K, device="cuda", generator=g).to(bf16)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(bf16)
    b = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda", dtype=bf16)
    r = torch.randn(M, N, device="cuda", generator=g).to(bf16) if res else None
    args = L.GemmArgs(a.data_ptr(), w.data_ptr(), out.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, None, M, N, K, K, K, N,
                      N if res else 0, 0, 0, act, 0, tile, None, 0, 0, 0, 0, 0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lib.v3a_gemm_bf16_nt
    for _ in range(3):
        assert f(C.byref(args), st) == 0
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f(C.byref(args), st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best, out, (a, w, b, r)


if __name__ == "__main__":
    ACT = int(__import__('os').environ.get('ACT', '0'))
    tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(len(names)))
    shapes = ((8192, 1536, 1536, True), (8192, 3072, 1536, False), (8192, 8960, 1536, False), (8192, 1536, 8960, True),
              (1536, 8192, 1536, False), (4096, 1536, 1536, True), (8192, 5120, 5120, False), (8192, 13824, 5120, False),
              (8192, 8192, 8192, False))
    if len(sys.argv) > 2 and "x" in sys.argv[2]:   # custom shapes: MxNxK[r];MxNxK...
        shapes = [tuple(int(v) for v in sh.rstrip("r").split("x")) + (sh.endswith("r"),) for sh in sys.argv[2].split(";")]
    elif len(sys.argv) > 2:
        shapes = [shapes[int(i)] for i in sys.argv[2].split(",")]
    for (M, N, K, res) in shapes:
        row, ref = {}, None
        for tile in tiles:
            g.manual_seed(1234)
            us, out, _ = run(M, N, K, tile, res=res, act=ACT)
            if ref is None:
                ref = out.clone()
            row[names[tile]] = dict(tf=round(2 * M * N * K / us / 1e6), us=round(us, 1), bit_equal=bool(torch.equal(out, ref)))
        print(json.dumps(dict(M=M, N=N, K=K, res=res, tiles=row)), flush=True)
