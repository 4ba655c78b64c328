"""e4m3 GEMM (v3a_gemm_fp8_nt): TFLOP/s per tile on the Wan-14B / Wan-1.3B projection shapes, and a check of the row quantiser and the
GEMM against a torch emulation of the same arithmetic (e4m3 values as fp32, fp32 matmul).  Development instrument (gpurun)."""
import json
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vist3a_amd import lib as L
from vist3a_amd import ops

bf16 = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
lib = L.load()
names = [lib.v3a_gemm_fp8_tile_name(t).decode() for t in range(lib.v3a_gemm_fp8_num_tiles())]


def emu_quant(x):
    amax = x.float().abs().amax(dim=1)
    sc = amax.clamp_min(1e-12) / torch.full_like(amax, 448.0)   # (tensor / python scalar multiplies by a reciprocal on the GPU)
    q = (x.float() * (torch.ones_like(sc) / sc)[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q, sc


def timeit(fn, iters=20, rounds=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


if __name__ == "__main__":
    shapes = [(8192, 13824, 5120), (8192, 5120, 13824), (8192, 5120, 5120), (8192, 8960, 1536), (8192, 1536, 8960), (8192, 1536, 1536),
              (1000, 520, 384)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in sh.split("x")) for sh in sys.argv[1].split(";")]
    for (M, N, K) in shapes:
        a = torch.randn(M, K, device="cuda", generator=g).to(bf16)
        w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(bf16)
        b = torch.randn(N, device="cuda", generator=g)
        a8, sa = ops.quantize_fp8_rows(a)
        w8, sw = ops.quantize_fp8_rows(w)
        qa, ra = emu_quant(a)
        qw, rw = emu_quant(w)
        row = {"M": M, "N": N, "K": K,
               "quant_bytes_equal": bool(torch.equal(a8, qa.view(torch.uint8)) and torch.equal(w8, qw.view(torch.uint8))),
               "quant_scale_equal": bool(torch.equal(sa, ra) and torch.equal(sw, rw)), "tiles": {}}
        ref = ((qa.float() @ qw.float().T) * (ra[:, None] * rw[None, :]) + b).to(bf16)
        row["quant_us"] = round(timeit(lambda: ops.quantize_fp8_rows(a, a8, sa)), 1)
        for t, nm in enumerate(names):
            out = torch.empty(M, N, device="cuda", dtype=bf16)
            fn = lambda: ops.gemm(a8, w8, b, out=out, a_scale=sa, w_scale=sw, tile=t)
            us = timeit(fn)
            err = ((out.float() - ref.float()).norm() / ref.float().norm()).item()
            row["tiles"][nm] = {"us": round(us, 1), "tf": round(2.0 * M * N * K / us / 1e6), "rel": float(f"{err:.2e}"),
                                "max_abs": float(f"{(out.float() - ref.float()).abs().max().item():.2e}")}
        print(json.dumps(row))
