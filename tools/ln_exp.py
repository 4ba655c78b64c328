import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops
bf16 = torch.bfloat16
def timeit(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M, d = 8192, 1536
x = torch.randn(M, d, device="cuda").to(bf16)
out = torch.empty_like(x)
w, b = torch.randn(d, device="cuda"), torch.randn(d, device="cuda")
mod = torch.randn(2, 6, d, device="cuda")
print("ln modulated    us", round(timeit(lambda: ops.layernorm(x, out=out, scale=mod[:, 1], shift=mod[:, 0], rows_per_batch=4096)), 1))
print("ln affine       us", round(timeit(lambda: ops.layernorm(x, out=out, weight=w, bias=b)), 1))
print("ln plain        us", round(timeit(lambda: ops.layernorm(x, out=out)), 1))
qk = torch.randn(M, 2 * d, device="cuda").to(bf16)
rope = torch.randn(4096, 64, 2, device="cuda")
print("rms+rope strided us", round(timeit(lambda: ops.rmsnorm_rope(qk[:, :d], w, out=qk[:, :d], rope=rope, head_dim=128, tokens_per_batch=4096)), 1))
print("rms no rope     us", round(timeit(lambda: ops.rmsnorm_rope(x, w, out=x)), 1))
# after a GEMM wrote x (dirty lines in other XCD L2s)
a = torch.randn(M, d, device="cuda").to(bf16); ww = torch.randn(d, d, device="cuda").to(bf16) * 0.02
def chain():
    ops.gemm(a, ww, None, out=x, residual=x)
    ops.layernorm(x, out=out, scale=mod[:, 1], shift=mod[:, 0], rows_per_batch=4096)
t_chain = timeit(chain); t_g = timeit(lambda: ops.gemm(a, ww, None, out=x, residual=x))
print("gemm+ln chain us", round(t_chain, 1), "gemm alone", round(t_g, 1), "=> ln after gemm", round(t_chain - t_g, 1))
print("ln weight only  us", round(timeit(lambda: ops.layernorm(x, out=out, weight=w)), 1))
print("ln affine rpb   us", round(timeit(lambda: ops.layernorm(x, out=out, weight=w, bias=b, rows_per_batch=4096)), 1))
w2, b2 = w.clone(), b.clone()
print("ln affine fresh us", round(timeit(lambda: ops.layernorm(x, out=out, weight=w2, bias=b2)), 1))
xf = x.float(); outf = torch.empty_like(xf)
print("ln affine f32io us", round(timeit(lambda: ops.layernorm(xf, out=outf, weight=w, bias=b)), 1))
print("ln plain  f32io us", round(timeit(lambda: ops.layernorm(xf, out=outf)), 1))
