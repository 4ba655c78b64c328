"""GPU-side kernel check + micro-benchmark (run on the MI355X box through gpurun).

Compares every C-ABI kernel against an fp32 PyTorch computation of the same op on the same bf16 inputs and
prints achieved TFLOP/s or GB/s.  Not part of the product path; a development instrument."""
from __future__ import annotations

import argparse
import json
import math
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vist3a_amd import lib as L
from vist3a_amd import ops

dev = "cuda"
bf16, f32 = torch.bfloat16, torch.float32
RESULTS = []


def log(**kw):
    RESULTS.append(kw)
    print(json.dumps(kw), flush=True)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item(), (a - b).abs().max().item()


def gemm_ref(a, w, bias, act, residual, scale, rpb, round_after_scale, bias_row, out_f32):
    v = a.float() @ w.float().t()
    if bias is not None:
        v = v + (bias[:, None] if bias_row else bias[None, :])
    v = v.to(bf16).float()
    if act == L.ACT_GELU_TANH:
        v = torch.nn.functional.gelu(v, approximate="tanh").to(bf16).float()
    elif act == L.ACT_GELU_ERF:
        v = torch.nn.functional.gelu(v).to(bf16).float()
    elif act == L.ACT_SILU:
        v = torch.nn.functional.silu(v).to(bf16).float()
    elif act == L.ACT_RELU:
        v = torch.relu(v)
    if scale is not None:
        if scale.dim() == 2:
            v = v * scale.repeat_interleave(rpb, dim=0)[: v.shape[0]]
        else:
            v = v * scale[None]
        if round_after_scale:
            v = v.to(bf16).float()
    if residual is not None:
        v = v + residual.float()
    return v if out_f32 else v.to(bf16)


def check_gemm():
    g = torch.Generator(device=dev).manual_seed(0)
    nt = L.load().v3a_gemm_num_tiles()
    cases = [
        dict(M=512, N=768, K=256),
        dict(M=300, N=200, K=128),  # ragged M, N
        dict(M=1029, N=1024, K=1024, act=L.ACT_GELU_ERF),
        dict(M=512, N=384, K=192 + 64, act=L.ACT_GELU_TANH, bias=True),
        dict(M=640, N=256, K=128, bias=True, res="bf16", scale="batch", rpb=320),
        dict(M=520, N=256, K=128, bias=True, res="f32", scale="col", round_after_scale=True, out_f32=True),
        dict(M=384, N=1000, K=64, bias=True, bias_row=True),
    ]
    for tile in range(nt):
        for c in cases:
            M, N, K = c["M"], c["N"], c["K"]
            a = torch.randn(M, K, device=dev, generator=g).to(bf16)
            w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(bf16)
            bias = torch.randn(M if c.get("bias_row") else N, device=dev, generator=g) if c.get("bias") else None
            res = None
            if c.get("res"):
                res = torch.randn(M, N, device=dev, generator=g).to(bf16 if c["res"] == "bf16" else f32)
            scale = None
            rpb = c.get("rpb", 0)
            if c.get("scale") == "batch":
                scale = torch.randn((M + rpb - 1) // rpb, N, device=dev, generator=g)
            elif c.get("scale") == "col":
                scale = torch.randn(N, device=dev, generator=g)
            kw = dict(act=c.get("act", 0), residual=res, scale=scale, rows_per_batch=rpb,
                      round_after_scale=c.get("round_after_scale", False), out_f32=c.get("out_f32", False),
                      bias_row=c.get("bias_row", False))
            out = ops.gemm(a, w, bias, tile=tile, **kw)
            torch.cuda.synchronize()
            ref = gemm_ref(a, w, bias, kw["act"], res, scale, rpb, kw["round_after_scale"], kw["bias_row"], kw["out_f32"])
            r, mx = relerr(out, ref)
            log(test="gemm", tile=L.load().v3a_gemm_tile_name(tile).decode(), case={k: v for k, v in c.items()},
                rel=r, maxabs=mx, ok=bool(r < 4e-3 and math.isfinite(r)))


def bench_gemm(shapes=None):
    g = torch.Generator(device=dev).manual_seed(1)
    nt = L.load().v3a_gemm_num_tiles()
    shapes = shapes or [
        (8192, 3072, 1536), (8192, 1536, 1536), (1536, 8192, 1536), (8192, 8960, 1536), (8192, 1536, 8960),
        (13377, 3072, 1024), (13377, 4096, 1024), (13377, 1024, 4096), (4096, 4096, 4096),
    ]
    for (M, N, K) in shapes:
        a = torch.randn(M, K, device=dev, generator=g).to(bf16)
        w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(bf16)
        bias = torch.randn(N, device=dev, generator=g)
        out = torch.empty(M, N, device=dev, dtype=bf16)
        fl = 2.0 * M * N * K
        t = timeit(lambda: torch.nn.functional.linear(a, w, bias.to(bf16)))
        log(test="gemm_bench", M=M, N=N, K=K, impl="torch_linear(hipblaslt)", ms=t * 1e3, tflops=fl / t / 1e12)
        ref = torch.nn.functional.linear(a.float(), w.float(), bias).to(bf16)
        for tile in list(range(nt)) + [-1]:
            t = timeit(lambda: ops.gemm(a, w, bias, out=out, tile=tile))
            r, _ = relerr(out, ref)
            log(test="gemm_bench", M=M, N=N, K=K, impl="hip", tile=(L.load().v3a_gemm_tile_name(tile).decode() if tile >= 0 else "auto"),
                ms=t * 1e3, tflops=fl / t / 1e12, rel=r)


def attn_inputs(B, H, Nq, Nk, D, g):
    q = torch.randn(B, Nq, H * D, device=dev, generator=g).to(bf16)
    k = torch.randn(B, Nk, H * D, device=dev, generator=g).to(bf16)
    v = torch.randn(B, Nk, H * D, device=dev, generator=g).to(bf16)
    nkp = (Nk + 63) // 64 * 64
    vt = torch.zeros(H * D, B * nkp, device=dev, dtype=bf16)
    vt.view(H * D, B, nkp)[:, :, :Nk] = v.permute(2, 0, 1)
    return q, k, v, vt, nkp


def attn_ref(q, k, v, B, H, D):
    qf = q.float().view(B, -1, H, D).transpose(1, 2)
    kf = k.float().view(B, -1, H, D).transpose(1, 2)
    vf = v.float().view(B, -1, H, D).transpose(1, 2)
    o = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf)
    return o.transpose(1, 2).reshape(B, -1, H * D)


def run_attn(q, k, vt, nkp, B, H, Nq, Nk, D, out=None):
    if out is None:
        out = torch.empty(B * Nq, H * D, device=dev, dtype=bf16)
    ops.attention(q.view(B * Nq, H * D), k.view(B * Nk, H * D), vt, out, B=B, H=H, Nq=Nq, Nk=Nk, D=D,
                  q_batch_stride=Nq * H * D, k_batch_stride=Nk * H * D, vt_batch_stride=nkp,
                  o_batch_stride=Nq * H * D)
    return out


def check_attn():
    g = torch.Generator(device=dev).manual_seed(2)
    for (B, H, Nq, Nk, D) in [(1, 2, 256, 256, 128), (2, 3, 200, 333, 128), (2, 2, 384, 512, 64), (3, 2, 1029, 1029, 64),
                              (1, 1, 128, 64, 128), (1, 2, 130, 1, 64)]:
        q, k, v, vt, nkp = attn_inputs(B, H, Nq, Nk, D, g)
        out = run_attn(q, k, vt, nkp, B, H, Nq, Nk, D).view(B, Nq, H * D)
        torch.cuda.synchronize()
        ref = attn_ref(q, k, v, B, H, D)
        r, mx = relerr(out, ref)
        log(test="attn", B=B, H=H, Nq=Nq, Nk=Nk, D=D, rel=r, maxabs=mx, ok=bool(r < 1e-2 and math.isfinite(r)))
    # spiked scores: force large running-max jumps between tiles
    B, H, Nq, Nk, D = 1, 1, 128, 512, 128
    q, k, v, vt, nkp = attn_inputs(B, H, Nq, Nk, D, g)
    k[0, 300] = (q[0, 5].float() * 4).to(bf16)
    k[0, 70] = (q[0, 77].float() * 2).to(bf16)
    out = run_attn(q, k, vt, nkp, B, H, Nq, Nk, D).view(B, Nq, H * D)
    ref = attn_ref(q, k, v, B, H, D)
    r, mx = relerr(out, ref)
    log(test="attn_spike", rel=r, maxabs=mx, ok=bool(r < 1e-2 and math.isfinite(r)))


def bench_attn():
    g = torch.Generator(device=dev).manual_seed(3)
    for (B, H, Nq, Nk, D) in [(2, 12, 4096, 4096, 128), (2, 12, 4096, 512, 128), (1, 16, 13377, 13377, 64), (13, 16, 1029, 1029, 64),
                              (2, 12, 6144, 6144, 128)]:
        q, k, v, vt, nkp = attn_inputs(B, H, Nq, Nk, D, g)
        out = torch.empty(B * Nq, H * D, device=dev, dtype=bf16)
        fl = 4.0 * B * H * Nq * Nk * D
        t = timeit(lambda: run_attn(q, k, vt, nkp, B, H, Nq, Nk, D, out), iters=10)
        log(test="attn_bench", B=B, H=H, Nq=Nq, Nk=Nk, D=D, impl="hip", ms=t * 1e3, tflops=fl / t / 1e12)
        qh = q.view(B, Nq, H, D).transpose(1, 2)
        kh = k.view(B, Nk, H, D).transpose(1, 2)
        vh = v.view(B, Nk, H, D).transpose(1, 2)
        try:
            t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qh, kh, vh), iters=10)
            log(test="attn_bench", B=B, H=H, Nq=Nq, Nk=Nk, D=D, impl="torch_sdpa", ms=t * 1e3, tflops=fl / t / 1e12)
        except Exception as e:  # noqa: BLE001
            log(test="attn_bench", impl="torch_sdpa", error=str(e)[:200])


def check_conv():
    g = torch.Generator(device=dev).manual_seed(9)
    F = torch.nn.functional
    cases = [
        # name, Cin, Cout, k, T,H,W, stride, pad(leading T,H,W), causal, ups2, replicate
        ("causal3x3x3_96", 96, 96, (3, 3, 3), 5, 24, 20, (1, 1, 1), (2, 1, 1), True, False, False),
        ("causal3x3x3_384_192", 384, 192, (3, 3, 3), 3, 16, 16, (1, 1, 1), (2, 1, 1), True, False, False),
        ("conv1x1_192_384", 192, 384, (1, 1, 1), 2, 16, 16, (1, 1, 1), (0, 0, 0), False, False, False),
        ("ups_conv3x3_192_96", 192, 96, (1, 3, 3), 3, 16, 12, (1, 1, 1), (0, 1, 1), False, True, False),
        ("timeconv_384_768", 384, 768, (3, 1, 1), 4, 8, 8, (1, 1, 1), (2, 0, 0), True, False, False),
        ("stitch_16_1024", 16, 1024, (5, 3, 3), 13, 16, 16, (1, 2, 2), (2, 1, 1), False, False, True),
        ("conv7x7_3_128", 3, 128, (1, 7, 7), 2, 28, 28, (1, 1, 1), (0, 3, 3), False, False, False),
        ("conv3x3_96_3", 96, 3, (3, 3, 3), 3, 32, 32, (1, 1, 1), (2, 1, 1), True, False, False),
        ("conv3x3s2_256", 256, 256, (1, 3, 3), 2, 32, 32, (1, 2, 2), (0, 1, 1), False, False, False),
    ]
    for (name, Cin, Cout, k, T, H, W, st, pd, causal, ups2, repl) in cases:
        w = torch.randn(Cout, Cin, *k, device=dev, generator=g) / math.sqrt(Cin * k[0] * k[1] * k[2])
        b = torch.randn(Cout, device=dev, generator=g)
        x = torch.randn(1, Cin, T, H, W, device=dev, generator=g).to(bf16)
        cw = ops.ConvWeight(w.to(bf16), b)
        xcl = torch.zeros(T, H, W, cw.CinP, device=dev, dtype=bf16)
        xcl[..., :Cin] = x[0].permute(1, 2, 3, 0)
        xin = x.float()
        if ups2:
            xin = F.interpolate(xin.transpose(1, 2).reshape(T, Cin, H, W), scale_factor=2.0, mode="nearest-exact").view(1, T, Cin, 2 * H, 2 * W).transpose(1, 2)
        mode = "replicate" if repl else "constant"
        if causal:
            xp = F.pad(xin, (pd[2], pd[2], pd[1], pd[1], pd[0], 0), mode=mode)
        else:
            xp = F.pad(xin, (pd[2], pd[2], pd[1], pd[1], pd[0], pd[0]), mode=mode)
        ref = F.conv3d(xp, w.to(bf16).float(), b, stride=st)
        res = torch.randn(ref.shape[2], ref.shape[3], ref.shape[4], cw.CoutP, device=dev, generator=g).to(bf16)
        y = ops.conv(xcl, cw, stride=st, pad=pd, ups2=ups2, replicate=repl, residual=res)
        torch.cuda.synchronize()
        refcl = (ref[0].permute(1, 2, 3, 0).to(bf16).float() + res[..., :Cout].float()).to(bf16)
        r, mx = relerr(y[..., :Cout], refcl)
        ok = bool(r < 4e-3 and tuple(y.shape[:3]) == tuple(ref.shape[2:]))
        if cw.CoutP != Cout:
            ok = ok and bool((y[..., Cout:].float() - res[..., Cout:].float()).abs().max() == 0)
        log(test="conv", name=name, oshape=list(y.shape), rel=r, maxabs=mx, ok=ok)
    # throughput at the VAE's dominant shapes
    for (Cin, Cout, T, H, W) in [(96, 96, 13, 512, 512), (192, 192, 13, 256, 256), (384, 384, 7, 128, 128)]:
        w = torch.randn(Cout, Cin, 3, 3, 3, device=dev, generator=g) / math.sqrt(Cin * 27)
        cw = ops.ConvWeight(w.to(bf16), torch.zeros(Cout, device=dev))
        x = torch.randn(T, H, W, Cin, device=dev, generator=g).to(bf16)
        out = torch.empty(T, H, W, Cout, device=dev, dtype=bf16)
        fl = 2.0 * T * H * W * Cout * Cin * 27
        for tile in [-1, 0, 2, 3, 4, 5]:
            t = timeit(lambda: ops.conv(x, cw, out=out, pad=(2, 1, 1), tile=tile), iters=5, warm=1)
            log(test="conv_bench", Cin=Cin, Cout=Cout, T=T, H=H, W=W, tile=tile, ms=t * 1e3, tflops=fl / t / 1e12)


def check_norms():
    g = torch.Generator(device=dev).manual_seed(4)
    for (M, d, rpb) in [(8192, 1536, 4096), (1029 * 3, 1024, 0), (100, 2048, 50), (64, 5120, 32), (77, 64, 0)]:
        x = (torch.randn(M, d, device=dev, generator=g) * 2 + 0.5).to(bf16)
        nb = (M + rpb - 1) // rpb if rpb else 1
        sc = torch.randn(nb, d, device=dev, generator=g) * 0.3
        sh = torch.randn(nb, d, device=dev, generator=g) * 0.3
        w = torch.randn(d, device=dev, generator=g)
        b = torch.randn(d, device=dev, generator=g)
        # modulated, no affine
        y = ops.layernorm(x, scale=sc, shift=sh, rows_per_batch=rpb or M, eps=1e-6)
        ln = torch.nn.functional.layer_norm(x.float(), (d,), eps=1e-6)
        idx = (torch.arange(M, device=dev) // (rpb or M))
        ref = (ln * (1 + sc[idx]) + sh[idx]).to(bf16)
        r, mx = relerr(y, ref)
        log(test="layernorm_mod", M=M, d=d, rel=r, maxabs=mx, ok=bool(r < 3e-3))
        # affine, f32 in / f32 out
        xf = x.float() * 1.37
        y = ops.layernorm(xf, weight=w, bias=b, eps=1e-5, out_dtype=f32)
        ref = torch.nn.functional.layer_norm(xf, (d,), w, b, eps=1e-5)
        r, mx = relerr(y, ref)
        log(test="layernorm_affine_f32", M=M, d=d, rel=r, maxabs=mx, ok=bool(r < 1e-5))
    for (B, N, H, hd) in [(2, 4096, 12, 128), (1, 300, 3, 128), (2, 512, 12, 128)]:
        d = H * hd
        M = B * N
        x = torch.randn(M, d, device=dev, generator=g).to(bf16)
        w = torch.randn(d, device=dev, generator=g)
        ang = torch.rand(N, hd // 2, device=dev, generator=g, dtype=torch.float64) * 6.28
        rope = torch.stack([ang.cos(), ang.sin()], -1).float().contiguous()
        y = ops.rmsnorm_rope(x, w, rope=rope, head_dim=hd, tokens_per_batch=N, eps=1e-6)
        xf = x.float()
        n = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * w
        nc = torch.view_as_complex(n.double().view(B, N, H, hd // 2, 2))
        fc = torch.polar(torch.ones_like(ang), ang)[None, :, None, :]
        ref = torch.view_as_real(nc * fc).reshape(M, d).to(bf16)
        r, mx = relerr(y, ref)
        log(test="rmsnorm_rope", B=B, N=N, rel=r, maxabs=mx, ok=bool(r < 3e-3))
        y2 = ops.rmsnorm_rope(x, w, eps=1e-6)
        r, mx = relerr(y2, n.to(bf16))
        log(test="rmsnorm", B=B, N=N, rel=r, maxabs=mx, ok=bool(r < 3e-3))
    # bandwidth
    M, d = 8192, 1536
    x = torch.randn(M, d, device=dev, generator=g).to(bf16)
    sc = torch.randn(2, d, device=dev, generator=g)
    out = torch.empty_like(x)
    t = timeit(lambda: ops.layernorm(x, out=out, scale=sc, shift=sc, rows_per_batch=4096), iters=50)
    log(test="layernorm_bw", M=M, d=d, us=t * 1e6, GBps=2 * M * d * 2 / t / 1e9)
    w = torch.randn(d, device=dev, generator=g)
    rope = torch.randn(4096, 64, 2, device=dev, generator=g)
    t = timeit(lambda: ops.rmsnorm_rope(x, w, out=out, rope=rope, head_dim=128, tokens_per_batch=4096), iters=50)
    log(test="rmsnorm_rope_bw", M=M, d=d, us=t * 1e6, GBps=2 * M * d * 2 / t / 1e9)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="check_gemm,check_attn,check_norms,bench_gemm,bench_attn")
    ap.add_argument("--out", default="gpurun_out/kcheck.jsonl")
    a = ap.parse_args()
    print(torch.cuda.get_device_name(0), L.load().v3a_build_info().decode(), flush=True)
    t0 = time.time()
    for name in a.what.split(","):
        try:
            globals()[name]()
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            log(test=name, error=str(e)[:300], ok=False)
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text("\n".join(json.dumps(r) for r in RESULTS))
    bad = [r for r in RESULTS if r.get("ok") is False]
    print(f"done in {time.time() - t0:.1f}s; {len(bad)} failing checks", flush=True)
    for r in bad:
        print("FAIL", json.dumps(r))
