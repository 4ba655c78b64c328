"""Time + check the flash-attention launch (default: the DiT self-attention, B=2, H=12, N=4096, D=128) on seeded N(0, 1/4) data.
  python tools/attn_time.py 2x12x4096 2x12x6144          shapes as BxHxN
  V3A_LIB=<other build>                                   A/B against another library build in a second process (vist3a_amd/lib.py)
  V3A_ATTN_SAVE=/tmp/prefix                               first process saves the output, later ones compare bit-for-bit with it and
                                                          print, for differing rows, which build is closer to an fp64 softmax
  V3A_ATTN_D=64  V3A_ATTN_PERIOD=1032,1029                the reconstruction's head dim / padded view layout (kv_period, kv_valid)"""
import sys, json, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import lib, ops
shapes = [(2, 12, 4096)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
D = int(os.environ.get("V3A_ATTN_D", "128"))
L = lib.load()
for B, H, N in shapes:
    d = H * D
    g = torch.Generator(device="cuda").manual_seed(0)
    q = (torch.randn(B * N, d, device="cuda", generator=g) * 0.5).bfloat16()
    k = (torch.randn(B * N, d, device="cuda", generator=g) * 0.5).bfloat16()
    v = torch.randn(B * N, d, device="cuda", generator=g).bfloat16()
    vt = torch.empty(d, B * N, device="cuda", dtype=torch.bfloat16)
    for b in range(B):
        vt[:, b * N:(b + 1) * N] = v[b * N:(b + 1) * N].t()
    outs = {}
    for which in (1,):
        o = torch.empty(B * N, d, device="cuda", dtype=torch.bfloat16)
        per = [int(x) for x in os.environ.get("V3A_ATTN_PERIOD", "0,0").split(",")]   # kv_period,kv_valid (the reconstruction's padded view layout)
        nk = int(os.environ.get("V3A_ATTN_NK", N))   # cross-attention: fewer keys than queries, with a per-key bias (the merged padding key)
        kb = torch.zeros(B, (nk + 63) // 64 * 64, device="cuda") if os.environ.get("V3A_ATTN_NK") else None
        if kb is not None: kb[:, nk - 1] = 6.0
        run = lambda: ops.attention(q, k, vt, o, B=B, H=H, Nq=N, Nk=nk, D=D, q_batch_stride=N * d, k_batch_stride=N * d, vt_batch_stride=N, o_batch_stride=N * d,
                                    kv_period=per[0], kv_valid=per[1], key_bias=kb)
        for _ in range(20): run()
        torch.cuda.synchronize()
        times = []
        for _ in range(6):   # the first rounds run on cold clocks: report all, quote the best
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run()
            e1.record(); torch.cuda.synchronize()
            times.append(round(e0.elapsed_time(e1) / 50 * 1e3, 1))
        us = min(times)
        qf, kf, vf = q[:N, :D].float(), k[:N, :D].float(), v[:N, :D].float()
        ref = torch.softmax(qf @ kf.t() * D ** -0.5, -1) @ vf
        rel = ((o[:N, :D].float() - ref).norm() / ref.norm()).item()
        outs[which] = o.clone()
        sv = os.environ.get("V3A_ATTN_SAVE")
        if sv:
            f = Path(f"{sv}_{B}x{H}x{N}.pt")
            if f.exists():
                ref_o = torch.load(f)
                bad = ((ref_o.float() - o.cpu().float()).abs().view(B, N, H, D) > 0).any(dim=3).nonzero()[:8]
                for bb, nn, hh in bad.tolist():   # which of the two is closer to an fp64 softmax of that row?
                    qd = q[bb * N + nn, hh * D:(hh + 1) * D].double(); kd = k[bb * N:(bb + 1) * N, hh * D:(hh + 1) * D].double(); vd = v[bb * N:(bb + 1) * N, hh * D:(hh + 1) * D].double()
                    r64 = torch.softmax(kd @ qd * D ** -0.5, 0) @ vd
                    e_saved = (ref_o.view(B, N, H, D)[bb, nn, hh].double().cuda() - r64).norm() / r64.norm()
                    e_this = (o.view(B, N, H, D)[bb, nn, hh].double() - r64).norm() / r64.norm()
                    print(json.dumps(dict(row=(bb, nn, hh), rel_saved=float(e_saved), rel_this=float(e_this))), flush=True)
                df = (ref_o.float() - o.cpu().float()).abs().view(B, N, H, D)
                print(json.dumps(dict(vs_saved_bit_identical=bool(torch.equal(ref_o, o.cpu())), differing=int((df > 0).sum()), max_abs=float(df.max()),
                                      per_batch_head=(df > 0).sum(dim=(1, 3)).tolist(), rows_differing=int((df > 0).any(dim=3).sum()),
                                      first=[(tuple(int(x) for x in ix), float(ref_o.view(B, N, H, D)[tuple(ix)]), float(o.cpu().view(B, N, H, D)[tuple(ix)])) for ix in (df > 0).nonzero()[:24]])), flush=True)
            else:
                torch.save(o.cpu(), f)
        print(json.dumps(dict(B=B, H=H, N=N, kernel=which, us=round(us, 1), rounds_us=times, tflops=round(4 * B * H * N * N * D / us / 1e6), rel=rel)), flush=True)
