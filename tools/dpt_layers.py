"""Per-launch times of the fp32-equivalent DPT heads at production size (13 views @448): every ops.* call inside ReconEngine.heads is
wrapped with HIP events (one synchronise at the end).  Prints the launches sorted by time with their shapes and useful TFLOP/s
(2 M N K of the fp32 convolution; the split kernel executes 3x that on the matrix pipe)."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops
from vist3a_amd.recon.engine import ReconCfg, ReconEngine
from vist3a_amd.recon.weights import random_recon_state_dict, round_aggregator_to_bf16

S = int(sys.argv[1]) if len(sys.argv) > 1 else 13
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
cfg = ReconCfg(dpt_precision=prec)
sd = round_aggregator_to_bf16(random_recon_state_dict(cfg, seed=0, device="cuda"))
eng = ReconEngine(cfg, sd)
del sd
H = W = 448
g = eng._geometry(S, H, W)
for t in g["taps"]:
    t.normal_()
img = torch.rand(S, H, W, 8, device="cuda")
pose = torch.tensor([[0.1, 0.2, 0.3, 0, 0, 0, 1, 1.0, 1.0]], device="cuda").repeat(S, 1)
log = []
names = ["conv_split", "conv", "bilinear_cl_pair", "bilinear_cl", "layernorm_pair", "layernorm", "split_f32", "depth_unproject"]
orig = {n: getattr(ops, n) for n in names}


def wrap(n):
    f = orig[n]

    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = f(*a, **k)
        e1.record()
        d = dict(op=n, e=(e0, e1))
        if n in ("conv_split", "conv"):
            x, cw = a[0], a[1]
            o = r if r.dtype == torch.float32 or n == "conv" else r[0]
            M = o.numel() // o.shape[-1]
            if k.get("out_rows"):
                M = (x[0] if n == "conv_split" else x).numel() // x.shape[-1]
            Kc = cw.k[0] * cw.k[1] * cw.k[2] * cw.CinP
            d.update(M=M, N=cw.CoutP, K=Kc, k=cw.k[1], flop=2.0 * M * cw.CoutP * Kc)
        else:
            d.update(bytes=sum(t.numel() * t.element_size() for t in list(a) + [r] if torch.is_tensor(t)))
        log.append(d)
        return r
    return w


for it in range(3):
    log.clear()
    for n in names:
        setattr(ops, n, wrap(n))
    eng.heads(g, S, H, W, img, pose)
    torch.cuda.synchronize()
    for n in names:
        setattr(ops, n, orig[n])
tot = 0.0
rows = []
for d in log:
    ms = d["e"][0].elapsed_time(d["e"][1])
    tot += ms
    d["ms"] = round(ms, 4)
    del d["e"]
    if "flop" in d:
        d["useful_TF"] = round(d["flop"] / ms / 1e9, 1)
        del d["flop"]
    else:
        d["GBs"] = round(d["bytes"] / ms / 1e6, 0)
    rows.append(d)
for d in sorted(rows, key=lambda r: -r["ms"]):
    print(json.dumps(d))
conv = sum(d["ms"] for d in rows if d["op"].startswith("conv"))
print(json.dumps(dict(total_ms=round(tot, 3), conv_ms=round(conv, 3), other_ms=round(tot - conv, 3), launches=len(rows), precision=prec)))
