"""Record every distinct tile-GEMM a scene issues (2 denoise steps + VAE decode + reconstruction), then time each one on every bf16 tile: auto
pick vs best, weighted by call count (DiT calls scaled to 50 steps).  Shows where v3a_gemm_pick_tile loses time."""
import sys, json, collections
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops, lib as L
from vist3a_amd.t23d import Text23DGS, synthetic_text_embeddings
from vist3a_amd.wan.dit import WAN_1_3B
lib = L.load()
names = [lib.v3a_gemm_tile_name(t).decode() for t in range(lib.v3a_gemm_num_tiles())]
seen = collections.OrderedDict()
real = lib.v3a_gemm_bf16_nt
import ctypes as C
calls = []
def spy(argp, stream):
    a = argp._obj
    key = (a.M, a.N, a.K, bool(a.bias), bool(a.residual), bool(a.scale), a.act, a.flags, bool(a.residual2), a.out_row_group > 0)
    if a.tile < 0 and a.split_k <= 1:
        seen[key] = seen.get(key, 0) + 1
    return real(argp, stream)
m = Text23DGS.synthetic(WAN_1_3B, seed=0, device="cuda")
pe, ne = synthetic_text_embeddings("cuda")
STEPS = 2
class Proxy:
    def __getattr__(self, n):
        return spy if n == "v3a_gemm_bf16_nt" else getattr(lib, n)
orig_load = L.load
L.load = lambda: Proxy()
try:
    m.generate(pe, ne, num_inference_steps=STEPS, generator=torch.Generator().manual_seed(0))
finally:
    L.load = orig_load
bf16 = torch.bfloat16
def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
tot_auto = tot_best = 0.0
g = torch.Generator(device="cuda").manual_seed(0)
for key, cnt in seen.items():
    M, N, K, hb, hr, hs, act, flags, hr2, scat = key
    if scat: continue
    a = torch.randn(M, K, device="cuda", generator=g).to(bf16); w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(bf16)
    brow = bool(flags & L.GEMM_BIAS_ROW)
    b = torch.randn(M if brow else N, device="cuda", generator=g) if hb else None
    r = torch.randn(M, N, device="cuda", generator=g).to(bf16) if hr else None
    out = torch.empty(M, N, device="cuda", dtype=bf16)
    kw = dict(act=act, residual=r, bias_row=brow)
    f = lambda t: ops.gemm(a, w, b, out=out, tile=t, **kw)
    auto_t = lib.v3a_gemm_pick_tile(M, N)
    row = {}
    for t, nm in enumerate(names):
        try:
            f(t)
        except Exception:
            continue
        row[nm.split("_w")[0].split("_l")[0]] = round(timeit(lambda: f(t)), 1)
    an = names[auto_t].split("_w")[0].split("_l")[0]
    bn = min(row, key=row.get)
    weight = cnt * (50 / STEPS if M in (8192, 1536) or N == 8192 else 1)   # DiT launches repeat every step
    tot_auto += weight * row[an]; tot_best += weight * row[bn]
    print(json.dumps(dict(M=M, N=N, K=K, act=act, res=hr, calls_per_scene=round(weight), auto=an, auto_us=row[an], best=bn, best_us=row[bn],
                          lose_us_per_scene=round(weight * (row[an] - row[bn])))))
print(json.dumps(dict(gemm_us_per_scene_auto=round(tot_auto), gemm_us_per_scene_best=round(tot_best))))
