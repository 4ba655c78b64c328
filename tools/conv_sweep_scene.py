"""Record every distinct implicit-GEMM convolution a scene issues (VAE decode + reconstruction), then time each one on every tile that has a
convolution form: auto pick vs best.  Shows where the tile heuristic for convolutions loses time."""
import sys, json, collections
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops, lib as L
from vist3a_amd.t23d import Text23DGS
from vist3a_amd.wan.dit import WAN_1_3B
lib = L.load()
names = [lib.v3a_gemm_tile_name(t).decode() for t in range(lib.v3a_gemm_num_tiles())]
seen = collections.OrderedDict()
real = ops.conv
def spy(x, cw, **kw):
    key = (tuple(x.shape), cw.Cout, cw.k, tuple(kw.get("stride", (1, 1, 1))), tuple(kw.get("pad", (0, 0, 0))), bool(kw.get("ups2", False)), bool(kw.get("replicate", False)),
           kw.get("out_size"), kw.get("residual") is not None, kw.get("act", 0))
    if key not in seen:
        seen[key] = [0, x.clone(), cw, {k: v for k, v in kw.items() if k in ("stride", "pad", "ups2", "replicate", "out_size", "act")}]
    seen[key][0] += 1
    return real(x, cw, **kw)
m = Text23DGS.synthetic(WAN_1_3B, seed=0, device="cuda")
from vist3a_amd.t23d import synthetic_text_embeddings
pe, ne = synthetic_text_embeddings("cuda")
ops.conv = spy
try:
    m.generate(pe, ne, num_inference_steps=2, generator=torch.Generator().manual_seed(0))
finally:
    ops.conv = real
def timeit(fn, iters=4):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
tot_auto = tot_best = 0.0
for key, (cnt, x, cw, kw) in seen.items():
    auto = timeit(lambda: real(x, cw, **kw))
    row = {}
    for t, nm in enumerate(names):
        if nm.startswith("pp"): continue
        try:
            real(x, cw, tile=t, **kw)
        except Exception:
            continue
        row[nm.split("_k64")[0]] = round(timeit(lambda: real(x, cw, tile=t, **kw)), 1)
    b = min(row, key=row.get)
    tot_auto += cnt * auto; tot_best += cnt * row[b]
    print(json.dumps(dict(x=key[0], Cout=key[1], k=key[2], stride=key[3], ups2=key[5], calls=cnt, auto_us=round(auto, 1), best=b, best_us=row[b],
                          lose_us_per_scene=round(cnt * (auto - row[b]), 1))))
print(json.dumps(dict(conv_us_per_scene_auto=round(tot_auto), conv_us_per_scene_best=round(tot_best))))
