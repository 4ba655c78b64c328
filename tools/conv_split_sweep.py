"""Sweep the implicit-GEMM tiles (and the halo form) of v3a_conv_split over the DPT heads' layer shapes at 13 views @448:
best-of-3 x 5 launches per (layer, tile).  Prints one JSON line per layer: {shape, auto_ms, per-tile ms}."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import lib as L, ops

S = 13
LAYERS = [  # (name, Cin, Cout, k, H, W, stride)
    ("proj 2048->256 1x1 @32", 2048, 256, 1, 32, 32, 1), ("proj 2048->1024 1x1 @32", 2048, 1024, 1, 32, 32, 1),
    ("down 1024->1024 3x3 s2 @32", 1024, 1024, 3, 32, 32, 2), ("rn3 1024->256 3x3 @16", 1024, 256, 3, 16, 16, 1),
    ("rn2 1024->256 3x3 @32", 1024, 256, 3, 32, 32, 1), ("fus 256->256 3x3 @16", 256, 256, 3, 16, 16, 1),
    ("fus 256->256 3x3 @32", 256, 256, 3, 32, 32, 1), ("fus 256->256 3x3 @64", 256, 256, 3, 64, 64, 1),
    ("out 256->256 1x1 @128", 256, 256, 1, 128, 128, 1), ("up0 256->1024 1x1 @32", 256, 1024, 1, 32, 32, 1),
    ("merger 8->128 7x7 @448", 8, 128, 7, 448, 448, 1), ("oc22 128->88 1x1 @448", 128, 88, 1, 448, 448, 1), ("oc22 32->8 1x1 @448", 32, 8, 1, 448, 448, 1),
]
only = sys.argv[1:] 
lib = L.load()
nt = lib.v3a_gemm_num_tiles()
g = torch.Generator(device="cuda").manual_seed(0)
for name, cin, cout, k, H, W, st in LAYERS:
    if only and not any(o in name for o in only):
        continue
    cw = ops.ConvWeightSplit(torch.randn(cout, cin, k, k) * (cin * k * k) ** -0.5, torch.randn(cout) * 0.1)
    x = torch.randn(2, S, H, W, cw.CinP, device="cuda", generator=g).to(torch.bfloat16)
    kw = dict(stride=(1, st, st), pad=(0, k // 2, k // 2))

    def t(tile):
        try:
            ops.conv_split(x, cw, tile=tile, **kw)
        except RuntimeError:
            return None
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.conv_split(x, cw, tile=tile, **kw)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        return round(best, 4)
    res = {"layer": name, "auto": t(-1)}
    for ti in range(nt):
        r = t(ti)
        if r is not None:
            res[lib.v3a_gemm_tile_name(ti).decode()] = r
    h = t(-2)
    if h is not None:
        res["halo"] = h
    print(json.dumps(res), flush=True)
