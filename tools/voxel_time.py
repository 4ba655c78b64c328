"""Voxel fusion (R13) at production size: M = 13 x 448^2 points, 84 feature columns; time and parity vs a torch scatter reference
for a clustered cloud (what seeded synthetic weights give: ~70 points per voxel) and a spread one (what real scenes give: 1-3)."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M, C = 13 * 448 * 448, 83
for name, spread in (("clustered", 0.03), ("spread", 0.25), ("very_spread", 1.5)):
    pts = (torch.randn(M, 3, device=dev, generator=g) * spread).contiguous()
    feat = torch.randn(M, C + 1, device=dev, generator=g).contiguous()
    v = ops.voxelize_fuse(pts, feat, C, C, 0.002)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        v = ops.voxelize_fuse(pts, feat, C, C, 0.002)
    e1.record(); torch.cuda.synchronize()
    U = v["keys"].shape[0]
    inv = v["inverse"].long()
    conf = feat[:, C]
    mx = torch.full((U,), -float("inf"), device=dev).scatter_reduce(0, inv, conf, "amax")
    ex = torch.exp(conf - mx[inv])
    den = torch.zeros(U, device=dev).index_add_(0, inv, ex) + 1e-6
    w = ex / den[inv]
    rf = torch.zeros(U, C, device=dev).index_add_(0, inv, feat[:, :C] * w[:, None])
    rp = torch.zeros(U, 3, device=dev).index_add_(0, inv, pts * w[:, None])
    ef = ((v["voxel_feat"][:, :C] - rf).norm() / rf.norm()).item()
    ep = ((v["voxel_pts"] - rp).norm() / rp.norm()).item()
    print(json.dumps(dict(case=name, M=M, U=U, pts_per_voxel=round(M / U, 2), ms_whole_call=round(e0.elapsed_time(e1) / 5, 3), rel_feat=ef, rel_pts=ep)), flush=True)
