"""Time the full-size stitched reconstruction forward (13 views @448) on the GPU box, stage by stage."""
import sys, json, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.recon.engine import ReconCfg, ReconEngine
from vist3a_amd.recon.weights import random_recon_state_dict, round_aggregator_to_bf16
S = int(sys.argv[1]) if len(sys.argv) > 1 else 13
cfg = ReconCfg()
t0 = time.time()
sd = round_aggregator_to_bf16(random_recon_state_dict(cfg, seed=0, device="cuda"))
eng = ReconEngine(cfg, sd)
del sd
torch.cuda.synchronize()
print("build", time.time() - t0, "s", flush=True)
H = W = 448
lat = torch.randn(1, 1024, S, 32, 32, device="cuda") * 0.5
img = torch.rand(1, 3, S, H, W, device="cuda") * 2 - 1
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(3):
    x, g = eng.token_workspace(S, H, W)
    e0 = ev()
    tok = lat[0].permute(1, 2, 3, 0).reshape(S, 1024, 1024).bfloat16()
    x.view(S, g["Pp"], 1024)[:, 5:5 + 1024] = (tok.float() + g["pos_patch"].float()[None]).bfloat16()
    img_cl = torch.zeros(S, H, W, 8, device="cuda", dtype=torch.bfloat16)
    img_cl[..., :3] = ((img[0].permute(1, 2, 3, 0) + 1) / 2)
    e1 = ev(); eng.backbone(g, S)
    e2 = ev(); poses = eng.camera(g, S)
    e3 = ev(); depth, dconf, pts, raw, ext, K = eng.heads(g, S, H, W, img_cl, poses[-1])
    e4 = ev()
    from vist3a_amd import ops
    v = ops.voxelize_fuse(pts.view(-1, 3), raw, 83, 83, cfg.voxel_size)
    e5 = ev(); gs = ops.gaussian_adapter(v["voxel_pts"], v["voxel_feat"], eng.sh_mask, 4, 1.0)
    e6 = ev(); torch.cuda.synchronize()
    ms = lambda a, b: round(a.elapsed_time(b), 2)
    print(json.dumps(dict(prep=ms(e0, e1), backbone=ms(e1, e2), camera=ms(e2, e3), heads=ms(e3, e4), voxel=ms(e4, e5), adapter=ms(e5, e6),
                          total=ms(e0, e6), U=int(v["voxel_pts"].shape[0]), finite=bool(torch.isfinite(gs["means"]).all() and torch.isfinite(depth).all()),
                          peak_GB=round(torch.cuda.max_memory_allocated() / 2**30, 1))), flush=True)
