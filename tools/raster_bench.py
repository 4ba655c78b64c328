"""Kernel-level timing of the rasteriser on a production-size synthetic scene (U Gaussians on a noisy sphere shell, 448^2)."""
import sys, time, json, math
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops

U = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.004
W = H = 448
g = torch.Generator(device="cuda").manual_seed(0)
d = torch.randn(U, 3, device="cuda", generator=g)
d = d / d.norm(dim=-1, keepdim=True)
means = d * (1.0 + 0.05 * torch.randn(U, 1, device="cuda", generator=g))
A = torch.randn(U, 3, 3, device="cuda", generator=g) * sigma
cov = (A @ A.transpose(1, 2) + 1e-8 * torch.eye(3, device="cuda")).contiguous()
sh = (torch.randn(U, 3, 25, device="cuda", generator=g) * 0.2).contiguous()
op = torch.rand(U, device="cuda", generator=g)
Cn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
views = []
for i in range(Cn):  # orbit
    a = 2 * math.pi * i / 132
    v = torch.eye(4); v[:3, :3] = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]]); v[2, 3] = 3.0
    views.append(v)
view = torch.stack(views)
K = torch.tensor([[400.0, 0, 224], [0, 400.0, 224], [0, 0, 1]])[None].repeat(Cn, 1, 1).contiguous()
campos = torch.linalg.inv(view)[:, :3, 3].contiguous().cuda()
view, K = view.cuda(), K.cuda()
ws = ops.GsWorkspace()
bg = torch.ones(3, device="cuda")
def frame():
    pr = ops.gs_project(means, cov, sh, view, campos, K, W, H)
    return ops.gs_rasterize(pr, op, W, H, background=bg, workspace=ws)
for _ in range(3):
    r = frame()
torch.cuda.synchronize()
n = 20
t0 = time.time()
for _ in range(n):
    r = frame()
torch.cuda.synchronize()
ms = (time.time() - t0) / n * 1e3 / Cn
print(json.dumps(dict(cameras_per_launch=Cn, U=U, sigma=sigma, ms_per_camera=ms, n_isect=r["n_isect"], alpha_mean=float(r["alpha"].mean()))))
