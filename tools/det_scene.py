"""Run-to-run determinism of the whole scene pipeline and of its stages at production size."""
import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.t23d import Text23DGS, synthetic_text_embeddings
m = Text23DGS.synthetic(seed=0)
pe, ne = synthetic_text_embeddings("cuda")
lat0 = torch.randn(1, 16, 4, 64, 64, generator=torch.Generator().manual_seed(1))
res = []
for _ in range(2):
    out, lat, clip = m.generate(pe, ne, latents=lat0, num_inference_steps=6)
    g = out.gaussians
    res.append(dict(lat=lat.clone(), clip=clip.clone(), means=g.means.clone(), cov=g.covariances.clone(), sh=g.harmonics.clone(), op=g.opacities.clone(),
                    pose=out.last_pred_pose_enc.clone(), depth=out.depth_dict["depth"].clone()))
torch.cuda.synchronize()
for k in res[0]:
    a, b = res[0][k], res[1][k]
    same = a.shape == b.shape and torch.equal(a, b)
    print(f"{k:6s} identical={same}" + ("" if same else f" shapes {tuple(a.shape)} {tuple(b.shape)} max|d| {float((a.float()-b.float()).abs().max()) if a.shape==b.shape else 'n/a'}"))
