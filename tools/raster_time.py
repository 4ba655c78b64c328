"""Time the orbit render of one synthetic scene: 132 cameras @448^2 over the Gaussians of the bench scene."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.t23d import Text23DGS, synthetic_text_embeddings
from vist3a_amd.misc.image_io import interpolate_camera_path

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = Text23DGS.synthetic(seed=0)
pe, ne = synthetic_text_embeddings("cuda")
lat0 = torch.randn(1, 16, 4, 64, 64, generator=torch.Generator().manual_seed(1))
out, _, _ = m.generate(pe, ne, latents=lat0, num_inference_steps=steps)
g = out.gaussians
U = g.means.shape[1]
ex, ix = interpolate_camera_path(out.pred_context_pose["extrinsic"], out.pred_context_pose["intrinsic"], 1, 10)
dec = m.stitched_decoder.stitched_3d_model.decoder
near = torch.ones(1, ex.shape[1], device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    o = dec.forward(g, ex, ix.float(), near * 0.1, near * 100, (448, 448))
    torch.cuda.synchronize(); dt = time.time() - t0
ni = dec.last_n_isect
print(json.dumps(dict(U=U, cameras=ex.shape[1], render_ms=dt * 1e3, ms_per_camera=dt * 1e3 / ex.shape[1], isect_mean=sum(ni) / len(ni), isect_max=max(ni),
                      alpha_mean=float(o.alpha.mean()), scales_median=float(g.scales.median()), finite=bool(torch.isfinite(o.color).all()))))
