"""BASELINE config #3 geometry on one GPU: 21 views (6 latent frames, 6144 DiT tokens), few denoise steps; twice, for determinism."""
import sys, time, json
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.t23d import Text23DGS, synthetic_text_embeddings, SceneTimes
m = Text23DGS.synthetic(seed=0)
pe, ne = synthetic_text_embeddings("cuda")
lat0 = torch.randn(1, 16, 6, 64, 64, generator=torch.Generator().manual_seed(2))
outs = []
for _ in range(2):
    st = SceneTimes()
    out, lat, clip = m.generate(pe, ne, latents=lat0, num_frames=21, num_inference_steps=4, timings=st)
    outs.append((lat.clone(), out.gaussians.means.clone(), out.last_pred_pose_enc.clone()))
torch.cuda.synchronize()
print(json.dumps(dict(views=int(out.last_pred_pose_enc.shape[1]), clip=list(clip.shape), gaussians=int(out.gaussians.means.shape[1]),
                      denoise_ms_per_step=round(st.denoise_ms / 4, 1), vae_ms=round(st.vae_ms, 1), recon_ms=round(st.recon_ms, 1),
                      deterministic=all(torch.equal(a, b) for a, b in zip(*outs)), finite=bool(torch.isfinite(out.gaussians.means).all()),
                      peak_GB=round(torch.cuda.max_memory_allocated() / 2**30, 1))))
