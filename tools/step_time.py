"""Where a denoise step goes: graphed DiT forward alone vs the whole pipeline step (CFG combine + UniPC update + casts)."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.wan.dit import WanDiT, WAN_1_3B, GraphedWanDiT
from vist3a_amd.wan.weights import random_dit_state_dict
from vist3a_amd.wan.pipeline import WanT2VPipeline
from vist3a_amd.wan.scheduler import UniPCMultistepScheduler
from vist3a_amd.t23d import synthetic_text_embeddings

cfg = WAN_1_3B
m = WanDiT(cfg, random_dit_state_dict(cfg, seed=0, device="cuda"))
gm = GraphedWanDiT(m)
pe, ne = synthetic_text_embeddings("cuda")
text = torch.cat([pe, ne], 0).contiguous()
lat = torch.randn(2, 16, 4, 64, 64, device="cuda").bfloat16()
t = torch.tensor([900, 900], device="cuda")
def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3
r = dict(eager_forward_ms=timeit(lambda: m(lat, t, text), 10), graph_forward_ms=timeit(lambda: gm(lat, t, text), 10))
for name, fn in (("eager", m), ("graph", gm)):
    pipe = WanT2VPipeline(fn, UniPCMultistepScheduler(flow_shift=5.0))
    lat0 = torch.randn(1, 16, 4, 64, 64)
    run = lambda: pipe(prompt_embeds=pe, negative_prompt_embeds=ne, height=512, width=512, num_frames=13, num_inference_steps=50, guidance_scale=7.5, latents=lat0)
    r[name + "_pipeline_step_ms"] = timeit(run, 2) / 50
print(json.dumps(r))
