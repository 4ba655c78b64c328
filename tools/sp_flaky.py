import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.wan.dit import WanDiT, WanDiTConfig
from vist3a_amd.wan.pipeline import WanT2VPipeline
from vist3a_amd.wan.scheduler import UniPCMultistepScheduler
from vist3a_amd.wan.seqpar import DenoisePlan, ThreadWorld
TINY = dict(num_attention_heads=2, attention_head_dim=128, ffn_dim=512, num_layers=2, text_dim=128, freq_dim=64)
from vist3a_amd.wan.weights import random_dit_state_dict
ocfg = WanDiTConfig(**TINY)
model = WanDiT(ocfg, random_dit_state_dict(ocfg, seed=3, device="cuda"), device="cuda")
g = torch.Generator().manual_seed(9)
pe = torch.randn(1, 32, ocfg.text_dim, generator=g) * 0.5
ne = torch.randn(1, 32, ocfg.text_dim, generator=g) * 0.5
lat0 = torch.randn(1, 16, 2, 16, 16, generator=g)
kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, height=128, width=128, num_frames=5, num_inference_steps=4, guidance_scale=6.0, latents=lat0)
ref = WanT2VPipeline(model, UniPCMultistepScheduler(flow_shift=5.0))(**kw)["frames"].clone()
import sys as _s
for world in (1, 2, 3, 4):
    bad = 0
    kw_w = dict(kw)
    if world == 3:
        lat3 = torch.randn(1, 16, 3, 16, 32, generator=torch.Generator().manual_seed(5))
        kw_w.update(latents=lat3, height=128, width=256, num_frames=9)
    ref_w = WanT2VPipeline(model, UniPCMultistepScheduler(flow_shift=5.0))(**kw_w)["frames"].clone()
    for it in range(30):
        if world == 1:
            outs = [WanT2VPipeline(model, UniPCMultistepScheduler(flow_shift=5.0))(**kw_w)["frames"].clone()]
        else:
            plans = DenoisePlan.from_threads(world)
            runner = ThreadWorld(world)
            outs = runner.run(lambda r: WanT2VPipeline(model, UniPCMultistepScheduler(flow_shift=5.0), plan=plans[r])(**kw_w)["frames"].clone())
        torch.cuda.synchronize()
        if any(float((o - ref_w).abs().max()) != 0 for o in outs):
            bad += 1
    print("world", world, "bad iterations", bad, "of 30", flush=True)
