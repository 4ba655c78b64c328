"""Time the full-size Wan VAE decode (T_lat=4 -> 13 x 512^2) on the GPU box."""
import sys, json, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.wan.vae import WanVAEConfig, WanVAEDecoder
from vist3a_amd.t23d import random_vae_decoder_state_dict, random_vae_encoder_state_dict
sd = random_vae_decoder_state_dict(WanVAEConfig(), seed=0)
dec = WanVAEDecoder(WanVAEConfig(), sd)
Tl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
z = torch.randn(1, 16, Tl, 64, 64, device="cuda")
for _ in range(2):
    o = dec.decode(z)[0]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 3
for _ in range(n):
    o = dec.decode(z)[0]
e1.record(); torch.cuda.synchronize()
print(json.dumps(dict(vae_decode_ms=e0.elapsed_time(e1) / n, shape=list(o.shape), finite=bool(torch.isfinite(o.float()).all()),
                      peak_mem_GB=torch.cuda.max_memory_allocated() / 2**30)))

# encoder (StitchVAE3D.forward path): 13 views @512^2 -> latent [1,32,4,64,64]
from vist3a_amd.wan.vae import WanVAEEncoder
enc = WanVAEEncoder(WanVAEConfig(), random_vae_encoder_state_dict(WanVAEConfig(), seed=1))
x = (torch.rand(1, 3, 1 + 4 * (Tl - 1), 512, 512, device="cuda") * 2 - 1)
for _ in range(2):
    p = enc.encode_params(x)
torch.cuda.synchronize()
e0.record()
for _ in range(n):
    p = enc.encode_params(x)
e1.record(); torch.cuda.synchronize()
print(json.dumps(dict(vae_encode_ms=e0.elapsed_time(e1) / n, shape=list(p.shape), finite=bool(torch.isfinite(p).all()))))
