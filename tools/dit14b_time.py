"""One CFG step of the full-size Wan-14B DiT (BASELINE config #4 geometry, single GPU): bf16, fp8 attention, fp8 attention + fp8 GEMMs."""
import sys, time, json
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.wan.dit import WanDiT, WAN_14B
from vist3a_amd.wan.weights import random_dit_state_dict
cfg = WAN_14B
sd = random_dit_state_dict(cfg, seed=0, device="cuda")
m = WanDiT(cfg, sd)
del sd
torch.cuda.empty_cache()
lat = torch.randn(2, 16, 4, 64, 64, device="cuda").bfloat16()
text = torch.zeros(2, 512, 4096, device="cuda")
text[0, :64] = torch.randn(64, 4096, device="cuda") * 0.1
text[1, :80] = torch.randn(80, 4096, device="cuda") * 0.1
t = torch.tensor([900, 900], device="cuda")
N, d, ffn, L, ctx = 4096, cfg.dim, cfg.ffn_dim, cfg.num_layers, 512
fl = 2 * L * (8 * N * d * d + 4 * N * N * d + 4 * N * d * d + 4 * N * ctx * d + 4 * N * d * ffn)
outs = {}
for attn, gemm in (("bf16", "bf16"), ("fp8", "bf16"), ("fp8", "fp8")):   # config #4: e4m3 MFMA for the attention and the block projections
    m.attn_dtype = attn
    if gemm == "fp8":
        m.enable_fp8_gemm()
    a = m(lat, t, text)[0].clone()
    b = m(lat, t, text)[0].clone()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): m(lat, t, text)
    torch.cuda.synchronize(); ms = (time.time() - t0) / 3 * 1e3
    outs[(attn, gemm)] = a
    print(json.dumps(dict(attention=attn, gemm=gemm, ms_per_cfg_pair=round(ms, 1), model_tflops=round(fl / ms / 1e9, 1), deterministic=bool(torch.equal(a, b)),
                          finite=bool(torch.isfinite(a.float()).all()), peak_GB=round(torch.cuda.max_memory_allocated() / 2**30, 1))), flush=True)
ref = outs[("bf16", "bf16")].float()
rel = lambda o: ((o.float() - ref).norm() / ref.norm()).item()
print(json.dumps(dict(fp8_attention_vs_bf16_forward_rel=rel(outs[("fp8", "bf16")]), fp8_attention_and_gemm_vs_bf16_forward_rel=rel(outs[("fp8", "fp8")]))))
