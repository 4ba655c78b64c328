"""VAE-decoder oracle pinned against the golden vector produced by the reference module itself."""
from pathlib import Path

import torch
from safetensors.torch import load_file

from oracle import wan_vae as OV

GOLD = Path(__file__).parent / "golden" / "vae_decode_tiny.safetensors"


def test_oracle_matches_reference_golden():
    g = load_file(str(GOLD))
    cfg = OV.WanVAEConfig(base_dim=16)
    sd = OV.make_weights(cfg, seed=11)
    out = OV.decode(sd, cfg, g["z"])
    assert out.shape == g["out"].shape == (1, 3, 9, 64, 64)
    assert torch.allclose(out, g["out"], atol=3e-5, rtol=1e-5), (out - g["out"]).abs().max()


def test_frame_count_and_causality():
    cfg = OV.WanVAEConfig(base_dim=16)
    sd = OV.make_weights(cfg, seed=3)
    z = torch.randn(1, 16, 4, 8, 8)
    full = OV.decode(sd, cfg, z)
    assert full.shape[2] == 1 + 4 * 3
    # causal in time: truncating the latent clip must not change the earlier frames
    part = OV.decode(sd, cfg, z[:, :, :2])
    assert torch.allclose(full[:, :, :5], part, atol=1e-5)


def test_production_plan():
    d0, plan = OV.WanVAEConfig().decoder_plan()
    assert d0 == 384
    assert plan == [(384, 384, "upsample3d"), (192, 384, "upsample3d"), (192, 192, "upsample2d"), (96, 96, None)]


ENC_GOLD = Path(__file__).parent / "golden" / "vae_encode_tiny.safetensors"


def test_encoder_oracle_matches_reference_golden():
    """Whole-clip restatement == the reference's chunked (1,4,4,...) cached `_encode`, on two clips (9 and 5 frames, one non-square)."""
    g = load_file(str(ENC_GOLD))
    cfg = OV.WanVAEConfig(base_dim=16)
    sd = OV.make_encoder_weights(cfg, seed=12)
    out = OV.encode(sd, cfg, g["x"])
    assert out.shape == g["out"].shape == (1, 32, 3, 8, 8)
    assert torch.allclose(out, g["out"], atol=2e-5, rtol=1e-5), (out - g["out"]).abs().max()
    out2 = OV.encode(sd, cfg, g["x2"])
    assert out2.shape == g["out2"].shape == (1, 32, 2, 4, 6)
    assert torch.allclose(out2, g["out2"], atol=2e-5, rtol=1e-5)


def test_encoder_plan_causality_and_posterior():
    dims, plan = OV.encoder_plan(OV.WanVAEConfig())
    assert dims == [96, 96, 192, 384, 384]
    assert [p for p in plan if p[0] == "down"] == [("down", 96, "downsample2d"), ("down", 192, "downsample3d"), ("down", 384, "downsample3d")]
    assert len(plan) == 11
    cfg = OV.WanVAEConfig(base_dim=16)
    sd = OV.make_encoder_weights(cfg, seed=4)
    x = torch.randn(1, 3, 9, 32, 32).clamp(-1, 1)
    full = OV.encode(sd, cfg, x)
    part = OV.encode(sd, cfg, x[:, :, :5])  # causal in time: dropping later frames leaves earlier latents unchanged
    assert full.shape[2] == 3 and torch.allclose(full[:, :, :2], part, atol=1e-5)
    one = OV.encode(sd, cfg, x[:, :, :1])   # a single image is a 1-frame clip
    assert torch.allclose(full[:, :, :1], one, atol=1e-5)
    params = torch.cat([torch.full((1, 2, 1, 1, 1), 3.0), torch.tensor([0.0, 40.0]).view(1, 2, 1, 1, 1)], 1)
    z = OV.posterior_sample(params, torch.ones(1, 2, 1, 1, 1))
    assert torch.allclose(z.flatten(), torch.tensor([3.0 + 1.0, 3.0 + float(torch.exp(torch.tensor(10.0)))]))  # logvar clamped to 20


def test_bf16_contract_mode_is_a_rounding_of_the_pinned_form():
    """emulate_bf16 = the same arithmetic with CUDA autocast's bf16 rounding points: off by default (the golden-pinned form is
    untouched), within bf16 noise of the fp32 form, every conv input / output bf16-representable, and the global switch restored."""
    g = load_file(str(GOLD))
    cfg = OV.WanVAEConfig(base_dim=16)
    sd = OV.make_weights(cfg, seed=11)
    ref = OV.decode(sd, cfg, g["z"])
    emu = OV.decode(sd, cfg, g["z"], emulate_bf16=True)
    assert not OV._EMU and torch.equal(OV.decode(sd, cfg, g["z"]), ref)
    rel = ((emu - ref).norm() / ref.norm()).item()
    assert 1e-3 < rel < 4e-2, rel                                    # ~35 bf16-rounded conv layers
    assert torch.equal(emu, emu.to(torch.bfloat16).float())          # the output of the last conv is a bf16 value (clamp keeps it one)
    esd = OV.make_encoder_weights(cfg, seed=12)
    ge = load_file(str(ENC_GOLD))
    e32, e16 = OV.encode(esd, cfg, ge["x"]), OV.encode(esd, cfg, ge["x"], emulate_bf16=True)
    rel = ((e16 - e32).norm() / e32.norm()).item()
    assert 1e-3 < rel < 4e-2, rel
    # flash contract: bf16 probabilities, fp32 row sum of the unrounded ones
    q, k, v = (torch.randn(2, 1, 50, 16).to(torch.bfloat16).float() for _ in range(3))
    a = OV.sdpa_bf16p(q, k, v)
    b = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    assert 1e-5 < ((a - b).norm() / b.norm()).item() < 5e-3
