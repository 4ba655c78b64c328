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
