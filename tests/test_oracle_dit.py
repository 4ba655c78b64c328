"""Known-answer / self-consistency tests of the DiT oracle (parity unpinned by the reference: SURVEY.md §8c)."""
import math

import torch

from oracle import wan_dit as O

TINY = O.WanDiTConfig(num_attention_heads=2, attention_head_dim=128, ffn_dim=512, num_layers=2, text_dim=128, freq_dim=64)


def test_rope_split_and_unit_modulus():
    f = O.rope_freqs(O.WAN_1_3B)
    assert f.shape == (1024, 64) and f.dtype == torch.complex128
    g = O.rope_for_grid(O.WAN_1_3B, 4, 32, 32)
    assert g.shape == (4096, 64)
    assert torch.allclose(g.abs(), torch.ones_like(g.abs()))
    # t / h / w blocks: 22 / 21 / 21 complex dims; token (f,h,w)=(1,2,3) -> index 1*1024+2*32+3
    idx = 1 * 1024 + 2 * 32 + 3
    assert torch.allclose(g[idx, :22], f[1, :22]) and torch.allclose(g[idx, 22:43], f[2, 22:43]) and torch.allclose(g[idx, 43:], f[3, 43:])
    x = torch.randn(1, 2, 4096, 128)
    y = O.apply_rope(x, g)
    assert torch.allclose(y.norm(dim=-1), x.norm(dim=-1), rtol=1e-5)


def test_patchify_roundtrip_and_conv_equivalence():
    cfg = TINY
    lat = torch.randn(2, 16, 3, 8, 8)
    tok = O.patchify(cfg, lat)
    assert tok.shape == (2, 3 * 4 * 4, 64)
    w = torch.randn(cfg.dim, 16, 1, 2, 2)
    ref = torch.nn.functional.conv3d(lat, w, stride=(1, 2, 2)).flatten(2).transpose(1, 2)
    got = tok @ w.reshape(cfg.dim, -1).t()
    assert torch.allclose(ref, got, atol=1e-4)
    # proj_out features are (pt,ph,pw,C)-major: unpatchify(identity tokens) inverts a (pt,ph,pw,C) patchify
    t2 = lat.view(2, 16, 3, 1, 4, 2, 4, 2).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(2, 48, 64)
    assert torch.equal(O.unpatchify(cfg, t2, 3, 8, 8), lat)


def test_timestep_embedding_kat():
    e = O.timestep_embedding(torch.tensor([0, 999]), 256)
    assert e.shape == (2, 256)
    assert torch.allclose(e[0, :128], torch.ones(128)) and torch.allclose(e[0, 128:], torch.zeros(128))
    assert abs(e[1, 0].item() - math.cos(999.0)) < 1e-4 and abs(e[1, 128].item() - math.sin(999.0)) < 1e-4


def test_zero_gate_block_is_cross_attention_only_and_chunk_order():
    cfg = TINY
    sd = O.make_weights(cfg, seed=1)
    x = torch.randn(1, 32, cfg.dim)
    ctx = torch.randn(1, 16, cfg.dim)
    freqs = O.rope_for_grid(cfg, 2, 4, 4)
    # gates are chunks 2 and 5 of (shift, scale, gate, c_shift, c_scale, c_gate): zero them, kill cross-attn output
    sd["blocks.0.scale_shift_table"][:, 2] = 0
    sd["blocks.0.scale_shift_table"][:, 5] = 0
    sd["blocks.0.attn2.to_out.0.weight"].zero_()
    sd["blocks.0.attn2.to_out.0.bias"].zero_()
    y = O.block_forward(sd, cfg, 0, x, ctx, torch.zeros(1, 6, cfg.dim), freqs, False)
    assert torch.allclose(y, x, atol=1e-6)


def test_forward_shapes_and_bf16_emulation_close():
    cfg = TINY
    sd = O.make_weights(cfg, seed=2)
    lat = torch.randn(2, 16, 2, 8, 8)
    text = torch.randn(2, 64, cfg.text_dim) * 0.1
    text[:, 40:] = 0
    t = torch.tensor([900, 900])
    a = O.dit_forward(sd, cfg, lat, t, text)
    b = O.dit_forward(sd, cfg, lat, t, text, emulate_bf16=True)
    assert a.shape == lat.shape and torch.isfinite(a).all()
    rel = ((a - b).norm() / a.norm()).item()
    assert rel < 3e-2, rel
    # batch independence (CFG batching must equal two B=1 calls)
    a0 = O.dit_forward(sd, cfg, lat[:1], t[:1], text[:1])
    assert torch.allclose(a[:1], a0, atol=1e-5)
