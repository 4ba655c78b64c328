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


def test_flash_emulation_is_softmax_up_to_bf16_p_and_handles_spikes_masks_and_bias():
    """attention_flash_emulated restates the HIP flash kernel's rounding points; in exact arithmetic it IS softmax attention."""
    g = torch.Generator().manual_seed(3)
    q = torch.randn(2, 3, 70, 64, generator=g).bfloat16().float()
    k = torch.randn(2, 3, 200, 64, generator=g).bfloat16().float()
    v = torch.randn(2, 3, 200, 64, generator=g).bfloat16().float()
    k[0, 1, 150] = q[0, 1, 5] * 6     # the running maximum of one row jumps by far more than 2^8 at the third tile
    ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    for defer in (0.0, 8.0):
        assert rel(O.attention_flash_emulated(q, k, v, 0.125, defer=defer), ref) < 3e-3
    # per-key bias and key mask
    kb = torch.zeros(2, 1, 200)
    kb[..., 199] = math.log(313.0)
    ref_b = torch.softmax(q @ k.transpose(-1, -2) * 0.125 + kb[:, :, None, :], -1) @ v
    assert rel(O.attention_flash_emulated(q, k, v, 0.125, key_bias=kb), ref_b) < 3e-3
    mask = (torch.arange(200) % 40) < 37
    ref_m = torch.softmax((q @ k.transpose(-1, -2) * 0.125).masked_fill(~mask, float("-inf")), -1) @ v
    assert rel(O.attention_flash_emulated(q, k, v, 0.125, key_mask=mask), ref_m) < 3e-3
    # a ragged last wave (70 = 2 x 32 + 6 rows) gives the rows it has the same result as when they are computed alone in their wave
    alone = O.attention_flash_emulated(q[..., 64:, :], k, v, 0.125, defer=0.0)
    assert torch.allclose(O.attention_flash_emulated(q, k, v, 0.125, defer=0.0)[..., 64:, :], alone, atol=1e-6)


def test_merged_padding_key_is_exact_and_option_switches_compose():
    cfg, sd = TINY, O.make_weights(TINY, seed=3)
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(2, 16, 1, 8, 8, generator=g)
    text = torch.randn(2, 96, cfg.text_dim, generator=g) * 0.5
    text[0, 20:] = 0
    text[1, 33:] = 0
    t = torch.tensor([500, 500])
    Lk, kb = O.merged_padding_keys(text)
    assert Lk == 40 and kb.shape == (2, 40) and abs(kb[0, 39].item() - math.log(96 - 39)) < 1e-6 and kb[:, :39].abs().max() == 0
    full = O.dit_forward(sd, cfg, lat, t, text)
    merged = O.dit_forward(sd, cfg, lat, t, text, merge_padding=True)
    assert torch.allclose(full, merged, atol=2e-5, rtol=1e-5)
    assert O.merged_padding_keys(torch.randn(1, 16, 8)) is None
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    emu = O.dit_forward(sd, cfg, lat, t, text, emulate_bf16=True)
    assert rel(O.dit_forward(sd, cfg, lat, t, text, emulate_bf16=True, flash=True, merge_padding=True), emu) < 1e-2
    assert rel(O.dit_forward(sd, cfg, lat, t, text, emulate_bf16=True, fp16_norm=True), emu) < 5e-3
    # the cached-context cross-attention (ctx_vo: sum_h bf16(P_h) bf16(V_h Wo_h^T)) is the same map with other rounding points: as far
    # from the fp32 forward as the reference order of operations is, and within bf16 noise of it
    con = O.dit_forward(sd, cfg, lat, t, text, emulate_bf16=True, flash=True, merge_padding=True)
    vo = O.dit_forward(sd, cfg, lat, t, text, emulate_bf16=True, flash=True, merge_padding=True, ctx_vo=True)
    assert rel(vo, con) < 4e-3 and rel(vo, full) < 1.3 * rel(con, full) + 1e-4


def test_unmerged_lora_equals_merged_weights_in_fp32():
    """Wx + (alpha/r) B(Ax) == (W + (alpha/r) BA) x: the oracle's unmerged adapter path (what the reference executes) against the
    merge the product performs at load (vist3a_amd.wan.dit.merge_lora_into_state_dict, pure tensor algebra: importable without a GPU)."""
    from vist3a_amd.wan.dit import merge_lora_into_state_dict
    cfg, base = TINY, O.make_weights(TINY, seed=3)
    sd_l, peft = O.with_lora_adapter(base, cfg)
    merged = {k: v.clone() for k, v in base.items()}
    assert merge_lora_into_state_dict(merged, peft, alpha=16, r=8) == 8 * cfg.num_layers
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, 16, 1, 8, 8, generator=g)
    text = torch.randn(1, 24, cfg.text_dim, generator=g) * 0.5
    t = torch.tensor([300])
    a = O.dit_forward(sd_l, cfg, lat, t, text)
    b = O.dit_forward(merged, cfg, lat, t, text)
    plain = O.dit_forward(base, cfg, lat, t, text)
    rel = lambda x, y: ((x - y).norm() / y.norm()).item()
    assert rel(a, b) < 2e-6 and rel(a, plain) > 1e-2     # identical map; and the adapter really changes the output
