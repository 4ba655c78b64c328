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


def test_rope_scores_depend_on_relative_grid_offsets_only():
    """The published rotary scheme (complex multiplication of adjacent pairs, per-axis frequency blocks): the score of a query at grid
    position p and a key at position p' is a function of p - p' alone - shifting both tokens by the same (dt, dh, dw) changes nothing, and
    a shift along ONE axis changes only that axis' block of the inner product."""
    cfg = O.WAN_1_3B
    g = O.rope_for_grid(cfg, 6, 8, 8)                                  # [6*8*8, 64] complex
    idx = lambda t, h, w: (t * 8 + h) * 8 + w
    gen = torch.Generator().manual_seed(3)
    q = torch.randn(1, 1, 1, 128, generator=gen, dtype=torch.float64)
    k = torch.randn(1, 1, 1, 128, generator=gen, dtype=torch.float64)
    def score(pq, pk):
        a = O.apply_rope(q, g[idx(*pq)][None].to(torch.complex128))
        b = O.apply_rope(k, g[idx(*pk)][None].to(torch.complex128))
        return (a * b).sum().item()
    base = score((1, 2, 3), (0, 5, 1))
    for d in ((1, 0, 0), (0, 2, 0), (0, 0, 4), (3, 1, 2)):
        moved = score((1 + d[0], 2 + d[1], 3 + d[2]), (0 + d[0], 5 + d[1], 1 + d[2]))
        assert abs(moved - base) < 1e-9 * (1 + abs(base)), (d, moved, base)
    assert abs(score((1, 2, 3), (1, 5, 1)) - base) > 1e-6               # ... and it does depend on the offset itself
    # per-axis blocks: with only the h-block of q non-zero, moving the key along t or w is invisible
    qh = torch.zeros_like(q)
    qh[..., 44:86] = q[..., 44:86]                                      # pairs 22..42 = the 21 complex dims of the h axis
    def score_h(pk):
        a = O.apply_rope(qh, g[idx(1, 2, 3)][None].to(torch.complex128))
        b = O.apply_rope(k, g[idx(*pk)][None].to(torch.complex128))
        return (a * b).sum().item()
    assert abs(score_h((0, 5, 1)) - score_h((4, 5, 6))) < 1e-9 and abs(score_h((0, 5, 1)) - score_h((0, 6, 1))) > 1e-6


def test_forward_is_invariant_to_the_order_of_text_tokens_and_independent_across_the_batch():
    """Cross-attention carries no position information on the text side (the rotary embedding is applied in the self-attention only), so
    permuting the rows of encoder_hidden_states must not change the output; and batch items never mix (a batch of two = two batches of one)."""
    cfg = TINY
    sd = O.make_weights(cfg, seed=5)
    gen = torch.Generator().manual_seed(6)
    lat = torch.randn(2, 16, 2, 8, 8, generator=gen)
    text = torch.randn(2, 12, cfg.text_dim, generator=gen)
    t = torch.tensor([700, 300])
    y = O.dit_forward(sd, cfg, lat, t, text)
    perm = torch.randperm(12, generator=gen)
    y_perm = O.dit_forward(sd, cfg, lat, t, text[:, perm])
    assert torch.allclose(y, y_perm, atol=2e-5), (y - y_perm).abs().max()
    for b in range(2):
        yb = O.dit_forward(sd, cfg, lat[b:b + 1], t[b:b + 1], text[b:b + 1])
        assert torch.allclose(y[b:b + 1], yb, atol=2e-5)
    # ... while the latent tokens DO carry positions: rolling the latent along w is not a roll of the output
    y_roll = O.dit_forward(sd, cfg, lat.roll(2, dims=-1), t, text)
    assert not torch.allclose(y_roll.roll(-2, dims=-1), y, atol=1e-3)


def test_adaln_modulation_enters_as_scale_shift_gate_in_the_published_order():
    """scale_shift_table + time projection, chunked as (shift, scale, gate) for the self-attention and (shift, scale, gate) for the FFN:
    with every gate at zero and the cross-attention silenced a block is the identity WHATEVER the shifts and scales are; a non-zero FFN gate
    scales exactly the FFN branch (linear in the gate)."""
    cfg = TINY
    sd = O.make_weights(cfg, seed=8)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(1, 32, cfg.dim, generator=gen)
    ctx = torch.randn(1, 16, cfg.dim, generator=gen)
    freqs = O.rope_for_grid(cfg, 2, 4, 4)
    sd["blocks.0.attn2.to_out.0.weight"].zero_()
    sd["blocks.0.attn2.to_out.0.bias"].zero_()
    tproj = torch.randn(1, 6, cfg.dim, generator=gen)
    tab = sd["blocks.0.scale_shift_table"]
    tab[:, 2] = -tproj[:, 2]                                            # total self-attention gate = 0
    tab[:, 5] = -tproj[:, 5]                                            # total FFN gate = 0
    assert torch.allclose(O.block_forward(sd, cfg, 0, x, ctx, tproj, freqs, False), x, atol=1e-6)
    tab[:, 5] = -tproj[:, 5] + 0.5
    d1 = O.block_forward(sd, cfg, 0, x, ctx, tproj, freqs, False) - x
    tab[:, 5] = -tproj[:, 5] + 1.5
    d3 = O.block_forward(sd, cfg, 0, x, ctx, tproj, freqs, False) - x
    assert d1.abs().max() > 1e-3 and torch.allclose(d3, 3.0 * d1, rtol=1e-4, atol=1e-5)
