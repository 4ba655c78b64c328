"""GPU parity: HIP DiT forward / denoise loop (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerances (relative L2, ||a-b||/||b||):
  * vs oracle with bf16 rounding points emulated: 1.5e-2 for a 2-block forward.  The residual gap is the flash
    kernel's bf16 P matrix (2e-3 per attention, measured in tools/kcheck.py) and fp32 summation order; the
    reference's own CUDA SDPA has the same property.
  * vs the pure-fp32 oracle: 4e-2 (bf16 storage of activations, as in the reference under autocast).
"""
import pytest
import torch

from oracle import wan_dit as O

pytestmark = pytest.mark.gpu

# full-depth production forward (30 blocks, 4096 tokens), asserted at <= 2x what MI355X measured (profiles/r3/parity.json)
TOL_FULL_DEPTH = 1.8e-2     # measured 9.1e-3 (both oracles)
TOL_ONE_BLOCK = 3e-3        # measured ~1.5e-3 after one block

TINY = dict(num_attention_heads=2, attention_head_dim=128, ffn_dim=512, num_layers=2, text_dim=128, freq_dim=64)


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def tiny(hip_lib):
    from vist3a_amd.wan.dit import WanDiT, WanDiTConfig
    ocfg = O.WanDiTConfig(**TINY)
    sd = O.make_weights(ocfg, seed=3)
    sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}  # weights exactly representable in bf16
    model = WanDiT(WanDiTConfig(**TINY), sd, device="cuda")
    return ocfg, sd, model


@pytest.mark.parametrize("shape,L", [((2, 16, 2, 16, 16), 64), ((1, 16, 3, 8, 16), 40), ((2, 16, 1, 32, 32), 512)])
def test_forward_matches_oracle(tiny, parity, shape, L):
    ocfg, sd, model = tiny
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(shape, generator=g)
    text = torch.randn(shape[0], L, ocfg.text_dim, generator=g) * 0.5
    text[:, L // 2:] = 0
    t = torch.tensor([937] * shape[0])
    lat_b = lat.to(torch.bfloat16)
    ref_emu = O.dit_forward(sd, ocfg, lat_b.float(), t, text.to(torch.bfloat16).float(), emulate_bf16=True)
    ref_f32 = O.dit_forward(sd, ocfg, lat_b.float(), t, text.to(torch.bfloat16).float())
    out = model(lat_b.cuda(), t.cuda(), text.cuda(), return_dict=False)[0]
    torch.cuda.synchronize()
    assert out.shape == lat.shape and out.dtype == torch.bfloat16
    assert torch.isfinite(out.float()).all()
    r1, r2 = _rel(out, ref_emu), _rel(out, ref_f32)
    # the reference loads the DiT in fp16 (inference_t23d.py:73) and diffusers' RMSNorm casts q / k to that dtype before RoPE: one more
    # rounding point (fp16, 11 bits) in front of the bf16 one the HIP kernel has - its effect on the output, measured, not asserted on
    ref_h16 = O.dit_forward(sd, ocfg, lat_b.float(), t, text.to(torch.bfloat16).float(), emulate_bf16=True, fp16_norm=True)
    parity("dit_two_blocks_tiny", shape=str(shape), rel_vs_emu_oracle=r1, rel_vs_fp32_oracle=r2, rel_vs_fp16_rmsnorm_oracle=_rel(out, ref_h16),
           fp16_rmsnorm_oracle_vs_emu_oracle=_rel(ref_h16, ref_emu))
    print(f"rel vs emu-oracle {r1:.3e}  vs fp32-oracle {r2:.3e}  vs oracle with fp16 q/k RMSNorm output {_rel(out, ref_h16):.3e}")
    assert r1 < 3.4e-3, r1      # measured 1.4e-3 .. 1.7e-3 on MI355X
    assert r2 < 9.4e-3, r2      # measured 4.7e-3


def test_batch2_equals_two_batch1_calls(tiny):
    """CFG batching must not couple the two branches."""
    ocfg, sd, model = tiny
    g = torch.Generator().manual_seed(6)
    lat = torch.randn(2, 16, 2, 16, 16, generator=g).to(torch.bfloat16).cuda()
    text = (torch.randn(2, 64, ocfg.text_dim, generator=g) * 0.5).cuda()
    t = torch.tensor([500, 500]).cuda()
    both = model(lat, t, text)[0].clone()
    a = model(lat[:1], t[:1], text[:1].contiguous())[0].clone()
    b = model(lat[1:], t[1:], text[1:].contiguous())[0].clone()
    assert torch.equal(both[0], a[0]) and torch.equal(both[1], b[0])


def test_forward_is_invariant_to_the_order_of_the_real_text_tokens(tiny):
    """The cross-attention has no positions on the text side (tests/test_oracle_dit.py shows it for the oracle): permuting the REAL rows of a
    zero-padded prompt must leave the HIP forward unchanged up to summation order - through the merged padding key, the cached V.Wo^T
    operands and the probabilities kernel, whose key order it changes."""
    ocfg, sd, model = tiny
    g = torch.Generator().manual_seed(16)
    lat = torch.randn(2, 16, 2, 16, 16, generator=g).to(torch.bfloat16).cuda()
    text = torch.randn(2, 64, ocfg.text_dim, generator=g) * 0.5
    text[0, 23:] = 0
    text[1, 41:] = 0
    t = torch.tensor([650, 650]).cuda()
    y = model(lat, t, text.cuda())[0].float().clone()
    perm = text.clone()
    perm[0, :23] = text[0, torch.randperm(23, generator=g)]
    perm[1, :41] = text[1, torch.randperm(41, generator=g)]
    y2 = model(lat, t, perm.cuda())[0].float().clone()
    r = _rel(y2, y)
    assert r < 2e-3, r            # bf16 rounding of differently ordered sums; a position-dependent bug is O(1)
    # (moving a token into the padding region is ALSO a permutation of the same multiset of keys - interior zero rows stay ordinary keys, the
    # trailing run is merged - and must not change the output either; changing a token's VALUE must)
    moved = text.clone()
    moved[0, 0], moved[0, 30] = 0, text[0, 0]
    assert _rel(model(lat, t, moved.cuda())[0].float(), y) < 2e-3
    changed = text.clone()
    changed[0, 5] = -text[0, 5]
    assert _rel(model(lat, t, changed.cuda())[0].float(), y) > 1e-4


def test_denoise_loop_matches_oracle_loop(tiny):
    """4-step CFG denoise: product pipeline (HIP DiT + host UniPC) vs oracle DiT (bf16 points) + oracle UniPC."""
    from oracle.unipc import OracleUniPC
    from vist3a_amd.wan.pipeline import WanT2VPipeline
    from vist3a_amd.wan.scheduler import UniPCMultistepScheduler
    ocfg, sd, model = tiny
    g = torch.Generator().manual_seed(7)
    lat0 = torch.randn(1, 16, 2, 16, 16, generator=g)
    pe = torch.randn(1, 64, ocfg.text_dim, generator=g) * 0.5
    ne = torch.randn(1, 64, ocfg.text_dim, generator=g) * 0.5
    pe[:, 30:] = 0
    ne[:, 45:] = 0
    pe, ne = pe.to(torch.bfloat16).float(), ne.to(torch.bfloat16).float()
    steps, gs = 4, 7.5
    pipe = WanT2VPipeline(model, UniPCMultistepScheduler(flow_shift=5.0))
    out = pipe(prompt_embeds=pe.cuda(), negative_prompt_embeds=ne.cuda(), height=128, width=128, num_frames=5,
               num_inference_steps=steps, guidance_scale=gs, latents=lat0.clone())["frames"].cpu()
    sch = OracleUniPC(flow_shift=5.0)
    sch.set_timesteps(steps)
    x = lat0.clone()
    text = torch.cat([pe, ne], 0)
    for t in sch.timesteps:
        xin = x.to(torch.bfloat16).float().expand(2, -1, -1, -1, -1)
        v = O.dit_forward(sd, ocfg, xin, t.expand(2), text, emulate_bf16=True).to(torch.bfloat16)
        nz = v[1:2] + gs * (v[0:1] - v[1:2])
        x = sch.step(nz, x)
    r = _rel(out, x)
    print("denoise loop rel", r)
    assert out.dtype == torch.float32 and r < 1.2e-2, r   # measured 5.9e-3


@pytest.mark.parametrize("P,shape", [(2, (2, 16, 2, 16, 16)), (4, (1, 16, 2, 32, 16)), (8, (2, 16, 1, 32, 32))])
def test_seq_parallel_forward_is_bit_identical(tiny, P, shape):
    """P token shards (virtual ranks = threads on this GPU, same kernels, in-process all-gather) == the unsharded forward."""
    from vist3a_amd.wan.seqpar import ThreadWorld
    ocfg, sd, model = tiny
    g = torch.Generator().manual_seed(8)
    lat = torch.randn(shape, generator=g).to(torch.bfloat16).cuda()
    text = (torch.randn(shape[0], 64, ocfg.text_dim, generator=g) * 0.5).cuda()
    t = torch.tensor([500] * shape[0]).cuda()
    full = model(lat, t, text)[0].clone()
    w = ThreadWorld(P)
    outs = w.run(lambda r: model(lat, t, text, sp=w.group(r))[0].clone())
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, full)


def test_seq_parallel_rejects_ragged_split(tiny):
    from vist3a_amd.wan.seqpar import ThreadWorld
    ocfg, sd, model = tiny
    lat = torch.zeros(1, 16, 1, 6, 6, dtype=torch.bfloat16, device="cuda")  # 9 tokens
    with pytest.raises(ValueError):
        model(lat, torch.tensor([1]).cuda(), torch.zeros(1, 8, ocfg.text_dim, device="cuda"), sp=ThreadWorld(2).group(0))


@pytest.mark.parametrize("world", [2, 4, 3])
def test_denoise_plan_matches_single_gpu_loop(tiny, world):
    """CFG-parallel x sequence-parallel denoise (wan/seqpar.py) reproduces the single-GPU latents bit for bit."""
    from vist3a_amd.wan.pipeline import WanT2VPipeline
    from vist3a_amd.wan.scheduler import UniPCMultistepScheduler
    from vist3a_amd.wan.seqpar import DenoisePlan, ThreadWorld
    ocfg, sd, model = tiny
    g = torch.Generator().manual_seed(9)
    pe = torch.randn(1, 32, ocfg.text_dim, generator=g) * 0.5
    ne = torch.randn(1, 32, ocfg.text_dim, generator=g) * 0.5
    # 3 ranks: no CFG split, 3 token shards -> needs N % 24 == 0
    lat0 = torch.randn(1, 16, 3, 16, 32, generator=g) if world == 3 else torch.randn(1, 16, 2, 16, 16, generator=g)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, height=lat0.shape[3] * 8, width=lat0.shape[4] * 8,
              num_frames=(lat0.shape[2] - 1) * 4 + 1, num_inference_steps=4, guidance_scale=6.0, latents=lat0)
    ref = WanT2VPipeline(model, UniPCMultistepScheduler(flow_shift=5.0))(**kw)["frames"].clone()
    plans = DenoisePlan.from_threads(world)
    cfg_deg, sp_deg = DenoisePlan.layout(world)
    assert (cfg_deg, sp_deg) == ((2, world // 2) if world % 2 == 0 else (1, world))
    runner = ThreadWorld(world)
    outs = runner.run(lambda r: WanT2VPipeline(model, UniPCMultistepScheduler(flow_shift=5.0), plan=plans[r])(**kw)["frames"].clone())
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref)


def test_graph_replay_is_bit_identical_across_steps_and_prompts(tiny):
    """GraphedWanDiT: one captured hipGraph serves changing latents, timesteps and PROMPTS (persistent context buffers)."""
    from vist3a_amd.wan.dit import GraphedWanDiT
    ocfg, sd, model = tiny
    gm = GraphedWanDiT(model)
    g = torch.Generator().manual_seed(12)
    texts = [(torch.randn(2, 64, ocfg.text_dim, generator=g) * 0.5).cuda() for _ in range(2)]
    for rnd in range(2):
        for text in texts:
            for t in (900, 400):
                lat = torch.randn(2, 16, 2, 16, 16, generator=g).to(torch.bfloat16).cuda()
                tt = torch.tensor([t, t]).cuda()
                want = model(lat, tt, text)[0].clone()
                got = gm(lat, tt, text)[0].clone()
                assert torch.equal(got, want)
    assert len(gm._graphs) == 1
    # in-place edit of the prompt tensor is noticed (version counter), a different latent shape captures a second graph
    texts[0].mul_(0.5)
    lat = torch.randn(2, 16, 1, 16, 16, generator=g).to(torch.bfloat16).cuda()
    tt = torch.tensor([10, 10]).cuda()
    assert torch.equal(gm(lat, tt, texts[0])[0], model(lat, tt, texts[0])[0])
    assert len(gm._graphs) == 2


def test_graph_replay_follows_a_ctx_vo_mode_flip(tiny):
    """`WanDiT.ctx_vo` changes what the persistent prompt buffers hold (norm_q weight folded into the cached keys, V.Wo^T instead of V^T):
    a graph captured in one mode must not replay over the other mode's buffers.  The mode is part of the capture key."""
    from vist3a_amd.wan.dit import GraphedWanDiT
    ocfg, sd, model = tiny
    gm = GraphedWanDiT(model)
    g = torch.Generator().manual_seed(13)
    text = (torch.randn(2, 64, ocfg.text_dim, generator=g) * 0.5).cuda()
    text[:, 20:] = 0      # a merged padding key: the cached-context form is taken
    lat = torch.randn(2, 16, 2, 16, 16, generator=g).to(torch.bfloat16).cuda()
    tt = torch.tensor([500, 500]).cuda()
    keep = model.ctx_vo
    try:
        outs = {}
        for mode in (True, False, True, False):
            model.ctx_vo = mode
            want = model(lat, tt, text)[0].clone()
            got = gm(lat, tt, text)[0].clone()
            assert torch.equal(got, want), f"graph replay differs from eager with ctx_vo={mode}"
            outs.setdefault(mode, want)
            assert torch.equal(outs[mode], want)
        assert len(gm._graphs) == 2
        assert not torch.equal(outs[True], outs[False])     # the two orders of operations round differently: the flip is observable
    finally:
        model.ctx_vo = keep


def test_fused_qkv_projection_is_bit_identical_to_separate_launches(hip_lib):
    """`WanDiT.fused_qkv`: q | k | V^T of every block from one GEMM launch (transposed tail) - the forward equals the two-launch form bit for
    bit at production width and token count (two blocks, CFG batch 2), eagerly and through a hipGraph (the mode is part of the capture key)."""
    import dataclasses
    from vist3a_amd.wan.dit import WAN_1_3B, GraphedWanDiT, WanDiT
    from vist3a_amd.wan.weights import random_dit_state_dict
    cfg = dataclasses.replace(WAN_1_3B, num_layers=2, text_dim=512)
    m = WanDiT(cfg, random_dit_state_dict(cfg, seed=0, device="cuda"))
    g = torch.Generator().manual_seed(6)
    lat = torch.randn(2, 16, 4, 64, 64, generator=g).to(torch.bfloat16).cuda()
    text = (torch.randn(2, 512, 512, generator=g) * 0.1).cuda()
    text[:, 70:] = 0
    t = torch.tensor([650, 650]).cuda()
    seen = []
    from vist3a_amd import ops
    real = ops.gemm
    ops.gemm = lambda *a, **k: (seen.append(k.get("t_out") is not None), real(*a, **k))[1]
    try:
        assert not m.fused_qkv      # opt-in: no measured gain on the step
        m.fused_qkv = True
        a = m(lat, t, text)[0].clone()
        n_fused = sum(seen)
        m.fused_qkv = False
        seen.clear()
        b = m(lat, t, text)[0].clone()
        assert n_fused == 2 and sum(seen) == 0
    finally:
        ops.gemm = real
    assert torch.equal(a, b)
    gm = GraphedWanDiT(m)
    for mode in (True, False, True):
        m.fused_qkv = mode
        assert torch.equal(gm(lat, t, text)[0], a)
    assert len(gm._graphs) == 2
    m.fused_qkv = False


def test_wan14b_width_two_blocks_matches_oracle(hip_lib):
    """BASELINE config #4 geometry (Wan-14B: 40 heads x 128 = 5120 wide, FFN 13824) on two blocks: the GEMM tilings are ragged
    there (5120 / 192, 13824 / 192 are not integers) — same tolerances as the 1.3B-width forward."""
    from vist3a_amd.wan.dit import WanDiT, WanDiTConfig
    kw = dict(num_attention_heads=40, attention_head_dim=128, ffn_dim=13824, num_layers=2, text_dim=256, freq_dim=256)
    ocfg = O.WanDiTConfig(**kw)
    sd = {k: v.to(torch.bfloat16).float() for k, v in O.make_weights(ocfg, seed=4).items()}
    model = WanDiT(WanDiTConfig(**kw), sd, device="cuda")
    g = torch.Generator().manual_seed(14)
    lat = torch.randn(2, 16, 2, 16, 16, generator=g).to(torch.bfloat16)   # 128 tokens per item
    text = (torch.randn(2, 48, 256, generator=g) * 0.5).to(torch.bfloat16).float()
    t = torch.tensor([611, 611])
    ref = O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True)
    out = model(lat.cuda(), t.cuda(), text.cuda())[0]
    r = _rel(out, ref)
    print("14B-width rel vs emu-oracle", r)
    assert torch.isfinite(out.float()).all() and r < 1.06e-2, r   # measured 5.3e-3


def test_config4_14b_width_eight_blocks_match_oracle(hip_lib, parity):
    """BASELINE config #4 (Wan-14B stitched, fp8 MFMA attention) on EIGHT of its 40 blocks, 512 tokens per batch item, B = 2: the bf16
    mode against the contract oracle and the fp8-attention mode against the oracle with the e4m3 rounding points emulated - with the
    error-vs-depth curve of both, so a defect that only compounds shows."""
    from vist3a_amd.wan.dit import WanDiT, WanDiTConfig
    import fullsize_cases as FC
    import oracle_cache as OC
    case = FC.dit_config4_eight_blocks()
    sd, lat, text, t = case.sd, case.lat, case.text, case.t
    model = WanDiT(WanDiTConfig(num_attention_heads=40, attention_head_dim=128, ffn_dim=13824, num_layers=8, text_dim=256, freq_dim=256), sd, device="cuda")
    depths = FC.CONFIG4_DEPTHS
    # (two 8-block oracle forwards at 14B width: 1-2 minutes of host time - committed digest, tests/oracle_cache.py)
    od, live = OC.oracle(case.name, case.fingerprint, case.compute, sources=case.sources, case_fns=case.case_fns)
    taps16 = {L: od[f"bf16_depth{L}"] for L in depths}
    taps8 = {L: od[f"fp8_depth{L}"] for L in depths}
    _rel = OC.rel
    c16 = {L: _rel(model(lat.cuda(), t.cuda(), text.cuda(), num_layers=L)[0], taps16[L]) for L in depths}
    model.attn_dtype = "fp8"
    c8 = {L: _rel(model(lat.cuda(), t.cuda(), text.cuda(), num_layers=L)[0], taps8[L]) for L in depths}
    parity("dit_config4_14B_width_8_blocks", bf16_vs_contract_by_depth={str(k): v for k, v in c16.items()},
           fp8_attention_vs_e4m3_oracle_by_depth={str(k): v for k, v in c8.items()})
    print("config #4, 14B width, by depth: bf16", {k: f"{v:.2e}" for k, v in c16.items()}, "fp8 attention", {k: f"{v:.2e}" for k, v in c8.items()})
    # measured on MI355X: bf16 3.5e-3 / 4.9e-3 / 6.6e-3 / 8.4e-3 after 1 / 2 / 4 / 8 blocks, fp8 attention 3.6e-3 / 5.2e-3 / 7.1e-3 / 9.1e-3
    # (one block at this width is 2.3x the 1.3B-width figure: 3.3x longer bf16 reductions; the curve grows like sqrt(depth), no faster)
    assert c16[1] < 7e-3 and c8[1] < 7.2e-3, (c16, c8)
    assert c16[8] < 1.7e-2 and c8[8] < 1.8e-2, (c16, c8)
    assert c16[8] < 3.2 * c16[1] and c8[8] < 3.2 * c8[1], (c16, c8)      # sqrt(8) = 2.8: an error that compounds faster than rounding noise fails


def test_config4_full_depth_14b_forward_is_deterministic_and_finite(hip_lib, parity):
    """The whole 40-block Wan-14B forward at config #4's geometry (13 views @512 = 4096 tokens, CFG batch 2), bf16 and fp8-attention modes:
    finite, bounded, bit-identical run to run, and the two modes within fp8's price of each other (no oracle at this size - 28 GB of
    weights; size-independent properties only)."""
    from vist3a_amd.wan.dit import WAN_14B, WanDiT
    from vist3a_amd.wan.weights import random_dit_state_dict
    import dataclasses
    cfg = dataclasses.replace(WAN_14B, text_dim=512)
    model = WanDiT(cfg, random_dit_state_dict(cfg, seed=0, device="cuda"))
    assert cfg.num_layers == 40 and cfg.num_attention_heads == 40
    g = torch.Generator().manual_seed(46)
    lat = torch.randn(1, 16, 4, 64, 64, generator=g).to(torch.bfloat16).cuda().expand(2, -1, -1, -1, -1).contiguous()
    text = (torch.randn(2, 512, 512, generator=g) * 0.1).cuda()
    text[0, 77:] = 0
    text[1, 9:] = 0
    t = torch.tensor([700, 700]).cuda()
    a16 = model(lat, t, text)[0].clone()
    b16 = model(lat, t, text)[0].clone()
    model.attn_dtype = "fp8"
    a8 = model(lat, t, text)[0].clone()
    b8 = model(lat, t, text)[0].clone()
    torch.cuda.synchronize()
    assert torch.equal(a16, b16) and torch.equal(a8, b8)
    assert torch.isfinite(a16.float()).all() and torch.isfinite(a8.float()).all()
    assert not torch.equal(a16[0], a16[1])                         # the two prompts condition the two batch items differently
    shift = _rel(a8, a16)
    rms = a16.float().pow(2).mean().sqrt().item()
    parity("dit_config4_14B_40_blocks_N4096", rms=rms, fp8_attention_vs_bf16=shift, deterministic=True, peak_GB=torch.cuda.max_memory_allocated() / 2 ** 30)
    print(f"14B 40-block forward: rms {rms:.3f}, fp8-attention vs bf16 {shift:.2e}, peak {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GB")
    assert 1e-3 < rms < 1e3 and 0 < shift < 0.2
    del model
    torch.cuda.empty_cache()


def test_21_view_latent_shapes(tiny):
    """BASELINE config #3 geometry: 21 views = 6 latent frames (6144 tokens at 64x64 latents; here 6 x 16 x 16 = 384)."""
    ocfg, sd, model = tiny
    g = torch.Generator().manual_seed(21)
    lat = torch.randn(2, 16, 6, 32, 32, generator=g).to(torch.bfloat16)
    text = (torch.randn(2, 40, ocfg.text_dim, generator=g) * 0.5).to(torch.bfloat16).float()
    t = torch.tensor([333, 333])
    ref = O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True)
    out = model(lat.cuda(), t.cuda(), text.cuda())[0]
    r = _rel(out, ref)
    print("21-view tiny rel", r)
    assert out.shape == lat.shape and r < 3.5e-3, r


@pytest.mark.parametrize("Nk", [4096, 512, 1000])
def test_attention_is_run_to_run_deterministic_at_full_size(hip_lib, Nk):
    """Regression: with the single-buffered V^T tile a wave used to pass the end-of-tile barrier with a fragment read still in
    flight (hipcc sinks the last MFMA and its lgkmcnt wait below a bare s_barrier) while another wave's DMA refilled the buffer —
    visible only at production size as whole 32-row groups differing between runs."""
    from vist3a_amd import ops
    B, H, N, D = 2, 12, 4096, 128
    d = H * D
    g = torch.Generator(device="cuda").manual_seed(Nk)
    q = (torch.randn(B * N, d, device="cuda", generator=g) * 0.5).bfloat16()
    k = (torch.randn(B * Nk, d, device="cuda", generator=g) * 0.5).bfloat16()
    Lp = (Nk + 63) // 64 * 64
    vt = torch.zeros(d, B * Lp, device="cuda", dtype=torch.bfloat16)
    for b in range(B):
        vt[:, b * Lp: b * Lp + Nk] = torch.randn(d, Nk, device="cuda", generator=g).bfloat16()
    outs = []
    for _ in range(5):
        o = torch.empty(B * N, d, device="cuda", dtype=torch.bfloat16)
        ops.attention(q, k, vt, o, B=B, H=H, Nq=N, Nk=Nk, D=D, q_batch_stride=N * d, k_batch_stride=Nk * d, vt_batch_stride=Lp,
                      o_batch_stride=N * d)
        outs.append(o)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    qf, kf, vf = q[:N, :D].float(), k[:Nk, :D].float(), vt[:D, :Nk].float().t()
    ref = torch.softmax(qf @ kf.t() * D ** -0.5, -1) @ vf
    assert _rel(outs[0][:N, :D], ref) < 3e-3


def test_full_size_dit_forward_is_deterministic_and_batch_independent(hip_lib):
    """Two production-width blocks at 4096 tokens: identical across runs, and the CFG pair (B=2) equals two B=1 forwards."""
    from vist3a_amd.wan.dit import WAN_1_3B, WanDiT
    from vist3a_amd.wan.weights import random_dit_state_dict
    import dataclasses
    cfg = dataclasses.replace(WAN_1_3B, num_layers=2)
    m = WanDiT(cfg, random_dit_state_dict(cfg, seed=0, device="cuda"))
    g = torch.Generator().manual_seed(1)
    text = (torch.randn(2, 512, 4096, generator=g) * 0.1).cuda()
    lat = torch.randn(2, 16, 4, 64, 64, generator=g).bfloat16().cuda()
    t = torch.tensor([900, 900]).cuda()
    a = m(lat, t, text)[0].clone()
    b = m(lat, t, text)[0].clone()
    one = torch.cat([m(lat[i:i + 1], t[i:i + 1], text[i:i + 1].contiguous())[0].clone() for i in range(2)], 0)
    assert torch.equal(a, b) and torch.equal(a, one)


def test_prompt_context_cache_survives_address_reuse(tiny):
    """Regression: the per-prompt cross-attention cache was keyed by data_ptr; a new prompt whose embedding tensor was allocated at
    the address of the previous (freed) one got the OLD prompt's K / V."""
    from vist3a_amd.wan.dit import WanDiT, WanDiTConfig
    ocfg, sd, model = tiny
    fresh = WanDiT(WanDiTConfig(**TINY), sd, device="cuda")
    g = torch.Generator().manual_seed(33)
    lat = torch.randn(2, 16, 2, 16, 16, generator=g).to(torch.bfloat16).cuda()
    t = torch.tensor([700, 700]).cuda()
    a_cpu = torch.randn(2, 64, ocfg.text_dim, generator=g) * 0.5
    b_cpu = torch.randn(2, 64, ocfg.text_dim, generator=g) * 0.5
    text = a_cpu.cuda()
    ptr = text.data_ptr()
    model(lat, t, text)
    del text
    text = b_cpu.cuda()            # the caching allocator hands the same block back
    same_addr = text.data_ptr() == ptr
    out = model(lat, t, text)[0].clone()
    want = fresh(lat, t, b_cpu.cuda())[0]
    assert torch.equal(out, want), f"stale prompt context (address reused: {same_addr})"


def test_prompt_context_slots_are_bounded_and_evicted_graphs_are_dropped(tiny):
    """The per-(batch, prompt length) context buffers are an LRU of `max_ctx_slots` (2) per host thread, not a cache that grows for ever (a slot
    is 283 MB at Wan-1.3B, 4.2 GB at Wan-14B): a third prompt length evicts the least recently used slot, the evicted length is recomputed
    correctly when it comes back, and a hipGraph captured over an evicted slot is dropped instead of replayed over freed buffers."""
    from vist3a_amd.wan.dit import GraphedWanDiT, WanDiT, WanDiTConfig
    ocfg, sd, _ = tiny
    model = WanDiT(WanDiTConfig(**TINY), sd, device="cuda")
    gm = GraphedWanDiT(model)
    g = torch.Generator().manual_seed(34)
    lat = torch.randn(2, 16, 2, 16, 16, generator=g).to(torch.bfloat16).cuda()
    t = torch.tensor([700, 700]).cuda()
    texts = {n: (torch.randn(2, n, ocfg.text_dim, generator=g) * 0.5).cuda() for n in (40, 64, 96)}
    first = {n: gm(lat, t, texts[n])[0].clone() for n in (40, 64)}
    assert len(model._ctx) == 2 and len(gm._graphs) == 2
    serial40 = model._ctx[next(k for k in model._ctx if k[1] == 40)][1][11]
    out96 = gm(lat, t, texts[96])[0].clone()                      # third length: evicts the slot of length 40 (least recently used)
    assert len(model._ctx) == 2 and {k[1] for k in model._ctx} == {64, 96}
    assert len(gm._graphs) == 2 and all(k[-1] != serial40 for k in gm._graphs)      # ... and its graph with it
    assert torch.equal(out96, model(lat, t, texts[96])[0])        # graph replay == eager
    again = gm(lat, t, texts[40])[0].clone()                      # the evicted length comes back: new slot, new capture, same result
    assert torch.equal(again, first[40]) and {k[1] for k in model._ctx} == {96, 40}
    assert torch.equal(gm(lat, t, texts[96])[0], out96)
    assert model.max_ctx_slots == 2


def test_merged_padding_keys_equal_full_cross_attention(tiny):
    """Zero-padded prompt: cross-attention over n real keys + ONE padding key with bias log(count) == attention over all 512 keys
    (exact in real arithmetic; bf16 P rounding differs: 2e-3), and an unpadded prompt takes the plain path."""
    from vist3a_amd.wan.dit import WanDiT, WanDiTConfig
    ocfg, sd, model = tiny
    plain = WanDiT(WanDiTConfig(**TINY), sd, device="cuda")
    plain.merge_padding_keys = False
    g = torch.Generator().manual_seed(44)
    lat = torch.randn(2, 16, 2, 16, 16, generator=g).to(torch.bfloat16).cuda()
    t = torch.tensor([500, 500]).cuda()
    text = torch.zeros(2, 512, ocfg.text_dim)
    text[0, :37] = torch.randn(37, ocfg.text_dim, generator=g) * 0.5
    text[1, :80] = torch.randn(80, ocfg.text_dim, generator=g) * 0.5
    text = text.cuda()
    a = model(lat, t, text)[0]
    b = plain(lat, t, text)[0]
    assert any(v[1][5] == 88 and v[1][6] for v in model._ctx.values())  # 80 real rows, padded to a multiple of 8: 88 keys
    assert _rel(a, b) < 3.5e-3   # (the merged prompt also takes the cached-context form of the cross-attention, the 512-key one the flash form)
    ref = O.dit_forward(sd, ocfg, lat.float().cpu(), t.cpu(), text.cpu().to(torch.bfloat16).float(), emulate_bf16=True)
    assert _rel(a, ref) < 3.5e-3
    full = (torch.randn(2, 512, ocfg.text_dim, generator=g) * 0.5).cuda()       # nothing to merge
    assert torch.equal(model(lat, t, full)[0], plain(lat, t, full)[0])
    one_pad = full.clone()
    one_pad[:, -1] = 0                                                           # a single zero row is not worth a merge
    assert torch.equal(model(lat, t, one_pad)[0], plain(lat, t, one_pad)[0])


def test_full_depth_production_size_forward_matches_oracle(hip_lib, parity):
    """The one full-size parity point: Wan-1.3B geometry (30 blocks, 12 x 128 heads, FFN 8960), 13 views @512 (4096 tokens), B=1,
    random bf16-representable weights, against oracle.dit_forward with the bf16 rounding points emulated (~35 s on the GPU box's
    host cores).  This is the only test in which the production GEMM tiles (ping-pong 256x192 / 192x256) and the three-per-CU hd128
    flash kernel run inside the model AND are compared with a reference.  Text context 77 tokens + padding to 512 (merged)."""
    import dataclasses
    from vist3a_amd.wan.dit import WAN_1_3B, WanDiT
    import fullsize_cases as FC
    import oracle_cache as OC
    case = FC.dit_full_depth()
    cfg = dataclasses.replace(WAN_1_3B, text_dim=512)   # UMT5 width 4096 only scales the (cached) context MLP
    ocfg, sd, lat, text, t = case.ocfg, case.sd, case.lat, case.text, case.t
    assert (ocfg.num_attention_heads, ocfg.ffn_dim, ocfg.num_layers, ocfg.freq_dim) == (cfg.num_attention_heads, cfg.ffn_dim, cfg.num_layers, cfg.freq_dim)
    model = WanDiT(cfg, sd, device="cuda")
    depths = FC.DIT_DEPTHS
    outs = {L: model(lat.cuda(), t.cuda(), text.cuda(), num_layers=L)[0].float().cpu() for L in depths}
    out = outs[30]
    torch.cuda.synchronize()
    # (the two 30-block oracle forwards take ~2 minutes of host time: committed digest, tests/oracle_cache.py)
    od, live = OC.oracle(case.name, case.fingerprint, case.compute, sources=case.sources, case_fns=case.case_fns)
    r, rc, floor = OC.rel(out, od["ref"]), OC.rel(out, od["ref_c"]), OC.rel_dd(od["ref_c"], od["ref"])
    curve = {L: OC.rel(outs[L], od[f"depth{L}"]) for L in depths}
    mx = (OC.digest(out) - od["ref"]).abs().max().item()
    parity("dit_full_depth_30_blocks_N4096", rel_vs_emu_oracle=r, rel_vs_contract_oracle=rc, oracle_contract_vs_exact_softmax=floor,
           rel_vs_contract_oracle_by_depth={str(k): v for k, v in curve.items()}, max_abs=mx, ref_rms=float(od["ref_rms"].item()),
           oracle="live" if live else "digest fixture")
    print(f"full-depth 30-block N=4096 forward: rel {r:.3e} (exact-softmax oracle), {rc:.3e} (kernel-contract oracle); the two oracles differ "
          f"by {floor:.3e}; by depth {', '.join(f'{k}: {v:.2e}' for k, v in curve.items())}; max abs {mx:.3e} ({'live' if live else 'fixture'})")
    assert torch.isfinite(out).all()
    # Round 3 finding: HIP, the exact-softmax oracle and the contract oracle are MUTUALLY ~9e-3 apart after 30 blocks - two restatements
    # of the same arithmetic that differ only in where P is rounded land as far from each other as the kernels land from either.  The
    # figure is the conditioning of 30 random-weight blocks times the bf16 ulp (error-vs-depth curve: 1e-3 after one block, growing with
    # depth), not a kernel error; the kernels themselves are pinned at <= 1e-3 of their contract in tests/test_kernels_gpu.py.
    assert curve[1] < TOL_ONE_BLOCK and curve[2] < 2 * TOL_ONE_BLOCK, curve
    assert r < TOL_FULL_DEPTH and rc < TOL_FULL_DEPTH, (r, rc)
    assert rc < 2.0 * floor, (rc, floor)      # no further from the oracle than the oracle's own two forms are from each other (x2)


def test_wan14b_width_fp8_attention_matches_e4m3_oracle(hip_lib, parity):
    """BASELINE config #4 precision mode at Wan-14B width (40 heads x 128, FFN 13824, two blocks): self-attention on the fp8 MFMA
    against the oracle with the same e4m3 rounding points emulated; same tolerance as the bf16 two-block forward.  Also reports how
    far the fp8 mode moves the output from the bf16 mode (the price of the mode, not an error of the kernel)."""
    from vist3a_amd.wan.dit import WanDiT, WanDiTConfig
    kw = dict(num_attention_heads=40, attention_head_dim=128, ffn_dim=13824, num_layers=2, text_dim=256, freq_dim=256)
    ocfg = O.WanDiTConfig(**kw)
    sd = {k: v.to(torch.bfloat16).float() for k, v in O.make_weights(ocfg, seed=4).items()}
    model = WanDiT(WanDiTConfig(**kw), sd, device="cuda")
    g = torch.Generator().manual_seed(14)
    lat = torch.randn(2, 16, 2, 16, 16, generator=g).to(torch.bfloat16)
    text = (torch.randn(2, 48, 256, generator=g) * 0.5).to(torch.bfloat16).float()
    t = torch.tensor([611, 611])
    out16 = model(lat.cuda(), t.cuda(), text.cuda())[0].clone()
    model.attn_dtype = "fp8"
    out8 = model(lat.cuda(), t.cuda(), text.cuda())[0].clone()
    ref8 = O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True, fp8_attn=True)
    r, shift = _rel(out8, ref8), _rel(out8, out16)
    parity("dit_14B_width_fp8_attention", rel_vs_e4m3_oracle=r, rel_fp8_mode_vs_bf16_mode=shift)
    print(f"14B-width fp8-attention forward: rel vs e4m3 oracle {r:.2e}; fp8 mode vs bf16 mode {shift:.2e}")
    assert torch.isfinite(out8.float()).all() and r < 1.26e-2, r   # measured 6.3e-3


def test_wan14b_width_fp8_gemm_matches_e4m3_oracle(hip_lib, parity):
    """BASELINE config #4 "fp8 GEMMs for the attention/FFN contractions": the block projections on e4m3 operands (per-token /
    per-output-channel scales) at Wan-14B width against the oracle emulating the same quantisation, alone and together with the
    fp8 attention; the distance from the bf16 mode is reported as the price of the mode."""
    from vist3a_amd.wan.dit import WanDiT, WanDiTConfig
    kw = dict(num_attention_heads=40, attention_head_dim=128, ffn_dim=13824, num_layers=2, text_dim=256, freq_dim=256)
    ocfg = O.WanDiTConfig(**kw)
    sd = {k: v.to(torch.bfloat16).float() for k, v in O.make_weights(ocfg, seed=4).items()}
    model = WanDiT(WanDiTConfig(**kw), sd, device="cuda")
    g = torch.Generator().manual_seed(14)
    lat = torch.randn(2, 16, 2, 16, 16, generator=g).to(torch.bfloat16)
    text = (torch.randn(2, 48, 256, generator=g) * 0.5).to(torch.bfloat16).float()
    t = torch.tensor([611, 611])
    out16 = model(lat.cuda(), t.cuda(), text.cuda())[0].clone()
    model.enable_fp8_gemm()
    outg = model(lat.cuda(), t.cuda(), text.cuda())[0].clone()
    refg = O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True, fp8_gemm=True)
    model.attn_dtype = "fp8"
    outga = model(lat.cuda(), t.cuda(), text.cuda())[0].clone()
    refga = O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True, fp8_gemm=True, fp8_attn=True)
    r1, r2, shift = _rel(outg, refg), _rel(outga, refga), _rel(outga, out16)
    # The e4m3 quantiser between the layers is discontinuous: how far does the ORACLE move from itself when its input carries rounding-level
    # noise (5e-4 relative, then re-rounded to bf16 - what separates any two bf16 implementations after one layer)?  That is the floor two
    # restatements of this mode can agree to; the bf16 mode under the same noise is shown beside it.
    latp = (lat.float() * (1 + 1e-3 * torch.randn(lat.shape, generator=g))).to(torch.bfloat16)
    floor8 = _rel(O.dit_forward(sd, ocfg, latp.float(), t, text, emulate_bf16=True, fp8_gemm=True), refg)
    floor16 = _rel(O.dit_forward(sd, ocfg, latp.float(), t, text, emulate_bf16=True), O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True))
    parity("dit_14B_width_fp8_gemm", rel_vs_e4m3_oracle=r1, rel_with_fp8_attention_vs_e4m3_oracle=r2, rel_fp8_modes_vs_bf16_mode=shift,
           oracle_e4m3_mode_moves_under_rounding_level_input_noise=floor8, oracle_bf16_mode_moves_under_the_same_noise=floor16)
    print(f"14B-width fp8-GEMM forward: rel vs e4m3 oracle {r1:.2e} (+fp8 attention {r2:.2e}); fp8 modes vs bf16 mode {shift:.2e}; the e4m3 oracle "
          f"moves {floor8:.2e} under 5e-4 input noise (the bf16 oracle {floor16:.2e})")
    assert r1 < 2.0 * floor8 and r2 < 2.0 * floor8, (r1, r2, floor8)   # measured 3.2e-2 against a 3.3e-2 self-distance: conditioning, not a kernel error
    # Two pipelines with a discontinuous quantiser between their layers: a 1-ulp bf16 difference in an activation near an e4m3 tie
    # moves that element by a whole e4m3 step (2^-3 relative), so ~3 % of the quantised elements differ between kernel path and
    # oracle and the forward can only be bounded to a few e-2 here.  The GEMM itself is pinned on identical quantised operands to
    # bf16 rounding (< 1.2e-3, bit-identical across tiles) by tests/test_kernels_gpu.py::test_gemm_fp8_every_tile_matches_e4m3_emulation.
    assert torch.isfinite(outga.float()).all() and r1 < 6e-2 and r2 < 6e-2, (r1, r2)
    assert shift < 9e-2, shift     # measured 4.6e-2
    # config #4 shards the denoise: per-token quantisation is token-local, so the sequence-parallel forward with e4m3 GEMMs (bf16
    # attention over the gathered slabs) equals the unsharded one bit for bit
    from vist3a_amd.wan.seqpar import ThreadWorld
    model.attn_dtype, model.sp_kv_split = "bf16", 1
    tw = ThreadWorld(2)
    outs = tw.run(lambda r: model(lat.cuda(), t.cuda(), text.cuda(), sp=tw.group(r))[0].clone())
    assert all(torch.equal(o, outg) for o in outs)
    # ... and with the e4m3 attention too: the ranks all-gather e4m3 K | V^T slabs and the fp8 flash kernel walks them in place
    model.attn_dtype = "fp8"
    outs = tw.run(lambda r: model(lat.cuda(), t.cuda(), text.cuda(), sp=tw.group(r))[0].clone())
    assert all(torch.equal(o, outga) for o in outs)


@pytest.mark.parametrize("width", ["1.3b", "14b"])
def test_time_tables_bit_identical_at_production_width(hip_lib, width):
    """WanDiT.time_tables (the time conditioning of a whole 50-step schedule as ONE batched pass of the embedder, handed to the fused
    loop as stride-0 batch views of one [L, S, 6, d] table) against the per-step `_time_conditioning(t.expand(B))` at the production
    widths: its bit-identity rests on the skinny / tile GEMM results being independent of the number of rows M (K = freq_dim = 256 in
    the first GEMM, K = d in the others) - asserted here at d = 1536 and d = 5120, not only on a tiny model."""
    import dataclasses
    from vist3a_amd.wan.dit import WAN_1_3B, WAN_14B, WanDiT
    from vist3a_amd.wan.scheduler import UniPCMultistepScheduler
    from vist3a_amd.wan.weights import random_dit_state_dict
    cfg = dataclasses.replace(WAN_1_3B if width == "1.3b" else WAN_14B, num_layers=2, text_dim=256)
    m = WanDiT(cfg, random_dit_state_dict(cfg, seed=1, device="cuda"))
    sch = UniPCMultistepScheduler(flow_shift=5.0)
    sch.set_timesteps(50, device="cuda")
    tabs = m.time_tables(sch.timesteps, 2)
    assert len(tabs) == 50
    for i in (0, 1, 17, 49):
        temb, mod = m._time_conditioning(sch.timesteps[i].to(torch.float32).expand(2).contiguous(), 2)
        assert torch.equal(tabs[i][0], temb) and torch.equal(tabs[i][1], mod), i
    assert tabs[3][1].stride(1) == 0 and tabs[3][1].shape == (2, 2, 6, cfg.dim)    # a view: nothing materialised per step / batch item


def test_cached_context_cross_attention_matches_flash_form_and_oracle(hip_lib, parity):
    """WanDiT.ctx_vo (default on): attn2 = sum_h softmax_h(q K^T) (V_h Wo_h^T) + bo with (V_h Wo_h^T) cached per prompt, against the
    reference's order of operations (flash attention, then to_out: ctx_vo = False) at production width, B = 2, 4096 tokens, two
    blocks, zero-padded 64 / 80-token prompts - and both against the oracle with the respective rounding points.  The two HIP forms must
    be as close to each other as the oracle's two forms are (the deviation costs bf16 noise, nothing else), each within the two-block
    tolerance of its own contract oracle."""
    import dataclasses
    from vist3a_amd.wan.dit import WAN_1_3B, WanDiT
    cfg = dataclasses.replace(WAN_1_3B, num_layers=2, text_dim=512)
    ocfg = O.WanDiTConfig(num_attention_heads=cfg.num_attention_heads, attention_head_dim=cfg.attention_head_dim, ffn_dim=cfg.ffn_dim,
                          num_layers=2, text_dim=cfg.text_dim, freq_dim=cfg.freq_dim)
    sd = {k: v.to(torch.bfloat16).float() for k, v in O.make_weights(ocfg, seed=21).items()}
    model = WanDiT(cfg, sd, device="cuda")
    assert model.ctx_vo
    g = torch.Generator().manual_seed(22)
    lat = torch.randn(2, 16, 4, 64, 64, generator=g).to(torch.bfloat16)
    text = (torch.randn(2, 512, cfg.text_dim, generator=g) * 0.5).to(torch.bfloat16).float()
    text[0, 64:] = 0
    text[1, 80:] = 0
    t = torch.tensor([650, 650])
    out_vo = model(lat.cuda(), t.cuda(), text.cuda())[0].float().cpu()
    ent = next(iter(model._ctx.values()))[1]
    assert ent[7] is not None and ent[8] == 96 and ent[5] == 88      # 88 merged keys per head, padded to 96: K = 12 x 96 = 1152
    model.ctx_vo = False
    model._ctx.clear()
    out_fl = model(lat.cuda(), t.cuda(), text.cuda())[0].float().cpu()
    assert next(iter(model._ctx.values()))[1][7] is None
    model.ctx_vo = True
    with torch.no_grad():
        ref_vo = O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True, flash=True, merge_padding=True, ctx_vo=True)
        ref_fl = O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True, flash=True, merge_padding=True)
    r = dict(hip_vo_vs_oracle_vo=_rel(out_vo, ref_vo), hip_flash_vs_oracle_flash=_rel(out_fl, ref_fl), hip_vo_vs_hip_flash=_rel(out_vo, out_fl),
             oracle_vo_vs_oracle_flash=_rel(ref_vo, ref_fl), hip_vo_vs_oracle_flash=_rel(out_vo, ref_fl))
    parity("dit_cached_context_cross_attention_two_blocks", **r)
    print("cached-context cross-attention (1.3B width, B = 2, N = 4096, 2 blocks):", {k: f"{v:.2e}" for k, v in r.items()})
    assert torch.isfinite(out_vo).all()
    assert r["hip_vo_vs_oracle_vo"] < 2 * TOL_ONE_BLOCK and r["hip_flash_vs_oracle_flash"] < 2 * TOL_ONE_BLOCK, r
    assert r["hip_vo_vs_hip_flash"] < 2.0 * max(r["oracle_vo_vs_oracle_flash"], 1e-3), r


def test_config4_wan14b_two_blocks_at_4096_tokens_matches_oracle(hip_lib, parity):
    """BASELINE config #4 AT SIZE: Wan-14B width (40 heads x 128 = 5120, FFN 13824), the CFG pair (B = 2) of a 13-view scene
    (N = 4096 tokens, M = 8192 GEMM rows: the ragged 5120 / 13824 tilings at their real M), 512-row zero-padded prompts - two blocks
    in bf16 against the oracle with the kernel contracts emulated (bf16 rounding points, bf16-P flash tiles, merged padding key),
    and one block in the config's fp8-attention mode against the oracle with the e4m3 rounding points."""
    from vist3a_amd.wan.dit import WanDiT, WanDiTConfig
    import fullsize_cases as FC
    import oracle_cache as OC
    case = FC.dit_config4_two_blocks()
    ocfg, sd, lat, text, t = case.ocfg, case.sd, case.lat, case.text, case.t
    model = WanDiT(WanDiTConfig(num_attention_heads=40, attention_head_dim=128, ffn_dim=13824, num_layers=2, text_dim=512, freq_dim=256), sd, device="cuda")
    out16 = model(lat.cuda(), t.cuda(), text.cuda())[0].float().cpu()
    out16_1 = model(lat.cuda(), t.cuda(), text.cuda(), num_layers=1)[0].float().cpu()
    model.attn_dtype = "fp8"
    out8 = model(lat.cuda(), t.cuda(), text.cuda(), num_layers=1)[0].float().cpu()    # the fp8 mode on ONE block (half the oracle time)
    model.attn_dtype = "bf16"
    torch.cuda.synchronize()
    # (the two oracle forwards take 85 + 53 s of host time: committed digest, tests/oracle_cache.py)
    od, live = OC.oracle(case.name, case.fingerprint, case.compute, sources=case.sources, case_fns=case.case_fns)
    r16, r8, shift = OC.rel(out16, od["ref16"]), OC.rel(out8, od["ref8"]), _rel(out8, out16_1)
    parity("dit_config4_14B_N4096_B2_two_blocks", rel_bf16_vs_contract_oracle=r16, rel_fp8_attention_vs_e4m3_oracle=r8,
           fp8_attention_mode_vs_bf16_mode=shift, oracle="live" if live else "digest fixture")
    print(f"config #4 at size (14B width, N=4096, B=2, 2 blocks): bf16 vs contract oracle {r16:.2e}; fp8 attention vs e4m3 oracle {r8:.2e}; "
          f"fp8 mode moves the output by {shift:.2e} ({'live' if live else 'fixture'})")
    assert out16.shape == lat.shape and torch.isfinite(out16).all() and torch.isfinite(out8).all()
    assert r16 < 6e-3 and r8 < 1.3e-2, (r16, r8)    # the N = 6144 / 1.3B-width two-block forward measured 2.6e-3; 14B width at 128 tokens 5.3e-3 / 6.3e-3


@pytest.mark.parametrize("P", [4, 8])
def test_seq_parallel_production_width_reads_gathered_slabs_in_place(hip_lib, P):
    """Production width and token count (4096 tokens, 2 blocks) over P virtual ranks: every shard holds 4096 / P keys (a multiple of
    64), so the flash kernel walks the all-gathered per-rank [K | V^T] slabs in place (`kv_seg`), with no reassembly copy between the
    all-gather and the attention launch - and the result stays bit-identical to the unsharded forward."""
    import dataclasses
    from vist3a_amd import ops
    from vist3a_amd.wan.dit import WAN_1_3B, WanDiT
    from vist3a_amd.wan.seqpar import ThreadWorld
    from vist3a_amd.wan.weights import random_dit_state_dict
    cfg = dataclasses.replace(WAN_1_3B, num_layers=2)
    m = WanDiT(cfg, random_dit_state_dict(cfg, seed=0, device="cuda"))
    g = torch.Generator().manual_seed(3)
    text = (torch.randn(1, 512, 4096, generator=g) * 0.1).cuda()
    text[:, 70:] = 0
    lat = torch.randn(1, 16, 4, 64, 64, generator=g).bfloat16().cuda()
    t = torch.tensor([700]).cuda()
    full = m(lat, t, text)[0].clone()
    seen, splits = [], []
    real = ops.attention
    def spy(*a, **kw):
        seen.append(kw.get("kv_seg", 0))
        splits.append(kw.get("kv_split", 1))
        return real(*a, **kw)
    ops.attention = spy
    try:
        w = ThreadWorld(P)
        m.sp_kv_split = 1     # every workgroup walks all keys in the unsharded order: bit-identical
        outs = w.run(lambda r: m(lat, t, text, sp=w.group(r))[0].clone())
        m.sp_kv_split = None  # default: the keys of a query block divided among workgroups so that the small launch fills the chip
        outs_split = w.run(lambda r: m(lat, t, text, sp=w.group(r))[0].clone())
        torch.cuda.synchronize()
    finally:
        ops.attention = real
    assert 4096 // P in seen, "self-attention did not take the in-place slab path"
    assert max(splits) > 1, "the default sequence-parallel path did not split the keys"
    for o in outs:
        assert torch.equal(o, full)
    for o in outs_split:
        assert torch.equal(o, outs_split[0]) and _rel(o, full) < 5e-3, _rel(o, full)   # bf16 rounding of the merged softmax


def test_sharded_forward_replays_from_a_hipgraph_bit_identically(hip_lib):
    """GraphedWanDiT(capture_sp=True): a rank's sequence-parallel forward at production width (1024 of 4096 tokens: P = 4, two blocks,
    both the exact and the default key-split / split-K modes) captured in a hipGraph WITH its per-block all-gathers and replayed on new
    inputs equals the eager sharded forward bit for bit.  The group is a device-side stand-in (every segment = the local slab, as in
    tools/sp_rank_time.py) because one GPU cannot host P capturing ranks; that RCCL's all-gather itself captures and replays is
    tests/test_boundary_gpu.py::test_rccl_all_gather_is_capturable_in_a_hipgraph, and the real multi-rank graph run is the second leg of
    test_rccl_scene_parallel_denoise_matches_single_gpu."""
    import dataclasses
    from vist3a_amd.wan.dit import WAN_1_3B, GraphedWanDiT, WanDiT
    from vist3a_amd.wan.weights import random_dit_state_dict

    class _Done:
        def wait(self):
            return None

    class SelfGather:
        graph_safe = True

        def __init__(self, world):
            self.world, self.rank, self.calls = world, 1, 0

        def all_gather(self, out, inp):
            self.calls += 1
            out.view(self.world, -1).copy_(inp.reshape(1, -1).expand(self.world, -1))
            return _Done()

    cfg = dataclasses.replace(WAN_1_3B, num_layers=2)
    m = WanDiT(cfg, random_dit_state_dict(cfg, seed=0, device="cuda"))
    gm = GraphedWanDiT(m, capture_sp=True)
    g = torch.Generator().manual_seed(5)
    text = (torch.randn(1, 512, 4096, generator=g) * 0.1).cuda()
    text[:, 70:] = 0
    for mode in (1, None):
        m.sp_kv_split = mode
        sp = SelfGather(4)
        for k, tval in enumerate((700, 300, 40)):
            lat = torch.randn(1, 16, 4, 64, 64, generator=g).bfloat16().cuda()
            t = torch.tensor([tval]).cuda()
            want = m(lat, t, text, sp=sp)[0].clone()
            calls = sp.calls
            got = gm(lat, t, text, sp=sp)[0].clone()
            if k > 0:
                assert sp.calls == calls, "the replay went through the Python forward"
            assert torch.equal(got, want)
    assert len(gm._graphs) == 2
    # a group that synchronises through the host is never captured
    from vist3a_amd.wan.seqpar import ThreadWorld
    assert ThreadWorld(2).group(0).graph_safe is False


def test_config4_seq_parallel_14b_width_P8_e4m3_slabs_in_place(hip_lib):
    """BASELINE config #4's sharding at its own width: Wan-14B (5120 wide, 40 heads), 4096 tokens over P = 8 virtual ranks (512 keys per
    rank), fp8 attention - the ranks all-gather E4M3 [K | V^T] slabs (half the bytes) and the fp8 flash kernel walks them in place;
    bit-identical to the unsharded fp8-attention forward in the exact mode, and so is the bf16 path over bf16 slabs."""
    import dataclasses
    from vist3a_amd import ops
    from vist3a_amd.wan.dit import WAN_14B, WanDiT
    from vist3a_amd.wan.seqpar import ThreadWorld
    from vist3a_amd.wan.weights import random_dit_state_dict
    cfg = dataclasses.replace(WAN_14B, num_layers=1, text_dim=512)
    m = WanDiT(cfg, random_dit_state_dict(cfg, seed=0, device="cuda"))
    g = torch.Generator().manual_seed(5)
    text = (torch.randn(1, 512, 512, generator=g) * 0.1).cuda()
    text[:, 70:] = 0
    lat = torch.randn(1, 16, 4, 64, 64, generator=g).bfloat16().cuda()
    t = torch.tensor([700]).cuda()
    P = 8
    w = ThreadWorld(P)
    m.sp_kv_split = 1
    seen = []
    real16, real8 = ops.attention, ops.attention_fp8
    ops.attention = lambda *a, **k: (seen.append(("bf16", k.get("kv_seg", 0))), real16(*a, **k))[1]
    ops.attention_fp8 = lambda *a, **k: (seen.append(("fp8", k.get("kv_seg", 0))), real8(*a, **k))[1]
    try:
        full16 = m(lat, t, text)[0].clone()
        outs16 = w.run(lambda r: m(lat, t, text, sp=w.group(r))[0].clone())
        m.attn_dtype = "fp8"
        full8 = m(lat, t, text)[0].clone()
        outs8 = w.run(lambda r: m(lat, t, text, sp=w.group(r))[0].clone())
        torch.cuda.synchronize()
    finally:
        ops.attention, ops.attention_fp8 = real16, real8
    assert ("bf16", 4096 // P) in seen and ("fp8", 4096 // P) in seen, "the sharded self-attention did not walk the gathered slabs in place"
    assert all(torch.equal(o, full16) for o in outs16)
    assert all(torch.equal(o, full8) for o in outs8)
    assert not torch.equal(full8, full16) and _rel(full8, full16) < 5e-2


def test_lora_adapter_forward_matches_unmerged_oracle(hip_lib, parity):
    """SURVEY A10.  The reference runs the DiT with its peft adapter UNMERGED (inference_t23d.py:74-78; targets attn1 / attn2
    {to_q, to_k, to_v, to_out.0}, r = 8, alpha = 16: train_vdm.py:369-384): y = bf16(W x) + bf16(bf16(B bf16(A x)) alpha/r), five bf16
    roundings per adapted projection under CUDA autocast.  The HIP path folds W + (alpha/r) B A in fp32 at load and rounds the sum to
    bf16 once.  Production width (12 x 128, FFN 8960), 1024 tokens, two blocks: HIP-merged vs the oracle that keeps the adapter
    unmerged with those rounding points; the gap between the oracle's own two forms is recorded beside it."""
    import dataclasses
    from vist3a_amd.wan.dit import WAN_1_3B, WanDiT, merge_lora_into_state_dict
    cfg = dataclasses.replace(WAN_1_3B, text_dim=256, num_layers=2)
    ocfg = O.WanDiTConfig(num_attention_heads=12, attention_head_dim=128, ffn_dim=8960, num_layers=2, text_dim=256, freq_dim=256)
    base = {k: v.to(torch.bfloat16).float() for k, v in O.make_weights(ocfg, seed=21).items()}
    sd_l, peft = O.with_lora_adapter(base, ocfg, std=0.05)          # |(alpha/r) B A| ~ 1.4e-2 per entry against |W| ~ 2e-2: a strong adapter
    merged = {k: v.clone() for k, v in base.items()}
    assert merge_lora_into_state_dict(merged, peft, alpha=16, r=8) == 16
    model = WanDiT(cfg, merged, device="cuda")
    g = torch.Generator().manual_seed(22)
    lat = torch.randn(1, 16, 1, 64, 64, generator=g).to(torch.bfloat16)
    text = (torch.randn(1, 96, 256, generator=g) * 0.5).to(torch.bfloat16).float()
    t = torch.tensor([620])
    out = model(lat.cuda(), t.cuda(), text.cuda())[0].float().cpu()
    kw = dict(emulate_bf16=True, flash=True)
    with torch.no_grad():
        ref_unmerged = O.dit_forward(sd_l, ocfg, lat.float(), t, text, **kw)
        ref_merged = O.dit_forward(merged, ocfg, lat.float(), t, text, **kw)
        ref_plain = O.dit_forward(base, ocfg, lat.float(), t, text, **kw)
    r_u, r_m, gap, effect = _rel(out, ref_unmerged), _rel(out, ref_merged), _rel(ref_merged, ref_unmerged), _rel(ref_unmerged, ref_plain)
    parity("dit_lora_adapter_two_blocks", rel_hip_merged_vs_oracle_unmerged=r_u, rel_hip_merged_vs_oracle_merged=r_m,
           oracle_merged_vs_unmerged=gap, adapter_effect_on_output=effect)
    print(f"LoRA forward: HIP(merged) vs oracle(unmerged) {r_u:.2e}, vs oracle(merged) {r_m:.2e}; oracle merged vs unmerged {gap:.2e}; "
          f"adapter moves the output by {effect:.2e}")
    assert effect > 10 * r_u                  # the adapter matters, so a dropped or mis-scaled one cannot pass
    assert r_m < 5.4e-3 and r_u < 6.6e-3, (r_m, r_u)   # measured 2.7e-3 / 3.3e-3 (the oracle's own two forms: 3.3e-3 apart)
