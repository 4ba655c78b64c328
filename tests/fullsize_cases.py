"""The three production-size oracle cases whose outputs are committed as digests (tests/oracle_cache.py): seeded inputs, weights and the
oracle computation of each, with NO dependence on the GPU or on the HIP library - `tests/golden/make_fullsize_oracle.py` runs exactly
these on host cores to write the digests, and the `-m gpu` tests build the same case to get the same inputs and the digest's fingerprint.
Each case carries the oracle modules whose SOURCE is hashed into the fingerprint: editing oracle/recon.py or oracle/wan_dit.py (or this
file) invalidates the committed digest loudly instead of leaving a stale golden that still "matches"."""
from __future__ import annotations

import time
from types import SimpleNamespace

import torch

import oracle_cache as OC
from oracle import recon as R
from oracle import wan_dit as O

def _ov():
    from oracle import wan_vae as OV
    return OV


RECON_MH = dict(C=128, heads=2, n_dino=22, depth=24, cam_heads=4, cam_trunk=2, features=32, oc=(16, 32, 64, 64))
DIT_DEPTHS = (1, 2, 4, 8, 16, 30)


def recon_full_weights():
    """Width-1024 / 16-head reconstruction weights from the oracle's seeded generator (LayerScale 0.3 in the aggregator: every block
    contributes, unlike the 0.01 of bench.py's reference-style initialisation), aggregator values bf16-representable like the
    reference's bf16-stored aggregator (anysplat.py:144)."""
    from vist3a_amd.recon.weights import round_aggregator_to_bf16
    cfg = R.ReconCfg()
    assert (cfg.C, cfg.heads, cfg.cam_heads) == (1024, 16, 16)
    return cfg, round_aggregator_to_bf16(R.make_recon_weights(cfg, seed=51))


def recon_full(weights=None) -> SimpleNamespace:
    """13 views @448, width 1024: tests/test_fullsize_gpu.py::test_full_size_reconstruction_matches_oracle"""
    ocfg, sd = weights if weights is not None else recon_full_weights()
    g = torch.Generator().manual_seed(52)
    w = (torch.randn(1024, 16, 5, 3, 3, generator=g) * 0.08).to(torch.bfloat16).float()
    b = torch.randn(1024, generator=g) * 0.1
    lat = torch.randn(1, 16, 4, 64, 64, generator=g)
    img = (torch.rand(1, 3, 13, 448, 448, generator=g) * 2 - 1).to(torch.bfloat16).float()
    S, H = 13, 448

    def compute():
        t0 = time.time()
        feat_c = R.stitch_conv(R.upsample_T(lat), w, b, (1, 2, 2), (2, 1, 1), emulate_bf16=True)
        ctaps = R.backbone(sd, feat_c, 1, S, (H, H), ocfg.heads, ocfg.n_dino, ocfg.depth, emulate_bf16=True)
        t1 = time.time()
        ora = R.recon_forward(sd, ocfg, feat_c, img, toks=ctaps)
        t2 = time.time()
        # plain fp32 backbone as well (informational: how far the reference's own rounding points move the taps)
        ftaps = R.backbone(sd, R.stitch_conv(R.upsample_T(lat), w, b, (1, 2, 2), (2, 1, 1)), 1, S, (H, H), ocfg.heads, ocfg.n_dino, ocfg.depth)
        t3 = time.time()
        d = {f"tap{i}": c[0] for i, c in enumerate(ctaps)}
        d.update({f"tap{i}_fp32": c[0] for i, c in enumerate(ftaps)})
        d.update(pose=ora["pred_pose_enc_list"][-1], depth=ora["depth"], depth_conf=ora["depth_conf"], raw_gs=ora["raw_gs"][:, :, :83],
                 gs_conf=ora["raw_gs"][:, :, 83], c2w=ora["pred_context_pose"]["extrinsic"], intrinsic=ora["pred_context_pose"]["intrinsic"],
                 voxels=ora["gaussians"]["means"].shape[1], seconds_backbone_contract=t1 - t0, seconds_heads=t2 - t1, seconds_backbone_fp32=t3 - t2)
        return d

    fp = OC.checksum(lat, img, w, b, sd["encoder.aggregator.frame_blocks.0.attn.qkv.weight"], sd["encoder.gaussian_param_head.scratch.output_conv2.2.weight"])
    return SimpleNamespace(name="recon_full_C1024_S13", fingerprint=fp, sources=(R,), case_fns=(recon_full_weights, recon_full), compute=compute, ocfg=ocfg, sd=sd, w=w, b=b, lat=lat, img=img, S=S, H=H)


def recon_config3() -> SimpleNamespace:
    """21 views @448 at width 128: tests/test_fullsize_gpu.py::test_config3_21_view_reconstruction_layout_matches_oracle"""
    ocfg = R.ReconCfg(**RECON_MH)
    sd = R.make_recon_weights(ocfg, seed=71)
    g = torch.Generator().manual_seed(72)
    w = torch.randn(128, 16, 5, 3, 3, generator=g) * 0.08
    b = torch.randn(128, generator=g) * 0.1
    S, H = 21, 448
    lat = torch.randn(1, 16, 6, 64, 64, generator=g)
    img = torch.rand(1, 3, S, H, H, generator=g) * 2 - 1

    def compute():
        feat = R.stitch_conv(R.upsample_T(lat), w, b, (1, 2, 2), (2, 1, 1), emulate_bf16=True)
        ora = R.recon_forward(sd, ocfg, feat, img, emulate_bf16=True)                       # the reference's GPU rounding points, fp32 heads
        dev = R.recon_forward(sd, ocfg, feat, img, dpt_bf16=True, toks=ora["taps"])         # + bf16 DPT heads (the opt-in dpt_precision="bf16")
        o32 = R.recon_forward(sd, ocfg, R.stitch_conv(R.upsample_T(lat), w, b, (1, 2, 2), (2, 1, 1)), img)   # plain fp32 (informational)
        d = {}
        for tag, o in (("c", ora), ("d", dev), ("f", o32)):
            d.update({f"{tag}_pose": o["pred_pose_enc_list"][-1], f"{tag}_depth": o["depth"], f"{tag}_depth_conf": o["depth_conf"],
                      f"{tag}_raw_gs": o["raw_gs"][:, :, :83], f"{tag}_raw_all": o["raw_gs"]})
        d.update({f"c_tap{i}": c[0] for i, c in enumerate(ora["taps"])})
        d.update({f"f_tap{i}": c[0] for i, c in enumerate(o32["taps"])})
        d["voxels"] = ora["gaussians"]["means"].shape[1]
        return d

    fp = OC.checksum(lat, img, w, b, sd["encoder.aggregator.frame_blocks.0.attn.qkv.weight"])
    return SimpleNamespace(name="recon_config3_S21_width128", fingerprint=fp, sources=(R,), case_fns=(recon_config3,), compute=compute, ocfg=ocfg, sd=sd, w=w, b=b, lat=lat, img=img,
                           S=S, H=H)


def dit_full_depth() -> SimpleNamespace:
    """Wan-1.3B geometry, 30 blocks, 4096 tokens: tests/test_dit_gpu.py::test_full_depth_production_size_forward_matches_oracle"""
    ocfg = O.WanDiTConfig(num_attention_heads=12, attention_head_dim=128, ffn_dim=8960, num_layers=30, text_dim=512, freq_dim=256)
    sd = {k: v.to(torch.bfloat16).float() for k, v in O.make_weights(ocfg, seed=11).items()}
    g = torch.Generator().manual_seed(12)
    lat = torch.randn(1, 16, 4, 64, 64, generator=g).to(torch.bfloat16)
    text = (torch.randn(1, 512, ocfg.text_dim, generator=g) * 0.5).to(torch.bfloat16).float()
    text[:, 77:] = 0
    t = torch.tensor([700])

    def compute():
        ref = O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True)
        # the same forward with the CONTRACT differences of the HIP path emulated as well: bf16 P per 64-key flash tile (every flash
        # kernel has that term, the reference's SDPA included), the merged zero-padding key of the cross-attention, and the
        # cross-attention's cached-context order of operations (ctx_vo)
        taps = {L: None for L in DIT_DEPTHS}
        ref_c = O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True, flash=True, merge_padding=True, ctx_vo=True, depth_outputs=taps)
        d = dict(ref=ref, ref_c=ref_c, ref_rms=ref.pow(2).mean().sqrt().item())
        d.update({f"depth{L}": taps[L] for L in DIT_DEPTHS})
        return d

    fp = OC.checksum(lat, text, sd["blocks.0.attn1.to_q.weight"], sd["blocks.29.ffn.net.2.weight"])
    return SimpleNamespace(name="dit_full_depth_30_blocks_N4096", fingerprint=fp, sources=(O,), case_fns=(dit_full_depth,), compute=compute, ocfg=ocfg, sd=sd, lat=lat, text=text, t=t)


def dit_config4_two_blocks() -> SimpleNamespace:
    """BASELINE config #4 at size - Wan-14B width, the CFG pair of a 13-view scene (B = 2, 4096 tokens), two blocks:
    tests/test_dit_gpu.py::test_config4_wan14b_two_blocks_at_4096_tokens_matches_oracle"""
    ocfg = O.WanDiTConfig(num_attention_heads=40, attention_head_dim=128, ffn_dim=13824, num_layers=2, text_dim=512, freq_dim=256)
    sd = {k: v.to(torch.bfloat16).float() for k, v in O.make_weights(ocfg, seed=4).items()}
    g = torch.Generator().manual_seed(44)
    lat = torch.randn(2, 16, 4, 64, 64, generator=g).to(torch.bfloat16)
    text = (torch.randn(2, 512, 512, generator=g) * 0.5).to(torch.bfloat16).float()
    text[0, 64:] = 0
    text[1, 80:] = 0
    t = torch.tensor([611, 611])

    def compute():
        t0 = time.time()
        ref16 = O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True, flash=True, merge_padding=True, ctx_vo=True)
        t1 = time.time()
        ref8 = O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True, fp8_attn=True, flash=True, merge_padding=True, ctx_vo=True, num_layers=1)
        return dict(ref16=ref16, ref8=ref8, seconds_bf16_two_blocks=t1 - t0, seconds_fp8_one_block=time.time() - t1)

    fp = OC.checksum(lat, text, sd["blocks.0.attn1.to_q.weight"], sd["blocks.1.ffn.net.2.weight"])
    return SimpleNamespace(name="dit_config4_14B_two_blocks_N4096_B2", fingerprint=fp, sources=(O,), case_fns=(dit_config4_two_blocks,), compute=compute, ocfg=ocfg, sd=sd, lat=lat, text=text, t=t)


def vae_full() -> SimpleNamespace:
    """Wan VAE decoder at production size (base_dim 96, latent [1,16,4,64,64] -> 13 x 512^2), contract oracle:
    tests/test_fullsize_gpu.py::test_full_size_vae_decode_matches_oracle"""
    from oracle import wan_vae as OV
    cfg = OV.WanVAEConfig()
    sd = OV.make_weights(cfg, seed=31)
    z = torch.randn(1, 16, 4, 64, 64, generator=torch.Generator().manual_seed(32))

    def compute():
        t0 = time.time()
        ref = OV.decode(sd, cfg, z, emulate_bf16=True)
        return dict(ref=ref, clamped_fraction=(ref.abs() >= 1.0).float().mean().item(), seconds=time.time() - t0)

    fp = OC.checksum(z, sd["decoder.conv_in.weight"], sd["decoder.conv_out.weight"])
    return SimpleNamespace(name="vae_full_base96_13x512", fingerprint=fp, sources=(OV,), case_fns=(vae_full,), compute=compute, cfg=cfg, sd=sd, z=z)


CONFIG4_DEPTHS = (1, 2, 4, 8)


def dit_config4_eight_blocks() -> SimpleNamespace:
    """BASELINE config #4 on eight of its 40 blocks, 14B width, B = 2 x 512 tokens, bf16 contract and e4m3-attention oracles with their
    error-vs-depth taps: tests/test_dit_gpu.py::test_config4_14b_width_eight_blocks_match_oracle"""
    ocfg = O.WanDiTConfig(num_attention_heads=40, attention_head_dim=128, ffn_dim=13824, num_layers=8, text_dim=256, freq_dim=256)
    sd = {k: v.to(torch.bfloat16).float() for k, v in O.make_weights(ocfg, seed=44).items()}
    g = torch.Generator().manual_seed(45)
    lat = torch.randn(2, 16, 1, 32, 64, generator=g).to(torch.bfloat16)
    text = (torch.randn(2, 96, 256, generator=g) * 0.5).to(torch.bfloat16).float()
    text[0, 60:] = 0
    text[1, 70:] = 0
    t = torch.tensor([611, 611])

    def compute():
        taps16, taps8 = {L: None for L in CONFIG4_DEPTHS}, {L: None for L in CONFIG4_DEPTHS}
        O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True, flash=True, merge_padding=True, ctx_vo=True, depth_outputs=taps16)
        O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True, fp8_attn=True, merge_padding=True, ctx_vo=True, depth_outputs=taps8)
        d = {f"bf16_depth{L}": taps16[L] for L in CONFIG4_DEPTHS}
        d.update({f"fp8_depth{L}": taps8[L] for L in CONFIG4_DEPTHS})
        return d

    fp = OC.checksum(lat, text, sd["blocks.0.attn1.to_q.weight"], sd["blocks.7.ffn.net.2.weight"])
    return SimpleNamespace(name="dit_config4_14B_eight_blocks", fingerprint=fp, sources=(O,), case_fns=(dit_config4_eight_blocks,), compute=compute,
                           ocfg=ocfg, sd=sd, lat=lat, text=text, t=t)


CASES = {"vae_full": vae_full, "dit_config4_eight_blocks": dit_config4_eight_blocks, "recon_full": recon_full, "recon_config3": recon_config3, "dit_full_depth": dit_full_depth, "dit_config4_two_blocks": dit_config4_two_blocks}
# digest name -> (oracle modules, case functions) whose source it depends on (what each case passes as `sources` / `case_fns`; lets a CPU
# test check every committed digest against the current sources without building the cases' gigabytes of weights)
SOURCES = {"recon_full_C1024_S13": ((R,), (recon_full_weights, recon_full)), "recon_config3_S21_width128": ((R,), (recon_config3,)),
           "dit_full_depth_30_blocks_N4096": ((O,), (dit_full_depth,)), "dit_config4_14B_two_blocks_N4096_B2": ((O,), (dit_config4_two_blocks,)),
           "dit_config4_14B_eight_blocks": ((O,), (dit_config4_eight_blocks,)), "vae_full_base96_13x512": ((_ov(),), (vae_full,))}
