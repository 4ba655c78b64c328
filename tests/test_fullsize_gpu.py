"""PRODUCTION-SIZE parity (`-m gpu`): the stages whose outputs round 2 only ever compared at reduced width.

  * Wan VAE decoder, base_dim 96, latent [1,16,4,64,64] -> 13 x 512^2       vs oracle.wan_vae.decode        (utils/wan_utils.py:745-1117)
  * stitched Conv3d + AnySplat reconstruction, width 1024 / 16 heads, 13 views @448
                                                                            vs oracle.recon.recon_forward   (models/anysplat_stitched.py:167-525)
  * BASELINE config #3 (21 views): a production-width two-block DiT forward at N = 6144 tokens vs the oracle, the 21-view
    reconstruction layout (21 x 1032 rows, 21 609 keys in the global attention) at full resolution vs the oracle at reduced width,
    and a production-width S = 21 run checked through size-independent properties.

The VAE / DiT oracles run live on the GPU box's host cores (~1 minute); the two reconstruction oracles (11 + 4.5 minutes live) come from
committed digests of the same oracle code's outputs (tests/oracle_cache.py; V3A_LIVE_ORACLE=1 runs them live).  Nothing here reads
/root/reference.  Every measured error goes to parity.json through the `parity` fixture; asserts sit at <= 2x the value measured on MI355X."""
import pytest
import torch

from oracle import recon as R
from oracle import wan_dit as O
from oracle import wan_vae as OV
import fullsize_cases as FC
import oracle_cache as OC

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def test_full_size_vae_decode_matches_oracle(hip_lib, parity):
    """Primary gate: the HIP decoder against the oracle with the reference's GPU rounding points (`emulate_bf16`: CUDA-autocast bf16
    convs / SDPA, fp32 WanRMS_norm + SiLU; oracle/wan_vae.py docstring) at production size.  The fp32 figure is informational and
    taken on a quarter-size clip (same 96-wide network, latent [1,16,2,32,32] -> 5 x 256^2) where both oracle forms run in seconds:
    it shows how far bf16 rounding alone moves this network (oracle-contract vs oracle-fp32) next to HIP vs either."""
    import time
    from vist3a_amd.wan.vae import WanVAEConfig, WanVAEDecoder
    cfg = OV.WanVAEConfig()
    assert cfg.base_dim == 96
    sd = OV.make_weights(cfg, seed=31)
    dec = WanVAEDecoder(WanVAEConfig(), sd)
    zs = torch.randn(1, 16, 2, 32, 32, generator=torch.Generator().manual_seed(33))
    outs = dec.decode(zs.cuda(), return_dict=False)[0].float().cpu()
    with torch.no_grad():
        rs32, rsc = OV.decode(sd, cfg, zs), OV.decode(sd, cfg, zs, emulate_bf16=True)
    small = dict(rel_vs_contract=_rel(outs, rsc), rel_vs_fp32=_rel(outs, rs32), contract_vs_fp32=_rel(rsc, rs32))
    case = FC.vae_full()
    assert all(torch.equal(case.sd[k], sd[k]) for k in ("decoder.conv_in.weight", "decoder.conv_out.weight"))     # the same seeded weights
    z = case.z
    out = dec.decode(z.cuda(), return_dict=False)[0].float().cpu()
    torch.cuda.synchronize()
    # (the production-size contract oracle takes 1-1.5 minutes of host time: committed digest, tests/oracle_cache.py)
    od, live = OC.oracle(case.name, case.fingerprint, case.compute, sources=case.sources, case_fns=case.case_fns)
    assert out.shape == (1, 3, 13, 512, 512)
    r = OC.rel(out, od["ref"])
    sat = float(od["clamped_fraction"].item())
    mx = (OC.digest(out) - od["ref"]).abs().max().item()       # (over the digest's 16 k sampled positions)
    t_oracle = float(od["seconds"].item())
    parity("vae_decode_full_size_base96", rel_vs_contract_oracle=r, max_abs=mx, clamped_fraction=sat, quarter_size=small, oracle_seconds=t_oracle)
    print(f"full-size VAE decode (base_dim 96, 13 x 512^2): rel vs contract oracle {r:.3e} max abs {mx:.3e} clamped {sat:.3f} "
          f"(oracle {t_oracle:.0f} s); quarter size: vs contract {small['rel_vs_contract']:.3e}, vs fp32 {small['rel_vs_fp32']:.3e}, "
          f"contract vs fp32 {small['contract_vs_fp32']:.3e}")
    assert torch.isfinite(out).all() and sat < 0.5
    # MI355X, round 4: 1.48e-2 vs the contract oracle (1.34e-2 vs fp32 in round 3), quarter size 1.51e-2 / 1.35e-2 - while the oracle's own
    # two forms are 1.37e-2 apart: 35 bf16 conv layers put ANY two restatements of this decoder (different fp32 summation order is enough)
    # ~1.4e-2 from each other, i.e. the rounding-point emulation cannot bring two implementations closer than the network's conditioning
    # (2^-9 / sqrt(3) per rounding x sqrt(~100 roundings along a path)).  The kernel-level errors are 6e-5 (tests/test_kernels_gpu.py).
    # Gates: <= 2x measured, and HIP no further from the contract oracle than 2x the oracle's contract-vs-fp32 spread.
    assert r < VAE_FULL_GATE and small["rel_vs_contract"] < VAE_FULL_GATE, (r, small)
    assert small["rel_vs_contract"] < 2.0 * small["contract_vs_fp32"] and r < 2.0 * small["contract_vs_fp32"], (r, small)
    assert small["rel_vs_fp32"] < 2.6e-2        # informational figure, round-3 gate (1.3e-2 measured at full size)


VAE_FULL_GATE = 2.9e-2    # <= 2x the measured HIP-vs-contract figure (1.48e-2 on MI355X, profiles/r4/parity.json)


# (`recon_full`: the width-1024 / 16-head reconstruction weights, a session fixture of tests/conftest.py shared with
#  tests/test_production_blocks_gpu.py)


def _stitched(sd, rcfg_kw, C, res=512):
    from vist3a_amd.models.anysplat_stitched import AnySplatWeights
    from vist3a_amd.models.stitched_model import StitchVAE3D
    from vist3a_amd.models.stitching_layer_builder import parse_conv_spec
    from vist3a_amd.recon.engine import ReconCfg
    return StitchVAE3D(None, AnySplatWeights(dict(sd), ReconCfg(**rcfg_kw)), "cuda", "enc_blocks_2",
                       parse_conv_spec(f"conv3d_k5x3x3_o{C}_s1x2x2_p2x1x1"), resolution=res)


def test_full_size_reconstruction_matches_oracle(recon_full, parity):
    """13 views @448, width 1024, 16 heads x 64, 22 DINO + 24 frame + 24 global blocks, camera / depth / Gaussian heads, voxel fusion.
    Primary gate: against the oracle with the reference's GPU rounding points (`emulate_bf16`: bf16 Linear / SDPA / LayerScale outputs,
    fp32 LayerNorm, bf16 DINO stream, fp32 aggregator stream, fp32 heads - oracle/recon.py docstring).  One oracle pass (backbone in
    contract mode, fp32 heads on its taps); the figures against the plain fp32 oracle (round 3: taps 9.2e-3 .. 7.1e-3) are informational and
    taken in the 21-view layout test below, where all three oracle forms run in a minute."""
    case = FC.recon_full(recon_full)
    ocfg, sd, w, b, lat, img, S, H = case.ocfg, case.sd, case.w, case.b, case.lat, case.img, case.S, case.H
    model = _stitched(sd, {}, 1024)
    model.stitching_layer.weight.data, model.stitching_layer.bias.data = w, b
    eo, anchor, conf, dconf = model.forward_with_latent(lat.cuda(), img.cuda(), train=True)
    torch.cuda.synchronize()
    eng = model.stitched_3d_model.engine()
    _, geo = eng.token_workspace(S, H, H)
    taps = [t.view(S, geo["Pp"], -1)[:, :geo["P"]].float().cpu() for t in geo["taps"]]
    ora, live = OC.oracle(case.name, case.fingerprint, case.compute, sources=case.sources, case_fns=case.case_fns)
    tap_c = [OC.rel(t, ora[f"tap{i}"]) for i, t in enumerate(taps)]
    tap_32 = [OC.rel(t, ora[f"tap{i}_fp32"]) for i, t in enumerate(taps)]
    floor = [OC.rel_dd(ora[f"tap{i}"], ora[f"tap{i}_fp32"]) for i in range(len(taps))]
    e = dict(pose=OC.rel(eo.pred_pose_enc_list[-1], ora["pose"]), depth=OC.rel(eo.depth_dict["depth"], ora["depth"]),
             depth_conf=OC.rel(dconf, ora["depth_conf"]), raw_gs=OC.rel(anchor, ora["raw_gs"]), gs_conf=OC.rel(conf, ora["gs_conf"]),
             c2w=OC.rel(eo.pred_context_pose["extrinsic"], ora["c2w"]), intrinsic=OC.rel(eo.pred_context_pose["intrinsic"], ora["intrinsic"]))
    U, Uo = eo.gaussians.means.shape[1], int(ora["voxels"].item())
    secs = {k[8:]: round(float(v.item()), 1) for k, v in ora.items() if k.startswith("seconds_")}
    parity("recon_full_size_C1024_H16_S13", taps_vs_contract=tap_c, taps_vs_fp32=tap_32, taps_contract_vs_fp32=floor, voxels=U, voxels_oracle=Uo,
           oracle="live" if live else "digest fixture", oracle_seconds=secs, **e)
    print("full-size recon taps vs contract", [f"{t:.2e}" for t in tap_c], "vs fp32", [f"{t:.2e}" for t in tap_32], "contract vs fp32",
          [f"{t:.2e}" for t in floor], {k: f"{v:.2e}" for k, v in e.items()}, "voxels", U, "oracle", Uo, "live" if live else "fixture", secs)
    assert all(torch.isfinite(t).all() for t in taps)
    # MI355X (round 4, contract oracle): taps 9.7e-3 / 8.6e-3 / 8.1e-3 / 7.5e-3, pose 2.8e-3, depth 4.0e-3, depth_conf 1.8e-3, raw_gs 9.4e-3,
    # gs_conf 1.4e-2, c2w 5.2e-3 - the same as against the plain fp32 oracle in round 3 (9.2e-3 .. 7.1e-3): like the DiT (test_dit_gpu.py,
    # full-depth test), HIP, the contract oracle and the fp32 oracle are MUTUALLY ~1e-2 apart after 70 bf16 blocks - rounding-point
    # conditioning, not a kernel error.  Gates: <= 2x measured, and HIP no further from the contract oracle than 2x the oracle's own
    # contract-vs-fp32 spread.
    assert max(tap_c) < RECON_GATES["taps"], tap_c
    assert all(c < 2.0 * f for c, f in zip(tap_c, floor)), (tap_c, floor)
    for k in ("pose", "depth", "depth_conf", "raw_gs", "gs_conf", "c2w", "intrinsic"):
        assert e[k] < RECON_GATES[k], (k, e[k])
    assert abs(U - Uo) <= 0.03 * Uo


RECON_GATES = dict(taps=1.9e-2, pose=5.6e-3, depth=7.9e-3, depth_conf=3.5e-3, raw_gs=1.9e-2, gs_conf=2.8e-2, c2w=1.0e-2, intrinsic=1e-4)


def test_config3_21_view_dit_forward_matches_oracle(hip_lib, parity):
    """BASELINE config #3 geometry at production width: 21 views = 6 latent frames x 32 x 32 = 6144 tokens, two blocks, against the oracle
    with the kernel contracts emulated (bf16 rounding points, bf16-P flash tiles, merged padding key)."""
    import dataclasses
    from vist3a_amd.wan.dit import WAN_1_3B, WanDiT
    cfg = dataclasses.replace(WAN_1_3B, text_dim=512, num_layers=2)
    ocfg = O.WanDiTConfig(num_attention_heads=12, attention_head_dim=128, ffn_dim=8960, num_layers=2, text_dim=512, freq_dim=256)
    sd = {k: v.to(torch.bfloat16).float() for k, v in O.make_weights(ocfg, seed=61).items()}
    model = WanDiT(cfg, sd, device="cuda")
    g = torch.Generator().manual_seed(62)
    lat = torch.randn(2, 16, 6, 64, 64, generator=g).to(torch.bfloat16)
    text = (torch.randn(2, 512, 512, generator=g) * 0.5).to(torch.bfloat16).float()
    text[0, 64:] = 0
    text[1, 80:] = 0
    t = torch.tensor([450, 450])
    out = model(lat.cuda(), t.cuda(), text.cuda())[0].float().cpu()
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.dit_forward(sd, ocfg, lat.float(), t, text, emulate_bf16=True, flash=True, merge_padding=True, ctx_vo=True)
    r = _rel(out, ref)
    parity("dit_config3_N6144_two_blocks", rel_vs_contract_oracle=r)
    print(f"config #3 DiT (N=6144, 2 blocks): rel vs contract oracle {r:.2e}")
    assert out.shape == lat.shape and torch.isfinite(out).all()
    assert r < 5.2e-3, r     # measured 2.6e-3 (round 3 also ran the plain fp32 oracle: 5.0e-3, profiles/r3/parity.json)


RECON_MH = FC.RECON_MH


def test_config3_21_view_reconstruction_layout_matches_oracle(hip_lib, parity):
    """The 21-view token layout at FULL resolution (21 x 1032 padded rows, 1029 valid; global attention over 21 609 keys with the
    per-view mask; 21 x 448^2 = 4.2 M points into the voxeliser) at width 128 / two heads so that the oracle finishes in a minute."""
    case = FC.recon_config3()
    ocfg, sd, w, b, lat, img, S, H = case.ocfg, case.sd, case.w, case.b, case.lat, case.img, case.S, case.H
    model = _stitched(sd, RECON_MH, 128)
    model.stitching_layer.weight.data, model.stitching_layer.bias.data = w, b
    eo, anchor, conf, dconf = model.forward_with_latent(lat.cuda(), img.cuda(), train=True)
    torch.cuda.synchronize()
    eng = model.stitched_3d_model.engine()
    _, geo = eng.token_workspace(S, H, H)
    taps = [t_.view(S, geo["Pp"], -1)[:, :geo["P"]].float().cpu() for t_ in geo["taps"]]
    od, live = OC.oracle(case.name, case.fingerprint, case.compute, sources=case.sources, case_fns=case.case_fns)
    errs = lambda tag: dict(pose=OC.rel(eo.pred_pose_enc_list[-1], od[tag + "_pose"]), depth=OC.rel(eo.depth_dict["depth"], od[tag + "_depth"]),
                            depth_conf=OC.rel(dconf, od[tag + "_depth_conf"]), raw_gs=OC.rel(anchor, od[tag + "_raw_gs"]))
    e, ed, e32 = errs("c"), errs("d"), errs("f")
    nt = len(taps)
    tap_c, tap_32 = [OC.rel(a, od[f"c_tap{i}"]) for i, a in enumerate(taps)], [OC.rel(a, od[f"f_tap{i}"]) for i, a in enumerate(taps)]
    floor = [OC.rel_dd(od[f"c_tap{i}"], od[f"f_tap{i}"]) for i in range(nt)]
    price = dict(depth=OC.rel_dd(od["d_depth"], od["c_depth"]), raw_gs=OC.rel_dd(od["d_raw_all"], od["c_raw_all"]))
    U, Uo = eo.gaussians.means.shape[1], int(od["voxels"].item())
    parity("recon_config3_S21_448_width128", voxels=U, voxels_oracle=Uo, vs_contract=e, vs_contract_with_bf16_dpt_heads=ed, vs_fp32=e32,
           bf16_dpt_heads_move_the_oracle_by=price, taps_vs_contract=tap_c, taps_vs_fp32=tap_32, taps_contract_vs_fp32=floor,
           oracle="live" if live else "digest fixture")
    print("config #3 recon layout (S=21 @448, width 128): vs contract", {k: f"{v:.2e}" for k, v in e.items()}, "vs contract + bf16 DPT heads",
          {k: f"{v:.2e}" for k, v in ed.items()}, "price of the bf16 heads", {k: f"{v:.2e}" for k, v in price.items()}, "vs fp32", {k: f"{v:.2e}" for k, v in e32.items()},
          "taps vs contract", [f"{x:.2e}" for x in tap_c], "vs fp32", [f"{x:.2e}" for x in tap_32], "contract vs fp32", [f"{x:.2e}" for x in floor],
          "voxels", U, "oracle", Uo, "live" if live else "fixture")
    assert all(c < 2.0 * f for c, f in zip(tap_c, floor)), (tap_c, floor)   # measured 1.2e-2 .. 1.0e-2 against a 1.1e-2 oracle spread
    assert e["pose"] < 9e-3 and e["depth"] < 7.4e-3 and e["depth_conf"] < 3.8e-3 and e["raw_gs"] < 1.65e-2   # round 3 vs fp32: 4.6e-3 / 3.7e-3 / 1.9e-3 / 8.2e-3
    assert abs(U - Uo) <= 0.03 * Uo      # measured -1.3 %


def test_config3_21_view_production_width_properties(recon_full, parity):
    """S = 21 at width 1024 (no oracle at this size within the time budget): size-independent properties - run-to-run bit identity of
    the whole stitched forward, every point lands in exactly one voxel (counts sum to 21 x 448^2), confidences >= 1, quaternions
    unit, scales inside the adapter's clamp, and the first 13 views' per-view depth statistics finite and positive."""
    ocfg, sd = recon_full
    model = _stitched(sd, {}, 1024)
    g = torch.Generator().manual_seed(82)
    model.stitching_layer.weight.data = torch.randn(1024, 16, 5, 3, 3, generator=g) * 0.08
    model.stitching_layer.bias.data = torch.randn(1024, generator=g) * 0.1
    S, H = 21, 448
    lat = torch.randn(1, 16, 6, 64, 64, generator=g).cuda()
    img = (torch.rand(1, 3, S, H, H, generator=g) * 2 - 1).cuda()
    raw = {}
    real_package = model.stitched_3d_model.package

    def spy(out, *args, **kw):   # the engine's raw outputs (voxel integer data) on their way into EncoderOutput
        raw.update(out)
        return real_package(out, *args, **kw)
    model.stitched_3d_model.package = spy
    a = model.forward_with_latent(lat, img, train=False)
    means_a, depth_a = a.gaussians.means.clone(), a.depth_dict["depth"].clone()
    counts, inverse, keys = raw["voxel_counts"].clone(), raw["voxel_inverse"].clone(), raw["voxel_keys"].clone()
    bb = model.forward_with_latent(lat, img, train=False)
    assert torch.equal(means_a, bb.gaussians.means) and torch.equal(depth_a, bb.depth_dict["depth"])
    assert torch.equal(counts, raw["voxel_counts"]) and torch.equal(inverse, raw["voxel_inverse"])
    assert int(counts.sum()) == S * H * H and int(inverse.max()) == keys.shape[0] - 1 and int(inverse.min()) == 0
    assert torch.equal(torch.bincount(inverse.long(), minlength=keys.shape[0]).to(counts.dtype), counts)
    k = keys.long().cpu()
    lin = (k[:, 0] + 2 ** 20) * 2 ** 42 + (k[:, 1] + 2 ** 20) * 2 ** 21 + (k[:, 2] + 2 ** 20)
    assert torch.all(lin[1:] > lin[:-1])      # unique, lexicographically sorted voxel keys
    gs = a.gaussians
    U = gs.means.shape[1]
    assert torch.isfinite(gs.means).all() and torch.isfinite(gs.covariances).all() and torch.isfinite(gs.harmonics).all()
    assert (depth_a > 0).all() and torch.isfinite(depth_a).all()
    qn = gs.rotations.norm(dim=-1)
    assert (qn - 1).abs().max() < 1e-4
    assert gs.scales.min() >= 0 and gs.scales.max() <= 0.3 + 1e-6
    assert (gs.opacities >= 0).all() and (gs.opacities <= 1).all()
    assert 0 < U <= S * H * H
    parity("recon_config3_S21_production_width_properties", gaussians=U, points=S * H * H, deterministic=True)
