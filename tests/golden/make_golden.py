"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own modules (stub-imported from
/root/reference, only possible in the build container) on seeded inputs and on weights produced by the oracle's
`make_weights` functions (so the test side can regenerate identical weights without shipping them).

    python tests/golden/make_golden.py [name ...]

Each generator also asserts that the oracle restatement reproduces the reference output (fp32, tight tolerance)
before writing the fixture, so a drifting oracle cannot silently re-bless itself."""
from __future__ import annotations

import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT))

import torch
from safetensors.torch import save_file

import _ref_import

_ref_import.install()


def _save(name, tensors):
    out = HERE / f"{name}.safetensors"
    save_file({k: v.contiguous() for k, v in tensors.items()}, str(out))
    print(f"wrote {out} ({out.stat().st_size / 1024:.0f} KiB)")


def vae_decode_tiny():
    """AutoencoderKLWan(base_dim=16)._decode on z[1,16,3,8,8] -> [1,3,9,64,64] (chunked, cached reference path)."""
    import utils.wan_utils as W
    from oracle import wan_vae as OV
    cfg = OV.WanVAEConfig(base_dim=16)
    sd = OV.make_weights(cfg, seed=11)
    vae = W.AutoencoderKLWan(base_dim=16)
    missing = vae.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if k.startswith(("decoder.", "post_quant_conv"))]
    assert not missing.unexpected_keys
    z = torch.randn(1, 16, 3, 8, 8, generator=torch.Generator().manual_seed(21))
    with torch.no_grad():
        ref = vae._decode(z, return_dict=False)[0]
        mine = OV.decode(sd, cfg, z)
    err = (ref - mine).abs().max().item()
    sat = (ref.abs() >= 1.0).float().mean().item()
    print(f"vae_decode_tiny: oracle vs reference max abs err {err:.2e}; clamped fraction {sat:.3f}")
    assert err < 2e-5 and sat < 0.2
    _save("vae_decode_tiny", {"z": z, "out": ref})


def vae_encode_tiny():
    """AutoencoderKLWan(base_dim=16)._encode on x[1,3,9,64,64] -> [1,32,3,8,8] (chunks of 1,4,4 frames, cached reference path)."""
    import utils.wan_utils as W
    from oracle import wan_vae as OV
    cfg = OV.WanVAEConfig(base_dim=16)
    sd = OV.make_encoder_weights(cfg, seed=12)
    vae = W.AutoencoderKLWan(base_dim=16)
    missing = vae.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if k.startswith(("encoder.", "quant_conv"))]
    assert not missing.unexpected_keys
    x = torch.randn(1, 3, 9, 64, 64, generator=torch.Generator().manual_seed(22)).clamp(-1, 1)
    with torch.no_grad():
        ref = vae._encode(x)
        mine = OV.encode(sd, cfg, x)
        # a second, odd-shaped clip (5 frames = chunks 1+4, non-square) keeps the chunk/cache equivalence honest
        x2 = torch.randn(1, 3, 5, 32, 48, generator=torch.Generator().manual_seed(23)).clamp(-1, 1)
        ref2, mine2 = vae._encode(x2), OV.encode(sd, cfg, x2)
    err, err2 = (ref - mine).abs().max().item(), (ref2 - mine2).abs().max().item()
    print(f"vae_encode_tiny: oracle vs reference max abs err {err:.2e} / {err2:.2e}; out {tuple(ref.shape)} {tuple(ref2.shape)}")
    assert err < 2e-5 and err2 < 2e-5
    _save("vae_encode_tiny", {"x": x, "out": ref, "x2": x2, "out2": ref2})


def stitch_tiny():
    """stitched_model.py:92-107 trilinear T-upsample + stitching_layer_builder.py ConvSpec.build (replicate pad)."""
    from models.stitching_layer_builder import parse_conv_spec
    from oracle import recon as R
    spec = parse_conv_spec("conv3d_k5x3x3_o64_s1x2x2_p2x1x1")
    layer = spec.build(in_channels=16)
    g = torch.Generator().manual_seed(31)
    with torch.no_grad():
        layer.weight.copy_(torch.randn(layer.weight.shape, generator=g) / 27.0)
        layer.bias.copy_(torch.randn(layer.bias.shape, generator=g) * 0.1)
    lat = torch.randn(1, 16, 3, 8, 8, generator=g)
    T = lat.shape[2]
    with torch.no_grad():
        up = torch.nn.functional.interpolate(lat, size=[(T - 1) * 4 + 1, 8, 8], mode="trilinear", align_corners=True)
        ref = layer(up)
        mine = R.stitch_conv(R.upsample_T(lat), layer.weight, layer.bias, (1, 2, 2), (2, 1, 1))
    err = (ref - mine).abs().max().item()
    print(f"stitch_tiny: oracle vs reference {err:.2e}; out {tuple(ref.shape)}")
    assert err < 1e-5 and layer.padding_mode == "replicate"
    _save("stitch_tiny", {"latent": lat, "weight": layer.weight.detach(), "bias": layer.bias.detach(), "out": ref})


def stitch_dilated_grouped():
    """stitching_layer_builder.py:21-42: `ConvSpec.build(in_channels, groups=)` hands dilation and groups to nn.Conv3d(padding_mode="replicate").
    Two layers of the reference's own builder on one T-upsampled latent: a dilated one (`_d1x2x2`: the spec grammar's optional field) and a
    dilated + grouped one (groups = 4).  Inputs, parameters ([Cout, Cin / groups, *k] for the grouped layer) and outputs become the fixture."""
    from models.stitching_layer_builder import parse_conv_spec
    g = torch.Generator().manual_seed(41)
    lat = torch.randn(1, 16, 3, 12, 12, generator=g)
    T = lat.shape[2]
    out = {"latent": lat}
    with torch.no_grad():
        up = torch.nn.functional.interpolate(lat, size=[(T - 1) * 4 + 1, 12, 12], mode="trilinear", align_corners=True)
        for name, spec_s, groups in (("dil", "conv3d_k3x3x3_o64_s1x2x2_p1x2x2_d1x2x2", 1), ("dilgrp", "conv3d_k3x3x3_o64_s1x1x1_p2x2x2_d2x2x2", 4)):
            spec = parse_conv_spec(spec_s)
            layer = spec.build(in_channels=16, groups=groups)
            layer.weight.copy_(torch.randn(layer.weight.shape, generator=g) / 12.0)
            layer.bias.copy_(torch.randn(layer.bias.shape, generator=g) * 0.1)
            assert layer.padding_mode == "replicate" and layer.groups == groups and tuple(layer.dilation) == tuple(spec.dilation)
            ref = layer(up)
            out.update({f"{name}_weight": layer.weight.detach(), f"{name}_bias": layer.bias.detach(), f"{name}_out": ref})
            print(f"stitch_dilated_grouped: {spec_s} groups {groups}: weight {tuple(layer.weight.shape)} out {tuple(ref.shape)}")
    _save("stitch_dilated_grouped", out)


def _bare(cls):
    import torch.nn as nn
    o = cls.__new__(cls)
    nn.Module.__init__(o)
    return o


RECON_TINY = dict(C=64, heads=1, n_dino=22, depth=24, cam_heads=2, cam_trunk=2, features=32, oc=(16, 32, 64, 64))


def build_reference_stitched(cfg, sd):
    """Assemble the reference's AnySplatStitched object at reduced width WITHOUT its checkpoint-downloading
    constructors (from_pretrained), from the reference's own sub-module classes, and load `sd` into it."""
    import types as _t
    from functools import partial

    import torch.nn as nn
    from models.anysplat_stitched import AnySplatStitched
    from third_party_model.anysplat.src.model.encoder.anysplat import EncoderAnySplat
    from third_party_model.anysplat.src.model.encoder.common.gaussian_adapter import GaussianAdapterCfg, UnifiedGaussianAdapter
    from third_party_model.anysplat.src.model.encoder.heads.vggt_dpt_gs_head import VGGT_DPT_GS_Head
    from third_party_model.anysplat.src.model.encoder.vggt.heads.camera_head import CameraHead
    from third_party_model.anysplat.src.model.encoder.vggt.heads.dpt_head import DPTHead
    from third_party_model.anysplat.src.model.encoder.vggt.layers.attention import MemEffAttention
    from third_party_model.anysplat.src.model.encoder.vggt.layers.block import Block
    from third_party_model.anysplat.src.model.encoder.vggt.layers.vision_transformer import DinoVisionTransformer
    from third_party_model.anysplat.src.model.encoder.vggt.models.aggregator import Aggregator

    C = cfg.C
    agg = Aggregator(img_size=518, patch_size=14, embed_dim=C, depth=cfg.depth, num_heads=cfg.heads, patch_embed="conv")
    agg.use_checkpoint = False
    dino = DinoVisionTransformer(img_size=518, patch_size=14, embed_dim=C, depth=cfg.n_dino, num_heads=cfg.heads, mlp_ratio=4,
                                 block_fn=partial(Block, attn_class=MemEffAttention), num_register_tokens=4,
                                 interpolate_antialias=True, interpolate_offset=0.0, block_chunks=0, init_values=1.0)
    del dino.patch_embed  # what convert_model_to_stitched_model does (blocks already at the post-deletion count)
    agg.patch_embed = dino
    enc = _bare(EncoderAnySplat)
    enc.aggregator = agg
    enc.camera_head = CameraHead(dim_in=2 * C, trunk_depth=cfg.cam_trunk, num_heads=cfg.cam_heads)
    for blk in enc.camera_head.trunk:
        pass
    enc.depth_head = DPTHead(dim_in=2 * C, output_dim=2, activation="exp", conf_activation="expp1", features=cfg.features,
                             out_channels=list(cfg.oc))
    enc.gaussian_adapter = UnifiedGaussianAdapter(GaussianAdapterCfg(gaussian_scale_min=0.5, gaussian_scale_max=15.0, sh_degree=cfg.sh_degree))
    enc.raw_gs_dim = 1 + enc.gaussian_adapter.d_in
    enc.gaussian_param_head = VGGT_DPT_GS_Head(dim_in=2 * C, patch_size=(14, 14), output_dim=enc.raw_gs_dim + 1, activation="norm_exp",
                                               conf_activation="expp1", features=256, out_channels=list(cfg.oc))
    enc.voxel_size = cfg.voxel_size
    enc.cfg = _t.SimpleNamespace(pred_head_type="depth", render_conf=False, voxelize=cfg.voxelize, opacity_conf=False, conf_threshold=0.1,
                                 opacity_mapping=_t.SimpleNamespace(initial=0.0, final=0.0, warm_up=1))
    model = _bare(AnySplatStitched)
    model.encoder = enc
    model.grad_checkpointing = False
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys[:5]
    miss = [k for k in res.missing_keys if "mask_token" not in k and "patch_embed.patch_embed" not in k]
    assert not miss, miss[:8]
    return model.eval()


def recon_tiny():
    """The reference's AnySplatStitched.forward at reduced width (C=64) but full depth (22 DINO + 24x2 aggregator blocks),
    2 views @28x28, voxelisation active.  Pins the whole R3-R17 restatement end to end."""
    from oracle import recon as R
    cfg = R.ReconCfg(**RECON_TINY)
    sd = R.make_recon_weights(cfg, seed=41)
    model = build_reference_stitched(cfg, sd)
    g = torch.Generator().manual_seed(42)
    S, H, W = 2, 28, 28
    lat = torch.randn(1, cfg.C, S, H // 14, W // 14, generator=g)
    img = torch.rand(1, 3, S, H, W, generator=g) * 2 - 1
    import contextlib, io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        ref = model(lat, img, False)
        mine = R.recon_forward(sd, cfg, lat, img)
    gs = ref.gaussians

    def chk(name, a, b, tol=2e-4):
        e = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-9)
        print(f"  {name:22s} rel-max err {e:.2e}  shape {tuple(b.shape)}")
        assert e < tol, name

    print("recon_tiny: oracle vs reference")
    chk("pose_enc", mine["pred_pose_enc_list"][-1], ref.pred_pose_enc_list[-1])
    chk("depth", mine["depth"], ref.depth_dict["depth"])
    chk("means", mine["gaussians"]["means"], gs.means)
    chk("scales", mine["gaussians"]["scales"], gs.scales)
    chk("rotations", mine["gaussians"]["rotations"], gs.rotations)
    chk("harmonics", mine["gaussians"]["harmonics"], gs.harmonics)
    chk("opacities", mine["gaussians"]["opacities"], gs.opacities)
    chk("covariances", mine["gaussians"]["covariances"], gs.covariances, 2e-3)
    chk("c2w", mine["pred_context_pose"]["extrinsic"], ref.pred_context_pose["extrinsic"])
    chk("intrinsic", mine["pred_context_pose"]["intrinsic"], ref.pred_context_pose["intrinsic"])
    U = gs.means.shape[1]
    print(f"  voxels U={U} of M={S * H * W}")
    _save("recon_tiny", {
        "latent": lat, "image": img,
        "pose_enc_list": torch.stack(ref.pred_pose_enc_list), "depth": ref.depth_dict["depth"],
        "means": gs.means, "scales": gs.scales, "rotations": gs.rotations, "harmonics": gs.harmonics.half(),
        "opacities": gs.opacities, "c2w": ref.pred_context_pose["extrinsic"], "intrinsic": ref.pred_context_pose["intrinsic"],
        "scene_scale": ref.infos["scene_scale"].reshape(1),
    })


RECON_MH = dict(C=128, heads=2, n_dino=22, depth=24, cam_heads=4, cam_trunk=2, features=32, oc=(16, 32, 64, 64))


def recon_mh():
    """recon_tiny at width 128 = TWO heads of 64 (the production head size): pins the per-head layout of qkv / q_norm / k_norm / RoPE2D /
    attention (vggt/layers/attention.py:49-80, rope.py:154-188) and a 4-head camera trunk, which a one-head model cannot see.
    3 views @28x42 (non-square, so hp != wp in the RoPE position grid)."""
    from oracle import recon as R
    cfg = R.ReconCfg(**RECON_MH)
    sd = R.make_recon_weights(cfg, seed=43)
    model = build_reference_stitched(cfg, sd)
    g = torch.Generator().manual_seed(44)
    S, H, W = 3, 28, 42
    lat = torch.randn(1, cfg.C, S, H // 14, W // 14, generator=g)
    img = torch.rand(1, 3, S, H, W, generator=g) * 2 - 1
    import contextlib, io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        ref = model(lat, img, False)
        mine = R.recon_forward(sd, cfg, lat, img)
    gs = ref.gaussians

    def chk(name, a, b, tol=2e-4):
        e = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-9)
        print(f"  {name:22s} rel-max err {e:.2e}  shape {tuple(b.shape)}")
        assert e < tol, name

    print("recon_mh: oracle vs reference")
    chk("pose_enc", mine["pred_pose_enc_list"][-1], ref.pred_pose_enc_list[-1])
    chk("depth", mine["depth"], ref.depth_dict["depth"])
    chk("means", mine["gaussians"]["means"], gs.means)
    chk("scales", mine["gaussians"]["scales"], gs.scales)
    chk("opacities", mine["gaussians"]["opacities"], gs.opacities)
    chk("c2w", mine["pred_context_pose"]["extrinsic"], ref.pred_context_pose["extrinsic"])
    # a one-head restatement of the same weights must NOT reproduce it (the fixture really sees the head split)
    import copy
    c1 = copy.copy(cfg)
    c1.heads, c1.cam_heads = 1, 1
    sd1 = dict(sd)
    for k in list(sd1):
        if "attn.q_norm" in k or "attn.k_norm" in k:
            sd1[k] = sd1[k].repeat(2)
    with torch.no_grad():
        wrong = R.recon_forward(sd1, c1, lat, img)
    d = (wrong["depth"] - ref.depth_dict["depth"]).abs().max().item() / ref.depth_dict["depth"].abs().max().item()
    print(f"  one-head restatement differs in depth by {d:.2e} (must be large)")
    assert d > 1e-2
    _save("recon_mh", {
        "latent": lat, "image": img,
        "pose_enc_list": torch.stack(ref.pred_pose_enc_list), "depth": ref.depth_dict["depth"],
        "means": gs.means, "scales": gs.scales, "rotations": gs.rotations, "opacities": gs.opacities,
        "c2w": ref.pred_context_pose["extrinsic"], "intrinsic": ref.pred_context_pose["intrinsic"],
        "scene_scale": ref.infos["scene_scale"].reshape(1),
    })


def voxel_collide():
    """EncoderAnySplat.voxelizaton_with_fusion on points engineered to collide (several points per voxel, negative
    coordinates, exact .5 rounding ties): integer keys / inverse / counts are the bit-exact contract."""
    from third_party_model.anysplat.src.model.encoder.anysplat import EncoderAnySplat
    from oracle import recon as R
    g = torch.Generator().manual_seed(51)
    V, C, H, W = 2, 7, 16, 16
    base = torch.randint(-6, 6, (V, 3, H, W), generator=g).float() * 0.002
    jitter = (torch.rand(V, 3, H, W, generator=g) - 0.5) * 0.0019
    pts = base + jitter
    pts[0, :, 0, :4] = torch.tensor([0.001, -0.001, 0.003]).view(3, 1)  # exact half-voxel ties (round-half-even)
    feat = torch.randn(V, C, H, W, generator=g)
    conf = torch.randn(V, H, W, generator=g) * 2
    enc = _bare(EncoderAnySplat)
    with torch.no_grad():
        rp, rf = enc.voxelizaton_with_fusion(feat, pts, 0.002, conf)
        vp, vf, keys, inv, cnt = R.voxelize_with_fusion(feat, pts, 0.002, conf)
    e1, e2 = (rp - vp).abs().max().item(), (rf - vf).abs().max().item()
    print(f"voxel_collide: U={keys.shape[0]} of M={V * H * W}; max count {cnt.max().item()}; err pts {e1:.2e} feats {e2:.2e}")
    assert e1 < 1e-6 and e2 < 1e-5 and cnt.max() >= 3
    _save("voxel_collide", {"pts": pts, "feat": feat, "conf": conf, "voxel_pts": rp, "voxel_feats": rf,
                            "keys": keys, "inverse": inv.to(torch.int32), "counts": cnt.to(torch.int32)})


def recon_tiny_conf():
    """Same reduced-width model as recon_tiny with encoder.cfg.voxelize = False and render_conf = True: the confidence-quantile
    mask branch of AnySplatStitched.forward (anysplat_stitched.py:381-387, 441-446).  Stores the depth confidences, the quantile,
    the mask and the compacted Gaussian means / opacities (boolean-mask order = row-major)."""
    from oracle import recon as R
    cfg = R.ReconCfg(**RECON_TINY)
    sd = R.make_recon_weights(cfg, seed=41)
    model = build_reference_stitched(cfg, sd)
    model.encoder.cfg.voxelize, model.encoder.cfg.render_conf, model.encoder.cfg.conf_threshold = False, True, 0.1
    g = torch.Generator().manual_seed(42)
    S, H, W = 2, 28, 28
    lat = torch.randn(1, cfg.C, S, H // 14, W // 14, generator=g)
    img = torch.rand(1, 3, S, H, W, generator=g) * 2 - 1
    import contextlib, io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        ref, _anchor, _conf, dconf = model(lat, img, True)   # train=True also returns depth_conf (anysplat_stitched.py:501-514)
    q = torch.quantile(dconf.flatten(0, 1), 0.1)
    mask = dconf > q
    assert torch.equal(mask, ref.depth_dict["conf_valid_mask"])
    gs = ref.gaussians
    assert gs.means.shape[1] == int(mask.sum())
    print(f"recon_tiny_conf: kept {int(mask.sum())} of {mask.numel()} points; quantile {q.item():.6f}")
    _save("recon_tiny_conf", {"latent": lat, "image": img, "depth_conf": dconf, "quantile": q.reshape(1), "mask": mask.to(torch.uint8),
                              "means": gs.means, "opacities": gs.opacities, "scales": gs.scales})


def recon_tiny_conf_b2():
    """recon_tiny_conf with a BATCH of two scenes (b = 2): the reference takes the render_conf quantile over `depth_conf.flatten(0, 1)`
    WITHOUT a dim (anysplat_stitched.py:381-387) - one threshold spanning the batch - then masks every scene with it and pads the per-scene
    lists to the larger count (:441-453).  Stores both scenes' inputs, the depth confidences, the batch quantile, the mask, the kept counts and
    the padded Gaussian means / opacities."""
    from oracle import recon as R
    cfg = R.ReconCfg(**RECON_TINY)
    sd = R.make_recon_weights(cfg, seed=41)
    model = build_reference_stitched(cfg, sd)
    model.encoder.cfg.voxelize, model.encoder.cfg.render_conf, model.encoder.cfg.conf_threshold = False, True, 0.25
    g = torch.Generator().manual_seed(43)
    S, H, W = 2, 28, 28
    lat = torch.randn(2, cfg.C, S, H // 14, W // 14, generator=g)
    lat[1] *= 1.7                                   # a second scene with a different confidence distribution: unequal kept counts
    img = torch.rand(2, 3, S, H, W, generator=g) * 2 - 1
    img[1] *= 0.4
    import contextlib, io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        ref, _anchor, _conf, dconf = model(lat, img, True)
    q = torch.quantile(dconf.flatten(0, 1), 0.25)
    mask = dconf > q
    assert torch.equal(mask, ref.depth_dict["conf_valid_mask"])
    kept = mask.flatten(1).sum(1)
    gs = ref.gaussians
    assert gs.means.shape[:2] == (2, int(kept.max())) and kept[0] != kept[1]
    per_scene_q = [torch.quantile(dconf[b].flatten(), 0.25).item() for b in range(2)]
    print(f"recon_tiny_conf_b2: kept {kept.tolist()} of {mask[0].numel()} points per scene; batch quantile {q.item():.6f} (per-scene {per_scene_q})")
    _save("recon_tiny_conf_b2", {"latent": lat, "image": img, "depth_conf": dconf, "quantile": q.reshape(1), "mask": mask.to(torch.uint8),
                                 "kept": kept, "means": gs.means, "opacities": gs.opacities, "scales": gs.scales})


DENOISE_LOOP_BLOCK_SHA256 = "097e7953bc87ac4d7da05ef37ee830a6a838ba17f9436763f3e9ec6890727a62"   # dedented train_vdm.py:586-624 as reviewed


def denoise_loop_ref():
    """SURVEY row A0.  The pipeline class the reference calls (diffusers WanPipeline) is not in the image, but the reference holds ONE in-tree
    spelling of the same CFG denoise loop: /root/reference/train_vdm.py:586-624 (B = 2 batching [cond | uncond], `pred.chunk(2)` order,
    `noise_uncond + g (noise_pred - noise_uncond)`, `scheduler.step(noise.float(), t, latents.float())`, `latents / latents_std +
    latents_mean`).  Its source lines are read from the reference file HERE, dedented and executed around a stub transformer
    (tests/stub_transformer.py) and the oracle's UniPC scheduler; inputs and outputs become the fixture.  What this pins is the loop;
    the scheduler arithmetic inside `step` stays "unpinned" (oracle/unipc.py header)."""
    import contextlib
    import inspect
    import textwrap
    from types import SimpleNamespace
    sys.path.insert(0, str(ROOT / "tests"))
    from stub_transformer import StubTransformer
    import utils.wan_utils as W
    from oracle.unipc import OracleUniPC
    lines = (Path(_ref_import.REF) / "train_vdm.py").read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if "for i, t in tqdm(enumerate(timesteps)):" in l)
    end = next(i for i, l in enumerate(lines) if "latents = latents / latents_std + latents_mean" in l)
    assert 580 < start < end < 640, (start, end)        # train_vdm.py:586-624
    block = textwrap.dedent("\n".join(lines[start:end + 1]))
    # /root/reference is untrusted content and these lines are about to be EXECUTED with the developer's privileges: they must be exactly
    # the 39 lines that were read and reviewed when the fixture was written, not "whatever sits between the two marker lines today".
    import hashlib
    got = hashlib.sha256(block.encode()).hexdigest()
    if got != DENOISE_LOOP_BLOCK_SHA256:
        raise SystemExit(f"train_vdm.py:{start + 1}-{end + 1} changed (sha256 {got}, reviewed {DENOISE_LOOP_BLOCK_SHA256}): refusing to exec it.\n"
                         "Review the new text, then update DENOISE_LOOP_BLOCK_SHA256:\n" + block)
    sig = inspect.signature(W.AutoencoderKLWan.__init__).parameters
    mean, std = sig["latents_mean"].default, sig["latents_std"].default
    steps, shift, guidance, text_dim = 10, 5.0, 5.25, 32
    g = torch.Generator().manual_seed(77)
    lat0 = torch.randn(1, 16, 3, 8, 8, generator=g)
    pe, ne = torch.randn(1, 12, text_dim, generator=g), torch.randn(1, 12, text_dim, generator=g)
    stub = StubTransformer(text_dim)
    osch = OracleUniPC(flow_shift=shift)
    osch.set_timesteps(steps)

    class Sched:   # diffusers' call surface over the oracle scheduler
        timesteps = osch.timesteps

        @staticmethod
        def step(model_output, t, sample, return_dict=False):
            assert model_output.dtype == torch.float32 and sample.dtype == torch.float32
            return (osch.step(model_output, sample),)

    ns = dict(torch=torch, tqdm=lambda it: it, accelerator=SimpleNamespace(autocast=contextlib.nullcontext),
              pipeline=SimpleNamespace(transformer=stub, scheduler=Sched), timesteps=osch.timesteps, t_train=torch.tensor([]),
              latents=lat0.clone(), prompt_embeds=pe, neg_prompt_embed=ne, guidance_scale=guidance,
              latents_mean=torch.tensor(mean).view(1, 16, 1, 1, 1).float(), latents_std=1.0 / torch.tensor(std).view(1, 16, 1, 1, 1).float())
    exec(compile(block, "train_vdm.py:586-624", "exec"), ns)
    out = ns["latents"]
    assert len(stub.calls) == steps and all(c == ((2, 16, 3, 8, 8), (2,), (2, 12, text_dim)) for c in stub.calls)
    # the loop up to (not including) the de-normalisation, re-derived from the final value
    final = (out - ns["latents_mean"]) * ns["latents_std"]
    print(f"denoise_loop_ref: executed train_vdm.py:{start + 1}-{end + 1}; |latents| {final.abs().mean():.3f}")
    _save("denoise_loop_ref", {"latents0": lat0, "prompt_embeds": pe, "negative_prompt_embeds": ne, "denormalised": out,
                               "config": torch.tensor([steps, shift, guidance, text_dim], dtype=torch.float64)})


GENERATORS = {f.__name__: f for f in [vae_decode_tiny, vae_encode_tiny, stitch_tiny, stitch_dilated_grouped, recon_tiny, recon_mh, recon_tiny_conf, recon_tiny_conf_b2, voxel_collide,
                                      denoise_loop_ref]}

if __name__ == "__main__":
    names = sys.argv[1:] or list(GENERATORS)
    for n in names:
        GENERATORS[n]()
