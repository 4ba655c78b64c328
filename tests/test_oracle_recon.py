"""Reconstruction oracle pinned against golden vectors produced by the reference's own modules
(tests/golden/make_golden.py: stitch_tiny, recon_tiny = the real AnySplatStitched.forward at reduced width, voxel_collide)."""
from pathlib import Path

import torch
from safetensors.torch import load_file

from oracle import recon as R

G = Path(__file__).parent / "golden"
RECON_TINY = dict(C=64, heads=1, n_dino=22, depth=24, cam_heads=2, cam_trunk=2, features=32, oc=(16, 32, 64, 64))


def test_stitch_golden():
    g = load_file(str(G / "stitch_tiny.safetensors"))
    out = R.stitch_conv(R.upsample_T(g["latent"]), g["weight"], g["bias"], (1, 2, 2), (2, 1, 1))
    assert torch.allclose(out, g["out"], atol=1e-5)
    # align_corners=True T-upsample is exact 1-D lerp with weights j/4
    lat = g["latent"]
    up = R.upsample_T(lat)
    assert torch.allclose(up[:, :, 4], lat[:, :, 1], atol=1e-6) and torch.allclose(up[:, :, 1], 0.75 * lat[:, :, 0] + 0.25 * lat[:, :, 1], atol=1e-6)


def test_recon_forward_golden():
    g = load_file(str(G / "recon_tiny.safetensors"))
    cfg = R.ReconCfg(**RECON_TINY)
    sd = R.make_recon_weights(cfg, seed=41)
    with torch.no_grad():
        o = R.recon_forward(sd, cfg, g["latent"], g["image"])
    gs = o["gaussians"]
    assert torch.allclose(torch.stack(o["pred_pose_enc_list"]), g["pose_enc_list"], atol=2e-5)
    assert torch.allclose(o["depth"], g["depth"], rtol=2e-4, atol=1e-5)
    assert gs["means"].shape == g["means"].shape
    assert torch.allclose(gs["means"], g["means"], atol=2e-5)
    assert torch.allclose(gs["scales"], g["scales"], rtol=1e-4, atol=1e-8)
    assert torch.allclose(gs["rotations"], g["rotations"], atol=2e-5)
    assert torch.allclose(gs["opacities"], g["opacities"], atol=2e-5)
    assert torch.allclose(gs["harmonics"], g["harmonics"].float(), atol=2e-3, rtol=2e-3)  # stored as fp16
    assert torch.allclose(o["pred_context_pose"]["extrinsic"], g["c2w"], atol=2e-5)
    assert torch.allclose(o["pred_context_pose"]["intrinsic"], g["intrinsic"], atol=2e-5)
    assert torch.allclose(o["scene_scale"].reshape(1), g["scene_scale"], rtol=1e-5)


RECON_MH = dict(C=128, heads=2, n_dino=22, depth=24, cam_heads=4, cam_trunk=2, features=32, oc=(16, 32, 64, 64))


def test_recon_multi_head_golden():
    """Two heads of 64 + a four-head camera trunk, 3 views @28x42 (tests/golden/make_golden.py::recon_mh, the reference's own
    AnySplatStitched.forward): pins the per-head split of qkv / q_norm / k_norm / RoPE2D that the one-head fixture cannot see."""
    g = load_file(str(G / "recon_mh.safetensors"))
    cfg = R.ReconCfg(**RECON_MH)
    sd = R.make_recon_weights(cfg, seed=43)
    with torch.no_grad():
        o = R.recon_forward(sd, cfg, g["latent"], g["image"])
    gs = o["gaussians"]
    assert torch.allclose(torch.stack(o["pred_pose_enc_list"]), g["pose_enc_list"], atol=2e-5)
    assert torch.allclose(o["depth"], g["depth"], rtol=2e-4, atol=1e-5)
    assert gs["means"].shape == g["means"].shape and torch.allclose(gs["means"], g["means"], atol=2e-5)
    assert torch.allclose(gs["scales"], g["scales"], rtol=1e-4, atol=1e-8)
    assert torch.allclose(gs["rotations"], g["rotations"], atol=2e-5)
    assert torch.allclose(gs["opacities"], g["opacities"], atol=2e-5)
    assert torch.allclose(o["pred_context_pose"]["extrinsic"], g["c2w"], atol=2e-5)
    assert torch.allclose(o["pred_context_pose"]["intrinsic"], g["intrinsic"], atol=2e-5)


def test_voxel_golden_bit_exact_integers():
    g = load_file(str(G / "voxel_collide.safetensors"))
    vp, vf, keys, inv, cnt = R.voxelize_with_fusion(g["feat"], g["pts"], 0.002, g["conf"])
    assert torch.equal(keys, g["keys"]) and torch.equal(inv.int(), g["inverse"]) and torch.equal(cnt.int(), g["counts"])
    assert torch.allclose(vp, g["voxel_pts"], atol=1e-6) and torch.allclose(vf, g["voxel_feats"], atol=1e-5)
    # lexicographic order and permutation invariance of the integer part
    k = keys.long()
    lin = (k[:, 0] + 2 ** 20) * 2 ** 42 + (k[:, 1] + 2 ** 20) * 2 ** 21 + (k[:, 2] + 2 ** 20)
    assert torch.all(lin[1:] > lin[:-1])
    assert cnt.sum().item() == g["pts"].numel() // 3


def test_small_pieces():
    # rope2d is identity at position 0 and norm preserving
    t = torch.randn(1, 2, 5, 64)
    pos = torch.tensor([[[0, 0], [1, 2], [3, 1], [7, 7], [2, 0]]])
    r = R.rope2d(t, pos)
    assert torch.allclose(r[:, :, 0], t[:, :, 0]) and torch.allclose(r.norm(dim=-1), t.norm(dim=-1), rtol=1e-5)
    # quaternion (xyzw) -> rotation: identity and orthonormality
    assert torch.allclose(R.quat_to_mat(torch.tensor([0.0, 0, 0, 1])), torch.eye(3))
    q = torch.randn(4, 4)
    M = R.quat_to_mat(q)
    assert torch.allclose(M @ M.transpose(-1, -2), torch.eye(3).expand(4, 3, 3), atol=1e-5)
    # opacity map with exponent 1 is the identity
    p = torch.rand(10)
    assert torch.allclose(R.map_pdf_to_opacity(p), p, atol=1e-7)
    # sh mask bands
    m = R.sh_mask(4)
    assert m[0] == 1 and abs(m[1].item() - 0.025) < 1e-9 and abs(m[24].item() - 0.1 * 0.25 ** 4) < 1e-9
    # slice_expand_and_flatten: first frame gets token 0, others token 1
    tok = torch.arange(2 * 3.0).view(1, 2, 1, 3)
    o = R.slice_expand_and_flatten(tok, 1, 4)
    assert torch.equal(o[0], tok[0, 0]) and all(torch.equal(o[i], tok[0, 1]) for i in (1, 2, 3))


def test_conf_mask_branch_golden():
    """voxelize=False + render_conf=True (anysplat_stitched.py:381-387, 441-446) against the reference's own forward."""
    g = load_file(str(G / "recon_tiny_conf.safetensors"))
    cfg = R.ReconCfg(**RECON_TINY, voxelize=False, render_conf=True, conf_threshold=0.1)
    sd = R.make_recon_weights(cfg, seed=41)
    with torch.no_grad():
        o = R.recon_forward(sd, cfg, g["latent"], g["image"])
    assert torch.allclose(o["depth_conf"], g["depth_conf"], rtol=2e-4, atol=1e-5)
    # the oracle's own confidences differ from the reference's in the last bits: compare the mask where the margin is not round-off
    far = (g["depth_conf"] - g["quantile"]).abs() > 1e-3
    assert torch.equal(o["conf_valid_mask"][far], g["mask"].bool()[far])
    if torch.equal(o["conf_valid_mask"], g["mask"].bool()):
        assert torch.allclose(o["gaussians"]["means"], g["means"], atol=2e-5)
        assert torch.allclose(o["gaussians"]["opacities"], g["opacities"], atol=2e-5)
    # the mask arithmetic itself, on the reference's confidences: index-exact
    q = torch.quantile(g["depth_conf"].flatten(0, 1), 0.1)
    assert torch.equal(q.reshape(1), g["quantile"]) and torch.equal(g["depth_conf"] > q, g["mask"].bool())


def test_bf16_contract_modes_are_roundings_of_the_pinned_form():
    """emulate_bf16 (CUDA-autocast rounding points in the backbone) and dpt_bf16 (the HIP path's bf16 DPT heads) are OFF by default -
    the golden-pinned fp32 form above is what runs - and move the outputs by bf16-noise amounts only; precomputed taps are honoured."""
    from vist3a_amd.recon.weights import round_aggregator_to_bf16
    g = load_file(str(G / "recon_mh.safetensors"))
    cfg = R.ReconCfg(**RECON_MH)
    sd = round_aggregator_to_bf16(R.make_recon_weights(cfg, seed=43))
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    with torch.no_grad():
        o32 = R.recon_forward(sd, cfg, g["latent"], g["image"])
        oe = R.recon_forward(sd, cfg, g["latent"], g["image"], emulate_bf16=True)
        od = R.recon_forward(sd, cfg, g["latent"], g["image"], emulate_bf16=True, dpt_bf16=True, toks=oe["taps"])
        again = R.recon_forward(sd, cfg, g["latent"], g["image"])
    assert not R._EMU and not R._DPT16
    assert torch.equal(again["depth"], o32["depth"])                       # switches restored: the default is the fp32 form
    t = [rel(a, b) for a, b in zip(oe["taps"], o32["taps"])]
    assert all(5e-4 < x < 5e-2 for x in t), t
    assert 1e-4 < rel(oe["depth"], o32["depth"]) < 3e-2
    assert all(torch.equal(a, b) for a, b in zip(od["taps"], oe["taps"]))  # toks= skipped the backbone
    assert torch.equal(od["pred_pose_enc_list"][-1], oe["pred_pose_enc_list"][-1])   # camera head stays fp32 under dpt_bf16
    assert 1e-4 < rel(od["depth"], oe["depth"]) < 3e-2                      # the bf16 DPT heads move the depth by bf16 noise
    # the flash contract against exact softmax attention
    q, k, v = (torch.randn(1, 2, 70, 64).to(torch.bfloat16).float() for _ in range(3))
    a, b = R.attention_bf16p(q, k, v, block_elems=70 * 64), torch.nn.functional.scaled_dot_product_attention(q, k, v)
    assert 1e-5 < rel(a, b) < 5e-3
