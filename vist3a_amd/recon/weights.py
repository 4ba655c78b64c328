"""Weight sources for the stitched AnySplat encoder: seeded synthetic weights of the exact production shapes under the
reference's parameter names (no checkpoint is reachable offline), a loader for real `lhjiang/anysplat`
safetensors, and the LoRA merge that `.eval()` performs in the reference (utils/lora_util/layers.py:161-165)."""
from __future__ import annotations

import math
from typing import Dict

import torch

from .engine import ReconCfg

SD = Dict[str, torch.Tensor]


def random_recon_state_dict(cfg: ReconCfg, seed: int = 0, device="cpu", n_pos: int = 1370, n_reg: int = 4, scene_like: bool = False) -> SD:
    """Linear/Conv ~ N(0, 1/fan_in) (keeps activations O(1) through 70 blocks), norm gamma ~ 1, LayerScale as the
    reference initialises it (DINO 1.0, aggregator / camera head 0.01), tokens ~ N(0, 0.02).
    scene_like: the camera head's last layer gets the bias of a plausible camera (identity rotation, 57 degree field of view) instead of a
    near-zero pose encoding.  With fov ~ 0 the intrinsics are fx = (W / 2) / (tan(fov / 2) + 1e-3) ~ 2e5 px: every ray is parallel, the
    13 x 448^2 points of a scene fall inside a 2 mm column and collapse into ~35 k voxels of ~76 points - a degenerate input for the
    voxel fusion / adapter / rasteriser tail.  A trained checkpoint spreads them over 1-2 M voxels; so does this bias (bench.py)."""
    g = torch.Generator(device=device).manual_seed(seed)
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g, device=device) * std
    sd: SD = {}

    def lin(name, o, i):
        sd[name + ".weight"] = rn(o, i, std=1 / math.sqrt(i))
        sd[name + ".bias"] = rn(o, std=0.02)

    def block(p, C, heads, qk_norm, ls):
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"] = 1 + rn(C, std=0.05)
            sd[p + n + ".bias"] = rn(C, std=0.02)
        lin(p + "attn.qkv", 3 * C, C)
        lin(p + "attn.proj", C, C)
        if qk_norm:
            for n in ("q_norm", "k_norm"):
                sd[p + f"attn.{n}.weight"] = 1 + rn(C // heads, std=0.05)
                sd[p + f"attn.{n}.bias"] = rn(C // heads, std=0.02)
        lin(p + "mlp.fc1", 4 * C, C)
        lin(p + "mlp.fc2", C, 4 * C)
        sd[p + "ls1.gamma"] = torch.full((C,), ls, device=device)
        sd[p + "ls2.gamma"] = torch.full((C,), ls, device=device)

    C = cfg.C
    a = "encoder.aggregator."
    pe = a + "patch_embed."
    sd[pe + "cls_token"] = rn(1, 1, C, std=0.02)
    sd[pe + "register_tokens"] = rn(1, n_reg, C, std=0.02)
    sd[pe + "mask_token"] = torch.zeros(1, C, device=device)
    sd[pe + "pos_embed"] = rn(1, n_pos, C, std=0.02)
    for i in range(cfg.n_dino):
        block(pe + f"blocks.{i}.", C, cfg.heads, False, 1.0)
    sd[pe + "norm.weight"] = 1 + rn(C, std=0.05)
    sd[pe + "norm.bias"] = rn(C, std=0.02)
    sd[a + "camera_token"] = rn(1, 2, 1, C, std=0.02)
    sd[a + "register_token"] = rn(1, 2, n_reg, C, std=0.02)
    for i in range(cfg.depth):
        block(a + f"frame_blocks.{i}.", C, cfg.heads, True, 0.01)
        block(a + f"global_blocks.{i}.", C, cfg.heads, True, 0.01)
    c, C2 = "encoder.camera_head.", 2 * C
    for j in range(cfg.cam_trunk):
        block(c + f"trunk.{j}.", C2, cfg.cam_heads, False, 0.01)
    for n in ("token_norm", "trunk_norm"):
        sd[c + n + ".weight"] = 1 + rn(C2, std=0.05)
        sd[c + n + ".bias"] = rn(C2, std=0.02)
    sd[c + "empty_pose_tokens"] = torch.zeros(1, 1, 9, device=device)
    lin(c + "embed_pose", C2, 9)
    lin(c + "poseLN_modulation.1", 3 * C2, C2)
    lin(c + "pose_branch.fc1", C2 // 2, C2)
    lin(c + "pose_branch.fc2", 9, C2 // 2)
    sd[c + "pose_branch.fc2.weight"] *= 0.1
    if scene_like:   # pose encoding = (T, quaternion xyzw, fov_h, fov_w): vggt/utils/pose_enc.py:65-130
        # (the head refines the encoding in four additive iterations, camera_head.py:118-160: a quarter of the target per iteration)
        sd[c + "pose_branch.fc2.bias"] = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.25, 0.25, 0.25], device=device)

    def conv(name, o, i, k, bias=True, tr=False):
        shape = (i, o, k, k) if tr else (o, i, k, k)
        sd[name + ".weight"] = rn(*shape, std=1 / math.sqrt(i * k * k))
        if bias:
            sd[name + ".bias"] = rn(o, std=0.02)

    def dpt(p, features, out_dim, gs):
        sd[p + "norm.weight"] = 1 + rn(C2, std=0.05)
        sd[p + "norm.bias"] = rn(C2, std=0.02)
        for i, oc in enumerate(cfg.oc):
            conv(p + f"projects.{i}", oc, C2, 1)
        conv(p + "resize_layers.0", cfg.oc[0], cfg.oc[0], 4, tr=True)
        conv(p + "resize_layers.1", cfg.oc[1], cfg.oc[1], 2, tr=True)
        conv(p + "resize_layers.3", cfg.oc[3], cfg.oc[3], 3)
        s = p + "scratch."
        for i, oc in enumerate(cfg.oc):
            conv(s + f"layer{i + 1}_rn", features, oc, 3, bias=False)
        for r in (1, 2, 3, 4):
            q = s + f"refinenet{r}."
            conv(q + "out_conv", features, features, 1)
            for u in (("resConfUnit1.", "resConfUnit2.") if r != 4 else ("resConfUnit2.",)):
                conv(q + u + "conv1", features, features, 3)
                conv(q + u + "conv2", features, features, 3)
        conv(s + "output_conv1", features // 2, features, 3)
        if gs:
            hf2 = 128 if out_dim > 50 else 32
            conv(p + "input_merger.0", hf2, 3, 7)
            conv(s + "output_conv2.0", hf2, 128, 3)
            conv(s + "output_conv2.2", out_dim, hf2, 1)
        else:
            conv(s + "output_conv2.0", 32, features // 2, 3)
            conv(s + "output_conv2.2", out_dim, 32, 1)
        sd[s + "output_conv2.2.weight"] *= 0.3

    dpt("encoder.depth_head.", cfg.features, 2, False)
    dpt("encoder.gaussian_param_head.", 256, 1 + 7 + 3 * (cfg.sh_degree + 1) ** 2 + 1, True)
    return sd


def round_aggregator_to_bf16(sd: SD) -> SD:
    """EncoderAnySplat stores the whole aggregator in bf16 (anysplat.py:144): make the values bf16-representable."""
    for k in list(sd):
        if k.startswith("encoder.aggregator."):
            sd[k] = sd[k].to(torch.bfloat16).to(sd[k].dtype)
    return sd


# Linear / Conv2d modules of the reference's stitched_3d_model that add_lora wraps but the inference forward never runs
# (anysplat.py:140-223: distillation copies, the point head of pred_head_type="point"; the rasteriser has no parameters).
UNUSED_PREFIXES = ("encoder.distill_aggregator.", "encoder.distill_camera_head.", "encoder.distill_depth_head.",
                   "encoder.point_head.", "decoder.")


def merge_lora(sd: SD, lora_sd: SD, alpha: float, r: int) -> int:
    """Apply the `lora` entry of a stitched checkpoint (/root/reference/model_stitching_training.py:59-72 saves
    `lora_state_dict(model, bias="lora_only")`, utils/lora_util/utils.py:35-54) to the base weights `sd`, the way the reference's
    evaluation loader does with add_lora + load_state_dict(strict=False) + .eval() (nvs_eval.py:37-50):
      * `<layer>.lora_A` [r,in] / `.lora_B` [out,r] (Conv2d: [r*k, in*k] / [out*k, r*k], view-ed to the kernel shape):
        W += (B @ A).view_as(W) * alpha / r  - what Linear.train(False) / ConvLoRA.train(False) do (layers.py:149-165,338-355);
      * every other key is a trained parameter of a LoRA-wrapped layer (its `bias`, also reachable as `<layer>.conv.bias` for a
        wrapped Conv2d) and REPLACES the base value.
    A key that names nothing in `sd` raises: a silently dropped bias (or a checkpoint whose keys carry a stray prefix such as
    `module.` / `stitched_3d_model.`) gives wrong reconstructions with no other symptom.  The only keys skipped (with a warning) are
    those under UNUSED_PREFIXES: modules the reference's add_lora wraps (it wraps every Linear / Conv2d of stitched_3d_model,
    utils/lora_util/utils.py:148-153) but this inference path never executes.  add_lora runs AFTER convert_model_to_stitched_model
    (nvs_eval.py:26-45), so a checkpoint never carries adapters for the dropped DINO blocks - those are not on the list.
    A checkpoint with adapter matrices that merges none of them raises too.
    Returns the number of merged matrices.  Pinned by tests/golden/lora_tiny.safetensors (made by the reference's own code)."""
    n, unknown, absent = 0, [], []
    layers = {k.rsplit(".", 1)[0] for k in sd}

    def layer_exists(key: str) -> bool:   # "<layer>.lora_A" / "<layer>.bias" / "<layer>.conv.bias": is <layer> part of this engine at all?
        base = key.rsplit(".", 1)[0]
        return base in layers or (base.endswith(".conv") and base[:-5] in layers)

    for key, v in lora_sd.items():
        if not layer_exists(key):
            if key.startswith(UNUSED_PREFIXES):
                absent.append(key)
            else:
                unknown.append(key)
            continue
        if key.endswith("lora_B"):
            if key[: -len("lora_B")] + "lora_A" not in lora_sd:
                unknown.append(key)
            continue
        if key.endswith("lora_A"):
            base = key[: -len("lora_A")]
            B = lora_sd.get(base + "lora_B")
            w = base + "weight"
            if B is None or w not in sd:
                unknown.append(key)
                continue
            sd[w] = (sd[w].float() + (B.float() @ v.float()).view(sd[w].shape) * (alpha / r)).to(sd[w].dtype)
            n += 1
            continue
        tgt = key if key in sd else key.replace(".conv.", ".")
        if tgt not in sd or sd[tgt].shape != v.shape:
            unknown.append(key)
            continue
        sd[tgt] = v.to(sd[tgt].dtype)
    if absent:
        import warnings
        warnings.warn(f"merge_lora: skipped {len(absent)} adapter tensors of layers that are not part of this engine "
                      f"(e.g. {absent[0]})", stacklevel=2)
    if unknown:
        raise KeyError(f"LoRA checkpoint keys without a target in the base weights: {unknown[:8]}{' ...' if len(unknown) > 8 else ''}")
    if n == 0 and any(k.endswith("lora_A") for k in lora_sd):
        raise KeyError("LoRA checkpoint holds adapter matrices but none of them names a layer of this model")
    return n
