"""`load_stitching_model(args) -> StitchVAE3D` — /root/reference/evaluation/novel_view_synthesis_bench/nvs_eval.py:21-63.

Reads the same `args` fields (feedforward_model, video_model, stitching_layer_location, stitching_layer_config, resolution,
initialization_weight_path, lora_config, checkpoint_path) and the same checkpoint dict
{lora, stitching_layer:{weight,bias}, mask_token, cls_token, register_tokens} (model_stitching_training.py:59-72).
LoRA is folded into the base weights at load (what `.eval()` does in the reference).  Offline sources: `--anysplat_weights`
local safetensors / `--checkpoint_path synthetic`."""
from __future__ import annotations

import torch

from ..recon.engine import ReconCfg
from ..recon.weights import random_recon_state_dict, round_aggregator_to_bf16
from ..utils.argument import parse_lora_mode
from .anysplat_stitched import AnySplatWeights
from .stitched_model import StitchVAE3D


def load_feedforward_model(args, device) -> AnySplatWeights:
    if args.feedforward_model != "anysplat":
        raise NotImplementedError(f"Feedforward model {args.feedforward_model} is not implemented.")
    cfg = ReconCfg()
    path = getattr(args, "anysplat_weights", None)
    if path:
        # a local .safetensors, or a hub-snapshot folder (`model.safetensors` + `config.json`, the layout AnySplat.from_pretrained /
        # PyTorchModelHubMixin reads, utils/utils_for_thirdparty.py:14-25).  The upstream config.json does not describe the VGGT
        # backbone (its width / depth are constants of the code); an optional "recon_cfg" object there overrides ReconCfg fields
        # (reduced-size checkpoints for tests).  The checkpoint is the FULL upstream model: all DINO blocks, the first k are dropped
        # by AnySplatStitched exactly as convert_model_to_stitched_model does (anysplat_stitched.py:158-165).
        import json
        import os
        from safetensors.torch import load_file
        if os.path.isdir(path):
            cj = os.path.join(path, "config.json")
            if os.path.exists(cj):
                cfg = ReconCfg(**json.load(open(cj)).get("recon_cfg", {}))
            path = os.path.join(path, "model.safetensors")
        sd = load_file(path)
        # the reference deletes the first k of however many DINO blocks the checkpoint holds (anysplat_stitched.py:158-165), for any k:
        # the block count the engine runs follows from the checkpoint and the stitching location, not from a default
        k = int(args.stitching_layer_location.split("_")[-1])
        pe = "encoder.aggregator.patch_embed.blocks."
        total = len({int(q[len(pe):].split(".")[0]) for q in sd if q.startswith(pe)})
        if total <= k:
            raise ValueError(f"--stitching_layer_location {args.stitching_layer_location} drops {k} DINO blocks, the checkpoint holds {total}")
        cfg.n_dino = total - k
        return AnySplatWeights(sd, cfg, n_total_dino_blocks=total)
    if getattr(args, "checkpoint_path", None) == "synthetic":
        return AnySplatWeights(round_aggregator_to_bf16(random_recon_state_dict(cfg, seed=2, device=str(device))), cfg)
    raise FileNotFoundError("AnySplat weights: pass --anysplat_weights <local .safetensors> (the HF hub id 'lhjiang/anysplat' the "
                            "reference downloads is unreachable offline) or --checkpoint_path synthetic")


def load_vae(args, device):
    from ..t23d import random_vae_decoder_state_dict
    from ..wan.vae import WanVAEConfig, WanVAEDecoder
    if args.video_model != "wan":
        raise NotImplementedError(f"Video diffusion model {args.video_model} is not implemented.")
    cfg = WanVAEConfig()
    model_dir = getattr(args, "model_id", None)
    import json
    import os
    if model_dir and os.path.isdir(os.path.join(model_dir, "vae")):
        from safetensors.torch import load_file
        cj = os.path.join(model_dir, "vae", "config.json")   # AutoencoderKLWan's registered config (utils/wan_utils.py:904-923)
        if os.path.exists(cj):
            c = json.load(open(cj))
            if c.get("attn_scales"):
                raise NotImplementedError("Wan VAE with attn_scales (attention inside the up blocks) is not implemented")
            cfg = WanVAEConfig(**{k: c[k] for k in ("base_dim", "z_dim", "dim_mult", "num_res_blocks", "temperal_downsample") if k in c})
        sd = load_file(os.path.join(model_dir, "vae", "diffusion_pytorch_model.safetensors"))
    elif getattr(args, "checkpoint_path", None) == "synthetic":
        sd = random_vae_decoder_state_dict(cfg, 1, str(device))
    else:
        raise FileNotFoundError(f"Wan VAE weights not found under {model_dir!r}/vae (no hub access offline)")
    return WanVAEDecoder(cfg, sd, device=device)


def load_stitching_model(args) -> StitchVAE3D:
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if device.type != "cuda":
        raise RuntimeError("the MI355X path has no CPU fallback")
    ff = load_feedforward_model(args, device)
    vae = load_vae(args, device)
    model = StitchVAE3D(diffusion_vae=vae, feedforward_model=ff, device=device, stitching_layer_location=args.stitching_layer_location,
                        stitching_layer_config=args.stitching_layer_config, resolution=args.resolution,
                        stitching_layer_init_path=args.initialization_weight_path)
    if args.checkpoint_path != "synthetic":
        sd = torch.load(args.checkpoint_path, weights_only=False, map_location="cpu")
        r, alpha = parse_lora_mode(args.lora_config)
        model.stitched_3d_model.load_lora(sd["lora"], alpha, r)
        model.stitching_layer.weight.data = sd["stitching_layer"]["weight"]
        model.stitching_layer.bias.data = sd["stitching_layer"]["bias"]
        pe = model.stitched_3d_model.encoder.aggregator.patch_embed
        pe.mask_token.data, pe.cls_token.data, pe.register_tokens.data = sd["mask_token"], sd["cls_token"], sd["register_tokens"]
    else:
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():
            model.stitching_layer.weight.copy_(torch.randn(model.stitching_layer.weight.shape, generator=g) * 0.02)
            model.stitching_layer.bias.copy_(torch.randn(model.stitching_layer.bias.shape, generator=g) * 0.02)
    return model
