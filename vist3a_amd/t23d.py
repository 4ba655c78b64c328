"""Text -> 3D-Gaussian-splat scene: the per-prompt body of /root/reference/inference_t23d.py:85-137 as one object.

    latents  = WanPipeline(..., output_type="latent")          50-step CFG denoise (DiT + UniPC)          :94-103
    latents  = latents * std + mean                              de-normalise                              :105-113
    samples  = vae.decode(latents)                               13 x 512^2 RGB                            :114
    feedfwd  = trilinear(samples -> (T, 448, 448), align=False)                                            :118-123
    output   = stitched_decoder.forward_with_latent(latents, feedfwd, train=False)                         :131-137

Everything stays on the device and, between the VAE decoder and the reconstruction heads, in channels-last bf16."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops
from .models.anysplat_stitched import AnySplatWeights
from .models.stitched_model import StitchVAE3D
from .models.stitching_layer_builder import parse_conv_spec
from .models.types import EncoderOutput
from .recon.engine import ReconCfg
from .recon.weights import random_recon_state_dict, round_aggregator_to_bf16
from .wan.dit import WAN_1_3B, GraphedWanDiT, WanDiT, WanDiTConfig
from .wan.pipeline import WanT2VPipeline, denormalize_latents
from .wan.scheduler import UniPCMultistepScheduler
from .wan.vae import WanVAEConfig, WanVAEDecoder
from .wan.weights import random_dit_state_dict


@dataclass
class SceneTimes:
    denoise_ms: float = 0.0
    vae_ms: float = 0.0
    recon_ms: float = 0.0


class Text23DGS:
    def __init__(self, transformer: WanDiT, vae: WanVAEDecoder, stitched_decoder: StitchVAE3D, flow_shift: float = 5.0,
                 feedforward_resolution: int = 448, device="cuda", graph: Optional[bool] = None):
        """graph: replay the DiT step from a captured hipGraph (wan/dit.py GraphedWanDiT; V3A_GRAPH=1 or graph=True).  Off by default:
        on MI355X the eager loop is already GPU-bound — measured 33.49 ms eager vs 33.52 ms replayed per CFG step, launches run
        ~a full step ahead of the GPU — so the graph buys nothing at 1.3B/4096 tokens; it matters for small-token or sharded runs."""
        self.device = torch.device(device)
        self.transformer, self.vae, self.stitched_decoder = transformer, vae, stitched_decoder
        if graph is None:
            graph = os.environ.get("V3A_GRAPH", "0") == "1"
        step_fn = GraphedWanDiT(transformer) if graph else transformer
        self.pipe = WanT2VPipeline(step_fn, UniPCMultistepScheduler(flow_shift=flow_shift), vae=vae, device=device)
        self.ff_res = feedforward_resolution
        # Scene-parallel runs (a DenoisePlan over > 1 ranks) also shard the VAE decode (H-strips) and the reconstruction (views) - bit-identical
        # to the unsharded stages over 2 / 4 / 8 VIRTUAL ranks and in a 2-process gloo exchange test, but not yet run on real multi-GPU
        # hardware (no such box in the build loop).  V3A_SCENE_SHARD_STAGES=0 (or shard_stages = False) keeps the 8 % of a scene these two
        # stages are replicated on every rank, as a fallback switch for the first real multi-GPU run.
        self.shard_stages = os.environ.get("V3A_SCENE_SHARD_STAGES", "1") != "0"

    @classmethod
    def synthetic(cls, dit_cfg: WanDiTConfig = WAN_1_3B, seed: int = 0, device="cuda", flow_shift: float = 5.0,
                  stitch_spec: str = "conv3d_k5x3x3_o1024_s1x2x2_p2x1x1", stitch_location: str = "enc_blocks_2",
                  recon_cfg: Optional[ReconCfg] = None, vae_cfg: Optional[WanVAEConfig] = None, resolution: int = 512):
        """Seeded random weights of the exact production shapes (no checkpoint is reachable offline)."""
        dit = WanDiT(dit_cfg, random_dit_state_dict(dit_cfg, seed=seed, device=device), device=device)
        vcfg = vae_cfg or WanVAEConfig()
        vae = WanVAEDecoder(vcfg, random_vae_decoder_state_dict(vcfg, seed + 1, device), device=device)
        rcfg = recon_cfg or ReconCfg()
        rsd = round_aggregator_to_bf16(random_recon_state_dict(rcfg, seed=seed + 2, device=device, scene_like=True))
        dec = StitchVAE3D(vae, AnySplatWeights(rsd, rcfg), device, stitch_location, parse_conv_spec(stitch_spec), resolution)
        g = torch.Generator().manual_seed(seed + 3)
        with torch.no_grad():
            dec.stitching_layer.weight.copy_(torch.randn(dec.stitching_layer.weight.shape, generator=g) * 0.02)
            dec.stitching_layer.bias.copy_(torch.randn(dec.stitching_layer.bias.shape, generator=g) * 0.02)
        return cls(dit, vae, dec, flow_shift=flow_shift, device=device)

    @torch.no_grad()
    def generate(self, prompt_embeds: torch.Tensor, negative_prompt_embeds: torch.Tensor, *, latents: Optional[torch.Tensor] = None,
                 generator: Optional[torch.Generator] = None, num_frames: int = 13, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, height: int = 512, width: int = 512, timings: Optional[SceneTimes] = None,
                 stage_flops: Optional[dict] = None):
        """-> (EncoderOutput, de-normalised latents, channels-last decoded clip [T,512,512,8] in [-1,1]).
        stage_flops (a dict): receives {"denoise" | "vae" | "recon": {kind: algorithmic FLOPs}} of the matrix-pipe launches each stage
        issued (ops.FlopMeter; eager launches only) - bench.py's `stage_roofline`."""
        ev = (lambda: _event()) if timings is not None else (lambda: None)

        def meter(stage):   # close the previous stage's meter, open the next one
            if stage_flops is None:
                return
            if ops._meter is not None and getattr(ops._meter, "stage", None):
                stage_flops[ops._meter.stage] = dict(ops._meter.by_kind)
            ops.set_flop_meter(None)
            if stage is not None:
                m = ops.FlopMeter()
                m.stage = stage
                ops.set_flop_meter(m)
        meter("denoise")
        e0 = ev()
        lat = self.pipe(prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds, height=height, width=width,
                        num_frames=num_frames, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                        latents=latents, generator=generator, output_type="latent")["frames"]
        lat = denormalize_latents(lat)
        e1 = ev()
        meter("vae")
        wg = getattr(self.pipe.plan, "world", None) if self.pipe.plan is not None else None
        if self.shard_stages and wg is not None and wg.world > 1 and lat.shape[3] % wg.world == 0:
            clip_cl = self.vae.decode_cl_sharded(lat, wg)      # one clip over the ranks of the scene: H-strips with exchanged halo rows
        else:
            clip_cl = self.vae.decode_cl(lat)
        ff_cl = ops.bilinear_cl(clip_cl, (self.ff_res, self.ff_res), align_corners=False)
        e2 = ev()
        meter("recon")
        if not self.shard_stages and getattr(self.stitched_decoder, "recon_group", None) is not None:
            self.stitched_decoder.recon_group = None
        out = self.stitched_decoder.forward_with_latent(lat, None, train=False, image_cl=ff_cl)
        e3 = ev()
        meter(None)
        if timings is not None:
            torch.cuda.synchronize()
            timings.denoise_ms, timings.vae_ms, timings.recon_ms = e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3)
        return out, lat, clip_cl


def _event():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def random_vae_decoder_state_dict(cfg: WanVAEConfig, seed: int = 0, device="cpu"):
    """Seeded Wan-VAE decoder (+post_quant_conv) weights under the reference's names (utils/wan_utils.py:745-1000)."""
    import math
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}

    def conv(name, o, i, k):
        sd[name + ".weight"] = torch.randn(o, i, *k, generator=g, device=device) / math.sqrt(i * math.prod(k))
        sd[name + ".bias"] = torch.randn(o, generator=g, device=device) * 0.02

    def gamma(name, c, nd):
        sd[name + ".gamma"] = 1 + 0.05 * torch.randn(c, *([1] * nd), generator=g, device=device)

    def res(p, i, o):
        gamma(p + "norm1", i, 3); conv(p + "conv1", o, i, (3, 3, 3)); gamma(p + "norm2", o, 3); conv(p + "conv2", o, o, (3, 3, 3))
        if i != o:
            conv(p + "conv_shortcut", o, i, (1, 1, 1))

    conv("post_quant_conv", cfg.z_dim, cfg.z_dim, (1, 1, 1))
    d0, plan = cfg.decoder_plan()
    d = "decoder."
    conv(d + "conv_in", d0, cfg.z_dim, (3, 3, 3))
    res(d + "mid_block.resnets.0.", d0, d0)
    gamma(d + "mid_block.attentions.0.norm", d0, 2)
    conv(d + "mid_block.attentions.0.to_qkv", 3 * d0, d0, (1, 1))
    conv(d + "mid_block.attentions.0.proj", d0, d0, (1, 1))
    res(d + "mid_block.resnets.1.", d0, d0)
    for i, (i_d, o_d, mode) in enumerate(plan):
        cur = i_d
        for j in range(cfg.num_res_blocks + 1):
            res(d + f"up_blocks.{i}.resnets.{j}.", cur, o_d)
            cur = o_d
        if mode is not None:
            conv(d + f"up_blocks.{i}.upsamplers.0.resample.1", o_d // 2, o_d, (3, 3))
            if mode == "upsample3d":
                conv(d + f"up_blocks.{i}.upsamplers.0.time_conv", 2 * o_d, o_d, (3, 1, 1))
    gamma(d + "norm_out", plan[-1][1], 3)
    conv(d + "conv_out", 3, plan[-1][1], (3, 3, 3))
    sd[d + "conv_out.weight"] *= 0.25
    return sd


def random_vae_encoder_state_dict(cfg: WanVAEConfig, seed: int = 0, device="cpu"):
    """Seeded Wan-VAE encoder (+quant_conv) weights under the reference's names (utils/wan_utils.py:534-662, 990)."""
    import math
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}

    def conv(name, o, i, k):
        sd[name + ".weight"] = torch.randn(o, i, *k, generator=g, device=device) / math.sqrt(i * math.prod(k))
        sd[name + ".bias"] = torch.randn(o, generator=g, device=device) * 0.02

    def gamma(name, c, nd):
        sd[name + ".gamma"] = 1 + 0.05 * torch.randn(c, *([1] * nd), generator=g, device=device)

    def res(p, i, o):
        gamma(p + "norm1", i, 3); conv(p + "conv1", o, i, (3, 3, 3)); gamma(p + "norm2", o, 3); conv(p + "conv2", o, o, (3, 3, 3))
        if i != o:
            conv(p + "conv_shortcut", o, i, (1, 1, 1))

    e = "encoder."
    dims = [cfg.base_dim * u for u in [1] + cfg.dim_mult]
    conv(e + "conv_in", dims[0], 3, (3, 3, 3))
    idx = 0
    for i, (i_d, o_d) in enumerate(zip(dims[:-1], dims[1:])):
        cur = i_d
        for _ in range(cfg.num_res_blocks):
            res(e + f"down_blocks.{idx}.", cur, o_d)
            cur = o_d
            idx += 1
        if i != len(cfg.dim_mult) - 1:
            p = e + f"down_blocks.{idx}."
            conv(p + "resample.1", o_d, o_d, (3, 3))
            if cfg.temperal_downsample[i]:
                conv(p + "time_conv", o_d, o_d, (3, 1, 1))
            idx += 1
    d = dims[-1]
    res(e + "mid_block.resnets.0.", d, d)
    gamma(e + "mid_block.attentions.0.norm", d, 2)
    conv(e + "mid_block.attentions.0.to_qkv", 3 * d, d, (1, 1))
    conv(e + "mid_block.attentions.0.proj", d, d, (1, 1))
    res(e + "mid_block.resnets.1.", d, d)
    gamma(e + "norm_out", d, 3)
    conv(e + "conv_out", 2 * cfg.z_dim, d, (3, 3, 3))
    conv("quant_conv", 2 * cfg.z_dim, 2 * cfg.z_dim, (1, 1, 1))
    return sd


def synthetic_text_embeddings(device="cuda", seed: int = 12413, L_pos: int = 64, L_neg: int = 80, max_len: int = 512, dim: int = 4096):
    """randn*0.1 rows for the first L tokens, zero rows after (the pipeline zero-pads to 512, no mask)."""
    g = torch.Generator().manual_seed(seed)
    pe = torch.zeros(1, max_len, dim)
    ne = torch.zeros(1, max_len, dim)
    pe[:, :L_pos] = torch.randn(1, L_pos, dim, generator=g) * 0.1
    ne[:, :L_neg] = torch.randn(1, L_neg, dim, generator=g) * 0.1
    return pe.to(device), ne.to(device)
