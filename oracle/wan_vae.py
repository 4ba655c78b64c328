"""ORACLE (test infrastructure).  CPU fp32 restatement of the Wan-2.1 VAE *decoder*  and, for the
image-conditioned entry `StitchVAE3D.forward`, of the *encoder*.

Follows /root/reference/utils/wan_utils.py (the vendored twin of diffusers' AutoencoderKLWan that
`pipe.vae.decode` runs at /root/reference/inference_t23d.py:114):
    WanCausalConv3d :96-147   WanRMS_norm :150-184   WanResample :202-330   WanResidualBlock :333-425
    WanAttentionBlock :428-475   WanMidBlock :478-531   WanUpBlock :667-749   WanDecoder3d :752-901
    AutoencoderKLWan._decode :1078-1117
    WanEncoder3d :534-662   AutoencoderKLWan._encode :1021-1047 (+ diffusers DiagonalGaussianDistribution)
PARITY PINNED: tests/golden/vae_decode_tiny.safetensors / vae_encode_tiny.safetensors are produced by running the reference
module itself (tests/golden/make_golden.py, stub-imported in the build container) on weights from `make_weights` /
`make_encoder_weights` below.

`emulate_bf16=True` (decode / encode) restates the SAME arithmetic with the rounding points the reference has when it runs under
`torch.autocast("cuda", dtype=torch.bfloat16)` (/root/reference/inference_t23d.py:87-114; op lists of CUDA autocast): every
conv3d / conv2d takes bf16 inputs and bf16 weights and returns bf16 (fp32 accumulation); `WanRMS_norm` (F.normalize: on autocast's
fp32 list) and the SiLU behind it run in fp32 and are rounded when the next conv casts its input; residual sums of two bf16
tensors are bf16; scaled_dot_product_attention runs on bf16 q / k / v with a bf16 probability operand (flash kernel) and returns
bf16; the nearest-exact upsample is exact.  The fp32 default stays the parity-pinned form (goldens); the emulation is the "kernel
contract" the HIP decoder is held to at production size (tests/test_fullsize_gpu.py).  It cannot be pinned by executing the
reference (no CUDA device here; CPU autocast has different op lists) - the rounding points are derived by reading.

The reference decodes one latent frame per call and threads a cache of the last two input frames through every
causal conv.  That is arithmetically a causal convolution over the whole frame sequence (two zero frames in
front), with ONE quirk that this restatement keeps: in `upsample3d` the first latent frame skips `time_conv`
("Rep" sentinel, :256-267) and the time_conv of the remaining frames never sees it (zero context instead, :283-297).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List

import torch
import torch.nn.functional as F


@dataclass
class WanVAEConfig:
    base_dim: int = 96
    z_dim: int = 16
    dim_mult: List[int] = field(default_factory=lambda: [1, 2, 4, 4])
    num_res_blocks: int = 2
    temperal_downsample: List[bool] = field(default_factory=lambda: [False, True, True])

    def decoder_plan(self):
        """[(in_dim, out_dim, upsample_mode)] per up block, as WanDecoder3d.__init__ (:783-812)."""
        dims = [self.base_dim * u for u in [self.dim_mult[-1]] + self.dim_mult[::-1]]
        tu = self.temperal_downsample[::-1]
        plan = []
        for i, (i_d, o_d) in enumerate(zip(dims[:-1], dims[1:])):
            if i > 0:
                i_d = i_d // 2
            mode = None
            if i != len(self.dim_mult) - 1:
                mode = "upsample3d" if tu[i] else "upsample2d"
            plan.append((i_d, o_d, mode))
        return dims[0], plan


_EMU = False   # set for the duration of decode(..., emulate_bf16=True) / encode(..., emulate_bf16=True)


def _r(x: torch.Tensor) -> torch.Tensor:
    """bf16 rounding point of the CUDA-autocast contract (identity in the fp32 default)"""
    return x.to(torch.bfloat16).float() if _EMU else x


def sdpa_bf16p(q, k, v):
    """softmax(q k^T / sqrt(d)) v with the probability operand rounded to bf16 and the row sum taken over the unrounded values -
    the contract of every flash kernel on bf16 inputs (the reference's CUDA SDPA included).  q, k, v [..., N, d] fp32 (bf16 values)."""
    s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    p = torch.exp(s - s.amax(-1, keepdim=True))
    l = p.sum(-1, keepdim=True)
    return (p.to(torch.bfloat16).float() @ v) / l


def causal_conv3d(x, w, b, pad):
    """x [B,C,T,H,W]; zero pad (W,W,H,H,2*pT,0) then valid conv (WanCausalConv3d.forward without cache)."""
    pt, ph, pw = pad
    return _r(F.conv3d(F.pad(_r(x), (pw, pw, ph, ph, 2 * pt, 0)), _r(w), _r(b)))


def conv2d(x, w, b, **kw):
    return _r(F.conv2d(_r(x), _r(w), _r(b), **kw))


def rms_norm(x, gamma, dim=1):
    return F.normalize(x, dim=dim) * (x.shape[dim] ** 0.5) * gamma


def res_block(sd, p, x):
    h = x
    if p + "conv_shortcut.weight" in sd:
        h = causal_conv3d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"], (0, 0, 0))
    x = F.silu(rms_norm(x, sd[p + "norm1.gamma"]))
    x = causal_conv3d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], (1, 1, 1))
    x = F.silu(rms_norm(x, sd[p + "norm2.gamma"]))
    x = causal_conv3d(x, sd[p + "conv2.weight"], sd[p + "conv2.bias"], (1, 1, 1))
    return _r(x + h)


def attn_block(sd, p, x):
    B, C, T, H, W = x.shape
    idn = x
    y = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    y = rms_norm(y, sd[p + "norm.gamma"])
    qkv = conv2d(y, sd[p + "to_qkv.weight"], sd[p + "to_qkv.bias"])
    qkv = qkv.reshape(B * T, 1, C * 3, -1).permute(0, 1, 3, 2).contiguous()
    q, k, v = qkv.chunk(3, dim=-1)
    y = _r(sdpa_bf16p(q, k, v)) if _EMU else F.scaled_dot_product_attention(q, k, v)
    y = y.squeeze(1).permute(0, 2, 1).reshape(B * T, C, H, W)
    y = conv2d(y, sd[p + "proj.weight"], sd[p + "proj.bias"])
    return _r(y.view(B, T, C, H, W).permute(0, 2, 1, 3, 4) + idn)


def resample(sd, p, x, mode):
    B, C, T, H, W = x.shape
    if mode == "upsample3d" and T > 1:
        rest = causal_conv3d(x[:, :, 1:], sd[p + "time_conv.weight"], sd[p + "time_conv.bias"], (1, 0, 0))  # [B,2C,T-1,H,W]
        rest = rest.reshape(B, 2, C, T - 1, H, W)
        rest = torch.stack((rest[:, 0], rest[:, 1]), 3).reshape(B, C, 2 * (T - 1), H, W)
        x = torch.cat([x[:, :, :1], rest], 2)
        T = x.shape[2]
    y = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    y = F.interpolate(y.float(), scale_factor=(2.0, 2.0), mode="nearest-exact")
    y = conv2d(y, sd[p + "resample.1.weight"], sd[p + "resample.1.bias"], padding=1)
    return y.view(B, T, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)


def decode(sd: Dict[str, torch.Tensor], cfg: WanVAEConfig, z: torch.Tensor, emulate_bf16: bool = False, trace: list = None) -> torch.Tensor:
    """AutoencoderKLWan._decode: z [B,16,T_lat,h,w] (de-normalised latents) -> video [B,3,1+4(T_lat-1),8h,8w] in [-1,1].
    emulate_bf16: the CUDA-autocast rounding points (module docstring).
    `trace` (a list): receives (name, input, output) of every residual block, the attention block and every upsampler - the inputs and
    expected outputs of per-layer teacher-forced comparisons."""
    global _EMU
    prev, _EMU = _EMU, bool(emulate_bf16)
    try:
        return _decode(sd, cfg, z, trace)
    finally:
        _EMU = prev


def _decode(sd, cfg, z, trace=None):
    sd = {k: v.float() for k, v in sd.items()}

    def rec(name, fn, x, *a):
        y = fn(sd, name, x, *a)
        if trace is not None:
            trace.append((name, x, y))
        return y

    x = causal_conv3d(z.float(), sd["post_quant_conv.weight"], sd["post_quant_conv.bias"], (0, 0, 0))
    d = "decoder."
    x = causal_conv3d(x, sd[d + "conv_in.weight"], sd[d + "conv_in.bias"], (1, 1, 1))
    x = rec(d + "mid_block.resnets.0.", res_block, x)
    x = rec(d + "mid_block.attentions.0.", attn_block, x)
    x = rec(d + "mid_block.resnets.1.", res_block, x)
    _, plan = cfg.decoder_plan()
    for i, (_, _, mode) in enumerate(plan):
        for j in range(cfg.num_res_blocks + 1):
            x = rec(d + f"up_blocks.{i}.resnets.{j}.", res_block, x)
        if mode is not None:
            x = rec(d + f"up_blocks.{i}.upsamplers.0.", resample, x, mode)
    x = F.silu(rms_norm(x, sd[d + "norm_out.gamma"]))
    x = causal_conv3d(x, sd[d + "conv_out.weight"], sd[d + "conv_out.bias"], (1, 1, 1))
    return torch.clamp(x, -1.0, 1.0)


def make_weights(cfg: WanVAEConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded decoder(+post_quant_conv) weights under the reference's state-dict names."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, o, i, k):
        fan = i * math.prod(k)
        sd[name + ".weight"] = torch.randn(o, i, *k, generator=g) / math.sqrt(fan)
        sd[name + ".bias"] = torch.randn(o, generator=g) * 0.05

    def gamma(name, c, nd):
        sd[name + ".gamma"] = 1 + 0.1 * torch.randn(c, *([1] * nd), generator=g)

    def res(p, i, o):
        gamma(p + "norm1", i, 3)
        conv(p + "conv1", o, i, (3, 3, 3))
        gamma(p + "norm2", o, 3)
        conv(p + "conv2", o, o, (3, 3, 3))
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
    sd[d + "conv_out.weight"] *= 0.25  # keep most of the output inside the final clamp(-1, 1)
    sd[d + "conv_out.bias"] *= 0.25
    return sd


# ----------------------------------------------------------------------------------------------- encoder
def encoder_plan(cfg: WanVAEConfig):
    """Flat `down_blocks` list of WanEncoder3d.__init__ (:571-590): [("res", in, out) | ("down", dim, mode)]."""
    dims = [cfg.base_dim * u for u in [1] + cfg.dim_mult]
    plan = []
    for i, (i_d, o_d) in enumerate(zip(dims[:-1], dims[1:])):
        cur = i_d
        for _ in range(cfg.num_res_blocks):
            plan.append(("res", cur, o_d))
            cur = o_d
        if i != len(cfg.dim_mult) - 1:
            plan.append(("down", o_d, "downsample3d" if cfg.temperal_downsample[i] else "downsample2d"))
    return dims, plan


def downsample(sd, p, x, mode):
    """WanResample downsample2d / downsample3d (:232-243, 303-329) in whole-clip form.  The reference feeds chunks of
    1, 4, 4, ... frames: the first chunk bypasses time_conv (its output is only cached), every later chunk runs the
    stride-2 (3,1,1) conv over [last cached frame, chunk] — i.e. out[0] = y[0], out[k] = conv(y[2k-2], y[2k-1], y[2k])."""
    B, C, T, H, W = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    y = conv2d(F.pad(y, (0, 1, 0, 1)), sd[p + "resample.1.weight"], sd[p + "resample.1.bias"], stride=2)
    y = y.view(B, T, C, y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)
    if mode == "downsample3d" and T > 1:
        rest = _r(F.conv3d(_r(y), _r(sd[p + "time_conv.weight"]), _r(sd[p + "time_conv.bias"]), stride=(2, 1, 1)))
        y = torch.cat([y[:, :, :1], rest], 2)
    return y


def encode(sd: Dict[str, torch.Tensor], cfg: WanVAEConfig, x: torch.Tensor, emulate_bf16: bool = False) -> torch.Tensor:
    """AutoencoderKLWan._encode: video [B,3,1+4n,H,W] in [-1,1] -> posterior parameters [B,2*z_dim,1+n,H/8,W/8]
    (mean | logvar).  Every causal conv over chunk+cache equals a causal conv over the whole clip (zero frames in front)."""
    global _EMU
    prev, _EMU = _EMU, bool(emulate_bf16)
    try:
        return _encode(sd, cfg, x)
    finally:
        _EMU = prev


def _encode(sd, cfg, x):
    sd = {k: v.float() for k, v in sd.items()}
    e = "encoder."
    x = causal_conv3d(x.float(), sd[e + "conv_in.weight"], sd[e + "conv_in.bias"], (1, 1, 1))
    _, plan = encoder_plan(cfg)
    for i, item in enumerate(plan):
        p = e + f"down_blocks.{i}."
        x = res_block(sd, p, x) if item[0] == "res" else downsample(sd, p, x, item[2])
    x = res_block(sd, e + "mid_block.resnets.0.", x)
    x = attn_block(sd, e + "mid_block.attentions.0.", x)
    x = res_block(sd, e + "mid_block.resnets.1.", x)
    x = F.silu(rms_norm(x, sd[e + "norm_out.gamma"]))
    x = causal_conv3d(x, sd[e + "conv_out.weight"], sd[e + "conv_out.bias"], (1, 1, 1))
    return causal_conv3d(x, sd["quant_conv.weight"], sd["quant_conv.bias"], (0, 0, 0))


def posterior_sample(params: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """diffusers DiagonalGaussianDistribution(params).sample(): mean + exp(0.5*clamp(logvar,-30,20)) * noise."""
    mean, logvar = torch.chunk(params, 2, dim=1)
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise


def make_encoder_weights(cfg: WanVAEConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded encoder(+quant_conv) weights under the reference's state-dict names."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, o, i, k):
        sd[name + ".weight"] = torch.randn(o, i, *k, generator=g) / math.sqrt(i * math.prod(k))
        sd[name + ".bias"] = torch.randn(o, generator=g) * 0.05

    def gamma(name, c, nd):
        sd[name + ".gamma"] = 1 + 0.1 * torch.randn(c, *([1] * nd), generator=g)

    def res(p, i, o):
        gamma(p + "norm1", i, 3)
        conv(p + "conv1", o, i, (3, 3, 3))
        gamma(p + "norm2", o, 3)
        conv(p + "conv2", o, o, (3, 3, 3))
        if i != o:
            conv(p + "conv_shortcut", o, i, (1, 1, 1))

    e = "encoder."
    dims, plan = encoder_plan(cfg)
    conv(e + "conv_in", dims[0], 3, (3, 3, 3))
    for i, item in enumerate(plan):
        p = e + f"down_blocks.{i}."
        if item[0] == "res":
            res(p, item[1], item[2])
        else:
            conv(p + "resample.1", item[1], item[1], (3, 3))
            if item[2] == "downsample3d":
                conv(p + "time_conv", item[1], item[1], (3, 1, 1))
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
