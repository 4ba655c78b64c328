"""Wan-2.1 VAE decoder on MI355X — the `pipe.vae.decode(latents, return_dict=False)[0]`
call of /root/reference/inference_t23d.py:114 (architecture: /root/reference/utils/wan_utils.py:752-901, 1078-1117).

MI355X-first restructuring (identical arithmetic, different schedule):
  * the reference decodes ONE latent frame per call and threads a two-frame cache through ~33 causal convs; here the
    whole clip is decoded in one pass — every WanCausalConv3d becomes one implicit-GEMM launch over all frames with
    causal (leading) zero padding, so the GEMM M dimension is T·H·W instead of H·W;
  * activations are channels-last bf16 [T,H,W,C]: the conv K dimension (taps x Cin) is contiguous per pixel and is
    gathered straight into LDS by the kernel's DMA — no im2col buffer, no NCHW<->NHWC traffic between layers;
  * RMS-norm + SiLU is one read-once/write-once kernel; nearest-exact 2x upsampling is folded into the following
    conv's gather (the 4x larger tensor is never written); residual adds ride in conv epilogues;
  * the "Rep" quirk of upsample3d is kept: the first frame bypasses time_conv and is invisible to it.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from .. import lib as L
from .. import ops

bf16, f32 = torch.bfloat16, torch.float32


@dataclass
class WanVAEConfig:
    base_dim: int = 96
    z_dim: int = 16
    dim_mult: List[int] = field(default_factory=lambda: [1, 2, 4, 4])
    num_res_blocks: int = 2
    temperal_downsample: List[bool] = field(default_factory=lambda: [False, True, True])

    def decoder_plan(self):
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


class _Res:
    def __init__(self, sd, p, dev):
        g = lambda k: sd[k].reshape(-1).to(device=dev, dtype=f32).contiguous()
        cw = lambda n: ops.ConvWeight(sd[p + n + ".weight"], sd[p + n + ".bias"], device=dev)
        self.g1, self.g2 = g(p + "norm1.gamma"), g(p + "norm2.gamma")
        self.c1, self.c2 = cw("conv1"), cw("conv2")
        self.sc = cw("conv_shortcut") if p + "conv_shortcut.weight" in sd else None

    def __call__(self, x, hook=None):
        """hook(point, tensor) (tests): the block's own tensors at its four bf16 rounding points - "n1" = SiLU(RMS-norm(x)), "y1" = conv1,
        "n2" = SiLU(RMS-norm(y1)), "out" = conv2 + shortcut - and "in" / "skip" (the input and the shortcut branch)"""
        hk = hook if hook is not None else (lambda *a: None)
        h = x if self.sc is None else ops.conv(x, self.sc)
        hk("in", x); hk("skip", h)
        n = ops.rownorm_act(x, self.g1, mode=1, act=L.ACT_SILU)
        hk("n1", n)
        y = ops.conv(n, self.c1, pad=(2, 1, 1))
        hk("y1", y)
        n = ops.rownorm_act(y, self.g2, mode=1, act=L.ACT_SILU)
        hk("n2", n)
        o = ops.conv(n, self.c2, pad=(2, 1, 1), residual=h)
        hk("out", o)
        return o


class _Attn:
    """WanAttentionBlock (wan_utils.py:428-475): one C-wide head per frame — QK^T GEMM -> row softmax -> PV GEMM."""

    def __init__(self, sd, a, dev):
        g = lambda k: sd[k].reshape(-1).to(device=dev, dtype=f32).contiguous()
        C = sd[a + "proj.weight"].shape[0]
        self.C = C
        self.g = g(a + "norm.gamma")
        wqkv = sd[a + "to_qkv.weight"].reshape(3 * C, C)
        bqkv = sd[a + "to_qkv.bias"]
        self.wqk = wqkv[: 2 * C].to(device=dev, dtype=bf16).contiguous()
        self.bqk = bqkv[: 2 * C].to(device=dev, dtype=f32).contiguous()
        self.wv = wqkv[2 * C:].to(device=dev, dtype=bf16).contiguous()
        self.bv = bqkv[2 * C:].to(device=dev, dtype=f32).contiguous()
        self.wproj = sd[a + "proj.weight"].reshape(C, C).to(device=dev, dtype=bf16).contiguous()
        self.bproj = sd[a + "proj.bias"].to(device=dev, dtype=f32).contiguous()

    def __call__(self, x):
        T, H, W, C = x.shape
        HW = H * W
        x2 = x.view(T * HW, C)
        n = ops.rownorm_act(x, self.g, mode=1).view(T * HW, C)
        qk = ops.gemm(n, self.wqk, self.bqk)
        if HW % 8:
            raise ValueError(f"mid-block attention needs H*W % 8 == 0 (got {H}x{W})")
        HWp = (HW + 63) // 64 * 64  # the PV GEMM's K dimension: zero probabilities x zero V^T columns in the pad
        o = torch.empty(T * HW, C, device=x.device, dtype=bf16)
        s = torch.empty(HW, HW, device=x.device, dtype=f32)
        pm = torch.zeros(HW, HWp, device=x.device, dtype=bf16)
        vt = torch.zeros(C, HWp, device=x.device, dtype=bf16)
        for t in range(T):
            sl = slice(t * HW, (t + 1) * HW)
            ops.gemm(self.wv, n[sl], self.bv, out=vt[:, :HW], bias_row=True)
            ops.gemm(qk[sl, :C], qk[sl, C:], out=s, out_f32=True)
            ops.softmax_rows(s, C ** -0.5, out=pm[:, :HW])
            ops.gemm(pm, vt, out=o[sl])
        y = ops.gemm(o, self.wproj, self.bproj, residual=x2)
        return y.view(T, H, W, C)


class WanVAEDecoder:
    """decode(z) with the AutoencoderKLWan.decode signature subset the reference uses."""

    def __init__(self, cfg: WanVAEConfig, state_dict: Dict[str, torch.Tensor], device="cuda"):
        L.load()
        self.cfg, self.device, self.dtype = cfg, torch.device(device), bf16
        sd, dev = state_dict, self.device
        cw = lambda n: ops.ConvWeight(sd[n + ".weight"], sd[n + ".bias"], device=dev)
        g = lambda k: sd[k].reshape(-1).to(device=dev, dtype=f32).contiguous()
        d = "decoder."
        self.pq = cw("post_quant_conv")
        self.conv_in = cw(d + "conv_in")
        self.mid0 = _Res(sd, d + "mid_block.resnets.0.", dev)
        self.mid1 = _Res(sd, d + "mid_block.resnets.1.", dev)
        self.attn = _Attn(sd, d + "mid_block.attentions.0.", dev)
        _, plan = cfg.decoder_plan()
        self.ups = []
        for i, (_, o_d, mode) in enumerate(plan):
            res = [_Res(sd, d + f"up_blocks.{i}.resnets.{j}.", dev) for j in range(cfg.num_res_blocks + 1)]
            rs = tc = None
            if mode is not None:
                rs = cw(d + f"up_blocks.{i}.upsamplers.0.resample.1")
                if mode == "upsample3d":
                    tw, tb = sd[d + f"up_blocks.{i}.upsamplers.0.time_conv.weight"], sd[d + f"up_blocks.{i}.upsamplers.0.time_conv.bias"]
                    hc = tw.shape[0] // 2     # output channels [0, C) = odd frames, [C, 2C) = even frames of the doubled clip (decode_cl)
                    tc = [ops.ConvWeight(tw[h * hc:(h + 1) * hc], tb[h * hc:(h + 1) * hc], device=dev) for h in range(2)]
            self.ups.append((res, mode, rs, tc, o_d))
        self.g_out = g(d + "norm_out.gamma")
        self.conv_out = cw(d + "conv_out")
        # the full AutoencoderKLWan checkpoint also carries the encoder (needed only by StitchVAE3D.forward)
        self.encoder = WanVAEEncoder(cfg, sd, device=dev) if "encoder.conv_in.weight" in sd else None

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """AutoencoderKLWan.encode surface: video [1,3,1+4n,H,W] in [-1,1] -> .latent_dist (sample / mode)."""
        if self.encoder is None:
            raise RuntimeError("this VAE was built from a decoder-only state dict: no `encoder.*` / `quant_conv.*` weights to encode with")
        return self.encoder.encode(x, return_dict=return_dict)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = False):
        """AutoencoderKLWan.decode surface: [1,16,T_lat,h,w] -> ([1,3,T,8h,8w] bf16 in [-1,1],)"""
        y = self.decode_cl(z)
        video = y[..., :3].permute(3, 0, 1, 2).unsqueeze(0).contiguous()
        return video if return_dict else (video,)

    @staticmethod
    def _time_upsample(x: torch.Tensor, mode: str, tc) -> torch.Tensor:
        """the temporal half of WanResample upsample3d (wan_utils.py:260-300): frames doubled by time_conv, the first frame bypassing it"""
        T = x.shape[0]
        if mode != "upsample3d" or T <= 1:
            return x
        # time_conv emits 2C channels per frame t >= 1: the first C are frame 2t-1, the last C frame 2t of the doubled clip.  Two
        # convolutions over the halves of the output channels write those frames in place (row scatter of the GEMM epilogue:
        # pixel m of frame t-1 -> frame 1 + 2(t-1) + half) instead of one convolution plus two strided interleave copies.
        HW = x.shape[1] * x.shape[2]
        y = torch.empty((1 + 2 * (T - 1), *x.shape[1:]), device=x.device, dtype=bf16)
        y[0] = x[0]
        for half, tch in enumerate(tc):
            ops.conv(x[1:], tch, pad=(2, 0, 0), out=y, out_rows=(HW, HW, (1 + half) * HW))
        return y

    @classmethod
    def _upsample(cls, x: torch.Tensor, mode: str, rs, tc) -> torch.Tensor:
        """WanResample upsample2d / upsample3d (wan_utils.py:202-330) on a channels-last clip [T,H,W,C]"""
        return ops.conv(cls._time_upsample(x, mode, tc), rs, pad=(0, 1, 1), ups2=True)

    # ------------------------------------------------------------------ one clip over several ranks: H-strips with exchanged halo rows
    @staticmethod
    def _with_halo(n: torch.Tensor, group) -> torch.Tensor:
        """strip [T,h,W,C] -> [T,h+2,W,C]: the bottom row of the strip above and the top row of the strip below (zeros at the image
        border = the convolution's zero padding), exchanged in ONE small all-gather of every rank's two boundary rows"""
        P, r = group.world, group.rank
        T, h, W, C = n.shape
        edge = torch.stack([n[:, 0], n[:, -1]], 0).contiguous()
        gb = torch.empty(P, edge.numel(), device=n.device, dtype=n.dtype)
        group.all_gather(gb, edge).wait()
        zero = torch.zeros(T, 1, W, C, device=n.device, dtype=n.dtype)
        top = gb[r - 1].view(2, T, 1, W, C)[1] if r > 0 else zero
        bot = gb[r + 1].view(2, T, 1, W, C)[0] if r < P - 1 else zero
        return torch.cat([top, n, bot], 1)

    def _res_strip(self, rb: "_Res", x: torch.Tensor, group) -> torch.Tensor:
        T, h, W, _ = x.shape
        P = float(group.world)
        h0 = x if rb.sc is None else ops.conv(x, rb.sc)
        n = ops.rownorm_act(x, rb.g1, mode=1, act=L.ACT_SILU)
        y = ops.conv(self._with_halo(n, group), rb.c1, pad=(2, 0, 1), out_size=(T, h, W), form_scale=P)
        n = ops.rownorm_act(y, rb.g2, mode=1, act=L.ACT_SILU)
        return ops.conv(self._with_halo(n, group), rb.c2, pad=(2, 0, 1), out_size=(T, h, W), residual=h0, form_scale=P)

    @torch.no_grad()
    def decode_cl_sharded(self, z: torch.Tensor, group) -> torch.Tensor:
        """`decode_cl` of ONE clip over the `group.world` ranks of a scene-parallel run: the latent-resolution part (conv_in, the
        mid block with its per-frame attention over all positions: 3 % of the decoder's FLOPs) runs replicated, then every rank owns an
        H-STRIP of the image through the four up blocks.  A 3x3 convolution needs one row of its neighbours' strips: before each one the ranks
        all-gather their two boundary rows (<= 1.3 MB per rank, 30 exchanges per clip) and convolve the haloed strip VALID in H - the same
        kernels, tap order and k-order as the unsharded decode, so the strips are bit-identical to the rows of `decode_cl`
        (tests/test_vae_gpu.py::test_strip_sharded_decode_is_bit_identical).  The temporal dimension is not split (causal convolutions look
        back over the whole clip).  Every rank returns the full clip [T, 8h, 8w, 8]."""
        P, r = group.world, group.rank
        if z.dim() != 5 or z.shape[0] != 1:
            raise ValueError("expected z [1, z_dim, T, h, w]")
        x = z[0].permute(1, 2, 3, 0).to(device=self.device, dtype=bf16).contiguous()
        x = ops.conv(x, self.pq)
        x = ops.conv(x, self.conv_in, pad=(2, 1, 1))
        x = self.mid1(self.attn(self.mid0(x)))
        H0 = x.shape[1]
        if P == 1 or H0 % P:
            raise ValueError(f"{H0} latent rows do not split into {P} strips")
        rows = H0 // P
        x = x[:, r * rows:(r + 1) * rows].contiguous()
        for res, mode, rs, tc, C in self.ups:
            for rb in res:
                x = self._res_strip(rb, x, group)
            if mode is None:
                continue
            x = self._time_upsample(x, mode, tc)      # time_conv: kernel (3,1,1), no spatial extent - local to the strip
            T, h, W, _ = x.shape
            # nearest-exact 2x upsample fused into the conv's gather: the haloed strip is stored at the input resolution, output row j of the
            # 2h-row strip reads upsampled rows j + 1 + dh of the 2 (h + 2)-row upsampled haloed strip (pad_H = -1)
            x = ops.conv(self._with_halo(x, group), rs, pad=(0, -1, 1), ups2=True, out_size=(T, 2 * h, 2 * W), form_scale=float(P))
        n = ops.rownorm_act(x, self.g_out, mode=1, act=L.ACT_SILU)
        T, h, W, _ = x.shape
        y = ops.conv(self._with_halo(n, group), self.conv_out, pad=(2, 0, 1), out_size=(T, h, W), form_scale=float(P)).clamp_(-1.0, 1.0)
        gb = torch.empty(P, y.numel(), device=y.device, dtype=y.dtype)
        group.all_gather(gb, y.contiguous()).wait()
        return gb.view(P, T, h, W, y.shape[-1]).permute(1, 0, 2, 3, 4).reshape(T, P * h, W, y.shape[-1]).contiguous()

    @torch.no_grad()
    def decode_cl(self, z: torch.Tensor, hook=None) -> torch.Tensor:
        """Same decode, result left channels-last [T, 8h, 8w, 8] bf16 (RGB in channels 0-2, clamped to [-1,1]) — the layout
        the resize kernel and the reconstruction heads consume, so the decoded clip never takes an NCHW round trip.
        hook(name, point, tensor) (tests): every up-block residual block's tensors (`_Res.__call__`), name = "up_blocks.i.resnets.j" - the
        production decode's own stream, for per-layer teacher-forced comparisons."""
        if z.dim() != 5 or z.shape[0] != 1:
            raise ValueError("expected z [1, z_dim, T, h, w]")
        x = z[0].permute(1, 2, 3, 0).to(device=self.device, dtype=bf16).contiguous()  # [T,h,w,16]
        x = ops.conv(x, self.pq)
        x = ops.conv(x, self.conv_in, pad=(2, 1, 1))
        x = self.mid0(x)
        x = self.attn(x)
        x = self.mid1(x)
        for i, (res, mode, rs, tc, C) in enumerate(self.ups):
            for j, r in enumerate(res):
                x = r(x, None if hook is None else (lambda pt, t, nm=f"up_blocks.{i}.resnets.{j}": hook(nm, pt, t)))
            if mode is None:
                continue
            x = self._upsample(x, mode, rs, tc)
        n = ops.rownorm_act(x, self.g_out, mode=1, act=L.ACT_SILU)
        y = ops.conv(n, self.conv_out, pad=(2, 1, 1))  # [T,H,W,8] (3 real channels, 5 zero)
        return y.clamp_(-1.0, 1.0)


class LatentDist:
    """The slice of diffusers' DiagonalGaussianDistribution the reference touches (`.sample()`, `.mode()`)."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, lv = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(lv, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        gdev = generator.device if generator is not None else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=gdev, dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class EncoderOutputKL:
    def __init__(self, latent_dist: LatentDist):
        self.latent_dist = latent_dist


class WanVAEEncoder:
    """`AutoencoderKLWan.encode` for the image-conditioned entry `StitchVAE3D.forward` (SURVEY.md §8f rank 4;
    /root/reference/utils/wan_utils.py:534-662, 1021-1076).  Same restructuring as the decoder: the reference encodes chunks of
    1,4,4,... frames through per-conv caches; here the whole clip goes through each layer once — causal convs see zero frames in
    front, `downsample3d` keeps its first-chunk quirk (frame 0 bypasses the stride-2 time_conv: out[0] = y[0],
    out[k] = conv(y[2k-2], y[2k-1], y[2k]))."""

    def __init__(self, cfg: WanVAEConfig, state_dict: Dict[str, torch.Tensor], device="cuda"):
        L.load()
        self.cfg, self.device = cfg, torch.device(device)
        sd, dev = state_dict, self.device
        cw = lambda n: ops.ConvWeight(sd[n + ".weight"], sd[n + ".bias"], device=dev)
        e = "encoder."
        self.conv_in = cw(e + "conv_in")
        dims = [cfg.base_dim * u for u in [1] + cfg.dim_mult]
        self.blocks = []
        idx = 0
        for i in range(len(cfg.dim_mult)):
            for _ in range(cfg.num_res_blocks):
                self.blocks.append(("res", _Res(sd, e + f"down_blocks.{idx}.", dev)))
                idx += 1
            if i != len(cfg.dim_mult) - 1:
                p = e + f"down_blocks.{idx}."
                tc = cw(p + "time_conv") if cfg.temperal_downsample[i] else None
                self.blocks.append(("down", (cw(p + "resample.1"), tc)))
                idx += 1
        self.mid0 = _Res(sd, e + "mid_block.resnets.0.", dev)
        self.attn = _Attn(sd, e + "mid_block.attentions.0.", dev)
        self.mid1 = _Res(sd, e + "mid_block.resnets.1.", dev)
        self.g_out = sd[e + "norm_out.gamma"].reshape(-1).to(device=dev, dtype=f32).contiguous()
        self.conv_out = cw(e + "conv_out")
        self.quant = cw("quant_conv")
        self.z2 = sd["quant_conv.weight"].shape[0]
        self.in_pad = self.conv_in.CinP

    @torch.no_grad()
    def encode_params(self, x: torch.Tensor) -> torch.Tensor:
        """video [1,3,1+4n,H,W] in [-1,1] -> posterior parameters [1,2*z_dim,1+n,H/8,W/8] fp32 (mean | logvar)."""
        if x.dim() != 5 or x.shape[0] != 1 or x.shape[1] != 3:
            raise ValueError("expected x [1, 3, T, H, W]")
        T, H, W = x.shape[2:]
        h = torch.zeros(T, H, W, self.in_pad, device=self.device, dtype=bf16)
        h[..., :3] = x[0].permute(1, 2, 3, 0).to(device=self.device, dtype=bf16)
        h = ops.conv(h, self.conv_in, pad=(2, 1, 1))
        for kind, blk in self.blocks:
            if kind == "res":
                h = blk(h)
                continue
            rs, tc = blk
            t, hh, ww = h.shape[:3]
            h = ops.conv(h, rs, stride=(1, 2, 2), pad=(0, 0, 0), out_size=(t, hh // 2, ww // 2))  # ZeroPad2d((0,1,0,1)) + stride 2
            if tc is not None and t > 1:
                rest = ops.conv(h, tc, stride=(2, 1, 1), pad=(0, 0, 0))
                h = torch.cat([h[:1], rest], 0)
        h = self.mid1(self.attn(self.mid0(h)))
        n = ops.rownorm_act(h, self.g_out, mode=1, act=L.ACT_SILU)
        h = ops.conv(n, self.conv_out, pad=(2, 1, 1))
        h = ops.conv(h, self.quant, out_f32=True)
        return h[..., : self.z2].permute(3, 0, 1, 2).unsqueeze(0).contiguous()

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        dist = LatentDist(self.encode_params(x))
        return EncoderOutputKL(dist) if return_dict else (dist,)
