"""Stitched Conv3d -> AnySplat reconstruction forward on MI355X.

Host orchestration over the C-ABI kernels; mirrors what /root/reference/models/anysplat_stitched.py:167-525 computes,
re-laid-out for the GPU instead of for nn.Module composition:

  * tokens live in ONE padded buffer [S * Pp, C] (Pp = P rounded up to 8 rows) for the whole backbone: frame attention
    is the batched (B=S) case of the flash kernel, global attention the B=1 case with a per-frame key mask — no
    view/reshape traffic between the 48 alternating blocks, and V is always produced transposed by the projection GEMM;
  * the stitching Conv3d writes its output straight into that buffer (channels-last conv output IS the "(b v) (h w) c"
    token layout) with the bicubic positional embedding added in the conv epilogue;
  * DINO residual stream bf16, aggregator residual stream fp32, LayerScale + residual inside GEMM epilogues — the
    rounding points of the reference under CUDA autocast;
  * DPT heads run channels-last: 3x3 / 1x1 / 7x7 convs as implicit GEMMs with ReLU, residual and second-residual
    epilogues (a RefineNet fusion block is 5 launches), ConvTranspose(k=s) as a 1x1 GEMM + pixel shuffle, the 1x1
    out_conv hoisted in front of the bilinear upsample (they commute; 4x fewer FLOPs);
  * camera head in fp32 (13 tokens: weight-bandwidth bound skinny kernels) because every 3-D point inherits its error;
  * depth activation + unprojection, voxel sort/unique/fuse and the Gaussian adapter are single-pass kernels.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .. import lib as L
from .. import ops

bf16, f32 = torch.bfloat16, torch.float32
SD = Dict[str, torch.Tensor]


class ReconCfg:
    def __init__(self, C=1024, heads=16, n_dino=22, depth=24, cam_heads=16, cam_trunk=4, features=256,
                 oc=(256, 512, 1024, 1024), patch=14, voxel_size=0.002, voxelize=True, sh_degree=4,
                 taps=(4, 11, 17, 23), opacity_exponent=1.0, render_conf=False, conf_threshold=0.1, dpt_precision="f32"):
        self.C, self.heads, self.n_dino, self.depth = C, heads, n_dino, depth
        self.cam_heads, self.cam_trunk, self.features, self.oc = cam_heads, cam_trunk, features, list(oc)
        self.patch, self.voxel_size, self.voxelize, self.sh_degree = patch, voxel_size, voxelize, sh_degree
        self.taps, self.opacity_exponent = tuple(taps), opacity_exponent
        self.render_conf, self.conf_threshold = render_conf, conf_threshold   # voxelize=False branch only (anysplat_stitched.py:381-387)
        # "f32": the DPT heads in fp32-equivalent split-bf16 arithmetic - the reference's precision (autocast off, anysplat_stitched.py:335);
        # "bf16": plain bf16 MFMA convolutions with bf16 activations (faster; a documented deviation, opt-in only)
        if dpt_precision not in ("f32", "bf16"):
            raise ValueError("dpt_precision must be 'f32' or 'bf16'")
        self.dpt_precision = dpt_precision


def uv_pos_embed(C: int, ph: int, pw: int, W: int, H: int, ratio: float = 0.1) -> torch.Tensor:
    """Constant sin/cos UV embedding added by the DPT heads ([ph*pw, C] f32): create_uv_grid + position_grid_to_embed
    (vggt/heads/utils.py:11-108, dpt_head.py:267-277), omega_0 = 100, float64 angles."""
    ar = W / H
    diag = (ar ** 2 + 1.0) ** 0.5
    sx, sy = ar / diag, 1.0 / diag
    xs = torch.linspace(-sx * (pw - 1) / pw, sx * (pw - 1) / pw, steps=pw, dtype=f32)
    ys = torch.linspace(-sy * (ph - 1) / ph, sy * (ph - 1) / ph, steps=ph, dtype=f32)
    uu, vv = torch.meshgrid(xs, ys, indexing="xy")

    def sincos(d, pos):
        om = 1.0 / 100 ** (torch.arange(d // 2, dtype=torch.double) / (d / 2.0))
        out = pos.reshape(-1).double()[:, None] * om[None]
        return torch.cat([out.sin(), out.cos()], 1).float()

    emb = torch.cat([sincos(C // 2, uu), sincos(C // 2, vv)], -1)
    return (emb * ratio).contiguous()


def rope2d_table(max_pos: int, freq: float = 100.0) -> torch.Tensor:
    """[max_pos, 16, 2] (cos, sin) for head_dim 64 (32 per axis, 16 frequencies): rope.py:86-109."""
    inv = 1.0 / (freq ** (torch.arange(0, 32, 2).float() / 32))
    ang = torch.arange(max_pos, dtype=f32)[:, None] * inv[None]
    return torch.stack([ang.cos(), ang.sin()], -1).contiguous()


def interpolate_pos_encoding(pos_embed: torch.Tensor, w: int, h: int, patch: int) -> torch.Tensor:
    """DinoVisionTransformer.interpolate_pos_encoding (vision_transformer.py:184-216): bicubic + antialias, fp32.
    A per-resolution constant, evaluated once on the host at model build."""
    pe = pos_embed.float().cpu()
    N = pe.shape[1] - 1
    M = int(math.sqrt(N))
    w0, h0 = w // patch, h // patch
    if w0 * h0 == N and w == h:
        return pe
    dim = pe.shape[-1]
    pp = torch.nn.functional.interpolate(pe[:, 1:].reshape(1, M, M, dim).permute(0, 3, 1, 2), mode="bicubic", antialias=True, size=(w0, h0))
    return torch.cat((pe[:, :1], pp.permute(0, 2, 3, 1).reshape(1, -1, dim)), 1)


class _Block:
    """Weights of one ViT block, packed for the fused-QK / transposed-V schedule."""

    def __init__(self, sd: SD, p: str, dev, qk_norm: bool):
        W = lambda k: sd[k].to(device=dev, dtype=bf16).contiguous()
        Fv = lambda k: sd[k].to(device=dev, dtype=f32).contiguous()
        C = sd[p + "attn.proj.weight"].shape[0]
        wqkv, bqkv = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        self.wqk = wqkv[: 2 * C].to(device=dev, dtype=bf16).contiguous()
        self.bqk = bqkv[: 2 * C].to(device=dev, dtype=f32).contiguous()
        self.wv = wqkv[2 * C:].to(device=dev, dtype=bf16).contiguous()
        self.bv = bqkv[2 * C:].to(device=dev, dtype=f32).contiguous()
        self.wo, self.bo = W(p + "attn.proj.weight"), Fv(p + "attn.proj.bias")
        self.n1w, self.n1b, self.n2w, self.n2b = Fv(p + "norm1.weight"), Fv(p + "norm1.bias"), Fv(p + "norm2.weight"), Fv(p + "norm2.bias")
        self.w1, self.b1, self.w2, self.b2 = W(p + "mlp.fc1.weight"), Fv(p + "mlp.fc1.bias"), W(p + "mlp.fc2.weight"), Fv(p + "mlp.fc2.bias")
        self.ls1, self.ls2 = Fv(p + "ls1.gamma"), Fv(p + "ls2.gamma")
        if qk_norm:
            self.qw, self.qb = Fv(p + "attn.q_norm.weight"), Fv(p + "attn.q_norm.bias")
            self.kw, self.kb = Fv(p + "attn.k_norm.weight"), Fv(p + "attn.k_norm.bias")


class _BlockF32:
    """Camera-head trunk block (fp32, a dozen tokens)."""

    def __init__(self, sd: SD, p: str, dev):
        Fv = lambda k: sd[k].to(device=dev, dtype=f32).contiguous()
        for n in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                  "attn.proj.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias", "ls1.gamma", "ls2.gamma"):
            setattr(self, n.replace(".", "_"), Fv(p + n))


class _DPT:
    def __init__(self, sd: SD, p: str, dev, gs: bool, split: bool):
        CW = ops.ConvWeightSplit if split else ops.ConvWeight
        self.split = split
        cw = lambda n, bias=True: CW(sd[p + n + ".weight"], sd.get(p + n + ".bias") if bias else None, device=dev)
        Fv = lambda k: sd[k].to(device=dev, dtype=f32).contiguous()
        self.nw, self.nb = Fv(p + "norm.weight"), Fv(p + "norm.bias")
        self.proj = [cw(f"projects.{i}") for i in range(4)]
        self.oc = [c.Cout for c in self.proj]
        # ConvTranspose2d(k = stride): out[(y*k+dy),(x*k+dx),co] = sum_ci in[y,x,ci] W[ci,co,dy,dx] + b  ==  for every dy a 1x1 conv to
        # the k*co channels (dx, co) whose output row of pixel (y, x) is row y*k+dy of the upsampled image: the pixel shuffle is the
        # GEMM epilogue's row scatter (out_rows), k launches writing the final layout instead of one launch plus a 109 MB permute copy
        self.up = []
        for i, k in ((0, 4), (1, 2)):
            w = sd[p + f"resize_layers.{i}.weight"]  # [ci, co, k, k]
            b = sd[p + f"resize_layers.{i}.bias"]
            wk = w.permute(2, 3, 1, 0)               # [dy, dx, co, ci]
            per_dy = [CW(wk[dy].reshape(k * w.shape[1], w.shape[0], 1, 1), b.repeat(k), device=dev) for dy in range(k)]
            self.up.append((per_dy, k, w.shape[1]))
        self.down = cw("resize_layers.3")
        s = "scratch."
        self.rn = [cw(s + f"layer{i + 1}_rn", bias=False) for i in range(4)]
        self.fus = {}
        for r in (1, 2, 3, 4):
            q = s + f"refinenet{r}."
            d = dict(out=cw(q + "out_conv"), c21=cw(q + "resConfUnit2.conv1"), c22=cw(q + "resConfUnit2.conv2"))
            if r != 4:
                d.update(c11=cw(q + "resConfUnit1.conv1"), c12=cw(q + "resConfUnit1.conv2"))
            self.fus[r] = d
        self.oc1 = cw(s + "output_conv1")
        self.oc20, self.oc22 = cw(s + "output_conv2.0"), cw(s + "output_conv2.2")
        self.merger = cw("input_merger.0") if gs else None
        if gs and split:
            # the 7x7 image convolution (3 -> 128 channels) with its seven dw taps unrolled into channels: the 3 real channels padded to 8 make a
            # K of 7 x 7 x 8 = 392 (x 3 products = 1176) of which 147 are real; [dw][c] = 21 -> 24 channels under a 7 x 1 kernel is K = 168 (x 3 = 504)
            w = sd[p + "input_merger.0.weight"].detach().float()              # [co, 3, dh, dw]
            w2 = torch.zeros(w.shape[0], 24, 7, 1)
            w2[:, :21, :, 0] = w.permute(0, 3, 1, 2).reshape(w.shape[0], 21, 7).cpu()   # channel dw * 3 + c
            self.merger = CW(w2, sd.get(p + "input_merger.0.bias"), device=dev)
        self.pos_cache: Dict[tuple, torch.Tensor] = {}

    def pos(self, C, ph, pw, W, H, dev):
        k = (C, ph, pw, W, H)
        if k not in self.pos_cache:
            self.pos_cache[k] = uv_pos_embed(C, ph, pw, W, H).to(dev)
        return self.pos_cache[k]


class ReconEngine:
    def __init__(self, cfg: ReconCfg, sd: SD, device="cuda"):
        L.load()
        self.cfg, self.dev = cfg, torch.device(device)
        dev, a = self.dev, "encoder.aggregator."
        pe = a + "patch_embed."
        Fv = lambda k: sd[k].to(device=dev, dtype=f32).contiguous()
        cpu = lambda k: sd[k].detach().float().cpu()
        self.cls_token, self.register_tokens, self.pos_embed = cpu(pe + "cls_token"), cpu(pe + "register_tokens"), cpu(pe + "pos_embed")
        self.dino = [_Block(sd, pe + f"blocks.{i}.", dev, False) for i in range(cfg.n_dino)]
        self.dino_nw, self.dino_nb = Fv(pe + "norm.weight"), Fv(pe + "norm.bias")
        self.camera_token, self.register_token = cpu(a + "camera_token"), cpu(a + "register_token")
        self.frame = [_Block(sd, a + f"frame_blocks.{i}.", dev, True) for i in range(cfg.depth)]
        self.glob = [_Block(sd, a + f"global_blocks.{i}.", dev, True) for i in range(cfg.depth)]
        c = "encoder.camera_head."
        self.cam_trunk = [_BlockF32(sd, c + f"trunk.{j}.", dev) for j in range(cfg.cam_trunk)]
        self.cam = {k: Fv(c + k) for k in ("token_norm.weight", "token_norm.bias", "trunk_norm.weight", "trunk_norm.bias",
                                           "poseLN_modulation.1.weight", "poseLN_modulation.1.bias", "pose_branch.fc1.weight",
                                           "pose_branch.fc1.bias", "pose_branch.fc2.weight", "pose_branch.fc2.bias", "embed_pose.bias")}
        ew = sd[c + "embed_pose.weight"].float()
        self.cam["embed_pose.weight"] = torch.nn.functional.pad(ew, (0, 3)).to(dev).contiguous()  # K 9 -> 12 (16-byte rows)
        self.cam["empty_pose_tokens"] = torch.nn.functional.pad(sd[c + "empty_pose_tokens"].float().reshape(1, 9), (0, 3)).to(dev)
        split = cfg.dpt_precision == "f32"
        self.depth_head = _DPT(sd, "encoder.depth_head.", dev, False, split)
        self.gs_head = _DPT(sd, "encoder.gaussian_param_head.", dev, True, split)
        dsh = (cfg.sh_degree + 1) ** 2
        m = torch.ones(dsh)
        for d in range(1, cfg.sh_degree + 1):
            m[d ** 2:(d + 1) ** 2] = 0.1 * 0.25 ** d
        self.sh_mask = m.to(dev)
        self._geo: Dict[tuple, dict] = {}
        self.batch_conf = False     # True while the scenes of a batch run one by one: the render_conf quantile is the batch assembly's (see _tail)

    # ------------------------------------------------------------------ per-resolution constants / workspaces
    def geometry_constants(self, S, H, W):
        """per-resolution CONSTANTS for S views (token grid, positional embedding of the patch tokens, special-token rows, rotary table): no
        workspace.  What a caller needs that only lays tokens out (the view-sharded path's full-scene token buffer) - `_geometry` adds the
        S-view workspaces (0.7-1 GB at 13 views, width 1024) on top of these."""
        key = ("const", S, H, W)
        g = self._geo.get(key)
        if g is not None:
            return g
        cfg, dev = self.cfg, self.dev
        hp, wp = H // cfg.patch, W // cfg.patch
        hw, C = hp * wp, cfg.C
        nsp = 1 + self.register_tokens.shape[1]
        P = hw + nsp
        Pp = (P + 7) // 8 * 8
        pe = interpolate_pos_encoding(self.pos_embed, W, H, cfg.patch)  # [1, 1+hw, C] f32
        cls_row = (self.cls_token.to(bf16).float().reshape(1, C) + pe[0, :1].to(bf16).float()).to(bf16)
        g = dict(hp=hp, wp=wp, hw=hw, nsp=nsp, P=P, Pp=Pp, M=S * Pp,
                 pos_patch=pe[0, 1:].to(bf16).to(dev).contiguous(),
                 special_dino=torch.cat([cls_row, self.register_tokens.reshape(-1, C).to(bf16)], 0).to(dev),
                 rope=rope2d_table(max(hp, wp) + 2).to(dev))
        cam0 = torch.cat([self.camera_token[0, 0], self.register_token[0, 0]], 0)
        cam1 = torch.cat([self.camera_token[0, 1], self.register_token[0, 1]], 0)
        sp = torch.stack([cam0] + [cam1] * (S - 1), 0)  # [S, nsp, C]
        g["special_agg"] = sp.to(bf16).float().to(dev)
        self._geo[key] = g
        return g

    def _geometry(self, S, H, W, tag=None):
        """workspaces + per-resolution constants for S views; `tag` separates the workspaces of virtual ranks (threads) sharing this engine"""
        key = (S, H, W) if tag is None else (S, H, W, tag)
        g = self._geo.get(key)
        if g is not None:
            return g
        cfg, dev = self.cfg, self.dev
        g = dict(self.geometry_constants(S, H, W))      # (the constant tensors are shared, the dict is this workspace's own)
        M, C = g["M"], cfg.C
        z = lambda *s, dt=bf16: torch.zeros(*s, device=dev, dtype=dt)
        g.update(x=z(M, C), xf=z(M, C, dt=f32), n=z(M, C), qk=z(M, 2 * C), vt=z(C, M + 64), ao=z(M, C), h=z(M, 4 * C),
                 taps=[z(M, 2 * C, dt=f32) for _ in cfg.taps])
        self._geo[key] = g
        return g

    # ------------------------------------------------------------------ transformer blocks
    def _attn(self, g, blk, S, glob: bool, rope: bool, shard=None):
        cfg = self.cfg
        C, H, P, Pp, M = cfg.C, cfg.heads, g["P"], g["Pp"], g["M"]
        ops.gemm(g["n"], blk.wqk, blk.bqk, out=g["qk"])
        ops.gemm(blk.wv, g["n"], blk.bv, out=g["vt"][:, :M], bias_row=True)
        if rope:
            ops.qknorm_rope2d(g["qk"], C, blk.qw, blk.qb, blk.kw, blk.kb, g["rope"], Pp, g["nsp"], P, g["wp"], 1e-5)
        q, k = g["qk"][:, :C], g["qk"][:, C:]
        if glob and shard is not None:
            # view-sharded global attention (aggregator.py:347-373 attends over the tokens of ALL views): this rank's K rows and V^T columns
            # go out in ONE all-gather per block, every rank reassembles the full [S Pp, C] K and [C, S Pp] V^T in view order, and the local
            # views' queries walk all keys - the same keys in the same order through the same kernel as the unsharded launch: bit-identical
            sh = shard
            Mx = sh["maxS"] * Pp
            pack, gbuf, kf, vtf = sh["pack"], sh["gbuf"], sh["kfull"], sh["vtfull"]
            self.kv_pack(pack, k, g["vt"][:, :M], Mx)
            sh["group"].all_gather(gbuf, pack).wait()
            self.kv_unpack(gbuf, sh["views"], Pp, Mx, kf, vtf)
            Mt = sh["S"] * Pp
            ops.attention(q, kf, vtf, g["ao"], B=1, H=H, Nq=M, Nk=Mt, D=C // H, q_batch_stride=0, k_batch_stride=0,
                          vt_batch_stride=0, o_batch_stride=0, kv_period=Pp if Pp != P else 0, kv_valid=P)
        elif glob:
            ops.attention(q, k, g["vt"], g["ao"], B=1, H=H, Nq=M, Nk=M, D=C // H, q_batch_stride=0, k_batch_stride=0,
                          vt_batch_stride=0, o_batch_stride=0, kv_period=Pp if Pp != P else 0, kv_valid=P)
        else:
            ops.attention(q, k, g["vt"], g["ao"], B=S, H=H, Nq=P, Nk=P, D=C // H, q_batch_stride=Pp * 2 * C,
                          k_batch_stride=Pp * 2 * C, vt_batch_stride=Pp, o_batch_stride=Pp * C)

    def _block(self, g, blk, x, S, glob, rope, eps, xo=None, shard=None):
        """x: residual stream (bf16 for DINO, f32 for the aggregator).  The block's output goes to `xo` (default: x, in place); with
        xo != x the attention branch already lands in xo (residual read from x), so x is left untouched - how the tapped aggregator
        blocks write their output straight into the [M, 2C] tap buffers (row stride 2C) instead of being copied there afterwards."""
        f = x.dtype == f32
        xo = x if xo is None else xo
        ops.layernorm(x, out=g["n"], weight=blk.n1w, bias=blk.n1b, eps=eps)
        self._attn(g, blk, S, glob, rope, shard)
        ops.gemm(g["ao"], blk.wo, blk.bo, out=xo, residual=x, scale=blk.ls1, round_after_scale=True, out_f32=f)
        ops.layernorm(xo, out=g["n"], weight=blk.n2w, bias=blk.n2b, eps=eps)
        ops.gemm(g["n"], blk.w1, blk.b1, out=g["h"], act=L.ACT_GELU_ERF)
        ops.gemm(g["h"], blk.w2, blk.b2, out=xo, residual=xo, scale=blk.ls2, round_after_scale=True, out_f32=f)

    def backbone(self, g, S, shard=None, hook=None):
        """x (bf16 tokens incl. DINO specials) -> tapped [M, 2C] f32 intermediates.  `shard` (forward_sharded): S = this rank's views.
        hook(kind, index, "in" | "out", buffer) (tests): called with the LIVE residual-stream buffer in front of and behind every block
        (kind in "dino" / "frame" / "global"; rows in the padded [S, Pp, .] layout, row stride = buffer.stride(0)) - the production
        forward's own stream, for per-block teacher-forced comparisons; the callee copies what it wants to keep."""
        hk = hook if hook is not None else (lambda *a: None)
        cfg = self.cfg
        C, Pp, nsp = cfg.C, g["Pp"], g["nsp"]
        x, xf = g["x"], g["xf"]
        x.view(S, Pp, C)[:, :nsp] = g["special_dino"]
        for i, blk in enumerate(self.dino):
            hk("dino", i, "in", x)
            self._block(g, blk, x, S, False, False, 1e-6)
            hk("dino", i, "out", x)
        ops.layernorm(x, out=xf, weight=self.dino_nw, bias=self.dino_nb, eps=1e-6)
        xf.view(S, Pp, C)[:, :nsp] = g["special_agg"] if shard is None else shard["special_agg"]
        # The residual stream walks THROUGH the tap buffers (anysplat_stitched.py:249-325 concatenates the frame and global intermediates
        # of the tapped layers): a tapped frame block writes its output into the left half of the tap, the global block reads it there
        # and writes the right half, the next frame block reads that and returns to xf.  No concatenation copies (8 x 55 MB per scene).
        cur, ti = xf, 0
        for li in range(cfg.depth):
            tap = li in cfg.taps
            dst = g["taps"][ti][:, :C] if tap else xf
            hk("frame", li, "in", cur)
            self._block(g, self.frame[li], cur, S, False, True, 1e-5, xo=dst)
            hk("frame", li, "out", dst)
            cur = dst
            dst = g["taps"][ti][:, C:] if tap else xf
            hk("global", li, "in", cur)
            self._block(g, self.glob[li], cur, S, True, True, 1e-5, xo=dst, shard=shard)
            hk("global", li, "out", dst)
            cur = dst
            ti += int(tap)
        return g["taps"]

    # ------------------------------------------------------------------ camera head (fp32)
    def camera(self, g, S, iters: int = 4, pose_tokens: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
        """pose_tokens [S, 2C] f32: the camera-token rows of the last tap (default: read from this workspace's taps)"""
        cfg, cam = self.cfg, self.cam
        C2, Hh = 2 * cfg.C, cfg.cam_heads
        pt = g["taps"][-1].view(S, g["Pp"], C2)[:, 0].contiguous() if pose_tokens is None else pose_tokens.contiguous()
        pt = ops.layernorm(pt, weight=cam["token_norm.weight"], bias=cam["token_norm.bias"], eps=1e-5, out_dtype=f32)
        pred, outs = None, []
        for _ in range(iters):
            inp = cam["empty_pose_tokens"].expand(S, -1).contiguous() if pred is None else torch.nn.functional.pad(pred, (0, 3))
            mi = ops.linear_f32(inp, cam["embed_pose.weight"], cam["embed_pose.bias"], act=L.ACT_SILU)
            mod = ops.linear_f32(mi, cam["poseLN_modulation.1.weight"], cam["poseLN_modulation.1.bias"])  # [S, 3*C2]
            ln = ops.layernorm(pt, scale=mod[:, C2:2 * C2], shift=mod[:, :C2], rows_per_batch=1, eps=1e-6, out_dtype=f32)
            x = mod[:, 2 * C2:] * ln + pt
            for b in self.cam_trunk:
                n = ops.layernorm(x, weight=b.norm1_weight, bias=b.norm1_bias, eps=1e-5, out_dtype=f32)
                qkv = ops.linear_f32(n, b.attn_qkv_weight, b.attn_qkv_bias)
                ao = ops.attention_small_f32(qkv, Hh)
                x = ops.linear_f32(ao, b.attn_proj_weight, b.attn_proj_bias, gamma=b.ls1_gamma, residual=x)
                n = ops.layernorm(x, weight=b.norm2_weight, bias=b.norm2_bias, eps=1e-5, out_dtype=f32)
                hmid = ops.linear_f32(n, b.mlp_fc1_weight, b.mlp_fc1_bias, act=L.ACT_GELU_ERF)
                x = ops.linear_f32(hmid, b.mlp_fc2_weight, b.mlp_fc2_bias, gamma=b.ls2_gamma, residual=x)
            n = ops.layernorm(x, weight=cam["trunk_norm.weight"], bias=cam["trunk_norm.bias"], eps=1e-5, out_dtype=f32)
            hmid = ops.linear_f32(n, cam["pose_branch.fc1.weight"], cam["pose_branch.fc1.bias"], act=L.ACT_GELU_ERF)
            d = ops.linear_f32(hmid, cam["pose_branch.fc2.weight"], cam["pose_branch.fc2.bias"])
            pred = d if pred is None else pred + d
            outs.append(torch.cat([pred[:, :7], torch.relu(pred[:, 7:])], -1))
        return outs

    # ------------------------------------------------------------------ DPT trunk (channels-last bf16)
    @staticmethod
    def _conv_bf16(g, S):
        """ops.conv with the kernel-FORM hint of a view-sharded forward: a rank that holds S of the scene's g["form_frames"] views chooses
        halo-tile vs implicit GEMM as the whole scene would (the two forms differ in fp32 summation order), like `form_frames` does on the
        fp32-equivalent path.  (At the production DPT widths no bf16 layer has a halo form - 256 channels are not a multiple of 96 - so
        this only matters for other head widths.)"""
        ff = g.get("form_frames")
        fs = (ff / S) if (ff and S) else None
        return lambda x_, cw_, **kw: ops.conv(x_, cw_, form_scale=fs, **kw)

    def _dpt_trunk(self, g, hd: _DPT, S, H, W):
        cfg, dev = self.cfg, self.dev
        hp, wp, hw, Pp, nsp = g["hp"], g["wp"], g["hw"], g["Pp"], g["nsp"]
        C2 = 2 * cfg.C
        lv = []
        cv = self._conv_bf16(g, S)
        for i, tap in enumerate(g["taps"]):
            n = ops.layernorm(tap, weight=hd.nw, bias=hd.nb, eps=1e-5, M=S * hw, in_rows=(hw, Pp - hw, nsp)).view(S, hp, wp, C2)
            x = cv(n, hd.proj[i], residual=hd.pos(hd.proj[i].CoutP, hp, wp, W, H, dev), res_row_mod=hw)
            if i < 2:
                per_dy, k, co = hd.up[i]
                up = torch.empty(S, hp * k, wp, k * co, device=dev, dtype=bf16)   # = [S, hp*k, wp*k, co]
                for dy, cwt in enumerate(per_dy):   # pixel m = (s, y, x) -> row ((s*hp + y)*k + dy)*wp + x of the [.., k*co] matrix
                    cv(x, cwt, out=up, out_rows=(wp, (k - 1) * wp, dy * wp))
                x = up.view(S, hp * k, wp * k, co)
            elif i == 3:
                x = cv(x, hd.down, stride=(1, 2, 2), pad=(0, 1, 1))
            lv.append(cv(x, hd.rn[i], pad=(0, 1, 1), relu_out=True))  # ReLU: only ever consumed through relu()
        R = L.ACT_RELU

        def rcu2_out(f, s, size):
            c1 = cv(s, f["c21"], pad=(0, 1, 1), act=R)
            o = cv(c1, f["c22"], pad=(0, 1, 1), residual=s)
            o = cv(o, f["out"])  # 1x1 out_conv commutes with the bilinear resize that follows it in the reference
            return ops.bilinear_cl(o, size, align_corners=True)

        o = rcu2_out(hd.fus[4], lv[3], lv[2].shape[1:3])
        for r, l in ((3, lv[2]), (2, lv[1]), (1, lv[0])):
            f = hd.fus[r]
            c1 = cv(l, f["c11"], pad=(0, 1, 1), act=R)
            s = cv(c1, f["c12"], pad=(0, 1, 1), residual=l, residual2=o, relu_out=True)  # relu(x0 + RCU1(x1))
            size = lv[r - 2].shape[1:3] if r > 1 else (l.shape[1] * 2, l.shape[2] * 2)
            o = rcu2_out(f, s, size)
        return cv(o, hd.oc1, pad=(0, 1, 1))

    # ------------------------------------------------------------------ DPT trunk, fp32-equivalent (pairs of bf16 planes)
    def _dpt_trunk_f32(self, g, hd: _DPT, S, H, W):
        """The same launches as _dpt_trunk on (hi, lo) pairs: ops.conv_split / layernorm_pair / bilinear_cl_pair keep every value in fp32
        between the MFMA accumulators and the next layer's operands (dpt_head.py:185-309 under autocast off)."""
        cfg, dev = self.cfg, self.dev
        hp, wp, hw, Pp, nsp = g["hp"], g["wp"], g["hw"], g["Pp"], g["nsp"]
        C2 = 2 * cfg.C
        lv = []
        ff = g.get("form_frames")       # view-sharded forward: kernel-form decisions follow the scene's frame count, not the shard's
        cs = lambda x_, cw_, **kw: ops.conv_split(x_, cw_, form_frames=ff, **kw)
        for i, tap in enumerate(g["taps"]):
            n = ops.layernorm_pair(tap, weight=hd.nw, bias=hd.nb, eps=1e-5, M=S * hw, in_rows=(hw, Pp - hw, nsp)).view(2, S, hp, wp, C2)
            x = cs(n, hd.proj[i], residual=hd.pos(hd.proj[i].CoutP, hp, wp, W, H, dev), res_row_mod=hw)
            if i < 2:
                per_dy, k, co = hd.up[i]
                up = torch.empty(2, S, hp * k, wp, k * co, device=dev, dtype=bf16)
                for dy, cwt in enumerate(per_dy):
                    cs(x, cwt, out=up, out_rows=(wp, (k - 1) * wp, dy * wp))
                x = up.view(2, S, hp * k, wp * k, co)
            elif i == 3:
                x = cs(x, hd.down, stride=(1, 2, 2), pad=(0, 1, 1))
            lv.append(cs(x, hd.rn[i], pad=(0, 1, 1), relu_out=True))
        R = L.ACT_RELU

        def rcu2_out(f, s, size):
            c1 = cs(s, f["c21"], pad=(0, 1, 1), act=R)
            o = cs(c1, f["c22"], pad=(0, 1, 1), residual=s)
            o = cs(o, f["out"])
            return ops.bilinear_cl_pair(o, size, align_corners=True)

        o = rcu2_out(hd.fus[4], lv[3], lv[2].shape[2:4])
        for r, l in ((3, lv[2]), (2, lv[1]), (1, lv[0])):
            f = hd.fus[r]
            c1 = cs(l, f["c11"], pad=(0, 1, 1), act=R)
            s = cs(c1, f["c12"], pad=(0, 1, 1), residual=l, residual2=o, relu_out=True)
            size = lv[r - 2].shape[2:4] if r > 1 else (l.shape[2] * 2, l.shape[3] * 2)
            o = rcu2_out(f, s, size)
        return cs(o, hd.oc1, pad=(0, 1, 1))

    def _heads_f32(self, g, S, H, W, img: torch.Tensor, cam: torch.Tensor):
        """img: [S,H,W,8] f32 (3 real channels, in [0,1])."""
        dev = self.dev
        R = L.ACT_RELU
        ff = g.get("form_frames")
        cs = lambda x_, cw_, **kw: ops.conv_split(x_, cw_, form_frames=ff, **kw)
        d = self.depth_head
        o = self._dpt_trunk_f32(g, d, S, H, W)
        up = ops.bilinear_cl_pair(o, (H, W), align_corners=True, table=d.pos(o.shape[-1], H, W, W, H, dev))
        c = cs(up, d.oc20, pad=(0, 1, 1), act=R)
        raw = cs(c, d.oc22, out_f32=True).view(S * H * W, -1)
        depth, dconf, pts = ops.depth_unproject(raw, cam, S, H, W)
        q = self.gs_head
        o = self._dpt_trunk_f32(g, q, S, H, W)
        # image with the dw taps of the 7x7 kernel unrolled into channels (see _DPT): u[s, h, w, dw * 3 + c] = img[s, h, w + dw - 3, c]
        u = torch.nn.functional.pad(img[..., :3], (0, 0, 3, 3)).unfold(2, 7, 1).permute(0, 1, 2, 4, 3).reshape(S, H, W, 21)
        di = cs(ops.split_f32(torch.nn.functional.pad(u, (0, 3)).contiguous()), q.merger, pad=(0, 3, 0), act=R)
        up = ops.bilinear_cl_pair(o, (H, W), align_corners=True, add=di, table=q.pos(o.shape[-1], H, W, W, H, dev))
        c = cs(up, q.oc20, pad=(0, 1, 1), act=R)
        raw_gs = cs(c, q.oc22, out_f32=True).view(S * H * W, -1)
        return depth, dconf, pts, raw_gs

    def heads(self, g, S, H, W, img_cl: torch.Tensor, pose: torch.Tensor):
        """img_cl: the context image in [0,1], channels-last [S,H,W,8] (3 real channels, the rest zero), f32 or bf16 (converted to what
        cfg.dpt_precision computes in).
        -> depth [S,H,W], depth_conf [S,H,W], pts [S,H,W,3], raw_gs [S*H*W, 88] f32, ext [S,3,4], K [S,3,3]."""
        dev = self.dev
        ext, K = pose_encoding_to_extri_intri(pose, (H, W))
        Rt = ext[:, :, :3].transpose(1, 2)
        tinv = -(Rt * ext[:, None, :, 3]).sum(-1)   # -R^T t, elementwise (a [S,3,3] @ [S,3,1] matmul would dispatch a BLAS kernel for 9 FMAs)
        cam = torch.cat([K[:, 0, 0:1], K[:, 1, 1:2], K[:, 0, 2:3], K[:, 1, 2:3], Rt.reshape(S, 9), tinv], 1).contiguous()
        if self.depth_head.split:
            return (*self._heads_f32(g, S, H, W, img_cl.float().contiguous(), cam), ext, K)
        img_cl = img_cl.to(bf16)
        R = L.ACT_RELU
        cv = self._conv_bf16(g, S)
        # depth head
        d = self.depth_head
        o = self._dpt_trunk(g, d, S, H, W)
        up = ops.bilinear_cl(o, (H, W), align_corners=True, table=d.pos(o.shape[-1], H, W, W, H, dev))
        c = cv(up, d.oc20, pad=(0, 1, 1), act=R)
        raw = cv(c, d.oc22, out_f32=True).view(S * H * W, -1)
        depth, dconf, pts = ops.depth_unproject(raw, cam, S, H, W)
        # gaussian-parameter head
        q = self.gs_head
        o = self._dpt_trunk(g, q, S, H, W)
        di = cv(img_cl, q.merger, pad=(0, 3, 3), act=R)
        up = ops.bilinear_cl(o, (H, W), align_corners=True, add=di, table=q.pos(o.shape[-1], H, W, W, H, dev))
        c = cv(up, q.oc20, pad=(0, 1, 1), act=R)
        raw_gs = cv(c, q.oc22, out_f32=True).view(S * H * W, -1)
        return depth, dconf, pts, raw_gs, ext, K

    # ------------------------------------------------------------------ full forward
    @torch.no_grad()
    def forward_tokens_filled(self, S: int, H: int, W: int, img_cl: torch.Tensor) -> dict:
        """Runs everything after the patch tokens have been written into the workspace (rows f*Pp + nsp + p)."""
        cfg = self.cfg
        g = self._geometry(S, H, W)
        self.backbone(g, S)
        poses = self.camera(g, S)
        depth, dconf, pts, raw_gs, ext, K = self.heads(g, S, H, W, img_cl, poses[-1])
        return self._tail(S, H, W, poses, depth, dconf, pts, raw_gs, ext, K)

    def _tail(self, S, H, W, poses, depth, dconf, pts, raw_gs, ext, K) -> dict:
        """per-pixel maps of all S views -> confidence mask / voxel fusion / Gaussian adapter (anysplat_stitched.py:381-453)"""
        cfg = self.cfg
        gsd = 1 + 7 + 3 * (cfg.sh_degree + 1) ** 2  # 83: density + raw gaussian; confidence sits in the next column
        out = dict(pred_pose_enc_list=poses, depth=depth, depth_conf=dconf, pts_all=pts, raw_gs=raw_gs, extrinsic_w2c=ext, intrinsic_px=K)
        M = S * H * W
        c = None
        if cfg.render_conf and not self.batch_conf:
            # the reference takes the quantile whenever render_conf is set (anysplat_stitched.py:381-387), also when the
            # voxel branch then ignores the mask for the Gaussians: depth_dict["conf_valid_mask"] is returned either way (:494).
            # (`batch_conf`: this scene is one of a batch - the quantile spans the batch and is taken by the batch assembly,
            # models/anysplat_stitched.py::assemble_batch, from the per-pixel maps returned here.)
            c = ops.conf_quantile_compact(dconf.reshape(M), cfg.conf_threshold, pts.view(M, 3), raw_gs, gsd)
            out["conf_valid"] = c["threshold"]
        if cfg.voxelize:
            v = ops.voxelize_fuse(pts.view(M, 3), raw_gs, gsd, gsd, cfg.voxel_size)
            vp, vf = v["voxel_pts"], v["voxel_feat"]
            out.update(voxel_keys=v["keys"], voxel_inverse=v["inverse"], voxel_counts=v["counts"])
        elif c is not None:
            vp, vf = c["pts"], c["feat"]
        else:
            vp, vf = pts.view(M, 3), raw_gs[:, :gsd]
        out["neural_pts"], out["neural_feats"] = vp, vf      # per-point / per-voxel inputs of the adapter (batched forwards pad and re-run it)
        out["gaussians"] = ops.gaussian_adapter(vp, vf, self.sh_mask, cfg.sh_degree, cfg.opacity_exponent)
        out["scene_scale"] = pts.view(-1, 3).norm(dim=-1).mean().clip(min=1e-8)
        out["num_points"] = M
        return out

    # ------------------------------------------------------------------ view-sharded forward (one scene over several ranks)
    @staticmethod
    def kv_pack(pack: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, Mx: int) -> None:
        """a rank's contribution to a global block's all-gather: [K rows [M, C] | V^T [C, Mx] (M valid columns)] in one flat buffer"""
        M, C = k.shape
        pack[: M * C].view(M, C).copy_(k)
        pack[Mx * C:].view(C, Mx)[:, :M].copy_(vt)

    @staticmethod
    def kv_unpack(gbuf: torch.Tensor, views, Pp: int, Mx: int, kfull: torch.Tensor, vtfull: torch.Tensor) -> None:
        """the all-gathered packs [P, 2 Mx C] -> K [S Pp, C] and V^T [C, >= S Pp] of ALL views in view order"""
        C = kfull.shape[1]
        for r, (v0, v1) in enumerate(views):
            Mr = (v1 - v0) * Pp
            if Mr:
                kfull[v0 * Pp: v1 * Pp].copy_(gbuf[r, : Mr * C].view(Mr, C))
                vtfull[:, v0 * Pp: v1 * Pp].copy_(gbuf[r, Mx * C:].view(C, Mx)[:, :Mr])

    @staticmethod
    def gather_rows(gbuf: torch.Tensor, views, rows_per_view: int, cols: int, max_views: int) -> torch.Tensor:
        """per-rank padded blocks [P, max_views * rows_per_view * cols] -> the rows of all views in view order"""
        return torch.cat([gbuf[r].view(max_views * rows_per_view, cols)[: (b - a) * rows_per_view] for r, (a, b) in enumerate(views)], 0)

    @staticmethod
    def shard_views(S: int, world: int) -> List[tuple]:
        """[(first view, one past last view)] per rank: contiguous runs, the first S % world ranks hold one view more"""
        base, extra = divmod(S, world)
        out, v = [], 0
        for r in range(world):
            n = base + (1 if r < extra else 0)
            out.append((v, v + n))
            v += n
        return out

    @torch.no_grad()
    def forward_sharded(self, S: int, H: int, W: int, tokens: torch.Tensor, img_cl: torch.Tensor, group, timings: Optional[dict] = None) -> dict:
        """The reconstruction of ONE scene over the `group.world` ranks of a scene-parallel run (BASELINE configs #3 / #4):
        views are split over ranks (13 -> 7 / 6, 4 / 3 / 3 / 3, ...).  DINO blocks, frame blocks and the DPT heads are per-view
        (aggregator.py:318-345, dpt_head.py: per-frame 2-D convolutions) and run on the local views only; every global block all-gathers
        the ranks' K | V^T (aggregator.py:347-373; _attn); the camera head (13 tokens, weight-streaming) runs replicated on the all-gathered
        camera-token rows; the per-pixel maps are all-gathered once and the voxel / adapter tail (3 ms) runs replicated.  Every row is
        computed by the same kernels in the same k-order as in `forward_tokens_filled`: the result is bit-identical to the unsharded forward
        (tests/test_recon_gpu.py::test_view_sharded_forward_is_bit_identical).
        tokens: [S * Pp, C] bf16 in the layout of `token_workspace` (all views; the stitching convolution is 2e10 FLOP and runs on every
        rank); img_cl: [S,H,W,8] all views.  Every rank returns the full scene."""
        import threading
        cfg, dev = self.cfg, self.dev
        P_, rk = group.world, group.rank
        views = self.shard_views(S, P_)
        v0, v1 = views[rk]
        Sl, maxS = v1 - v0, max(b - a for a, b in views)
        tid = threading.get_ident()
        gl = self._geometry(max(Sl, 1), H, W, tag=("shard", tid))       # (a rank without views still takes part in every collective)
        C, Pp, nsp, hw, hp, wp = cfg.C, gl["Pp"], gl["nsp"], gl["hw"], gl["hp"], gl["wp"]
        skey = ("shardbuf", S, H, W, P_, rk, tid)
        sh = self._geo.get(skey)
        if sh is None:
            z = lambda *s_, dt=bf16: torch.zeros(*s_, device=dev, dtype=dt)
            Mx = maxS * Pp
            cam0 = torch.cat([self.camera_token[0, 0], self.register_token[0, 0]], 0)
            cam1 = torch.cat([self.camera_token[0, 1], self.register_token[0, 1]], 0)
            sp_full = torch.stack([cam0] + [cam1] * (S - 1), 0).to(bf16).float().to(dev)       # view 0 carries the first-frame camera token
            npx = maxS * H * W
            sh = self._geo[skey] = dict(pack=z(2 * Mx * C), gbuf=z(P_, 2 * Mx * C), kfull=z(S * Pp, C), vtfull=z(C, S * Pp + 64),
                                        special_agg=sp_full[v0:v1], cpack=z(maxS, 2 * C, dt=f32), cgbuf=z(P_, maxS * 2 * C, dt=f32),
                                        opack=z(npx * OUT_COLS, dt=f32), ogbuf=z(P_, npx * OUT_COLS, dt=f32))
        sh.update(group=group, views=views, S=S, maxS=maxS)
        ev = (lambda: _ev()) if timings is not None else (lambda: None)
        e0 = ev()
        if Sl:
            gl["form_frames"] = S        # the halo / implicit-GEMM choice of the DPT convolutions follows the SCENE's frame count
            gl["x"].view(Sl, Pp, C)[:, nsp:nsp + hw] = tokens.view(S, Pp, C)[v0:v1, nsp:nsp + hw]
            self.backbone(gl, Sl, shard=sh)
            sh["cpack"][:Sl] = gl["taps"][-1].view(Sl, Pp, 2 * C)[:, 0]
        else:
            for _ in range(cfg.depth):   # stay in lockstep with the ranks that own views
                group.all_gather(sh["gbuf"], sh["pack"]).wait()
        group.all_gather(sh["cgbuf"], sh["cpack"]).wait()
        ptok = self.gather_rows(sh["cgbuf"], views, 1, 2 * C, maxS)     # [S, 2C] in view order
        e1 = ev()
        poses = self.camera(gl, S, pose_tokens=ptok)
        e2 = ev()
        op = sh["opack"]
        HW = H * W
        if Sl:
            depth, dconf, pts, raw_gs, _, _ = self.heads(gl, Sl, H, W, img_cl[v0:v1], poses[-1][v0:v1])
            o2 = op.view(maxS * HW, OUT_COLS)
            o2[: Sl * HW, 0], o2[: Sl * HW, 1] = depth.reshape(-1), dconf.reshape(-1)
            o2[: Sl * HW, 2:5] = pts.reshape(-1, 3)
            o2[: Sl * HW, 5:] = raw_gs
        e3 = ev()
        group.all_gather(sh["ogbuf"], op).wait()
        allv = self.gather_rows(sh["ogbuf"], views, HW, OUT_COLS, maxS)     # [S H W, 93] in view order
        depth, dconf = allv[:, 0].reshape(S, H, W).contiguous(), allv[:, 1].reshape(S, H, W).contiguous()
        pts, raw_gs = allv[:, 2:5].reshape(S, H, W, 3).contiguous(), allv[:, 5:].contiguous()
        ext, K = pose_encoding_to_extri_intri(poses[-1], (H, W))
        out = self._tail(S, H, W, poses, depth, dconf, pts, raw_gs, ext, K)
        e4 = ev()
        if timings is not None:
            torch.cuda.synchronize()
            timings.update(views=[Sl, S], backbone_ms=e0.elapsed_time(e1), camera_ms=e1.elapsed_time(e2), heads_ms=e2.elapsed_time(e3),
                           gather_and_tail_ms=e3.elapsed_time(e4))
        return out

    def token_buffer(self, S: int, H: int, W: int, tag) -> torch.Tensor:
        """a private [S * Pp, C] bf16 token buffer in the workspace layout (view-sharded forward: one per rank / virtual rank)"""
        g = self.geometry_constants(S, H, W)
        key = ("tokbuf", S, H, W, tag)
        t = self._geo.get(key)
        if t is None:
            t = self._geo[key] = torch.zeros(S * g["Pp"], self.cfg.C, device=self.dev, dtype=bf16)
        return t

    def token_workspace(self, S: int, H: int, W: int):
        """(x [S*Pp, C] bf16, geometry dict): the stitching conv writes patch tokens into x via out_rows=(hw, Pp-hw, nsp)."""
        g = self._geometry(S, H, W)
        return g["x"], g

    @torch.no_grad()
    def forward(self, context_latent: torch.Tensor, context_image: torch.Tensor) -> dict:
        """AnySplatStitched.forward inputs: context_latent [1,C,S,hp,wp] (stitch output), context_image [1,3,S,H,W] in [-1,1]."""
        _, C, S, hp, wp = context_latent.shape
        H, W = context_image.shape[-2:]
        x, g = self.token_workspace(S, H, W)
        tok = context_latent[0].permute(1, 2, 3, 0).reshape(S, hp * wp, C).to(device=self.dev, dtype=bf16)
        x.view(S, g["Pp"], C)[:, g["nsp"]:g["nsp"] + hp * wp] = (tok.float() + g["pos_patch"].float()[None]).to(bf16)
        img = (context_image[0].to(self.dev).float().permute(1, 2, 3, 0) + 1) / 2  # [S,H,W,3] in [0,1]
        img_cl = torch.zeros(S, H, W, 8, device=self.dev, dtype=f32)
        img_cl[..., :3] = img
        return self.forward_tokens_filled(S, H, W, img_cl)


OUT_COLS = 93    # per-pixel outputs a rank contributes to the final all-gather: depth, depth_conf, xyz, 88 raw head columns


def _ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def quat_to_mat(q: torch.Tensor) -> torch.Tensor:
    """xyzw quaternion -> rotation (vggt/utils/rotation.py:14-44)."""
    i, j, k, r = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def pose_encoding_to_extri_intri(pose: torch.Tensor, hw):
    """[S,9] (T, quat xyzw, fov_h, fov_w) -> world->camera [S,3,4], pixel intrinsics [S,3,3] (vggt/utils/pose_enc.py:65-130).
    A dozen 3x3 matrices: host-side tensor algebra, not a kernel."""
    T, quat, fov_h, fov_w = pose[..., :3], pose[..., 3:7], pose[..., 7], pose[..., 8]
    ext = torch.cat([quat_to_mat(quat), T[..., None]], -1)
    H, W = hw
    fy = (H / 2.0) / (torch.tan(fov_h / 2.0) + 1e-3)
    fx = (W / 2.0) / (torch.tan(fov_w / 2.0) + 1e-3)
    K = torch.zeros(pose.shape[:-1] + (3, 3), dtype=pose.dtype, device=pose.device)
    K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2], K[..., 2, 2] = fx, fy, W / 2, H / 2, 1.0
    return ext, K
