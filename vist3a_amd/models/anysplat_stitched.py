"""`AnySplatStitched` — drop-in for /root/reference/models/anysplat_stitched.py:143-525 on the inference path.

forward(context_latent[b, C, v, hp, wp], context_image[B, 3, S, H, W] in [-1,1], train) -> EncoderOutput
(train=True additionally returns anchor_feats [B,S,83,H,W], conf [B,S,H,W], depth_conf [B,S,H,W], :501-514).
The arithmetic runs in vist3a_amd.recon.engine.ReconEngine (HIP kernels); this class owns the reference's attribute
surface that callers touch: `.encoder.aggregator.patch_embed.{cls_token, register_tokens, mask_token}` (assigned by the
checkpoint loader, nvs_eval.py:53-61), `.encoder.cfg.voxelize` (model_stitching_training.py:331), `.decoder` (rasteriser,
out of scope: SURVEY.md §8f rank 1)."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch

from ..recon.engine import ReconCfg, ReconEngine
from .types import EncoderOutput, Gaussians


class AnySplatWeights:
    """What the reference calls `feedforward_model` (an `AnySplat` instance): here just its tensors + hyper-parameters."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: Optional[ReconCfg] = None, n_total_dino_blocks: Optional[int] = None):
        self.state_dict, self.cfg = state_dict, cfg or ReconCfg()
        self.n_total_dino_blocks = n_total_dino_blocks


class AnySplatStitched(torch.nn.Module):
    def __init__(self, model: AnySplatWeights, stitching_layer: str, device="cuda"):
        super().__init__()
        self.stitching_layer = stitching_layer
        self.stitched_layer_index = int(stitching_layer.split("_")[-1])  # "enc_blocks_k": DINO patch-embed + first k blocks dropped
        sd, cfg = dict(model.state_dict), model.cfg
        pe = "encoder.aggregator.patch_embed.blocks."
        present = sorted({int(k[len(pe):].split(".")[0]) for k in sd if k.startswith(pe)})
        if model.n_total_dino_blocks is not None and len(present) == model.n_total_dino_blocks:
            # full upstream checkpoint: delete the first k blocks and re-index, as convert_model_to_stitched_model (:158-165)
            k = self.stitched_layer_index
            moved = {}   # two passes: renaming in place would let "blocks.10." land on "blocks.8." before the old "blocks.8." has moved
            for key in [q for q in sd if q.startswith(pe)]:
                i = int(key[len(pe):].split(".")[0])
                v = sd.pop(key)
                if i >= k:
                    moved[pe + str(i - k) + key[len(pe) + len(str(i)):]] = v
            sd.update(moved)
            present = present[k:]
        if len(present) != cfg.n_dino:
            raise ValueError(f"state dict holds {len(present)} DINO blocks after stitching, config expects {cfg.n_dino}")
        self._sd, self._cfg, self._device = sd, cfg, torch.device(device)
        self._engine: Optional[ReconEngine] = None
        P = torch.nn.Parameter
        a = "encoder.aggregator.patch_embed."
        patch_embed = torch.nn.Module()
        patch_embed.cls_token = P(sd[a + "cls_token"].clone(), requires_grad=False)
        patch_embed.register_tokens = P(sd[a + "register_tokens"].clone(), requires_grad=False)
        patch_embed.mask_token = P(sd.get(a + "mask_token", torch.zeros(1, cfg.C)).clone(), requires_grad=False)
        aggregator = torch.nn.Module()
        aggregator.patch_embed = patch_embed
        self.encoder = torch.nn.Module()
        self.encoder.aggregator = aggregator
        self.encoder.cfg = SimpleNamespace(voxelize=cfg.voxelize, voxel_size=cfg.voxel_size, pred_head_type="depth",
                                           render_conf=cfg.render_conf, opacity_conf=False, conf_threshold=cfg.conf_threshold)
        self.encoder.raw_gs_dim = 1 + 7 + 3 * (cfg.sh_degree + 1) ** 2
        self._decoder = None
        self.grad_checkpointing = False

    @property
    def decoder(self):
        """Rasteriser (config/model/decoder/splatting_cuda.yaml: white background), built on first use."""
        if self._decoder is None:
            from .decoder_splatting import DecoderSplattingCUDA
            self._decoder = DecoderSplattingCUDA(background_color=(1.0, 1.0, 1.0), make_scale_invariant=False, device=self._device)
        return self._decoder

    def load_lora(self, lora_sd: Dict[str, torch.Tensor], alpha: float, r: int) -> int:
        from ..recon.weights import merge_lora
        self._engine = None
        return merge_lora(self._sd, lora_sd, alpha, r)

    def engine(self) -> ReconEngine:
        pe = self.encoder.aggregator.patch_embed
        a = "encoder.aggregator.patch_embed."
        ec = self.encoder.cfg
        dirty = (self._engine is None or self._cfg.voxelize != ec.voxelize or self._cfg.render_conf != ec.render_conf
                 or self._cfg.conf_threshold != ec.conf_threshold)
        for name in ("cls_token", "register_tokens"):
            if not torch.equal(getattr(pe, name).detach().cpu().float(), self._sd[a + name].detach().cpu().float()):
                self._sd[a + name] = getattr(pe, name).detach().clone()
                dirty = True
        if dirty:
            self._cfg.voxelize, self._cfg.render_conf = bool(ec.voxelize), bool(ec.render_conf)
            self._cfg.conf_threshold = float(ec.conf_threshold)
            self._engine = ReconEngine(self._cfg, self._sd, self._device)
        return self._engine

    def package(self, out: dict, S: int, H: int, W: int, train: bool):
        """raw engine outputs -> EncoderOutput exactly as anysplat_stitched.py:448-525 assembles it (batch dim b = 1)."""
        g = out["gaussians"]
        gauss = Gaussians(**{k: v[None] for k, v in g.items()})
        ext, K = out["extrinsic_w2c"][None], out["intrinsic_px"][None]
        pad = torch.tensor([0, 0, 0, 1.0], device=ext.device, dtype=ext.dtype).view(1, 1, 1, 4).repeat(1, S, 1, 1)
        Kn = torch.stack([K[:, :, 0] / W, K[:, :, 1] / H, K[:, :, 2]], 2)
        pose = dict(extrinsic=torch.cat([ext, pad], 2).inverse(), intrinsic=Kn)
        depth = out["depth"].view(1, S, H, W, 1)
        dconf = out["depth_conf"].view(1, S, H, W)
        U = g["means"].shape[0]
        # anysplat_stitched.py:381-387: depth_conf > torch.quantile(depth_conf, conf_threshold) under render_conf, else all-true
        valid = dconf > out["conf_valid"] if "conf_valid" in out else torch.ones_like(dconf, dtype=torch.bool)
        poses = [p[None] for p in out["pred_pose_enc_list"]]
        eo = EncoderOutput(gaussians=gauss, pred_pose_enc_list=poses, pred_context_pose=pose,
                           depth_dict=dict(depth=depth, conf_valid_mask=valid),
                           infos=dict(scene_scale=out["scene_scale"], voxelize_ratio=U / (H * W * S)), distill_infos=None,
                           last_pred_pose_enc=None if train else poses[-1])
        if not train:
            return eo
        gsd = self.encoder.raw_gs_dim
        raw = out["raw_gs"].view(1, S, H, W, -1).permute(0, 1, 4, 2, 3)
        return eo, raw[:, :, :gsd], raw[:, :, gsd], dconf

    @torch.no_grad()
    def forward(self, context_latent: torch.Tensor, context_image: torch.Tensor, train: bool = False):
        B, c, S, H, W = context_image.shape
        if context_latent.shape[0] != B:
            raise ValueError("context_latent and context_image disagree on the batch size")
        if B == 1:
            out = self.engine().forward(context_latent, context_image)
            return self.package(out, S, H, W, train)
        return self._forward_batched(context_latent, context_image, train)

    def _forward_batched(self, context_latent: torch.Tensor, context_image: torch.Tensor, train: bool):
        """b > 1 (anysplat_stitched.py:174-202 folds (b v) into the token batch; every scene is still reconstructed on its own views): one
        engine forward per scene, then the reference's batch assembly - per-scene voxel lists padded to the largest count with features
        -1e10 / points -1e4 (:440-453: a padded row has density sigmoid(-1e10) = 0, i.e. opacity 0), the Gaussian adapter over the padded
        rows, `scene_scale` taken over the whole batch (:411-412).  The `render_conf` quantile spans the batch as well (:381-387:
        `torch.quantile(depth_conf.flatten(0, 1), t)` has no `dim`): the per-scene forwards leave it to `assemble_batch`."""
        eng = self.engine()
        B, _, S, H, W = context_image.shape
        with self.scenes_of_a_batch():
            outs = [self.keep_scene(eng.forward(context_latent[b:b + 1], context_image[b:b + 1])) for b in range(B)]
        return self.assemble_batch(outs, S, H, W, train)

    def scenes_of_a_batch(self):
        """context manager around the per-scene engine forwards of a b > 1 call: the engine skips its per-scene confidence quantile"""
        import contextlib
        eng = self.engine()

        @contextlib.contextmanager
        def cm():
            prev, eng.batch_conf = eng.batch_conf, True
            try:
                yield
            finally:
                eng.batch_conf = prev
        return cm()

    _KEEP = ("pred_pose_enc_list", "depth", "depth_conf", "pts_all", "raw_gs", "extrinsic_w2c", "intrinsic_px", "neural_pts", "neural_feats")

    @classmethod
    def keep_scene(cls, o: dict) -> dict:
        """copies of what the batch assembly needs from one engine forward (the engine reuses its buffers for the next scene)"""
        return {k: ([t.clone() for t in o[k]] if isinstance(o[k], list) else o[k].clone()) for k in cls._KEEP}

    def assemble_batch(self, outs, S: int, H: int, W: int, train: bool):
        """per-scene engine outputs (keep_scene) -> the reference's batched EncoderOutput (anysplat_stitched.py:411-525)"""
        from .. import ops
        eng = self.engine()
        B = len(outs)
        valid = None
        if eng.cfg.render_conf:
            # anysplat_stitched.py:381-387: ONE threshold = the quantile of every scene's depth confidences together, then each scene's own mask;
            # without the voxel branch the kept rows (boolean-mask order = row-major, :441-446) are the scene's Gaussians.  One device
            # sort + compaction over the concatenated maps (v3a_conf_quantile_compact: exact fp32 rank up to 2^24 pixels per batch).
            M = S * H * W
            gsd = outs[0]["neural_feats"].shape[1]
            conf_all = torch.cat([o["depth_conf"].reshape(M) for o in outs]).contiguous()
            c = ops.conf_quantile_compact(conf_all, eng.cfg.conf_threshold, torch.cat([o["pts_all"].reshape(M, 3) for o in outs]),
                                          torch.cat([o["raw_gs"] for o in outs]), gsd)
            valid = torch.stack([o["depth_conf"].reshape(S, H, W) for o in outs], 0) > c["threshold"]
            if not eng.cfg.voxelize:
                kept = valid.view(B, -1).sum(1).tolist()
                for o, p_, f_ in zip(outs, torch.split(c["pts"], kept), torch.split(c["feat"], kept)):
                    o["neural_pts"], o["neural_feats"] = p_, f_
        U = max(o["neural_feats"].shape[0] for o in outs)
        dev = outs[0]["neural_feats"].device
        gs = []
        for o in outs:
            n, nf = o["neural_feats"].shape
            feats = torch.full((U, nf), -1e10, device=dev, dtype=torch.float32)
            pts = torch.full((U, 3), -1e4, device=dev, dtype=torch.float32)
            feats[:n], pts[:n] = o["neural_feats"], o["neural_pts"]
            gs.append(ops.gaussian_adapter(pts, feats, eng.sh_mask, eng.cfg.sh_degree, eng.cfg.opacity_exponent))
        gauss = Gaussians(**{k: torch.stack([g[k] for g in gs], 0) for k in gs[0]})
        ext = torch.stack([o["extrinsic_w2c"] for o in outs], 0)
        K = torch.stack([o["intrinsic_px"] for o in outs], 0)
        pad = torch.tensor([0, 0, 0, 1.0], device=dev, dtype=ext.dtype).view(1, 1, 1, 4).repeat(B, S, 1, 1)
        Kn = torch.stack([K[:, :, 0] / W, K[:, :, 1] / H, K[:, :, 2]], 2)
        pose = dict(extrinsic=torch.cat([ext, pad], 2).inverse(), intrinsic=Kn)
        depth = torch.stack([o["depth"] for o in outs], 0).view(B, S, H, W, 1)
        dconf = torch.stack([o["depth_conf"] for o in outs], 0).view(B, S, H, W)
        poses = [torch.stack([o["pred_pose_enc_list"][i] for o in outs], 0) for i in range(len(outs[0]["pred_pose_enc_list"]))]
        scale = torch.stack([o["pts_all"].reshape(-1, 3) for o in outs], 0).norm(dim=-1).mean().clip(min=1e-8)
        eo = EncoderOutput(gaussians=gauss, pred_pose_enc_list=poses, pred_context_pose=pose,
                           depth_dict=dict(depth=depth, conf_valid_mask=valid if valid is not None else torch.ones_like(dconf, dtype=torch.bool)),
                           infos=dict(scene_scale=scale, voxelize_ratio=U / (H * W * S)), distill_infos=None,
                           last_pred_pose_enc=None if train else poses[-1])
        if not train:
            return eo
        gsd = self.encoder.raw_gs_dim
        raw = torch.stack([o["raw_gs"] for o in outs], 0).view(B, S, H, W, -1).permute(0, 1, 4, 2, 3)
        return eo, raw[:, :, :gsd], raw[:, :, gsd], dconf
