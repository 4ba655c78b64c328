"""`StitchVAE3D` — drop-in for /root/reference/models/stitched_model.py:12-182 on the inference path.

Same constructor signature and attributes (`.device`, `.diffusion_vae`, `.stitching_layer`, `.stitched_3d_model`,
`.pre_upsample_layer`, `.vae_latent_dimension`, `.feedforward_dimension`); `forward_with_latent(latent, feedforward_image,
train=False)` is the entry /root/reference/inference_t23d.py:133-137 calls.  On MI355X the three stages the reference chains
through NCTHW tensors (trilinear T-upsample -> Conv3d -> token rearrange) collapse into: one lerp kernel that emits the
channels-last bf16 clip, and one implicit-GEMM conv whose epilogue adds the positional embedding and scatters rows
straight into the reconstruction engine's token buffer."""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops
from .anysplat_stitched import AnySplatStitched, AnySplatWeights
from .stitching_layer_builder import ConvSpec, StitchingConv


class StitchVAE3D(torch.nn.Module):
    def __init__(self, diffusion_vae, feedforward_model, device, stitching_layer_location: str, stitching_layer_config: ConvSpec,
                 resolution: int, stitching_layer_init_path: Optional[str] = None):
        super().__init__()
        self.device = torch.device(device)
        self.diffusion_vae = diffusion_vae
        # WAN VAE: C=16 latents, 8x spatial stride; T is nominal (stitched_model.py:49-64: only H, W are used)
        self.vae_latent_dimension = torch.tensor([13, 16, resolution // 8, resolution // 8])
        if not isinstance(feedforward_model, AnySplatWeights):
            raise NotImplementedError(f"Feedforward model preprocessing not implemented for {type(feedforward_model)}")
        self.stitched_3d_model = AnySplatStitched(feedforward_model, stitching_layer_location, device=self.device)
        self.feedforward_dimension = torch.tensor([13, feedforward_model.cfg.C, 448, 448])
        if not isinstance(stitching_layer_config, ConvSpec):
            raise TypeError("stitching_layer_config must be a ConvSpec (parse_conv_spec)")
        self.stitching_layer: StitchingConv = stitching_layer_config.build(in_channels=int(self.vae_latent_dimension[1]))
        if stitching_layer_init_path is not None:
            sd = torch.load(stitching_layer_init_path, map_location="cpu", weights_only=False)
            sd = sd.get("state_dict", sd)
            self.stitching_layer.load_state_dict(sd, strict=True)
        self._packed = None
        self._packed_key = None

    def pre_upsample_layer(self, x: torch.Tensor) -> torch.Tensor:
        """[1,16,Tl,h,w] -> [1,16,4(Tl-1)+1,h,w]: trilinear align_corners=True, H/W unchanged (stitched_model.py:92-107)."""
        cl = ops.latent_upsample_t_cl(x[0].to(device=self.device, dtype=torch.float32).contiguous())
        return cl.permute(3, 0, 1, 2).unsqueeze(0).float()

    def _conv_weight(self) -> ops.ConvWeight:
        w, b = self.stitching_layer.weight, self.stitching_layer.bias
        key = (w._version, w.data_ptr(), None if b is None else (b._version, b.data_ptr()))
        if self._packed is None or key != self._packed_key:
            st = self.stitching_layer
            wd = st.dense_weight() if hasattr(st, "dense_weight") else w.detach()      # (grouped layers: their block-diagonal dense form)
            wd = wd.reshape(wd.shape[0], wd.shape[1], *((1,) * (5 - wd.dim())), *wd.shape[2:]) if wd.dim() < 5 else wd
            self._packed, self._packed_key = ops.ConvWeight(wd, None if b is None else b.detach(), device=self.device,
                                                            dilation=getattr(st, "dilation3", (1, 1, 1))), key
        return self._packed

    @torch.no_grad()
    def vae_encoder_forward(self, images: torch.Tensor, decode: bool = False, generator: Optional[torch.Generator] = None):
        """stitched_model.py:122-137: latents = vae.encode(images).latent_dist.sample() (images [1,3,T,H,W] in [-1,1])."""
        latents = self.diffusion_vae.encode(images).latent_dist.sample(generator)
        feedforward_image = self.diffusion_vae.decode(latents)[0] if decode else None
        return latents, feedforward_image

    @torch.no_grad()
    def forward(self, images, feedforward_image, train=False, generator: Optional[torch.Generator] = None):
        """Image-conditioned entry (stitched_model.py:139-163; the NVS evaluation path): encode the views with the Wan VAE, sample
        the posterior (`generator` makes the draw reproducible; the reference uses the global CUDA RNG), then exactly
        `forward_with_latent`.  The reference never decodes here (decode=False), so `feedforward_image` is used as given."""
        latent, _ = self.vae_encoder_forward(images, decode=False, generator=generator)
        return self.forward_with_latent(latent, feedforward_image, train=train)

    @torch.no_grad()
    def forward_with_latent(self, latent: torch.Tensor, feedforward_image: torch.Tensor, train: bool = False, image_cl: Optional[torch.Tensor] = None,
                            recon_group=None):
        """latent: de-normalised VAE latent [B,16,Tl,64,64]; feedforward_image [B,3,T,448,448] in [-1,1]
        (or, MI355X fast path, image_cl [T,448,448,8] in [-1,1] as produced by WanVAEDecoder.decode_cl + resize; [B,T,448,448,8] for B > 1).
        recon_group (default: the attribute `self.recon_group`, None = this GPU alone): the ranks of a scene-parallel run (seqpar DistGroup /
        ThreadGroup) - every rank calls with the SAME inputs, the reconstruction is split by views (ReconEngine.forward_sharded) and every rank
        gets the whole scene back."""
        grp = recon_group if recon_group is not None else getattr(self, "recon_group", None)
        if latent.shape[0] != 1:
            # b > 1 (stitched_model.py:165-173 is batch-agnostic): every scene goes through exactly the b = 1 path below - stitching conv
            # writing tokens in place, one engine forward - and the reference's batch assembly follows (AnySplatStitched.assemble_batch)
            B = latent.shape[0]
            if image_cl is not None and (image_cl.dim() != 5 or image_cl.shape[0] != B):
                raise ValueError("image_cl must be [B,T,H,W,8] for a batched latent")
            if image_cl is None and feedforward_image.shape[0] != B:
                raise ValueError("latent and feedforward_image disagree on the batch size")
            model = self.stitched_3d_model
            outs, shp = [], None
            with model.scenes_of_a_batch():     # (the render_conf quantile spans the batch: taken in assemble_batch)
                for b in range(B):
                    o, shp = self._scene(latent[b:b + 1], None if feedforward_image is None else feedforward_image[b:b + 1],
                                         None if image_cl is None else image_cl[b], grp)
                    outs.append(model.keep_scene(o))
            return model.assemble_batch(outs, *shp, train)
        out, (S, H, W) = self._scene(latent, feedforward_image, image_cl, grp)
        return self.stitched_3d_model.package(out, S, H, W, train)

    def _scene(self, latent: torch.Tensor, feedforward_image: Optional[torch.Tensor], image_cl: Optional[torch.Tensor], grp=None):
        """one scene: T-upsample -> stitching conv into the token workspace -> reconstruction engine; -> (raw engine outputs, (S, H, W))"""
        st = self.stitching_layer
        eng = self.stitched_3d_model.engine()
        lat_cl = ops.latent_upsample_t_cl(latent[0].to(device=self.device, dtype=torch.float32).contiguous())
        S = lat_cl.shape[0]
        if image_cl is None:
            H, W = feedforward_image.shape[-2:]
            image_cl = torch.zeros(S, H, W, 8, device=self.device, dtype=torch.float32)
            image_cl[..., :3] = feedforward_image[0].to(self.device).float().permute(1, 2, 3, 0)
        else:
            H, W = image_cl.shape[1:3]
        # context_image = (context_image + 1) / 2 in fp32 (anysplat_stitched.py:174-175), whatever dtype the image arrived in
        img01 = (image_cl.float() + 1) / 2
        img01[..., 3:] = 0
        sharded = grp is not None and grp.world > 1
        # (the view-sharded path lays the scene's tokens out in a private buffer and never touches the S-view workspace: constants only)
        x, g = (None, eng.geometry_constants(S, H, W)) if sharded else eng.token_workspace(S, H, W)
        hw, Pp, nsp = g["hw"], g["Pp"], g["nsp"]
        cw = self._conv_weight()
        oshape = [(lat_cl.shape[i] + 2 * st.padding3[i] - cw.k_eff[i]) // st.stride3[i] + 1 for i in range(3)]
        if oshape[0] != S or oshape[1] * oshape[2] != hw or cw.CoutP != eng.cfg.C:
            raise ValueError(f"stitching layer output {oshape}x{cw.Cout} does not match the {S}x{g['hp']}x{g['wp']}x{eng.cfg.C} token grid")
        if grp is not None and grp.world > 1:
            # one scene over several ranks (scene-parallel latency mode): the stitching convolution (2e10 FLOP) runs on every rank into a
            # private token buffer, the reconstruction is split by views (ReconEngine.forward_sharded)
            import threading
            x = eng.token_buffer(S, H, W, threading.get_ident())
        ops.conv(lat_cl, cw, out=x, stride=st.stride3, pad=st.padding3, out_size=tuple(oshape), replicate=True,
                 residual=g["pos_patch"], res_row_mod=hw, out_rows=(hw, Pp - hw, nsp))
        if grp is not None and grp.world > 1:
            times = {}
            out = eng.forward_sharded(S, H, W, x, img01.contiguous(), grp, timings=times)
            self.recon_shard_times = times      # (last scene, this rank)
            return out, (S, H, W)
        return eng.forward_tokens_filled(S, H, W, img01.contiguous()), (S, H, W)
