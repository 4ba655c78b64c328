"""Gaussian-splat renderer with the surface of the reference's `DecoderSplattingCUDA`
(/root/reference/third_party_model/anysplat/src/model/decoder/decoder_splatting_cuda.py:28-152), running on the HIP rasteriser
(csrc/raster.hip) instead of gsplat.  `forward(gaussians, extrinsics, intrinsics, near, far, image_shape)` -> DecoderOutput
with color [B,V,3,H,W] in [0,1], depth [B,V,H,W] (alpha-weighted z, not normalised) and alpha [B,V,H,W].

As in the reference: extrinsics are camera-to-world, intrinsics are normalised (row 0 x W, row 1 x H), `near`/`far` are
accepted and ignored (gsplat is called with near_plane=1e-10 and its default far plane), SH degree comes from the harmonics
width, the world covariances are passed explicitly, background = cfg.background_color on RGB only."""
from __future__ import annotations

from dataclasses import dataclass
from math import isqrt
from typing import Optional, Sequence

import torch

from .. import ops
from .types import Gaussians


@dataclass
class DecoderOutput:
    color: torch.Tensor
    depth: Optional[torch.Tensor]
    alpha: Optional[torch.Tensor]
    lod_rendering: Optional[dict] = None


class DecoderSplattingCUDA:
    def __init__(self, background_color: Sequence[float] = (1.0, 1.0, 1.0), make_scale_invariant: bool = False, device="cuda",
                 camera_batch: int = 12):
        self.camera_batch = camera_batch  # cameras per rasteriser launch (scratch = camera_batch * U * 44 B + sort buffers)
        self.make_scale_invariant = make_scale_invariant
        self.device = torch.device(device)
        self.background_color = torch.tensor(list(background_color), dtype=torch.float32, device=self.device)
        self._ws = ops.GsWorkspace()
        self.last_n_isect = []  # per rendered camera batch, for reporting

    def rendering_fn(self, gaussians: Gaussians, extrinsics, intrinsics, near, far, image_shape, depth_mode=None,
                     cam_rot_delta=None, cam_trans_delta=None, cov_ignore: bool = False) -> DecoderOutput:
        if cov_ignore:
            raise NotImplementedError("cov_ignore=True (quaternion/scale path of gsplat) is not on the inference route")
        B, V = intrinsics.shape[:2]
        H, W = image_shape
        dev = self.device
        self.last_n_isect = []
        imgs, depths, alphas = [], [], []
        for i in range(B):
            means = gaussians.means[i].float().contiguous()
            cov = gaussians.covariances[i].float().contiguous()
            sh = gaussians.harmonics[i].float().contiguous()  # [U,3,K]: read in place, no permute copy
            op = gaussians.opacities[i].reshape(-1).float().contiguous()
            sh_degree = isqrt(sh.shape[-1]) - 1
            # tiny per-camera matrices on the host, exactly the reference's operations: c2w -> inverse -> (gsplat) inverse again
            w2c = torch.linalg.inv(extrinsics[i].float().cpu())
            c2w = torch.linalg.inv(w2c)
            K = intrinsics[i].float().cpu().clone()
            K[:, 0] = K[:, 0] * W
            K[:, 1] = K[:, 1] * H
            w2c_d, cam_d, K_d = w2c.contiguous().to(dev), c2w[:, :3, 3].contiguous().to(dev), K.contiguous().to(dev)
            ci, di, ai = [], [], []
            for j0 in range(0, V, self.camera_batch):  # the reference renders one camera per call; same arithmetic, batched
                sl = slice(j0, min(j0 + self.camera_batch, V))
                pr = ops.gs_project(means, cov, sh, w2c_d[sl].contiguous(), cam_d[sl].contiguous(), K_d[sl].contiguous(), W, H,
                                    sh_degree=sh_degree, sh_layout=1, near_plane=1e-10, far_plane=1e10, radius_clip=0.1, eps2d=0.3)
                r = ops.gs_rasterize(pr, op, W, H, background=self.background_color, clamp_rgb=True, workspace=self._ws)
                self.last_n_isect.append(r["n_isect"])
                ci.append(r["color"].permute(0, 3, 1, 2))
                di.append(r["depth"])
                ai.append(r["alpha"])
            imgs.append(torch.cat(ci))
            depths.append(torch.cat(di).squeeze())
            alphas.append(torch.cat(ai).squeeze())
        return DecoderOutput(torch.stack(imgs), torch.stack(depths), torch.stack(alphas), lod_rendering=None)

    def forward(self, gaussians: Gaussians, extrinsics, intrinsics, near, far, image_shape, depth_mode=None, cam_rot_delta=None,
                cam_trans_delta=None, cov_ignore: bool = False) -> DecoderOutput:
        return self.rendering_fn(gaussians, extrinsics, intrinsics, near, far, image_shape, depth_mode, cam_rot_delta,
                                 cam_trans_delta, cov_ignore)

    __call__ = forward
