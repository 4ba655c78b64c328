"""ORACLE (test infrastructure).  CPU fp32 restatement of the 3D-Gaussian rasterisation the reference calls right after
the text->3DGS path:

  /root/reference/third_party_model/anysplat/src/model/decoder/decoder_splatting_cuda.py:43-125
      -> `gsplat.rasterization(means, quats, scales, opacities, sh, w2c[j:j+1], K[j:j+1], W, H, sh_degree=4,
          render_mode="RGB+D", packed=False, near_plane=1e-10, backgrounds=white, radius_clip=0.1, covars=covariances,
          rasterize_mode="classic")`, one camera per call, colours clamped to [0,1] by the caller
  /root/reference/third_party_model/anysplat/src/misc/image_io.py:111-228  (camera path interpolation + video)

PARITY UNPINNED: the algorithm lives in the third-party wheel `gsplat==1.4.0` (/root/reference/requirements.txt:17), which is
absent from /root/reference and from this image, and the reference holds no test or golden vector for it.  This file
restates gsplat 1.4.0's published pipeline (fully_fused_projection -> spherical_harmonics -> isect_tiles -> stable radix
sort by (tile, depth bits) -> isect_offset_encode -> rasterize_to_pixels) and is guarded by analytic known-answer tests in
tests/test_oracle_raster.py (SH basis orthonormality, single-Gaussian closed forms, front-to-back compositing algebra,
tile-binning invariants).  Constants kept from gsplat 1.4.0: eps2d=0.3, radius = ceil(3*sqrt(lambda_max)) with the
discriminant floored at 0.01, tile 16x16, alpha = min(0.999, o*exp(-sigma)), skip alpha < 1/255, stop when the next
transmittance <= 1e-4, far_plane=1e10, depth channel = sum(vis_i * z_i) (mode "D", not normalised), background only on RGB.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

TILE = 16
f32 = torch.float32


# ----------------------------------------------------------------------------------------------- spherical harmonics
def sh_basis(dirs: torch.Tensor, degree: int) -> torch.Tensor:
    """Real SH basis values [N, (degree+1)^2] for unit directions, gsplat's `sh_coeffs_to_color_fast` polynomials
    (3DGS sign convention: Y1 = C1*(-y, z, -x))."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    out = [torch.full_like(x, 0.2820947917738781)]
    if degree >= 1:
        c1 = 0.48860251190292
        out += [-c1 * y, c1 * z, -c1 * x]
    if degree >= 2:
        z2 = z * z
        fTmp0B = -1.092548430592079 * z
        fC1 = x * x - y * y
        fS1 = 2.0 * x * y
        p6 = 0.9461746957575601 * z2 - 0.3153915652525201
        out += [0.5462742152960395 * fS1, fTmp0B * y, p6, fTmp0B * x, 0.5462742152960395 * fC1]
    if degree >= 3:
        fTmp0C = -2.285228997322329 * z2 + 0.4570457994644658
        fTmp1B = 1.445305721320277 * z
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        p12 = z * (1.865881662950577 * z2 - 1.119528997770346)
        out += [-0.5900435899266435 * fS2, fTmp1B * fS1, fTmp0C * y, p12, fTmp0C * x, fTmp1B * fC1, -0.5900435899266435 * fC2]
    if degree >= 4:
        fTmp0D = z * (-4.683325804901025 * z2 + 2.007139630671868)
        fTmp1C = 3.31161143515146 * z2 - 0.47308734787878
        fTmp2B = -1.770130769779931 * z
        fC3 = x * fC2 - y * fS2
        fS3 = x * fS2 + y * fC2
        p20 = 1.984313483298443 * z * p12 - 1.006230589874905 * p6
        out += [0.6258357354491763 * fS3, fTmp2B * fS2, fTmp1C * fS1, fTmp0D * y, p20, fTmp0D * x, fTmp1C * fC1,
                fTmp2B * fC2, 0.6258357354491763 * fC3]
    return torch.stack(out, dim=-1)


def sh_colors(means: torch.Tensor, campos: torch.Tensor, sh: torch.Tensor, degree: int) -> torch.Tensor:
    """rasterization(): dirs = means - camtoworld[:3,3]; colours = clamp_min(SH(dirs/|dirs|) + 0.5, 0).  sh: [U, K, 3]."""
    d = means - campos[None]
    d = d / d.norm(dim=-1, keepdim=True).clamp_min(1e-20)
    B = sh_basis(d, degree)  # [U, (deg+1)^2]
    nb = B.shape[1]
    col = torch.einsum("uk,ukc->uc", B, sh[:, :nb])
    return (col + 0.5).clamp_min(0.0)


# ----------------------------------------------------------------------------------------------- projection
def project(means: torch.Tensor, covars: torch.Tensor, viewmat: torch.Tensor, K: torch.Tensor, width: int, height: int,
            near_plane: float = 1e-10, far_plane: float = 1e10, radius_clip: float = 0.1, eps2d: float = 0.3) -> Dict[str, torch.Tensor]:
    """fully_fused_projection forward for one pinhole camera.  covars [U,3,3] world-space (symmetric; the upper triangle is
    what gsplat keeps).  Returns radii (int32, 0 = culled), means2d, depths, conics (a, b, c of the inverse 2D covariance)."""
    means, covars, viewmat, K = means.to(f32), covars.to(f32), viewmat.to(f32), K.to(f32)
    U = means.shape[0]
    R, t = viewmat[:3, :3], viewmat[:3, 3]
    mc = means @ R.T + t
    # upper triangle -> symmetric matrix (gsplat drops the lower half)
    iu = torch.triu_indices(3, 3)
    cs = torch.zeros(U, 3, 3, dtype=f32)
    cs[:, iu[0], iu[1]] = covars[:, iu[0], iu[1]]
    cs = cs + cs.transpose(1, 2) - torch.diag_embed(torch.diagonal(cs, dim1=1, dim2=2))
    cc = R[None] @ cs @ R.T[None]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x, y, z = mc[:, 0], mc[:, 1], mc[:, 2]
    in_z = (z >= near_plane) & (z <= far_plane)
    zs = torch.where(in_z, z, torch.ones_like(z))
    tan_fovx, tan_fovy = 0.5 * width / fx, 0.5 * height / fy
    lim_xp, lim_xn = (width - cx) / fx + 0.3 * tan_fovx, cx / fx + 0.3 * tan_fovx
    lim_yp, lim_yn = (height - cy) / fy + 0.3 * tan_fovy, cy / fy + 0.3 * tan_fovy
    rz = 1.0 / zs
    rz2 = rz * rz
    tx = zs * torch.minimum(lim_xp, torch.maximum(-lim_xn, x * rz))
    ty = zs * torch.minimum(lim_yp, torch.maximum(-lim_yn, y * rz))
    J = torch.zeros(U, 2, 3, dtype=f32)
    J[:, 0, 0] = fx * rz
    J[:, 0, 2] = -fx * tx * rz2
    J[:, 1, 1] = fy * rz
    J[:, 1, 2] = -fy * ty * rz2
    c2 = J @ cc @ J.transpose(1, 2)
    m2 = torch.stack([fx * x * rz + cx, fy * y * rz + cy], -1)
    a, b, c = c2[:, 0, 0] + eps2d, c2[:, 0, 1], c2[:, 1, 1] + eps2d
    det = a * c - b * b
    ok = in_z & (det > 0)
    dets = torch.where(ok, det, torch.ones_like(det))
    conics = torch.stack([c / dets, -b / dets, a / dets], -1)
    mid = 0.5 * (a + c)
    v1 = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.01))
    radius = torch.ceil(3.0 * torch.sqrt(v1))
    ok = ok & (radius > radius_clip)
    ok = ok & ~((m2[:, 0] + radius <= 0) | (m2[:, 0] - radius >= width) | (m2[:, 1] + radius <= 0) | (m2[:, 1] - radius >= height))
    radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)
    return dict(radii=radii, means2d=m2, depths=z, conics=conics)


# ----------------------------------------------------------------------------------------------- tile binning
def tile_bounds(means2d: torch.Tensor, radii: torch.Tensor, width: int, height: int) -> Tuple[torch.Tensor, torch.Tensor, int, int]:
    """isect_tiles: [tile_min, tile_max) per Gaussian, clamped to the tile grid (negative float->uint saturates to 0)."""
    tw, th = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    r = radii.to(f32) / TILE
    tm = means2d.to(f32) / TILE
    lim = torch.tensor([tw, th], dtype=f32)
    tmin = torch.minimum(torch.floor(tm - r[:, None]).clamp_min(0), lim).to(torch.int64)
    tmax = torch.minimum(torch.ceil(tm + r[:, None]).clamp_min(0), lim).to(torch.int64)
    vis = radii > 0
    tmin, tmax = tmin * vis[:, None], tmax * vis[:, None]
    return tmin, tmax, tw, th


def bin_and_sort(means2d, radii, depths, width: int, height: int):
    """-> (tile_offsets [ntiles+1] int64, flatten_ids [n_isect] int64): per tile the Gaussian indices in ascending order of
    the key (tile_id << 32 | float bits of depth); ties keep emission order = ascending Gaussian index (stable radix sort)."""
    tmin, tmax, tw, th = tile_bounds(means2d, radii, width, height)
    keys, ids = [], []
    dbits = depths.to(f32).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    for g in torch.nonzero(radii > 0).flatten().tolist():
        for i in range(int(tmin[g, 1]), int(tmax[g, 1])):
            for j in range(int(tmin[g, 0]), int(tmax[g, 0])):
                keys.append(((i * tw + j) << 32) | int(dbits[g]))
                ids.append(g)
    keys = np.asarray(keys, dtype=np.uint64)
    ids = np.asarray(ids, dtype=np.int64)
    order = np.argsort(keys, kind="stable")
    keys, ids = keys[order], ids[order]
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    offs = np.searchsorted(tiles, np.arange(tw * th + 1), side="left")
    return torch.from_numpy(offs.astype(np.int64)), torch.from_numpy(ids)


# ----------------------------------------------------------------------------------------------- compositing
def rasterize(means2d, conics, colors, opacities, width: int, height: int, tile_offsets, flatten_ids,
              background: Optional[torch.Tensor] = None):
    """rasterize_to_pixels forward.  colors [U, C] (C = 4 for RGB+D); background [C] or None.  -> (image [H,W,C], alpha [H,W])."""
    Cc = colors.shape[1]
    img = torch.zeros(height, width, Cc, dtype=f32)
    alpha_img = torch.zeros(height, width, dtype=f32)
    tw, th = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    means2d, conics, colors, opacities = means2d.to(f32), conics.to(f32), colors.to(f32), opacities.to(f32)
    for ti in range(th):
        for tj in range(tw):
            t = ti * tw + tj
            g = flatten_ids[int(tile_offsets[t]):int(tile_offsets[t + 1])]
            y0, x0 = ti * TILE, tj * TILE
            y1, x1 = min(y0 + TILE, height), min(x0 + TILE, width)
            py, px = torch.meshgrid(torch.arange(y0, y1, dtype=f32) + 0.5, torch.arange(x0, x1, dtype=f32) + 0.5, indexing="ij")
            P = py.numel()
            T = torch.ones(P, dtype=f32)
            acc = torch.zeros(P, Cc, dtype=f32)
            live = torch.ones(P, dtype=torch.bool)
            pxf, pyf = px.reshape(-1), py.reshape(-1)
            for gi in g.tolist():
                dx, dy = means2d[gi, 0] - pxf, means2d[gi, 1] - pyf
                sigma = 0.5 * (conics[gi, 0] * dx * dx + conics[gi, 2] * dy * dy) + conics[gi, 1] * dx * dy
                a = torch.clamp_max(opacities[gi] * torch.exp(-sigma), 0.999)
                use = live & (sigma >= 0) & (a >= 1.0 / 255.0)
                nT = T * (1.0 - a)
                stop = use & (nT <= 1e-4)
                live = live & ~stop
                use = use & ~stop
                vis = torch.where(use, a * T, torch.zeros_like(T))
                acc = acc + vis[:, None] * colors[gi][None]
                T = torch.where(use, nT, T)
                if not live.any():
                    break
            if background is not None:
                acc = acc + T[:, None] * background.to(f32)[None]
            img[y0:y1, x0:x1] = acc.view(y1 - y0, x1 - x0, Cc)
            alpha_img[y0:y1, x0:x1] = (1.0 - T).view(y1 - y0, x1 - x0)
    return img, alpha_img


def rasterization(means, covars, opacities, sh, viewmat, K, width: int, height: int, sh_degree: int = 4,
                  near_plane: float = 1e-10, far_plane: float = 1e10, radius_clip: float = 0.1, eps2d: float = 0.3,
                  background: Optional[torch.Tensor] = None):
    """One camera of gsplat.rasterization(render_mode="RGB+D", packed=False, rasterize_mode="classic", covars=...).
    sh [U, K, 3].  -> (render [H,W,4] = RGB (un-clamped) + accumulated depth, alpha [H,W], meta)"""
    pr = project(means, covars, viewmat, K, width, height, near_plane, far_plane, radius_clip, eps2d)
    campos = torch.linalg.inv(viewmat.to(f32))[:3, 3]
    col = sh_colors(means.to(f32), campos, sh.to(f32), sh_degree)
    col = torch.cat([col, pr["depths"][:, None]], -1)
    offs, ids = bin_and_sort(pr["means2d"], pr["radii"], pr["depths"], width, height)
    bg = None if background is None else torch.cat([background.to(f32), torch.zeros(1)])
    img, alpha = rasterize(pr["means2d"], pr["conics"], col, opacities, width, height, offs, ids, bg)
    return img, alpha, dict(pr, colors=col, tile_offsets=offs, flatten_ids=ids)


# ----------------------------------------------------------------------------------------------- camera path
def interpolate_camera_path(extrinsics: torch.Tensor, intrinsics: torch.Tensor, t: int = 10):
    """image_io.py:124-186: between neighbouring views, t extra poses — translation/intrinsics linear, rotation = linear blend
    re-orthonormalised by SVD (U V^T).  extrinsics [B,V,4,4], intrinsics [B,V,3,3] -> [B,(V-1)(t+1),...].  The reference's
    trailing "add the last frame" runs after the concatenation and so never reaches the renderer; kept that way."""
    b, V = extrinsics.shape[:2]
    ex, ix = [], []
    for i in range(V - 1):
        ex.append(extrinsics[:, i:i + 1])
        ix.append(intrinsics[:, i:i + 1])
        for j in range(1, t + 1):
            al = j / (t + 1)
            s, e = extrinsics[:, i], extrinsics[:, i + 1]
            tr = (1 - al) * s[:, :3, 3] + al * e[:, :3, 3]
            rot = ((1 - al) * s[:, :3, :3].reshape(b, 9) + al * e[:, :3, :3].reshape(b, 9)).reshape(b, 3, 3)
            u, _, v = torch.svd(rot)
            rot = torch.bmm(u, v.transpose(1, 2))
            m = torch.eye(4, dtype=extrinsics.dtype).unsqueeze(0).repeat(b, 1, 1)
            m[:, :3, :3] = rot
            m[:, :3, 3] = tr
            ex.append(m.unsqueeze(1))
            ix.append(((1 - al) * intrinsics[:, i] + al * intrinsics[:, i + 1]).unsqueeze(1))
    return torch.cat(ex, 1), torch.cat(ix, 1)
