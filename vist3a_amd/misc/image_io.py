"""Orbit video of a generated scene — `save_interpolated_video` / `save_video` of
/root/reference/third_party_model/anysplat/src/misc/image_io.py:89-228 with the same arguments and the same frames.

Container: the reference writes H.264 `gs.mp4` / `depth.mp4` through imageio-ffmpeg.  `save_video` does the same whenever
imageio(+ffmpeg) or OpenCV is importable; in this image neither exists, and the frames go into a Motion-JPEG AVI with the same
stem (`gs.avi`, `depth.avi`; PIL encodes each frame) — same frame count, order, size and fps."""
from __future__ import annotations

import io
import os
import struct
from pathlib import Path
from typing import Union

import numpy as np
import torch


def interpolate_camera_path(pred_extrinsics: torch.Tensor, pred_intrinsics: torch.Tensor, b: int, t: int = 10):
    """image_io.py:124-186: per neighbouring pair the first view + t in-betweens (translation / intrinsics linear, rotation =
    linear blend projected back to SO(3) by SVD).  The reference appends the last view only AFTER concatenating, so it is never
    rendered: (V-1)(t+1) cameras, kept."""
    ex, ix = [], []
    for i in range(pred_extrinsics.shape[1] - 1):
        ex.append(pred_extrinsics[:, i:i + 1])
        ix.append(pred_intrinsics[:, i:i + 1])
        s, e = pred_extrinsics[:, i], pred_extrinsics[:, i + 1]
        for j in range(1, t + 1):
            al = j / (t + 1)
            tr = (1 - al) * s[:, :3, 3] + al * e[:, :3, 3]
            rot = ((1 - al) * s[:, :3, :3].reshape(b, 9) + al * e[:, :3, :3].reshape(b, 9)).reshape(b, 3, 3)
            u, _, vh = torch.linalg.svd(rot.cpu())
            rot = torch.bmm(u, vh).to(rot.device)
            m = torch.eye(4, device=pred_extrinsics.device, dtype=pred_extrinsics.dtype).unsqueeze(0).repeat(b, 1, 1)
            m[:, :3, :3] = rot
            m[:, :3, 3] = tr
            ex.append(m.unsqueeze(1))
            ix.append(((1 - al) * pred_intrinsics[:, i] + al * pred_intrinsics[:, i + 1]).unsqueeze(1))
    return torch.cat(ex, dim=1), torch.cat(ix, dim=1)


def _h264_writer():
    """An H.264/mp4 writer if one is importable (what the reference uses: imageio + imageio-ffmpeg, src/misc/image_io.py:217-228;
    else OpenCV); None in this image, where neither exists."""
    try:
        import imageio.v2 as imageio  # noqa: F401
        import imageio_ffmpeg  # noqa: F401
        return "imageio"
    except Exception:  # noqa: BLE001
        pass
    try:
        import cv2  # noqa: F401
        return "cv2"
    except Exception:  # noqa: BLE001
        return None


def save_video(video: Union[torch.Tensor, np.ndarray], save_path: Union[str, Path], fps: int = 20, quality: int = 92) -> str:
    """video [T,3,H,W] in [0,1] (or uint8 [T,H,W,3]) -> `<save_path>` as H.264 mp4 when an encoder is importable (the reference's
    container, consumed by evaluation/gen_eval/utils.py:43-53), otherwise the same frames as a Motion-JPEG AVI next to it
    (`.avi` suffix).  Returns the path actually written."""
    if isinstance(video, torch.Tensor):
        video = (video.detach().float().clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy() * 255).astype(np.uint8)
    T, H, W, _ = video.shape
    save_path = Path(save_path)
    enc = _h264_writer()
    if enc == "imageio":
        import imageio.v2 as imageio
        with imageio.get_writer(str(save_path.with_suffix(".mp4")), fps=fps, codec="libx264", macro_block_size=1) as wr:
            for f in video:
                wr.append_data(f)
        return str(save_path.with_suffix(".mp4"))
    if enc == "cv2":
        import cv2
        wr = cv2.VideoWriter(str(save_path.with_suffix(".mp4")), cv2.VideoWriter_fourcc(*"mp4v"), fps, (W, H))
        for f in video:
            wr.write(f[..., ::-1].copy())
        wr.release()
        return str(save_path.with_suffix(".mp4"))
    from PIL import Image
    frames = []
    for f in video:
        buf = io.BytesIO()
        Image.fromarray(f).save(buf, format="JPEG", quality=quality)
        frames.append(buf.getvalue())
    _write_mjpeg_avi(save_path.with_suffix(".avi"), frames, W, H, fps)
    return str(save_path.with_suffix(".avi"))


def _chunk(tag: bytes, data: bytes) -> bytes:
    return tag + struct.pack("<I", len(data)) + data + (b"\x00" if len(data) & 1 else b"")


def _write_mjpeg_avi(path: Path, frames, W: int, H: int, fps: int) -> None:
    n = len(frames)
    maxb = max((len(f) for f in frames), default=0)
    avih = struct.pack("<IIIIIIIIIIIIII", 1000000 // fps, maxb * fps, 0, 0x10, n, 0, 1, maxb, W, H, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIIIhhhh", 0, 0, 0, 0, 1, fps, 0, n, maxb, 0xFFFFFFFF, 0, 0, 0, W, H)
    strf = struct.pack("<IiiHHIIiiII", 40, W, H, 1, 24, 0x47504A4D, W * H * 3, 0, 0, 0, 0)
    strl = b"strl" + _chunk(b"strh", strh) + _chunk(b"strf", strf)
    hdrl = b"hdrl" + _chunk(b"avih", avih) + _chunk(b"LIST", strl)
    movi, idx, off = b"movi", b"", 4
    for f in frames:
        c = _chunk(b"00dc", f)
        idx += b"00dc" + struct.pack("<III", 0x10, off, len(f))
        movi += c
        off += len(c)
    body = b"AVI " + _chunk(b"LIST", hdrl) + _chunk(b"LIST", movi) + _chunk(b"idx1", idx)
    path.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)


def save_interpolated_video(pred_extrinsics, pred_intrinsics, b, h, w, gaussians, save_path, decoder_func, t=10, name=None,
                            cov_ignore=False):
    """image_io.py:111-228.  Renders the interpolated orbit with `decoder_func.forward` and writes the colour and the
    turbo-mapped depth video.  -> (path of the colour video, path of the depth video)"""
    import matplotlib
    ex, ix = interpolate_camera_path(pred_extrinsics, pred_intrinsics, b, t)
    num_frames = ex.shape[1]
    dev = ex.device
    out = decoder_func.forward(gaussians, ex, ix.float(), torch.ones(1, num_frames, device=dev) * 0.1,
                               torch.ones(1, num_frames, device=dev) * 100, (h, w), cov_ignore=cov_ignore)
    video = out.color[0].clip(min=0, max=1)
    depth = out.depth[0]
    num_views = pred_extrinsics.shape[1]
    sub = depth[::num_views]
    lo, hi = sub.quantile(0.01), sub.quantile(0.99)
    dn = ((depth - lo) / (hi - lo)).cpu().numpy()
    col = matplotlib.colormaps["turbo"](dn)[..., :3]
    depth_colored = torch.from_numpy(col).permute(0, 3, 1, 2).clip(min=0, max=1)
    stem_c, stem_d = ("gs", "depth") if name is None else ("gs_optimized", "gs_optimized_depth")
    pd = save_video(depth_colored, os.path.join(save_path, f"{stem_d}.mp4"), fps=20)
    pc = save_video(video, os.path.join(save_path, f"{stem_c}.mp4"), fps=20)
    return pc, pd
