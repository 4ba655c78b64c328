"""gaussians.ply writer — on-disk format of /root/reference/third_party_model/anysplat/src/model/ply_export.py:26-74
(float32 vertex properties x,y,z,nx,ny,nz,f_dc_0..2,[f_rest_*],opacity,scale_0..2 (log),rot_0..3 (wxyz); binary little endian),
written with numpy only (plyfile is not installed).  SURVEY.md §8f rank 2."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch


def _attributes(num_rest: int):
    a = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(num_rest)]
    return a + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]


def export_ply(means, scales, rotations, harmonics, opacities, path: Path, shift_and_scale: bool = False, save_sh_dc_only: bool = True):
    if shift_and_scale:
        means = means - means.median(dim=0).values
        sf = means.abs().quantile(0.95, dim=0).max()
        means, scales = means / sf, scales / sf
    q = rotations.detach().cpu().double().numpy()
    try:  # same normalisation / representative as the reference (scipy round trip through a rotation matrix)
        from scipy.spatial.transform import Rotation as R
        q = R.from_matrix(R.from_quat(q).as_matrix()).as_quat()
    except ImportError:  # pragma: no cover
        q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    rot = np.stack((q[:, 3], q[:, 0], q[:, 1], q[:, 2]), -1).astype(np.float32)
    f_dc = harmonics[..., 0].detach().cpu().float().numpy()
    f_rest = harmonics[..., 1:].flatten(start_dim=1).detach().cpu().float().numpy()
    m = means.detach().cpu().float().numpy()
    cols = [m, np.zeros_like(m), f_dc] + ([] if save_sh_dc_only else [f_rest]) + [opacities[..., None].detach().cpu().float().numpy(),
                                                                                  scales.log().detach().cpu().float().numpy(), rot]
    data = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    names = _attributes(0 if save_sh_dc_only else f_rest.shape[1])
    assert data.shape[1] == len(names)
    path = Path(path)
    path.parent.mkdir(exist_ok=True, parents=True)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {data.shape[0]}\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(data.tobytes())
