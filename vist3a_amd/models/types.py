"""Boundary return types: `Gaussians` (/root/reference/third_party_model/anysplat/src/model/types.py:7-14) and
`EncoderOutput` (.../src/model/encoder/encoder.py:16-23)."""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch


@dataclass
class Gaussians:
    means: torch.Tensor        # [B, U, 3]
    covariances: torch.Tensor  # [B, U, 3, 3]
    harmonics: torch.Tensor    # [B, U, 3, d_sh]
    opacities: torch.Tensor    # [B, U]
    scales: torch.Tensor       # [B, U, 3]
    rotations: torch.Tensor    # [B, U, 4]  (xyzw)


@dataclass
class EncoderOutput:
    gaussians: Gaussians
    pred_pose_enc_list: Optional[List[torch.Tensor]] = None
    pred_context_pose: Optional[Dict[str, torch.Tensor]] = None
    depth_dict: Optional[Dict[str, torch.Tensor]] = None
    infos: Optional[Dict[str, Any]] = None
    distill_infos: Optional[Dict[str, Any]] = None
    last_pred_pose_enc: Optional[torch.Tensor] = None
