"""CLI contract of the text->3DGS driver: every flag of `inference_vist3a_argument()` in /root/reference/utils/argument.py:392-443
(built there from add_model_selection_args :58-78, add_stitching_args :234-270, add_common_data_args :140-160) with the same
names, types and defaults, plus MI355X-specific extras that default to the reference's behaviour."""
from __future__ import annotations

import argparse

from ..models.stitching_layer_builder import parse_conv_spec


def inference_vist3a_argument() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Inference on VIST3A argument", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    g = p.add_argument_group("Model selection")
    g.add_argument("--feedforward_model", type=str, default="anysplat", choices=["anysplat"], help="Feedforward model to use")
    g.add_argument("--video_model", type=str, default="wan", choices=["wan"], help="Video model to use")
    g = p.add_argument_group("Stitching")
    g.add_argument("--stitching_layer_location", type=str, default="enc_blocks_2", help="Location of the stitching layer in the feedforward model")
    g.add_argument("--initialization_weight_path", type=str, default=None, help="Path to the initialization weight for the stitching layer")
    g.add_argument("--stitching_layer_config", type=parse_conv_spec, default="conv3d_k5x3x3_o1024_s1x2x2_p2x1x1", metavar="CONV_SPEC")
    g.add_argument("--lora_config", type=str, default="r8,a16,d0.05,f0", help="r<rank>,a<alpha>,d<dropout>,b<bias>,t<targets>,f<0/1>")
    g = p.add_argument_group("Data (common)")
    g.add_argument("--resolution", type=int, default=512, help="Image resolution")
    g.add_argument("--feedforward_resolution", type=int, default=448, help="Image resolution for feedforward model")
    g = p.add_argument_group("Inference")
    g.add_argument("--model_id", default="Wan-AI/Wan2.1-T2V-1.3B-Diffusers", type=str)
    g.add_argument("--checkpoint_path", type=str, required=True, help="Path to the trained stitching model ('synthetic' = seeded random weights)")
    g.add_argument("--transformer_lora_path", type=str, required=True, help="Path to the LoRA weights for the transformer ('none' to skip)")
    g.add_argument("--input_texts_path", type=str, required=True, help="Path to input texts for inference")
    g.add_argument("--output_dir", type=str, default="inference_vist3a_results", help="Path to save inference results")
    g.add_argument("--num_frames", type=int, default=13, help="Number of frames to generate for each input text")
    g.add_argument("--flow_shift", type=float, default=5, help="Flow shift value for timesteps")
    g.add_argument("--cfg_scale", type=str, default="7.5", help="Classifier-free guidance scale(s)")
    g = p.add_argument_group("MI355X extras (not in the reference)")
    g.add_argument("--num_inference_steps", type=int, default=50, help="denoise steps (the reference hard-codes 50)")
    g.add_argument("--anysplat_weights", type=str, default=None, help="local AnySplat .safetensors (no HF hub access offline)")
    g.add_argument("--text_embeds_path", type=str, default=None,
                   help="torch file {prompt: [512,4096] embedding, '__negative__': ...}; UMT5-XXL itself is outside this path (SURVEY §8f)")
    g.add_argument("--synthetic_text", action="store_true", help="seeded synthetic text embeddings (weights-free smoke runs)")
    g.add_argument("--scene_parallel", action="store_true",
                   help="all ranks cooperate on each prompt (CFG-parallel x sequence-parallel DiT over RCCL) instead of striding prompts")
    g.add_argument("--no_video", action="store_true", help="skip the interpolated orbit render (gs.avi / depth.avi)")
    g.add_argument("--overwrite", action="store_true", help="reuse an existing output directory (the reference raises)")
    return p


def parse_lora_mode(spec: str):
    """'r64,a32,d0.0,f0' -> (r, alpha): the two numbers the merged-at-load LoRA needs.  Grammar and defaults (r=8, alpha=32) of
    /root/reference/utils/lora_util/utils.py:57-117 (`LoraConfig`, `parse_lora_mode`); pinned by tests/golden/lora_tiny.safetensors."""
    import re
    r, alpha = 8, 32
    pat = re.compile(r"(?P<key>[radbf t])(?:(?P<num>[\d.]+)|(?P<str>[^,]+))")
    for chunk in spec.split(","):
        c = chunk.strip().lower()
        if c in ("enc", "fix_head", "fixhead"):
            continue
        m = pat.fullmatch(c)
        if not m:
            raise ValueError(f"Bad LoRA chunk: {c!r}")
        k = m["key"]
        if k in "radf" and m["num"] is None:
            raise ValueError(f"Bad LoRA chunk: {c!r}")
        if k == "r":
            r = int(m["num"])
        elif k == "a":
            alpha = int(m["num"])
        elif k == "d":
            float(m["num"])
        elif k == "b":
            if m["str"] not in {"none", "all", "lora_only"}:
                raise ValueError("b chunk must be none|all|lora_only")
        elif k == "f":
            if bool(int(m["num"])):
                raise NotImplementedError("fan_in_fan_out LoRA layers (f1) are not supported by the merged-at-load path")
    return r, alpha
