"""Text -> 3DGS inference CLI — drop-in for /root/reference/inference_t23d.py (same flags, same output layout
`<output_dir>/<prompt[:100] sans '/'>/{prompt.txt, gaussians.ply, gs.mp4, depth.mp4}` — H.264 mp4 like the reference when imageio-ffmpeg
or OpenCV is importable; in this image neither exists and the same frames are written as Motion-JPEG gs.avi / depth.avi).

    python -m torch.distributed.run --nproc_per_node=K --master-addr 127.0.0.1 inference_t23d.py --checkpoint_path ... \
        --transformer_lora_path ... --input_texts_path prompts.txt

One process per GPU; prompts are strided over ranks (no inter-GPU traffic), exactly like the reference (:53-62)."""
from __future__ import annotations

import os
import sys
import zlib
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from vist3a_amd.models.loading import load_stitching_model  # noqa: E402
from vist3a_amd.misc.image_io import save_interpolated_video  # noqa: E402
from vist3a_amd.t23d import Text23DGS, synthetic_text_embeddings  # noqa: E402
from vist3a_amd.utils.argument import inference_vist3a_argument  # noqa: E402
from vist3a_amd.utils.dist_util import setup_dist, shard_prompts  # noqa: E402
from vist3a_amd.utils.ply_export import export_ply  # noqa: E402
from vist3a_amd.wan.dit import WAN_1_3B, WAN_14B, WanDiT  # noqa: E402
from vist3a_amd.wan.weights import load_dit_config, load_dit_state_dict, load_peft_lora, random_dit_state_dict  # noqa: E402

PROMPT_TEMPLATE = "The camera rotates around the scene, maintaining constant distance: `{}`. The orbiting trajectory captures 3D structure and consistency."
NEGATIVE_PROMPT = ("Background blur, Blurred background, Blurred scene, Artifacts, not aesthetic, not realistic, rendered noise, low quality movement, "
                   "low quality video, low quality image, deformed, disfigured, distorted, extra limbs, cloned face, skinny, glitchy, double torso, "
                   "extra arms, extra hands, mangled fingers, missing lips, ugly face, distorted legs, fused fingers, too many fingers, long neck")


def build_transformer(args, device):
    cfg = WAN_14B if "14B" in args.model_id else WAN_1_3B
    if args.checkpoint_path == "synthetic" and not os.path.isdir(args.model_id):
        sd = random_dit_state_dict(cfg, seed=0, device=str(device))
    else:
        cfg = load_dit_config(args.model_id, cfg)   # <model_id>/transformer/config.json, as diffusers' from_pretrained
        sd = load_dit_state_dict(args.model_id)
    if args.transformer_lora_path not in ("none", "", None):
        load_peft_lora(args.transformer_lora_path, sd)  # merged at load: W += (alpha/r) B A
    return WanDiT(cfg, sd, device=device)


def build_text_encoder(args, device):
    """(prompts, max_len) -> embeddings from `<model_id>/text_encoder` + `<model_id>/tokenizer` (the folders diffusers' WanPipeline
    loads, /root/reference/inference_t23d.py:71-83), or None when they are not on disk (no hub access here)."""
    te, tk = os.path.join(args.model_id, "text_encoder"), os.path.join(args.model_id, "tokenizer")
    if not (os.path.isdir(te) and os.path.isdir(tk)):
        return None
    import glob
    from safetensors.torch import load_file
    from transformers import AutoTokenizer
    from vist3a_amd.wan.text_encoder import UMT5Config, UMT5TextEncoder, make_pipeline_text_encoder
    sd = {}
    for f in sorted(glob.glob(os.path.join(te, "*.safetensors"))):
        sd.update(load_file(f))
    return make_pipeline_text_encoder(UMT5TextEncoder(UMT5Config(), sd, device=device), AutoTokenizer.from_pretrained(tk))


def main(args):
    setup_dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    device = torch.device(f"cuda:{rank % torch.cuda.device_count()}")
    torch.cuda.set_device(device)
    with open(args.input_texts_path, "r") as f:
        prompts = [line.strip() for line in f.readlines()]
    coop = args.scene_parallel and world > 1
    if not coop:  # the reference's split: whole prompts per rank, no collective
        prompts = shard_prompts(prompts, rank, world)
    gen = torch.Generator().manual_seed(12413)  # seed_everything(12413): one stream per process, consumed in prompt order
    transformer = build_transformer(args, device)
    stitched = load_stitching_model(args)
    scene = Text23DGS(transformer, stitched.diffusion_vae, stitched, flow_shift=args.flow_shift,
                      feedforward_resolution=args.feedforward_resolution, device=device)
    if coop:  # every rank works on the same prompt: CFG-parallel x sequence-parallel denoise, rank 0 writes
        from vist3a_amd.wan.seqpar import DenoisePlan
        scene.pipe.plan = DenoisePlan.from_dist()
        stitched.recon_group = scene.pipe.plan.world      # the reconstruction of the scene split by views over the same ranks
    embeds = torch.load(args.text_embeds_path, map_location="cpu") if args.text_embeds_path else None
    encode = None if (embeds is not None or args.synthetic_text) else build_text_encoder(args, device)
    for prompt in prompts:
        if encode is not None:  # the reference's own route: WanPipeline.encode_prompt(prompt, negative_prompt), 512 tokens, zero padded
            pe, ne = encode([PROMPT_TEMPLATE.format(prompt)], 512), encode([NEGATIVE_PROMPT], 512)
        elif embeds is not None:
            pe, ne = embeds[PROMPT_TEMPLATE.format(prompt)][None].to(device), embeds["__negative__"][None].to(device)
        elif args.synthetic_text:
            pe, ne = synthetic_text_embeddings(device, seed=zlib.crc32(prompt.encode()) % (2 ** 31))
        else:
            raise RuntimeError(f"no text encoder: {args.model_id}/text_encoder + /tokenizer are not on disk (no hub access here); pass "
                               "--text_embeds_path (precomputed UMT5 embeddings) or --synthetic_text")
        out, _, _ = scene.generate(pe, ne, generator=gen, num_frames=args.num_frames, num_inference_steps=args.num_inference_steps,
                                   guidance_scale=float(args.cfg_scale), height=args.resolution, width=args.resolution)
        if coop and rank != 0:
            continue
        save = Path(args.output_dir) / prompt[:100].replace("/", "")
        os.makedirs(save, exist_ok=args.overwrite)  # the reference raises when the directory exists (inference_t23d.py:126)
        (save / "prompt.txt").write_text(prompt)
        g = out.gaussians
        if not args.no_video:  # orbit video through the predicted context poses (reference :144-154)
            save_interpolated_video(out.pred_context_pose["extrinsic"], out.pred_context_pose["intrinsic"], 1, args.feedforward_resolution,
                                    args.feedforward_resolution, g, str(save), stitched.stitched_3d_model.decoder)
        export_ply(g.means[0], g.scales[0], g.rotations[0], g.harmonics[0], g.opacities[0], save / "gaussians.ply", save_sh_dc_only=True)
        print(f"[rank {rank}] {save}: {g.means.shape[1]} gaussians", flush=True)
    if world > 1:
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(inference_vist3a_argument().parse_args())
