"""Text->latent denoise loop (the generator half of /root/reference/inference_t23d.py:85-114).

`WanT2VPipeline.__call__` keeps the keyword surface of the diffusers==0.33.1 `WanPipeline.__call__` the reference
invokes (prompt / negative_prompt / height / width / num_frames / num_inference_steps / guidance_scale /
output_type="latent" -> {"frames": latents}).  Differences, all deliberate and documented in DESIGN.md:
  * classifier-free guidance runs the conditional and unconditional branches as ONE batch-2 forward (the in-tree
    training loop does the same: /root/reference/train_vdm.py:592-607) — doubles the GEMM M dimension;
  * the initial noise comes from an explicit torch.Generator / `latents=` argument (CUDA Philox noise of the
    reference is not reproducible across vendors);
  * prompts are encoded by whatever `text_encoder` callable is supplied (UMT5-XXL weights are not part of this
    path); `prompt_embeds=` bypasses it.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Union

import torch

from .scheduler import UniPCMultistepScheduler

# AutoencoderKLWan config constants (/root/reference/utils/wan_utils.py:925-960)
LATENTS_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
                0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
LATENTS_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
               3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


class WanT2VPipeline:
    vae_scale_factor_temporal = 4
    vae_scale_factor_spatial = 8

    def __init__(self, transformer, scheduler: UniPCMultistepScheduler, vae=None,
                 text_encoder: Optional[Callable[[List[str], int], torch.Tensor]] = None, device="cuda", plan=None):
        """`plan` (wan/seqpar.py DenoisePlan) spreads ONE scene over several ranks; None = everything on this GPU."""
        self.plan = plan
        self.transformer = transformer
        self.scheduler = scheduler
        self.vae = vae
        self.text_encoder = text_encoder
        self.device = torch.device(device)

    # One launch per step for guidance + UniPC + (un)patchify (csrc/denoise_step.hip); False = the tensor-op loop below.
    # The fused loop is taken only when `_can_fuse()` holds (a plain WanDiT, patch (1,2,2), no sequence-parallel plan, no callback,
    # solver_order <= 2 - the orders plan_step implements); a GraphedWanDiT transformer has no token buffers and runs the tensor-op
    # loop.  The fused loop drives the scheduler through plan_step() only: scheduler.step() is never called, so
    # `scheduler.model_outputs` / `.last_sample` are NOT updated by it (step_index / lower_order_nums are).
    fused = True
    # How the guided prediction enters `scheduler.step`.  False (default) = diffusers' WanPipeline, the call the inference path makes
    # (/root/reference/inference_t23d.py:94-103): the prediction stays in the transformer's dtype, so the scheduler's `sigma * v` is rounded to
    # bf16.  True = the reference's in-tree training roll-out, `scheduler.step(noise_pred.float(), t, latents.float())`
    # (/root/reference/train_vdm.py:620-622): fp32 product.  The two reference loops differ in exactly this cast (tensor-op loop only).
    scheduler_inputs_fp32 = False

    def _can_fuse(self, callback) -> bool:
        sch, tr = self.scheduler, self.transformer
        return (self.fused and not self.scheduler_inputs_fp32 and self.plan is None and callback is None and hasattr(tr, "token_buffers")
                and tuple(getattr(tr.cfg, "patch_size", ())) == (1, 2, 2) and hasattr(sch, "plan_step")
                and getattr(sch, "solver_order", 3) <= 2)

    def _fused_loop(self, latents: torch.Tensor, text: torch.Tensor, nb: int, shape: tuple, guidance: Optional[float]) -> torch.Tensor:
        """The denoise loop with everything between two DiT forwards in ONE kernel (ops.unipc_cfg_step) and the time conditioning of
        the whole schedule computed up front (WanDiT.time_tables): per step = one DiT forward + one launch.  Bit-identical to the
        tensor-op loop (tests/test_boundary_gpu.py::test_fused_denoise_loop_is_bit_identical_to_tensor_ops)."""
        from .. import ops
        dit, sch = self.transformer, self.scheduler
        bf16 = torch.bfloat16
        tok, out_tok = dit.token_buffers(nb, shape)
        tables = dit.time_tables(sch.timesteps, nb)
        cur = latents.contiguous().clone()
        last, m_a, m_b = torch.empty_like(cur), torch.empty_like(cur), torch.empty_like(cur)
        # first step's input tokens: the one patchify that is not produced by the step kernel
        B, C, T, H, W = (nb,) + tuple(shape[1:])
        x5 = cur.to(bf16).expand(nb, -1, -1, -1, -1).view(nb, C, T, 1, H // 2, 2, W // 2, 2).permute(0, 2, 4, 6, 1, 3, 5, 7)
        tok.view(nb, T, H // 2, W // 2, C, 1, 2, 2).copy_(x5)
        m1 = m2 = None
        for i in range(len(sch.timesteps)):
            dit.forward(None, None, text, tokens_in=True, tokens_out=True, latent_shape=(nb,) + tuple(shape[1:]), time_table=tables[i])
            c = sch.plan_step()
            m_out = m_b if m1 is m_a else m_a         # overwrites the older of the two x0 predictions (read by its own thread first)
            ops.unipc_cfg_step(out_tok, tok, cur, last if c["corr_order"] else None, m1, m2, m_out, last, cur, batch=nb, guidance=guidance,
                               coeffs=c)
            m2, m1 = m1, m_out
        return cur

    def encode_prompt(self, prompt: Union[str, List[str]], max_sequence_length: int = 512) -> torch.Tensor:
        if self.text_encoder is None:
            raise RuntimeError("no text encoder attached: pass prompt_embeds / negative_prompt_embeds "
                               "([B, 512, 4096], zero rows past the prompt length) or attach a UMT5 encoder callable")
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        return self.text_encoder(prompt, max_sequence_length).to(self.device)

    @torch.no_grad()
    def __call__(self, prompt=None, negative_prompt=None, height: int = 480, width: int = 832, num_frames: int = 81,
                 num_inference_steps: int = 50, guidance_scale: float = 5.0, generator: Optional[torch.Generator] = None,
                 latents: Optional[torch.Tensor] = None, prompt_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, output_type: str = "latent",
                 max_sequence_length: int = 512, callback: Optional[Callable] = None) -> Dict[str, torch.Tensor]:
        if num_frames % self.vae_scale_factor_temporal != 1:
            num_frames = num_frames // self.vae_scale_factor_temporal * self.vae_scale_factor_temporal + 1
        num_frames = max(num_frames, 1)
        if prompt_embeds is None:
            prompt_embeds = self.encode_prompt(prompt, max_sequence_length)
        do_cfg = guidance_scale > 1.0
        if do_cfg and negative_prompt_embeds is None:
            negative_prompt_embeds = self.encode_prompt(negative_prompt if negative_prompt is not None else "", max_sequence_length)
        B = prompt_embeds.shape[0]
        if B != 1:
            raise NotImplementedError("one prompt per call (the reference loops over prompts one at a time)")
        cfgc = self.transformer.cfg
        shape = (B, cfgc.in_channels, (num_frames - 1) // self.vae_scale_factor_temporal + 1,
                 height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial)
        if latents is None:
            g = generator if generator is not None else torch.Generator(device="cpu").manual_seed(torch.initial_seed() % (2 ** 63))
            latents = torch.randn(shape, generator=g, dtype=torch.float32, device=g.device)
        latents = latents.to(device=self.device, dtype=torch.float32)
        if tuple(latents.shape) != shape:
            raise ValueError(f"latents shape {tuple(latents.shape)} != {shape}")
        self.scheduler.set_timesteps(num_inference_steps, device=self.device)
        cfgp = self.plan.cfg if (self.plan is not None and do_cfg) else None
        sp = self.plan.sp if self.plan is not None else None
        if cfgp is not None:  # CFG-parallel: rank 0 of the pair runs the conditional branch, rank 1 the unconditional
            text = (prompt_embeds if cfgp.rank == 0 else negative_prompt_embeds).to(self.device).contiguous()
        elif do_cfg:
            text = torch.cat([prompt_embeds, negative_prompt_embeds], 0).to(self.device).contiguous()
        else:
            text = prompt_embeds.to(self.device).contiguous()
        nb = text.shape[0]
        pair = torch.empty((2,) + shape, device=self.device, dtype=torch.bfloat16) if cfgp is not None else None
        if self._can_fuse(callback):
            latents = self._fused_loop(latents, text, nb, shape, guidance_scale if do_cfg else None)
            if output_type == "latent":
                return {"frames": latents}
            if self.vae is None:
                raise RuntimeError("output_type != 'latent' needs a VAE")
            return {"frames": self.vae.decode(denormalize_latents(latents), return_dict=False)[0]}
        for i, t in enumerate(self.scheduler.timesteps):
            x_in = latents.to(torch.bfloat16).expand(nb, -1, -1, -1, -1)
            noise = self.transformer(x_in, t.expand(nb), text, return_dict=False, sp=sp)[0]
            if cfgp is not None:
                cfgp.all_gather(pair, noise.contiguous()).wait()
                n_c, n_u = pair[0], pair[1]
                noise = n_u + guidance_scale * (n_c - n_u)
            elif do_cfg:
                n_c, n_u = noise[0:1], noise[1:2]
                noise = n_u + guidance_scale * (n_c - n_u)  # bf16 arithmetic, as the reference pipeline
            if self.scheduler_inputs_fp32:
                noise = noise.float()
            latents = self.scheduler.step(noise, t, latents, return_dict=False)[0]
            if callback is not None:
                callback(i, t, latents)
        if output_type == "latent":
            return {"frames": latents}
        if self.vae is None:
            raise RuntimeError("output_type != 'latent' needs a VAE")
        video = self.vae.decode(denormalize_latents(latents), return_dict=False)[0]
        return {"frames": video}


def denormalize_latents(latents: torch.Tensor) -> torch.Tensor:
    """latents*std + mean per channel (/root/reference/inference_t23d.py:105-113: `latents / (1/std) + mean`)."""
    mean = torch.tensor(LATENTS_MEAN, device=latents.device, dtype=latents.dtype).view(1, -1, 1, 1, 1)
    inv_std = 1.0 / torch.tensor(LATENTS_STD, device=latents.device, dtype=latents.dtype).view(1, -1, 1, 1, 1)
    return latents / inv_std + mean
