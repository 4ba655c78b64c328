"""Wan-2.1 generator side of the VIST3A path (DiT denoiser, UniPC scheduler, pipeline, VAE decoder)."""
