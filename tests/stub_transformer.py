"""A deterministic stand-in for the DiT (test infrastructure): a cheap nonlinear map of (latents, timestep, prompt) with the call surface of
`pipeline.transformer(...)`, evaluated on the CPU in fp32 whatever device it is handed, so that the reference's denoise loop (executed by
tests/golden/make_golden.py::denoise_loop_ref) and the product's loop see bit-identical model outputs and every remaining difference is
the LOOP's: CFG batching order, chunk order, guidance arithmetic and dtype, scheduler call, latent de-normalisation."""
from types import SimpleNamespace

import torch


class StubTransformer:
    cfg = SimpleNamespace(in_channels=16, patch_size=(1, 2, 2))

    def __init__(self, text_dim: int = 32, seed: int = 5):
        g = torch.Generator().manual_seed(seed)
        self.mix = torch.randn(16, 16, generator=g) * 0.35
        self.proj = torch.randn(text_dim, 16, generator=g) * 0.3
        self.calls = []

    def __call__(self, hidden_states=None, timestep=None, encoder_hidden_states=None, return_dict=False, **_):
        dev = hidden_states.device
        x = hidden_states.detach().to("cpu").to(torch.bfloat16).float()         # the DiT sees bf16 latents under autocast
        t = timestep.detach().to("cpu").float().view(-1, 1, 1, 1, 1) / 1000.0
        c = encoder_hidden_states.detach().to("cpu").float().mean(1) @ self.proj     # [B,16]
        self.calls.append((tuple(x.shape), tuple(timestep.shape), tuple(encoder_hidden_states.shape)))
        y = torch.einsum("oc,bcthw->bothw", self.mix, x) * (0.5 + t) + 0.1 * torch.roll(x, 1, dims=-1) + c.view(c.shape[0], 16, 1, 1, 1)
        return (torch.tanh(y).to(torch.bfloat16).to(dev),)
