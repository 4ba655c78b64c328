"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own modules (stub-imported from
/root/reference, only possible in the build container) on seeded inputs and on weights produced by the oracle's
`make_weights` functions (so the test side can regenerate identical weights without shipping them).

    python tests/golden/make_golden.py [name ...]

Each generator also asserts that the oracle restatement reproduces the reference output (fp32, tight tolerance)
before writing the fixture, so a drifting oracle cannot silently re-bless itself."""
from __future__ import annotations

import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT))

import torch
from safetensors.torch import save_file

import _ref_import

_ref_import.install()


def _save(name, tensors):
    out = HERE / f"{name}.safetensors"
    save_file({k: v.contiguous() for k, v in tensors.items()}, str(out))
    print(f"wrote {out} ({out.stat().st_size / 1024:.0f} KiB)")


def vae_decode_tiny():
    """AutoencoderKLWan(base_dim=16)._decode on z[1,16,3,8,8] -> [1,3,9,64,64] (chunked, cached reference path)."""
    import utils.wan_utils as W
    from oracle import wan_vae as OV
    cfg = OV.WanVAEConfig(base_dim=16)
    sd = OV.make_weights(cfg, seed=11)
    vae = W.AutoencoderKLWan(base_dim=16)
    missing = vae.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if k.startswith(("decoder.", "post_quant_conv"))]
    assert not missing.unexpected_keys
    z = torch.randn(1, 16, 3, 8, 8, generator=torch.Generator().manual_seed(21))
    with torch.no_grad():
        ref = vae._decode(z, return_dict=False)[0]
        mine = OV.decode(sd, cfg, z)
    err = (ref - mine).abs().max().item()
    sat = (ref.abs() >= 1.0).float().mean().item()
    print(f"vae_decode_tiny: oracle vs reference max abs err {err:.2e}; clamped fraction {sat:.3f}")
    assert err < 2e-5 and sat < 0.2
    _save("vae_decode_tiny", {"z": z, "out": ref})


GENERATORS = {f.__name__: f for f in [vae_decode_tiny]}

if __name__ == "__main__":
    names = sys.argv[1:] or list(GENERATORS)
    for n in names:
        GENERATORS[n]()
