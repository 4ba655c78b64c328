"""Full-size oracle outputs as committed digests.

The CPU oracle needs 2 - 11 minutes per forward at production size (width-1024 reconstruction in its bf16-contract mode: 654 s on the GPU
box's 128 host threads) - three such tests took 1065 s of a 1317 s `pytest -m gpu` run, more than the driver's step limit allows.  The
oracle's result does not depend on the GPU, so those tests take it from a fixture: a DIGEST of every oracle output tensor (every `step`-th
element of the flattened tensor, ~16 k values each) written by the SAME test code with the live oracle (`tests/golden/make_fullsize_oracle.sh`,
run on the GPU box's host cores) and keyed to an exact integer checksum of the seeded inputs and weights, so a change of seeds, shapes or
generator invalidates it loudly.  The HIP outputs are sampled at the same indices; relative L2 over >= 16 k samples of a tensor agrees
with the full-tensor figure to ~1 %.  `V3A_LIVE_ORACLE=1` runs the oracle live instead (full cost), `V3A_WRITE_ORACLE=1` also rewrites
the fixture (into $V3A_ORACLE_OUT if set)."""
from __future__ import annotations

import os
from pathlib import Path
from typing import Callable, Dict

import torch

GOLD = Path(__file__).parent / "golden"
TARGET = 16384


def sample_index(numel: int) -> torch.Tensor:
    step = max(1, numel // TARGET)
    if step > 1 and step % 2 == 0:
        step += 1          # odd stride: does not lock onto power-of-two row lengths
    return torch.arange(0, numel, step)


def digest(t) -> torch.Tensor:
    """every step-th element of the contiguous flattened tensor, f32 on the CPU (scalars become one-element tensors)"""
    if not torch.is_tensor(t):
        return torch.tensor([float(t)], dtype=torch.float32)
    flat = t.detach().float().contiguous().reshape(-1).cpu()
    return flat[sample_index(flat.numel())].clone()


def rel(x, ref_digest: torch.Tensor) -> float:
    """relative L2 of the full tensor x, sampled at the digest's indices, against an oracle digest"""
    xs = digest(x)
    if xs.numel() != ref_digest.numel():
        raise ValueError(f"shape mismatch against the oracle digest: {xs.numel()} vs {ref_digest.numel()} samples")
    return ((xs - ref_digest).norm() / ref_digest.norm()).item()


def rel_dd(a: torch.Tensor, b: torch.Tensor) -> float:
    return ((a - b).norm() / b.norm()).item()


def checksum(*tensors) -> str:
    """exact, order-independent integer checksum of the bit patterns (fp32 / bf16 inputs and weights)"""
    s = 0
    for t in tensors:
        t = t.detach().contiguous().cpu()
        bits = t.view(torch.int32) if t.element_size() == 4 else t.view(torch.int16).to(torch.int32)
        s = (s * 1000003 + int(bits.to(torch.int64).sum().item()) + t.numel()) % (1 << 61)
    return str(s)


def oracle(name: str, fingerprint: str, compute: Callable[[], Dict[str, object]]):
    """-> ({key: digest}, live: bool).  `compute` runs the CPU oracle and returns full tensors / numbers."""
    from safetensors import safe_open
    from safetensors.torch import save_file
    path = GOLD / f"oracle_{name}.safetensors"
    live = os.environ.get("V3A_LIVE_ORACLE") == "1" or not path.exists()
    if not live:
        f = safe_open(str(path), "pt")
        meta = f.metadata() or {}
        if meta.get("fingerprint") != fingerprint:
            raise AssertionError(f"{path.name} was generated for other inputs (fingerprint {meta.get('fingerprint')} != {fingerprint}): "
                                 "re-run tests/golden/make_fullsize_oracle.sh")
        return {k: f.get_tensor(k) for k in f.keys()}, False
    with torch.no_grad():
        out = compute()
    dig = {k: digest(v) for k, v in out.items()}
    if os.environ.get("V3A_WRITE_ORACLE") == "1":
        dst = Path(os.environ.get("V3A_ORACLE_OUT") or GOLD)
        dst.mkdir(parents=True, exist_ok=True)
        save_file(dig, str(dst / path.name), metadata={"fingerprint": fingerprint, "torch": torch.__version__, "threads": str(torch.get_num_threads()),
                                                       "sampling": f"every max(1, numel // {TARGET}) (made odd) element of the flattened tensor"})
    return dig, True
