"""Process-group bring-up: one process per GPU, RCCL over xGMI (`backend="nccl"` is RCCL on ROCm), gloo on CPU-only hosts.
Mirrors /root/reference/utils/dist_util.py:25-72 (`setup_dist`, `is_main_process`); rendezvous always on 127.0.0.1 for
single-node runs.  The inference path needs NO data-path collective: ranks stride the prompt list (inference_t23d.py:62)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def setup_dist(backend: str | None = None) -> None:
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", os.environ["RANK"])) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device(f"cuda:{local}")
    dist.init_process_group(backend, **kw)


def is_main_process() -> bool:
    return (not dist.is_initialized()) or dist.get_rank() == 0


def shard_prompts(prompts, rank: int, world: int):
    """prompt_list[rank :: world] — the reference's data-parallel split (inference_t23d.py:62)."""
    return list(prompts)[rank::world]
