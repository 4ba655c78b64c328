"""Multi-GPU decomposition of ONE denoise (latency mode, BASELINE.json configs #3/#4).  The reference has no sequence-parallel
inference path (its ranks only stride the prompt list, /root/reference/inference_t23d.py:62); this is new, designed for xGMI.

The throughput path needs no collective (ranks stride the prompt list, utils/dist_util.py).  When a single scene must
finish sooner, the 50-step loop is split two ways:

  * CFG-parallel (2x, ~0.5 MB per step): the conditional and unconditional DiT branches run on two ranks (B=1 each
    instead of one B=2 batch) and swap their noise predictions once per step;
  * sequence-parallel inside each branch: every rank owns N/P latent tokens for the whole network.  Everything except
    self-attention is token-local.  For self-attention each rank projects K and V^T for ITS tokens, one all-gather
    hands everybody the full K / V^T, and the rank's N/P query rows run through the same flash kernel against all N
    keys.  (Ulysses would all-to-all Q, K, V and O — four exchanges of the activation per layer; on point-to-point xGMI
    one all-gather of K,V (2·N·d·2 B per layer, overlapped with the Q projection) is the cheaper pattern, and it keeps
    the head count unconstrained: 12 heads do not divide by 8 ranks.)

Rows are computed by the same kernels in the same k-order as the single-GPU path, so the sharded result is bit-identical
to it — that is what tests/test_dit_gpu.py::test_seq_parallel_* assert, using `ThreadWorld` (P virtual ranks as threads on
one GPU) because the dev loop has a single device; `DistGroup` is the RCCL/gloo backend used by real multi-process runs.
"""
from __future__ import annotations

import threading
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch


class _Done:
    def wait(self):
        return None


class DistGroup:
    """torch.distributed backend (RCCL over xGMI on GPUs, gloo on CPU)."""

    # the collectives are enqueued on the calling stream (no host rendezvous per call): GraphedWanDiT(capture_sp=True) may capture them
    graph_safe = True

    def __init__(self, ranks: List[int], my_global_rank: int, pg):
        self.ranks, self.world, self.rank, self.pg = list(ranks), len(ranks), list(ranks).index(my_global_rank), pg
        # measurement mode (bench.py --parallel scene, AFTER its timed region): every all-gather runs synchronously between two events
        # on the current stream, so its own duration is seen without the compute it normally hides under
        self.profile = False
        self.events: List[tuple] = []

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        """out [P, n] <- every rank's inp [n]; asynchronous: returns a handle whose wait() orders the current stream."""
        import torch.distributed as dist
        if self.profile and inp.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=self.pg, async_op=False)
            e1.record()
            self.events.append((e0, e1, out.numel() * out.element_size()))
            return _Done()
        return dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=self.pg, async_op=True)

    def profile_summary(self) -> dict:
        """{calls, ms, bytes} of the all-gathers recorded in measurement mode (synchronises the device)."""
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b, _ in self.events)
        out = dict(calls=len(self.events), ms=ms, gathered_bytes=sum(n for _, _, n in self.events))
        self.events = []
        return out


class ThreadWorld:
    """P virtual ranks as threads of one process sharing one device stream (validation on a 1-GPU box)."""

    def __init__(self, world: int):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots: List[Optional[torch.Tensor]] = [None] * world

    def group(self, rank: int) -> "ThreadGroup":
        return ThreadGroup(self, rank)

    def run(self, fn):
        """fn(rank) on every virtual rank; returns the per-rank results (re-raises the first failure)."""
        res, err = [None] * self.world, [None] * self.world

        def body(r):
            try:
                res[r] = fn(r)
            except BaseException as e:  # noqa: BLE001 - surfaced below
                err[r] = e
                self.barrier.abort()

        th = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in err:
            if e is not None and not isinstance(e, threading.BrokenBarrierError):
                raise e
        for e in err:
            if e is not None:
                raise e
        return res


class ThreadGroup:
    graph_safe = False   # host barriers between threads: cannot be captured

    def __init__(self, w: ThreadWorld, rank: int):
        self.w, self.rank, self.world = w, rank, w.world

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        w = self.w
        w.slots[self.rank] = inp
        w.barrier.wait()
        o = out.view(self.world, -1)
        for r in range(self.world):
            o[r].copy_(w.slots[r].reshape(-1))
        w.barrier.wait()
        return _Done()


@dataclass
class DenoisePlan:
    """How one scene's denoise is spread over ranks: `cfg` = 2-rank cond/uncond split, `sp` = token shards of the DiT."""
    sp: Optional[object] = None
    cfg: Optional[object] = None
    world: Optional[object] = None     # all ranks of the scene: the view-sharded reconstruction (recon/engine.py forward_sharded)

    @staticmethod
    def layout(world: int) -> Tuple[int, int]:
        """-> (cfg_degree, sp_degree): CFG-parallel first (communication-free 2x), the rest sequence-parallel."""
        if world < 1:
            raise ValueError("world must be >= 1")
        if world % 2 == 0:
            return 2, world // 2
        return 1, world

    @classmethod
    def from_dist(cls) -> "DenoisePlan":
        """Build the groups on the default torch.distributed world (every rank must call this)."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(), dist.get_rank()
        cfg_deg, sp_deg = cls.layout(world)
        sp = cfg = None
        for c in range(cfg_deg):  # rank = c*sp_deg + s
            ranks = list(range(c * sp_deg, (c + 1) * sp_deg))
            pg = dist.new_group(ranks) if sp_deg > 1 else None
            if rank in ranks and sp_deg > 1:
                sp = DistGroup(ranks, rank, pg)
        if cfg_deg == 2:
            for s in range(sp_deg):
                ranks = [s, s + sp_deg]
                pg = dist.new_group(ranks)
                if rank in ranks:
                    cfg = DistGroup(ranks, rank, pg)
        wg = DistGroup(list(range(world)), rank, None) if world > 1 else None     # (pg None = the default process group)
        return cls(sp=sp, cfg=cfg, world=wg)

    @classmethod
    def from_threads(cls, world: int) -> List["DenoisePlan"]:
        """The same layout on virtual ranks (one plan per thread)."""
        cfg_deg, sp_deg = cls.layout(world)
        sp_worlds = [ThreadWorld(sp_deg) for _ in range(cfg_deg)] if sp_deg > 1 else None
        cfg_worlds = [ThreadWorld(2) for _ in range(sp_deg)] if cfg_deg == 2 else None
        plans = []
        ww = ThreadWorld(world) if world > 1 else None
        for r in range(world):
            c, s = divmod(r, sp_deg)
            plans.append(cls(sp=sp_worlds[c].group(s) if sp_worlds else None,
                             cfg=cfg_worlds[s].group(c) if cfg_worlds else None, world=ww.group(r) if ww else None))
        return plans
