"""Per-rank compute time of the sequence-parallel DiT forward at the shard sizes of the latency mode (BASELINE configs #3/#4),
measured on ONE GPU with a mock group whose all-gather replicates the local slab (the values are meaningless; every kernel runs at
the shape a real rank would run it).  Compared with the unsharded forward / P = what perfect scaling of the compute would give."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vist3a_amd.wan.dit import WAN_1_3B, WAN_14B, WanDiT
from vist3a_amd.wan.weights import random_dit_state_dict


class _Done:
    def wait(self):
        return None


class SelfGather:
    def __init__(self, world):
        self.world, self.rank = world, 0

    def all_gather(self, out, inp):
        out.view(self.world, -1).copy_(inp.reshape(1, -1).expand(self.world, -1))
        return _Done()


def timed(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


if __name__ == "__main__":
    cfg = WAN_14B if "14b" in sys.argv[1:] else WAN_1_3B
    m = WanDiT(cfg, random_dit_state_dict(cfg, seed=0, device="cuda"))
    if "fp8" in sys.argv[1:]:   # config #4's precision mode: e4m3 block GEMMs and e4m3 attention over e4m3 K | V^T slabs
        m.enable_fp8_gemm()
        m.attn_dtype = "fp8"
    text = torch.zeros(1, 512, 4096, device="cuda")
    text[0, :64] = torch.randn(64, 4096, device="cuda") * 0.1
    t = torch.tensor([900], device="cuda")
    frames = 6 if "views21" in sys.argv[1:] else 4   # 21 views = 6 latent frames = 6144 tokens (BASELINE config #3)
    lat = torch.randn(1, 16, frames, 64, 64, device="cuda").bfloat16()
    import os
    if os.environ.get("SP_KV_SPLIT"):   # A/B of the key-split factor of the shard's attention (default: WanDiT._sp_split's choice)
        m.sp_kv_split = int(os.environ["SP_KV_SPLIT"])
    if os.environ.get("SP_ONLY"):   # profiling aid: only the sharded forward at P = SP_ONLY (rocprofv3 --kernel-trace --stats -- ...)
        sp = SelfGather(int(os.environ["SP_ONLY"]))
        print(json.dumps(dict(P=sp.world, ms=round(timed(lambda: m(lat, t, text, sp=sp)), 2))))
        sys.exit(0)
    full = timed(lambda: m(lat, t, text))
    ntok = frames * 1024
    rows = [dict(P=1, B=1, local_tokens=ntok, ms=round(full, 2))]
    for P in (2, 4, 8):
        sp = SelfGather(P)
        ms = timed(lambda: m(lat, t, text, sp=sp))
        rows.append(dict(P=P, B=1, local_tokens=ntok // P, ms=round(ms, 2), ideal_ms=round(full / P, 2), compute_scaling_eff=round(full / P / ms, 3)))
    # the same forwards replayed from a hipGraph: GPU-side time without the host launch path (~500 launches per forward)
    for P in (1, 2, 4, 8):
        sp = SelfGather(P) if P > 1 else None
        m(lat, t, text, sp=sp)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            m(lat, t, text, sp=sp)
        ms = timed(g.replay)
        rows.append(dict(P=P, B=1, local_tokens=ntok // P, graph_replay_ms=round(ms, 2)))
    for r in rows:
        print(json.dumps(r))
