"""Experiment: the CFG pair as ONE batch-2 DiT forward vs TWO batch-1 forwards in flight on two HIP streams (same kernels, M halved):
does overlapping one forward's kernel tails / prologues / store bursts with the other's main loops beat the larger GEMM M?"""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.wan.dit import WAN_1_3B, WanDiT
from vist3a_amd.wan.weights import random_dit_state_dict
from vist3a_amd.t23d import synthetic_text_embeddings
dev = "cuda"
cfg = WAN_1_3B
dit = WanDiT(cfg, random_dit_state_dict(cfg, seed=0, device=dev), device=dev)
pe, ne = synthetic_text_embeddings(dev)
text2 = torch.cat([pe, ne], 0).contiguous()
t0, t1 = pe.contiguous(), ne.contiguous()
shape = (16, 4, 64, 64)   # 4096 tokens
ts = torch.full((50,), 500, device=dev, dtype=torch.int64)
tab2, tab1 = dit.time_tables(ts, 2), dit.time_tables(ts, 1)
x2 = torch.randn((2,) + shape, device=dev).bfloat16()
x1 = x2[:1].contiguous()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

def batch2(n):
    for i in range(n):
        dit.forward(None, None, text2, tokens_in=True, tokens_out=True, latent_shape=(2,) + shape, time_table=tab2[i % 50])

def two(n):
    main = torch.cuda.current_stream()
    for i in range(n):
        e = torch.cuda.Event(); e.record(main)
        for s, t, lane in ((s0, t0, 0), (s1, t1, 1)):
            s.wait_event(e)
            with torch.cuda.stream(s):
                dit.forward(None, None, t, tokens_in=True, tokens_out=True, latent_shape=(1,) + shape, time_table=tab1[i % 50], lane=lane)
        for s in (s0, s1):
            e2 = torch.cuda.Event(); e2.record(s); main.wait_event(e2)

dit.forward(x2, ts[:1].expand(2), text2); dit.forward(x1, ts[:1], t0, lane=0); dit.forward(x1, ts[:1], t1, lane=1)
torch.cuda.synchronize()
graphs = {}
for name, fn in (("batch2", batch2), ("two_streams", two)):
    fn(2); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn(1)
    graphs[name] = g
def replay(name):
    return lambda n: [graphs[name].replay() for _ in range(n)]
for name, fn in (("batch2", batch2), ("two_streams", two), ("batch2_graph", replay("batch2")), ("two_streams_graph", replay("two_streams")),
                 ("batch2_graph", replay("batch2")), ("two_streams_graph", replay("two_streams"))):
    fn(5); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(30); b.record(); torch.cuda.synchronize()
    print(json.dumps({"mode": name, "ms_per_cfg_step": round(a.elapsed_time(b) / 30, 3)}), flush=True)
