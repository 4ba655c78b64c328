"""Kernel timeline of ONE DiT CFG step (B = 2, 4096 tokens) under rocprofv3 --kernel-trace: run as
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/step_trace.py ; python tools/step_trace.py --report DIR
The report lists per-kernel-shape time of the LAST forward, and the idle gaps between consecutive kernels."""
import csv, collections, glob, json, sys
from pathlib import Path

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"]) for r in csv.DictReader(open(f))))
    # the last forward = the kernels after the last patch-embedding marker gap: take the final 1/6 of the launches (6 forwards are run)
    n = len(rows) // 6
    last = rows[-n:]
    span = (last[-1][1] - last[0][0]) / 1e3
    busy = sum(e - s for s, e, _, _ in last) / 1e3
    gaps = [(last[i + 1][0] - last[i][1]) / 1e3 for i in range(len(last) - 1)]
    acc = collections.defaultdict(list)
    for s, e, nme, g in last:
        nme = nme.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        acc[(nme[:60], g)].append((e - s) / 1e3)
    print(json.dumps(dict(kernels=len(last), span_us=round(span, 1), busy_us=round(busy, 1), idle_us=round(span - busy, 1),
                          mean_gap_us=round(sum(gaps) / len(gaps), 2), gaps_over_5us=sum(g > 5 for g in gaps))))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:28]:
        print(f"{k[0]:62s} grid {k[1]:>8s} n {len(v):4d} avg_us {sum(v) / len(v):8.1f} sum_us {sum(v):9.1f} share {100 * sum(v) / busy:5.1f}%")
    sys.exit(0)

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.wan.dit import WAN_1_3B, WanDiT
from vist3a_amd.wan.weights import random_dit_state_dict
m = WanDiT(WAN_1_3B, random_dit_state_dict(WAN_1_3B, seed=0, device="cuda"))
import os
if os.environ.get("V3A_CTX_VO") == "0":
    m.ctx_vo = False
text = torch.zeros(2, 512, 4096, device="cuda")
text[0, :64] = torch.randn(64, 4096, device="cuda") * 0.1
text[1, :80] = torch.randn(80, 4096, device="cuda") * 0.1
t = torch.tensor([900, 900], device="cuda")
lat = torch.randn(2, 16, 4, 64, 64, device="cuda").bfloat16()
torch.cuda.synchronize()
for _ in range(6):
    m(lat, t, text)
torch.cuda.synchronize()
