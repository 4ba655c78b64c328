"""Kernel histogram of exactly ONE scene's worth of work (50-step denoise + VAE decode + reconstruction), no bench extras:
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/scene_trace.py ; python tools/scene_trace.py --report DIR
Three scenes are run; the report keeps every kernel between the END of the second scene's 50th unipc_cfg_step launch and the end of the
third scene's, i.e. the second scene's VAE + reconstruction and the third scene's denoise - one of each stage in steady state (the
first scene also builds the reconstruction engine: weight packing and uploads) - and lists the kernels that are NOT this library's
(torch elementwise / copy / fill launches: host glue) separately."""
import csv, collections, glob, json, sys
from pathlib import Path

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", "?")) for r in csv.DictReader(open(f))))
    steps = [i for i, r in enumerate(rows) if "unipc_cfg_step_kernel" in r[2]]
    n = len(steps) // 3
    first, last = steps[2 * n - 1] + 1, steps[-1] + 1
    sel = rows[first:last]
    busy = sum(e - s for s, e, _, _g in sel) / 1e3
    acc = collections.defaultdict(list)
    for s, e, nme, _g in sel:
        nme = nme.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        acc[nme[:110]].append((e - s) / 1e3)
    ours = lambda k: not (k.startswith("void at::") or k.startswith("__amd_rocclr") or k.startswith("at::") or "rocprim" in k)
    glue = {k: v for k, v in acc.items() if not ours(k) and "rocprim" not in k}
    print(json.dumps(dict(kernels=len(sel), busy_ms=round(busy / 1e3, 2), span_ms=round((sel[-1][1] - sel[0][0]) / 1e6, 2),
                          glue_launches=sum(len(v) for v in glue.values()), glue_ms=round(sum(sum(v) for v in glue.values()) / 1e3, 3),
                          glue_share_pct=round(100 * sum(sum(v) for v in glue.values()) / busy, 3))))
    # where the GPU idles: gaps > 20 us between consecutive kernels, by the kernel that follows them
    gaps = collections.defaultdict(lambda: [0, 0.0])
    for a, b in zip(sel[:-1], sel[1:]):
        g = (b[0] - a[1]) / 1e3
        if g > 20:
            k = b[2].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
            gaps[k][0] += 1; gaps[k][1] += g
    print("idle gaps > 20 us (count, total us) before:", {k: (v[0], round(v[1])) for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:8]})
    # glue launches inside the denoise loop (between two unipc steps) vs outside it, by (kernel, grid)
    inner = collections.Counter()
    for a, b in zip(steps[2 * n:-1], steps[2 * n + 1:]):
        for r in rows[a + 1:b]:
            k = r[2].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            if not ours(k):
                inner[(k[:60], r[3])] += 1
    print("glue launches per denoise step (kernel, grid): ", {f"{k[0]} g{k[1]}": round(v / max(1, len(steps) - 2 * n - 1), 2) for k, v in inner.most_common(12)})
    # the longest individual glue launches, with the library kernel that ran just before each (locates the call site)
    idx = {id(r): i for i, r in enumerate(sel)}
    big = sorted((r for r in sel if not ours(r[2].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))),
                 key=lambda r: r[0] - r[1])[:25]
    for r in big:
        i = idx[id(r)]
        prev = next((sel[j][2] for j in range(i - 1, -1, -1) if ours(sel[j][2].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))), "-")
        nxt = next((sel[j][2] for j in range(i + 1, len(sel)) if ours(sel[j][2].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))), "-")
        print(f"GLUE {(r[1] - r[0]) / 1e3:8.1f} us  grid {r[3]:>9s}  {r[2][:44]:44s} after {prev.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:34]:34s} before {nxt.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:40]}")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
        print(f"{'  ' if ours(k) else 'G '}{k:112s} n {len(v):5d} avg_us {sum(v) / len(v):9.1f} max_us {max(v):9.1f} sum_ms {sum(v) / 1e3:9.3f} {100 * sum(v) / busy:6.3f}%")
    sys.exit(0)

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd.t23d import Text23DGS, synthetic_text_embeddings
from vist3a_amd.wan.dit import WAN_1_3B
model = Text23DGS.synthetic(WAN_1_3B, seed=0, device="cuda")
pe, ne = synthetic_text_embeddings("cuda")
for i in range(3):
    lat0 = torch.randn(1, 16, 4, 64, 64, generator=torch.Generator().manual_seed(12413 + i))
    model.generate(pe, ne, latents=lat0, num_frames=13, num_inference_steps=50, guidance_scale=7.5)
torch.cuda.synchronize()
