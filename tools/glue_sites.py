"""Where the host-glue device launches of one scene come from.  torch ops that move or touch device tensors (copy_, contiguous, clone, to,
cat, __setitem__, arithmetic) are wrapped and logged with the calling repo line and the bytes involved; one scene is run after a warm-up
scene and the sites are listed by bytes.  A development instrument (the product path has no such wrappers)."""
import collections, sys, traceback
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from vist3a_amd.t23d import Text23DGS, synthetic_text_embeddings
from vist3a_amd.wan.dit import WAN_1_3B

log = collections.defaultdict(lambda: [0, 0])
active = False

def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "/vist3a_amd/" in fr.filename:
            return f"{fr.filename.split('/vist3a_amd/')[-1]}:{fr.lineno} {fr.line.strip()[:90]}"
    return "?"

def wrap(obj, name):
    orig = getattr(obj, name)
    def f(*a, **k):
        r = orig(*a, **k)
        if active:
            t = next((x for x in a if torch.is_tensor(x) and x.is_cuda), None)
            if t is None and torch.is_tensor(r) and r.is_cuda:
                t = r
            if t is not None:
                big = max([x.numel() * x.element_size() for x in list(a) + [r] if torch.is_tensor(x)] or [0])
                e = log[(name, site())]
                e[0] += 1; e[1] += big
        return r
    setattr(obj, name, f)

for n in ("copy_", "contiguous", "clone", "to", "float", "bfloat16", "__setitem__", "__add__", "__mul__", "__truediv__", "__sub__", "__radd__",
          "__rmul__", "zero_", "fill_", "clamp_", "expand", "repeat"):
    wrap(torch.Tensor, n)
for n in ("cat", "stack", "zeros", "empty", "zeros_like", "ones_like", "full"):
    wrap(torch, n)

model = Text23DGS.synthetic(WAN_1_3B, seed=0, device="cuda")
pe, ne = synthetic_text_embeddings("cuda")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
def scene(i):
    lat0 = torch.randn(1, 16, 4, 64, 64, generator=torch.Generator().manual_seed(12413 + i))
    model.generate(pe, ne, latents=lat0, num_frames=13, num_inference_steps=steps, guidance_scale=7.5)
scene(0)
torch.cuda.synchronize()
active = True
scene(1)
torch.cuda.synchronize()
active = False
print(f"wrapped torch calls on device tensors in one {steps}-step scene: {sum(v[0] for v in log.values())}")
for (n, s), (c, b) in sorted(log.items(), key=lambda kv: -kv[1][1])[:60]:
    if n in ("expand", "empty", "zeros") and b < 1 << 20:
        continue
    print(f"{b / 1e6:9.1f} MB  n={c:4d}  {n:12s} {s}")
