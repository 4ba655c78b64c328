"""Experiment: phase time stamps (s_memtime) of one tile of the pipelined one-wave-per-SIMD attention kernel (needs a -DV3A_PW_ABL=32.. build)."""
import ctypes, sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import lib, ops
L = lib.load()
B, H, N, D = 2, 16, 4096, 128
d = H * D
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B * N, d, device="cuda", generator=g) * 0.5).bfloat16()
k = (torch.randn(B * N, d, device="cuda", generator=g) * 0.5).bfloat16()
vt = torch.randn(d, B * N, device="cuda", generator=g).bfloat16()
o = torch.empty(B * N, d, device="cuda", dtype=torch.bfloat16)
L.v3a_attention_set_kernel(3)
for _ in range(5):
    ops.attention(q, k, vt, o, B=B, H=H, Nq=N, Nk=N, D=D, q_batch_stride=N * d, k_batch_stride=N * d, vt_batch_stride=N, o_batch_stride=N * d)
torch.cuda.synchronize()
raw = ctypes.CDLL(os.environ["V3A_LIB"])
buf = (ctypes.c_longlong * 32)()
print("rc", raw.v3a_debug_read(buf, 32))
t = list(buf)
print("phase A %d  rescale+gap %d  phase B %d  wait %d  barrier %d  | tile total %d (s_memtime ticks = 100 MHz? or shader clock)" % (t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]))
print("phase A steps:", [t[8 + i] - (t[0] if i == 0 else t[8 + i - 1]) for i in range(8)])
print("phase B steps:", [t[16 + i] - (t[2] if i == 0 else t[16 + i - 1]) for i in range(16)])
