import ctypes as C, sys, itertools
from pathlib import Path
import torch
lib = C.CDLL(str(Path(__file__).parent / "f8probe.so"))
dev = "cuda"
g = torch.Generator().manual_seed(0)
vals = torch.tensor([-2, -1.5, -1, -0.5, 0, 0.5, 1, 1.5, 2, 3])
A = vals[torch.randint(0, len(vals), (32, 64), generator=g)]
B = vals[torch.randint(0, len(vals), (32, 64), generator=g)]
ref = A @ B.t()                       # C[i][j] = sum_k A[i][k] B[j][k]
A8 = A.to(torch.float8_e4m3fn).view(torch.uint8)
B8 = B.to(torch.float8_e4m3fn).view(torch.uint8)
lay = {
    "L1 k=h*32+b": lambda h, b: h * 32 + b,
    "L2 k=(b/16)*32+h*16+b%16": lambda h, b: (b // 16) * 32 + h * 16 + b % 16,
    "L3 k=(b/8)*16+h*8+b%8": lambda h, b: (b // 8) * 16 + h * 8 + b % 8,
}
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, f in lay.items():
    pa = torch.zeros(64, 32, dtype=torch.uint8); pb = torch.zeros(64, 32, dtype=torch.uint8)
    for l in range(64):
        for b in range(32):
            pa[l, b] = A8[l & 31, f(l >> 5, b)]; pb[l, b] = B8[l & 31, f(l >> 5, b)]
    da, db = pa.to(dev), pb.to(dev)
    c = torch.zeros(64, 16, device=dev)
    lib.f8_mfma(C.c_void_p(da.data_ptr()), C.c_void_p(db.data_ptr()), C.c_void_p(c.data_ptr()), 0x7f7f7f7f, st)
    torch.cuda.synchronize()
    c = c.cpu()
    out = torch.zeros(32, 32)
    for l in range(64):
        for r in range(16):
            out[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = c[l, r]
    print(name, "max err as C[i][j]:", (out - ref).abs().max().item(), " as C[j][i]:", (out - ref.t()).abs().max().item())
# scale semantics: scale_a byte = 0x80 -> x2 ?
c = torch.zeros(64, 16, device=dev)
lib.f8_mfma(C.c_void_p(da.data_ptr()), C.c_void_p(db.data_ptr()), C.c_void_p(c.data_ptr()), 0x80808080, st)
torch.cuda.synchronize()
c2 = torch.zeros(64, 16, device=dev)
lib.f8_mfma(C.c_void_p(da.data_ptr()), C.c_void_p(db.data_ptr()), C.c_void_p(c2.data_ptr()), 0x7f7f7f7f, st)
torch.cuda.synchronize()
print("scale 0x80 / 0x7f ratio:", (c.sum() / c2.sum()).item())
# cvt check
x = torch.tensor([0.3, -1.7, 500.0, 1e-3, 0.0019, 0.001, 200.0, -0.06], device=dev)
y = torch.zeros(2, dtype=torch.int32, device=dev)
lib.f8_cvt(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 2, st)
torch.cuda.synchronize()
got = y.view(torch.uint8).view(torch.float8_e4m3fn).float().cpu()
print("cvt hw :", got.tolist())
print("cvt ref:", x.cpu().to(torch.float8_e4m3fn).float().tolist())
