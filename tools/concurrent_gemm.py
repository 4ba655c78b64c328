"""Are the GEMM kernels safe when two launches share the chip (two streams)?  Results must equal the serial ones bit for bit."""
import sys, math
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import torch
from vist3a_amd import ops
bf16 = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(8192, 1536, device="cuda", generator=g).to(bf16)
wqk = (torch.randn(3072, 1536, device="cuda", generator=g) / 39).to(bf16)
wv = (torch.randn(1536, 1536, device="cuda", generator=g) / 39).to(bf16)
bqk = torch.randn(3072, device="cuda", generator=g)
bv = torch.randn(1536, device="cuda", generator=g)
qk_ref = ops.gemm(x, wqk, bqk).clone()
vt_ref = ops.gemm(wv, x, bv, bias_row=True).clone()
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = [0, 0]
for it in range(40):
    qk = torch.empty_like(qk_ref); vt = torch.empty_like(vt_ref)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        ops.gemm(x, wqk, bqk, out=qk)
    with torch.cuda.stream(s2):
        ops.gemm(wv, x, bv, out=vt, bias_row=True)
    torch.cuda.synchronize()
    bad[0] += int(not torch.equal(qk, qk_ref)); bad[1] += int(not torch.equal(vt, vt_ref))
print("concurrent QK / V^T GEMMs differing from serial:", bad, "of 40")

# the same question for the HBM-bound q/k RMSNorm+RoPE pass running beside a GEMM, and for attention beside a GEMM
from vist3a_amd.wan.dit import rope_table, WAN_1_3B
rope = rope_table(WAN_1_3B, 4, 32, 32, "cuda")
nq = torch.randn(1536, device="cuda", generator=g)
qk0 = qk_ref.clone()
q_ref = qk0[:, :1536].clone()
ops.rmsnorm_rope(q_ref, nq, out=q_ref, rope=rope, head_dim=128, tokens_per_batch=4096, eps=1e-6)
torch.cuda.synchronize()
bad = [0, 0]
for it in range(40):
    qk = qk0.clone(); vt = torch.empty_like(vt_ref)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        ops.rmsnorm_rope(qk[:, :1536], nq, out=qk[:, :1536], rope=rope, head_dim=128, tokens_per_batch=4096, eps=1e-6)
    with torch.cuda.stream(s2):
        ops.gemm(wv, x, bv, out=vt, bias_row=True)
    torch.cuda.synchronize()
    bad[0] += int(not torch.equal(qk[:, :1536], q_ref)); bad[1] += int(not torch.equal(vt, vt_ref))
print("concurrent rmsnorm_rope / V^T GEMM differing from serial:", bad, "of 40")
