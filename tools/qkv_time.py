import sys, json, math
sys.path.insert(0, "/root/repo")
import torch
from vist3a_amd import ops
bf16=torch.bfloat16
g=torch.Generator(device="cuda").manual_seed(0)
M,d,K=8192,1536,1536
x=torch.randn(M,K,device="cuda",generator=g).to(bf16)
w=(torch.randn(3*d,K,device="cuda",generator=g)/39).to(bf16)
b=torch.randn(3*d,device="cuda",generator=g)
bqk=b[:2*d].contiguous(); bv=b[2*d:].contiguous()
qk=torch.empty(M,2*d,device="cuda",dtype=bf16); vt=torch.empty(d,M,device="cuda",dtype=bf16)
def t(fn,n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); best=1e9
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1)/n)
    return round(best*1e3,2)
def sep():
    ops.gemm(x,w[:2*d],bqk,out=qk); ops.gemm(w[2*d:],x,bv,out=vt,bias_row=True)
print(json.dumps(dict(fused_us=t(lambda: ops.gemm(x,w,b,out=qk,t_out=vt,t_col0=2*d)), separate_us=t(sep), qk_us=t(lambda: ops.gemm(x,w[:2*d],bqk,out=qk)),
  vt_us=t(lambda: ops.gemm(w[2*d:],x,bv,out=vt,bias_row=True)), n4608_plain_us=t(lambda: ops.gemm(x,w,b,tile=6)))))
